// The episode-resident rollout kernels (rollout.hip, widths <= 32) once more with the hidden layers on fp32 MFMA 16x16x4
// (MGP_RO_BF16 = 0: a k-ordered fmaf chain in fp32) instead of split-bf16 16x16x32: the arithmetic the product build's three
// bf16 pieces stand in for.  A checker, not a fast path: tests/test_gpu_headline_parity.py holds the product build to it on the
// same 256 episodes.  Entry points: mgp_rollout_f32ref_supported / _steps_ex / _image_floats / _image (include/mgp.h); the
// weight image is this build's own (fp32 fragments).
#define MGP_RO_BF16 0
#define MGP_RO_F32REF 1
#include "rollout.hip"
