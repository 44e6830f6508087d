// The episode-resident rollout kernels (rollout.hip) once more, for layer widths up to 64: sixteen k-steps per
// activation column instead of eight (a 68-float column stride, 20-float weight fragments, four m-tiles per layer run
// two at a time).  cfg/hidden_size.cfg sweeps widths 4 .. 128; 64 is the widest whose state still fits the LDS plan.
// Entry points: mgp_rollout_wide_supported_ / mgp_rollout_wide_steps_, reached through mgp_rollout_supported / _steps.
#define MGP_RO_KS 16
#define MGP_RO_WIDE 1
#include "rollout.hip"
