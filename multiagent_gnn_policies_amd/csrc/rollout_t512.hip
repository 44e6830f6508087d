// The episode-resident rollout kernel (rollout.hip) once more for launches with more episodes than the device has CUs: the
// headline instantiation -- N = 100, K = 3, the reference's policy shape [32, 32] compiled in, plain and collecting -- as a
// 512-thread workgroup of <= 80 KB of LDS, so that a CU holds TWO episodes and one episode's dependent phases run under the
// other's (rollout.hip: RO_T512).  Same arithmetic in the same order: the bits are the 1024-thread build's
// (tests/test_gpu_rollout.py::test_two_episodes_per_cu_build_is_bit_identical).  Entry points: mgp_rollout_t512_steps_ex_ /
// _collect_, reached through mgp_rollout_steps_ex / mgp_rollout_collect (MGP_RO_T512 = 0 / 1 forces the choice).
#define MGP_RO_T512 1
#include "rollout.hip"
