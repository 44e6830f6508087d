// The episode-resident rollout kernels (rollout.hip) a third time, for ONE hidden layer up to 128 wide (cfg/hidden_size.cfg:58,
// `hidden_size = 128`, n_layers = 1 in the reference's counting): eight m-tiles run two at a time on the 8-k-step operand of
// the aggregation tile, the 2-wide output layer straight from the 32 accumulator registers a lane holds.  N <= 128.
// Entry points: mgp_rollout_x128_*_, reached through mgp_rollout_supported / _steps_ex / _collect via the wide build.
#define MGP_RO_KS 8
#define MGP_RO_MAXMT 8
#define MGP_RO_X128 1
#include "rollout.hip"
