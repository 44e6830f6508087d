// The episode-resident rollout kernel (rollout.hip) for TWO hidden layers of up to 128 channels (cfg/hidden_size.cfg:81-82:
// hidden_size = 128, n_layers = 2) at the headline (N, K) = (100, 3).  The second layer's bf16 piece image alone is 96 KB; with the
// first layer's 24 KB and 59 KB of episode state that is more than a CU's 160 KB of LDS, so two of its four K blocks are streamed
// through one 24 KB LDS buffer by LDS-DMA -- block 3 under the multiplication of blocks 0 and 1, block 2 (of the next step) under
// the simulator phases (rollout.hip: RO_X2; layout: rollout_common.h RO_X2_*).  No Verlet lists (no LDS left for them).
// Entry points: mgp_rollout_x2_*_, reached through mgp_rollout_supported / _steps_ex / _image via the wide and 128-wide builds.
#define MGP_RO_KS 8
#define MGP_RO_MAXMT 8
#define MGP_RO_X2 1
#include "rollout.hip"
