// The episode-resident rollout kernel (rollout.hip) for THREE and more hidden layers of up to 128 channels (cfg/hidden_size.cfg:
// 104-106, 128-130: hidden_size = 128, n_layers = 3, 4) at the headline (N, K) = (100, 3).  Every K block of the layers behind the
// first (4 x 24 KB per layer) is streamed from the caller's image through a ring of three LDS buffers by LDS-DMA: two blocks in
// flight while the tile waves multiply a third, one workgroup barrier per block (rollout.hip: RO_XD).  The two-layer build
// (rollout_w128x2.hip) keeps half of its one streamed layer resident and is the faster form for that shape.
// Entry points: mgp_rollout_xd_*_, reached through the chain base -> wide -> x128 -> x2 -> here.
#define MGP_RO_KS 8
#define MGP_RO_MAXMT 8
#define MGP_RO_XD 1
#include "rollout.hip"
