/*
 * mgp.h -- C ABI of libmgp.so: the MI355X (gfx950) hot path of the multi-agent GNN
 * flocking policy.  Every entry point replaces a piece of arithmetic that the reference
 * (katetolstaya/multiagent_gnn_policies) dispatches to stock ATen ops or to the external
 * gym_flock package.  The reference has no FFI of its own; the interface each function
 * stands in for is cited as reference file:line.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer into caller-owned memory (e.g. torch tensors),
 *     row-major, contiguous unless strides are passed; strides are in ELEMENTS;
 *   - fp32 unless the name says f64; sizes are plain ints;
 *   - work is ENQUEUED on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     nothing synchronises, allocates or keeps state between calls -> re-entrant per stream,
 *     HIP-graph capturable;
 *   - return 0 on success or a negative MGP_E* code; never throws.  mgp_strerror() maps
 *     codes to text.
 */
#ifndef MGP_H
#define MGP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MGP_VERSION 340            /* 0.2.0: + mgp_rollout_steps_ex / _image / _carry_*, mgp_rollout_collect, mgp_replay_gather;
                                      0.2.1: + mgp_replay_gather_many; mgp_actor_fwd covers layer widths up to 128 at N <= 128;
                                      0.3.0: + mgp_p2p_* (one-shot gradient exchange), mgp_train_step_p2p, mgp_adam_step_filed;
                                      0.3.1: + mgp_rollout_f32ref_* (checker build of the resident kernels); a timed-out exchange
                                             skips Adam for the entries it missed (mgp_train_step_p2p: timeout semantics below);
                                      0.3.2: + mgp_replay_aggregate, mgp_train_step_agg / _grads_agg / _agg_supported (DAGGER updates on
                                             the aggregated first-layer input, operator slices never formed), mgp_flock_reset_check;
                                      0.3.3: + mgp_actor_fwd_deep / mgp_actor_deep_supported (inference with three or more hidden layers
                                             beyond the one-launch LDS plan: hidden_size 128 at n_layers 3, 4);
                                      0.3.4: no new entry point.  mgp_rollout_steps_ex / _collect at the headline shape run 512-thread
                                             workgroups, two episodes per CU, when B exceeds the device's CU count (same bits; MGP_RO_T512
                                             = 0 / 1 forces the choice); mgp_rollout_supported / _steps_ex / _image cover two and more
                                             hidden layers of up to 128 channels at (N, K) = (100, 3) -- those builds stream weight blocks
                                             from the image every step and return MGP_EUNSUPPORTED without a prebuilt one;
                                             mgp_train_step_p2p: timeout semantics spelled out (partial step; collective rollback is the caller's)
                                      0.4.0: + mgp_sparse_rollout_persistent / _status: mgp_sparse_rollout at K = 3, N <= 1024 runs its T
                                             steps as ONE launch of persistent workgroups (bit-identical outputs; see there) */

#define MGP_OK            0
#define MGP_EINVAL       -1        /* bad size / null pointer / unsupported combination */
#define MGP_EALIGN       -2        /* pointer not 4-byte aligned */
#define MGP_ELAUNCH      -3        /* hipLaunchKernel reported an error */
#define MGP_ENODEV       -4        /* no HIP device / wrong architecture */
#define MGP_EUNSUPPORTED -5        /* valid request the fused kernel does not cover: use the composed ops */

#define MGP_ACT_NONE 0
#define MGP_ACT_TANH 1

#define MGP_MAX_LAYERS 8

int         mgp_version(void);
const char* mgp_strerror(int code);
/* Kernel timing without marker packets: the NEXT mgp_rollout_steps / _steps_ex / _collect launch of the calling thread stamps
 * the given hipEvent_t handles (either may be NULL) with the kernel's own begin and end (hipExtLaunchKernel), then forgets
 * them.  The events must have been created with timing enabled; hipEventElapsedTime(start, stop) after completion is the
 * kernel's duration.  (An event recorded in front of a launch on an idle stream delays that launch by tens of microseconds.) */
int  mgp_set_launch_events(void* start_event, void* stop_event);

/* Text of the HIP error behind the calling thread's most recent MGP_ELAUNCH. */
const char* mgp_last_hip_error(void);
/* Fills name (<= cap bytes) with the device's gcnArchName; returns CU count or a negative code. */
int         mgp_device_info(char* name, int cap);

/* ------------------------------------------------------------------------------------
 * Graph-shift aggregation                        reference learner/actor.py:69-71
 *   Y[b,k,c,n] = sum_m X[b,k,c,m] * G[b,k,m,n]        (torch.matmul(x, delay_gso))
 * G is (B,K,N,N) contiguous.  X and Y are addressed through (b,k,c) strides with n
 * contiguous, so both the input layout (B,K,C,N) and the reference's permuted layout
 * (B,C,K,N) (actor.py:64,71) are served without a copy.
 * ------------------------------------------------------------------------------------ */
int mgp_agg_fwd(const float* X, const float* G, float* Y,
                int B, int K, int C, int N,
                long sxb, long sxk, long sxc,
                long syb, long syk, long syc,
                void* stream);

/* Backward of the aggregation w.r.t. X           autograd of actor.py:70
 *   dX[b,k,c,m] = sum_n dY[b,k,c,n] * G[b,k,m,n]                                   */
int mgp_agg_bwd_x(const float* dY, const float* G, float* dX,
                  int B, int K, int C, int N,
                  long sgb, long sgk, long sgc,      /* dY strides */
                  long sdb, long sdk, long sdc,      /* dX strides */
                  void* stream);

/* ------------------------------------------------------------------------------------
 * Per-agent dense layer = the reference's Conv2d with kernel (step,1), stride (step,1)
 *                                                 reference learner/actor.py:37-38,73-77
 *   out[b,o,t,n] = act( bias[o] + sum_c W[o,c] * in[b,c,t,n] )
 * `in` is addressed by (b,c,t) strides, n contiguous; `out` is (B,Cout,T,N) contiguous.
 * The (k,1) feature filter at layer ind_agg is the case Cin = C*K, T = 1 with
 * W (out, C, K, 1) viewed as (out, C*K).
 * ------------------------------------------------------------------------------------ */
int mgp_dense_fwd(const float* in, const float* W, const float* bias, float* out,
                  int B, int Cin, int Cout, int T, int N,
                  long sib, long sic, long sit,
                  int act, void* stream);

/* Backward of the dense layer                     autograd of actor.py:73-77
 *   delta = dOut * act'(out) ; dW[o,c] = sum delta*in ; db[o] = sum delta ;
 *   dIn[b,c,t,n] = sum_o W[o,c] * delta[b,o,t,n]   (dIn contiguous (B,Cin,T,N); may be NULL)
 * `workspace` must hold mgp_dense_bwd_workspace(...) floats.  Deterministic (no atomics). */
long mgp_dense_bwd_workspace(int B, int Cin, int Cout, int T, int N);
int  mgp_dense_bwd(const float* dOut, const float* out, const float* in, const float* W,
                   float* dW, float* db, float* dIn,
                   int B, int Cin, int Cout, int T, int N,
                   long sib, long sic, long sit,
                   int act, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused Actor forward, aggregation before layer 0 (ind_agg == 0: the only configuration
 * train.py reaches, gnn_dagger.py:43, gnn_cloning.py:41)
 *                                                 reference learner/actor.py:45-86
 *   out (B,1,nA,N) = MLP( X . G )   with X (B,K,F,N), G (B,K,N,N) contiguous.
 * W[i] / b[i]: HOST arrays of n_layers device pointers, W[i] shaped (dims[i+1], dims[i]*(i==0?K:1)),
 * dims = {F, h_1, ..., nA} (n_layers+1 ints).
 * `saved` (may be NULL for inference) receives what mgp_actor_bwd needs:
 *   [ Y (B, F*K, N) | Z_0 (B,h_1,N) | ... | Z_{L-2} (B,h_{L-1},N) ],  mgp_actor_saved_floats() floats.
 * Returns MGP_EUNSUPPORTED for shapes the fused kernel does not cover (caller composes
 * mgp_agg_fwd + mgp_dense_fwd instead).
 * ------------------------------------------------------------------------------------ */
long mgp_actor_saved_floats(const int* dims, int n_layers, int B, int K, int N);
/* 1 if the fused kernel covers (dims, K, N): F <= 8, F*K <= 64, every layer width <= 64, LDS plan fits. */
int  mgp_actor_supported(const int* dims, int n_layers, int K, int N);
int  mgp_actor_fwd(const float* X, const float* G,
                   const float* const* W, const float* const* b,
                   const int* dims, int n_layers,
                   float* out, float* saved,
                   int B, int K, int N, void* stream);

/* The same forward for THREE OR MORE hidden layers of which one is wider than 64 (cfg/hidden_size.cfg:104-106, 128-130:
 * n_layers 3 and 4 at hidden_size 128; reference learner/actor.py:45-86, inference only): one launch -- the aggregation and
 * the first two hidden layers as in mgp_actor_fwd, then every further hidden layer on the same workgroup with the weight image
 * in LDS rebuilt per layer (the activations stay in the matrix accumulators).  mgp_actor_deep_supported: F = 6, nA = 2,
 * 6 K <= 32, hidden widths multiples of 4 up to 128, 16 <= N <= 128, N % 4 == 0.  X, G and W[1 .. n_layers-2] 16-byte aligned. */
int  mgp_actor_deep_supported(const int* dims, int n_layers, int K, int N);
int  mgp_actor_fwd_deep(const float* X, const float* G,
                        const float* const* W, const float* const* b,
                        const int* dims, int n_layers,
                        float* out,
                        int B, int K, int N, void* stream);

/* Backward of the fused Actor forward (parameters only; X and G are leaves without grad
 * in DAGGER, gnn_dagger.py:83-92).  dW[i]/db[i]: HOST arrays of device pointers (overwritten). */
long mgp_actor_bwd_workspace(const int* dims, int n_layers, int B, int K, int N);
int  mgp_actor_bwd(const float* dOut, const float* saved,
                   const float* const* W, const int* dims, int n_layers,
                   float* const* dW, float* const* db,
                   int B, int K, int N, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * Delayed-GSO / delay-line update                 reference learner/state_with_delay.py:44-53
 *   G_next[b,0] = I ; G_next[b,j] = A[b] @ G_prev[b,j-1] (j>=1; 0 if !has_prev)
 *   Xd_next[b,0] = X_t[b] ; Xd_next[b,j] = Xd_prev[b,j-1]   (0 if !has_prev)
 * A (B,N,N), G_* (B,K,N,N), X_t (B,F,N), Xd_* (B,K,F,N).  G_prev[b,0] is the identity by
 * construction (state_with_delay.py:45) and is not read: G_next[b,1] = A[b] exactly.
 * G_prev/Xd_prev may be NULL iff has_prev == 0.  next buffers must not alias prev.
 * ------------------------------------------------------------------------------------ */
int mgp_gso_update(const float* A, const float* G_prev, float* G_next,
                   const float* X_t, const float* Xd_prev, float* Xd_next,
                   int B, int K, int F, int N, int has_prev, void* stream);

/* In-place form of mgp_gso_update for device-resident rollouts: the caller (mgp_flock_step with strided outputs)
 * has ALREADY put A_t into G_next[:,1] and X_t into Xd_next[:,0], and G_next[:,0] holds I from allocation.
 * Computes only G_next[b,j] = G_next[b,1] @ G_prev[b,j-1] for j >= 2 and Xd_next[b,j] = Xd_prev[b,j-1] for j >= 1;
 * with has_prev == 0 (first step of an episode) slices j >= 1 are zeroed instead (state_with_delay.py:44-53 with
 * prev_state=None).  Skips the A read-copy-write and the identity write of mgp_gso_update: 3 N^2 instead of
 * (2K-1) N^2 floats of traffic per episode-step at K = 3. */
int mgp_gso_advance(const float* G_prev, float* G_next, const float* Xd_prev, float* Xd_next,
                    int B, int K, int F, int N, int has_prev, void* stream);

/* curr_gso: powers of the current adjacency      reference learner/state_with_delay.py:38-41
 *   P[b,0] = I ; P[b,j] = A[b] @ P[b,j-1]                                            */
int mgp_gso_powers(const float* A, float* P, int B, int K, int N, void* stream);

/* ------------------------------------------------------------------------------------
 * Flocking simulation (gym_flock is NOT part of the reference tree: own spec FLOCK-SPEC v1,
 * DESIGN.md).  Call sites replaced: env.step gnn_dagger.py:163, env.env.controller
 * gnn_dagger.py:156 / gnn_baseline.py:16, observation tuple state_with_delay.py:22-26.
 * State x is fp64 (B,N,4) = (px,py,vx,vy); all pairwise arithmetic is fp64.
 * ------------------------------------------------------------------------------------ */
typedef struct MgpFlockParams {
    double comm_radius2;   /* squared communication radius                        */
    double dt;             /* integration step                                    */
    double action_gain;    /* step applies clip(u) * action_gain                  */
    double max_accel;      /* |u| clip                                            */
    double ctrl_gain;      /* controller output scale                             */
    double ctrl_clip;      /* controller raw clip                                 */
    double reward_scale;
    int    mean_pooling;   /* network = adj / max(deg,1) if nonzero               */
    int    n_leaders;      /* first n_leaders agents ignore u                     */
    int    centralized;    /* expert written by mgp_flock_step[_advance]: velocity term over ALL agents (the
                            * global teacher DAGGER imitates) if nonzero, else over radius neighbours only */
    unsigned int link_drop;/* link fading (FlockingStochastic-v0): a radius pair is connected iff its 32-bit fade
                            * hash >= link_drop = floor(P(drop) * 2^32); 0 = every radius link is up (no hashing) */
    unsigned int link_seed;/* mixed into the fade hash                            */
    int    reserved_;      /* keeps sizeof a multiple of 8                        */
} MgpFlockParams;

/* x_out (or x itself when x_out is NULL / == x) <- integrate(x, u), then observations of the new state.  With a
 * separate x_out (ping-pong state buffers) an episode is processed by several small workgroups, each integrating
 * the episode redundantly from x -- the fast form for device-resident rollouts; in place it is one workgroup per
 * 128 rows.  u is fp32 with batch stride 2N and element (i,a) at
 * i*su_agent + a*su_axis: (B,N,2) is (2,1); the Actor's output layout (B,1,2,N) is (1,N) -- no transpose
 * kernel between policy and simulator.  u may be NULL (= refresh observations only).
 *   A    (B,N,N) fp32  network matrix           (may be NULL)
 *   A64  (B,N,N) fp64  same, fp64               (may be NULL; gym facade)
 *   feat (B,6,N) fp32  features TRANSPOSED to the (F,N) layout state_with_delay.py:29 builds (may be NULL)
 *   feat64 (B,N,6) fp64 features in the env's own (N,6) layout (may be NULL; gym facade)
 *   reward (B) fp64    -(var vx + var vy) * reward_scale  (may be NULL)
 *   expert (B,N,2) fp32 expert action for the NEW state (may be NULL; centralised or not per p->centralized): a
 *                      closed form of the features (+ episode velocity sums), so the DAGGER label costs no second
 *                      pairwise pass
 *   sAb / sFb          batch strides (elements) of A / feat; 0 = dense (N*N / 6*N).  With sAb = K*N*N and
 *                      A = delay_gso_next + N*N the simulator writes the network matrix straight into slice 1 of
 *                      the next delayed-GSO buffer (feat likewise into delay_state_next[:,0]): see mgp_gso_advance */
int mgp_flock_step(double* x, double* x_out, const float* u, long su_agent, long su_axis,
                   float* A, double* A64, float* feat, double* feat64,
                   double* reward, float* expert, long sAb, long sFb,
                   const MgpFlockParams* p, int B, int N, void* stream);

/* Simulator step AND state transition in one launch (device-resident rollouts): bit-for-bit equal to
 *   mgp_flock_step(x, x_out, u, ..., A = G_next + N*N, sAb = K*N*N, feat = Xd_next, sFb = K*6*N, reward, expert, ...)
 *   followed by mgp_gso_advance(G_prev, G_next, Xd_prev, Xd_next, ...),
 * but the delayed-GSO product takes its neighbour lists from the simulator's membership bits instead of re-reading
 * and compacting the dense network rows, and one kernel boundary disappears.  Covered: x_out != x, u != NULL,
 * N % 4 == 0, N <= 128, K >= 2 (n_states = 6); otherwise MGP_EUNSUPPORTED (use the two calls above). */
int mgp_flock_step_advance(double* x, double* x_out, const float* u, long su_agent, long su_axis,
                           const float* G_prev, float* G_next, const float* Xd_prev, float* Xd_next,
                           double* reward, float* expert, const MgpFlockParams* p,
                           int B, int K, int N, int has_prev, void* stream);

/* Episode-resident closed-loop rollout: T policy steps for B episodes in ONE launch (one workgroup per episode, the
 * episode's state -- agent states, delay line, the neighbour lists of the last K-1 networks, weights -- resident in LDS).
 * Replaces T iterations of the reference's evaluation loop (test_model.py:38-44, gnn_dagger.py:194-203):
 *     u = Actor(delay_state, delay_gso)                    actor.py:45-86, ind_agg = 0       (== mgp_actor_fwd)
 *     x, reward = env.step(u) ; state = State(prev=state)  state_with_delay.py:44-53         (== mgp_flock_step_advance)
 * In place: x (B,N,4) fp64, G (B,K,N,N), Xd (B,K,6,N) hold the state before the call and the state T steps later
 * after it.  Slice 0 of G must be the identity (it is by construction; it is neither read nor written).
 *   action  (B,1,2,N) fp32  the LAST step's policy output (may be NULL)
 *   rewards (B,T)  fp64     reward of every step (may be NULL)
 * Inside the launch no dense operator exists: with G_j(t) = A_t G_{j-1}(t-1) (state_with_delay.py:44-47) tap j of the
 * aggregation is x_{t-j} A_t A_{t-1} ... A_{t-j+1}, evaluated left to right along the neighbour lists of the networks
 * the launch itself produced.  Products that reach back before the launch (its first K-1 steps) end with one dense
 * multiplication by the caller's slice, which may be ANY tensor; on exit the dense slices of the final state are
 * rebuilt the same way.  The simulator arithmetic is the stand-alone kernels' (fp64, bit-exact integration and
 * membership given the action); the aggregation associates differently from mgp_actor_fwd (same 1e-5 parity bound
 * against the reference forward).  Consequently chunking (T1 then T2 steps vs T1 + T2) agrees to fp32 rounding, not bit
 * for bit -- a launch boundary passes through the rounded dense slices -- and a closed loop amplifies that over steps.
 * Coverage (mgp_rollout_supported): dims[0] = 6, dims[n_layers] = 2, layer widths <= 64 (<= 32: a build with half the
 * activation tile), K <= 5, 4 <= N <= 256
 * (N > 128: a variant that keeps the K-1 networks as bit rows; its LDS plan must fit 160 KB: N = 200 with any K, N = 256 up to K = 4).  Anything else: MGP_EUNSUPPORTED -- use the two calls above. */
int mgp_rollout_supported(const int* dims, int n_layers, int K, int N);
int mgp_rollout_steps(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                      const int* dims, int n_layers, float* action, double* rewards,
                      const MgpFlockParams* p, int B, int K, int N, int T, void* stream);

/* The same launch for callers that launch REPEATEDLY (the chunked evaluation loops of gnn_dagger.py:190-232, test_model.py,
 * bench.py): what a launch costs besides its steps is then handed over instead of recomputed.
 *   image   prebuilt weight image (mgp_rollout_image: the MFMA fragment layout the kernel otherwise derives from W, b at
 *           every launch), 16-byte aligned, or NULL (then W, b are read; with an image W and b may be NULL)
 *   carry   B x mgp_rollout_carry_bytes(K, N) bytes: the operator history in FACTORED form -- per episode the membership
 *           bits (row-major, 2 x u64 per row for N <= 128, 4 beyond) and row weights of the last max(K-1, 1) networks
 *           A_t, A_{t-1}, ... (newest first).  An all-zero carry is the history of a reset observation.
 *   flags   MGP_RO_ENTER_CARRY  the history comes from `carry`; the dense slices G[:,1:] are NOT read
 *           MGP_RO_EXIT_CARRY   the history of the final state is written to `carry`
 *           MGP_RO_SKIP_DENSE   G[:,1:] is NOT rebuilt on exit (requires MGP_RO_EXIT_CARRY): the dense operator the
 *                               reference's contract names (state_with_delay.py:44-47) is then materialised on demand by
 *                               mgp_rollout_carry_to_dense -- same arithmetic, same summation order as the in-launch rebuild
 * With ENTER|EXIT the chain x_{t-j} A_t .. A_{t-j+1} never passes through rounded dense products, so ANY chunking of an
 * episode into launches is bit-identical to one launch.  A launch that enters from dense slices can only hand over a
 * complete history if it runs T >= K - 1 steps (else MGP_EINVAL with EXIT_CARRY / SKIP_DENSE).
 * x and Xd (B,K,6,N) are always read on entry and written on exit (they are exact: integration and features are fp64).
 * [0.3.4] Two and more hidden layers of more than 64 (up to 128) channels at (N, K) = (100, 3) stream weight blocks from `image`
 * every step: image == NULL returns MGP_EUNSUPPORTED there (build it once with mgp_rollout_image; mgp_rollout_collect does not
 * cover these shapes).  At the reference's policy shape a launch of more episodes than the device has CUs runs two episodes per
 * CU on 512-thread workgroups -- same results bit for bit. */
#define MGP_RO_ENTER_CARRY 1
#define MGP_RO_EXIT_CARRY 2
#define MGP_RO_SKIP_DENSE 4
int mgp_rollout_steps_ex(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                         const int* dims, int n_layers, float* action, double* rewards,
                         const MgpFlockParams* p, int B, int K, int N, int T, const float* image, void* carry, int flags,
                         void* stream);
/* DAGGER data collection on the same kernel (reference gnn_dagger.py:154-178, batched over B lock-step episodes): every
 * step (i) files the state it starts from as a compact FRAME -- features x_t (6,N) fp32, membership bits of its network
 * A_t (N x 2 u64, N x 4 beyond N = 128), the expert's action for it (2,N) fp32 (the label, :174-176), its age (steps since reset) -- into a ring
 * laid out [ring_steps][B], at ring step (ring_step0 + t) % ring_steps; (ii) is driven by the expert with probability
 * beta[b], else by the policy (:157-161), the coin being the counter-based hash dagger_coin(seed, episode[b], age)
 * (csrc/mgp_device.h; oracle/dagger_vec.py) -- no host RNG, no per-step host traffic.  The K-tap training state of a frame is
 * rebuilt from it and its K-1 predecessors in the ring (same lane) by mgp_replay_gather: 4.8 KB per transition at N = 100
 * instead of the 128 KB of a dense (delay_state, delay_gso) pair.
 * expert_io (B,2,N): in = expert action of the entry state, out = of the final state (chain launches through it).
 * Requires MGP_RO_ENTER_CARRY | MGP_RO_EXIT_CARRY (collection starts at a reset observation -- all-zero carry -- or continues
 * a collecting launch).  Bit rows: 2 x u64 per row for N <= 128, 4 beyond (N <= 256). */
typedef struct MgpCollect {
    float* feat;                 /* [ring_steps][B][6][N] */
    unsigned long long* bits;    /* [ring_steps][B][N][NW], NW = 2 (N <= 128) or 4 */
    float* label;                /* [ring_steps][B][2][N] */
    int* age;                    /* [ring_steps][B]       */
    float* expert_io;            /* (B,2,N) */
    const float* beta;           /* (B) */
    const unsigned int* episode; /* (B) global episode index: the coin stream of the lane */
    unsigned int seed;
    int age0;                    /* age of the entry state (all lanes advance in lock step) */
    int ring_step0;              /* ring step of the entry state's frame */
    int ring_steps;
} MgpCollect;
int mgp_rollout_collect(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                        const int* dims, int n_layers, double* rewards, const MgpFlockParams* p, int B, int K, int N, int T,
                        const float* image, void* carry, int flags, const MgpCollect* collect, void* stream);
/* Minibatch states from the frame ring: for batch element i with frame index r = idx[(cursor ? *cursor : 0) * Bt + i]
 * (frame index = ring_step * lanes + lane): X[i,k] = features of frame r - k*lanes (ring-wrapped) if age[r] >= k else 0;
 * Y[i] = label[r]; G[i,0] = I, G[i,j] = A_t A_{t-1} .. A_{t-j+1} from the bits (row weight 1/max(deg,1) or 1) if age[r] >= j
 * else 0 (state_with_delay.py:44-53 on the stored history).  idx int64 on the device, cursor int32 on the device or NULL.
 * REQUIRES symmetric membership (bit m of row n == bit n of row m): row i of the product is accumulated along bit ROWS,
 * reading row n of the stored bits as column n of A -- true for every network of FLOCK-SPEC v1 (radius test; link fading
 * hashes the unordered pair), wrong for a directed network (e.g. k nearest neighbours), which must not be filed as frames. */
int mgp_replay_gather(const float* feat, const unsigned long long* bits, const float* label, const int* age,
                      const long* idx, const int* cursor, int Bt, int lanes, int ring_steps, int K, int N, int mean_pooling,
                      float* X, float* G, float* Y, void* stream);
/* The same for `nb` consecutive minibatches in one launch: sample s < nb * Bt uses idx[(cursor ? *cursor : 0) * Bt + s] and
 * lands in X / G / Y slot s (buffers of nb * Bt samples) -- the gathers of a whole graph of updates do not depend on the
 * weights, so they need not sit between the updates. */
int mgp_replay_gather_many(const float* feat, const unsigned long long* bits, const float* label, const int* age,
                           const long* idx, const int* cursor, int Bt, int nb, int lanes, int ring_steps, int K, int N,
                           int mean_pooling, float* X, float* G, float* Y, void* stream);
long mgp_rollout_image_floats(const int* dims, int n_layers, int K, int N);      /* 0: shape not covered */
int mgp_rollout_image(const float* const* W, const float* const* b, const int* dims, int n_layers, int K, int N,
                      float* image, void* stream);
/* CHECKER build of the same kernels (csrc/rollout_f32ref.hip): hidden layers of <= 32 input channels on fp32 MFMA 16x16x4 -- a
 * k-ordered fp32 fmaf chain, the arithmetic the product build's split-bf16 layers (three bf16 pieces per operand, six of
 * the nine cross products) stand in for.  Same arguments, same state layout, same carry; layer widths <= 32 only (no
 * forwarding to the wide builds); `image` must come from mgp_rollout_f32ref_image (fp32 fragments), or be NULL.  Used by
 * tests/test_gpu_headline_parity.py to hold the product build to it on the same episodes; not a fast path. */
int mgp_rollout_f32ref_supported(const int* dims, int n_layers, int K, int N);
int mgp_rollout_f32ref_steps_ex(double* x, float* G, float* Xd, const float* const* W, const float* const* b,
                                const int* dims, int n_layers, float* action, double* rewards,
                                const MgpFlockParams* p, int B, int K, int N, int T, const float* image, void* carry,
                                int flags, void* stream);
long mgp_rollout_f32ref_image_floats(const int* dims, int n_layers, int K, int N);
int mgp_rollout_f32ref_image(const float* const* W, const float* const* b, const int* dims, int n_layers, int K, int N,
                             float* image, void* stream);
long mgp_rollout_carry_bytes(int K, int N);                                       /* per episode; 0: shape not covered */
/* G[:,j] = A_t A_{t-1} .. A_{t-j+1} for j = 1..K-1 from a carry (slice 0, the identity, is not touched). */
int mgp_rollout_carry_to_dense(const void* carry, float* G, int B, int K, int N, void* stream);

/* Acceptance statistics of M reset candidates (FLOCK-SPEC v1 section 3; gym_flock's reset loop -- draw until the flock is
 * connected enough and nobody overlaps -- is host control flow in the reference): pos (M,N,2) fp64 positions ->
 * min_degree[m] = min over agents of the number of others within the radius, r2_min[m] = smallest squared pair distance,
 * both exactly what numpy's fp64 evaluation gives (r2 = dx dx + dy dy, unfused).  The caller keeps the RNG stream, the draw
 * order and the accept rule (envs/flocking.py::sample_initial_states); N <= 8192. */
int mgp_flock_reset_check(const double* pos, int M, int N, double comm_radius2, int* min_degree, double* r2_min, void* stream);
/* Expert controller on the current x: u (B,N,2) fp32 and/or u64 (B,N,2) fp64 (either may be NULL). */
int mgp_flock_controller(const double* x, float* u, double* u64, const MgpFlockParams* p,
                         int centralized, int B, int N, void* stream);

/* ------------------------------------------------------------------------------------
 * DAGGER update helpers                            reference learner/gnn_dagger.py:91-93
 *   mgp_mse_grad : loss[0] = mean((pred-target)^2) ; dPred = 2 (pred-target) / n
 *   mgp_adam_step: torch.optim.Adam defaults on one flat fp32 buffer (step = 1-based count)
 * ------------------------------------------------------------------------------------ */
int mgp_mse_grad(const float* pred, const float* target, float* dPred, float* loss,
                 long n, void* stream);
int mgp_adam_step(float* param, const float* grad, float* m, float* v, long n,
                  float lr, float beta1, float beta2, float eps, int step, void* stream);
/* Same update with the 0-based step counter resident on the device (*step_dev is read, then incremented by a
 * trailing 1-thread kernel): no per-step host scalars, so a whole DAGGER update replays from one HIP graph. */
int mgp_adam_step_dev(float* param, const float* grad, float* m, float* v, long n,
                      float lr, float beta1, float beta2, float eps, int* step_dev, void* stream);

/* ------------------------------------------------------------------------------------
 * One DAGGER update in two launches          reference learner/gnn_dagger.py:85-93
 *   pred = actor(X, G) ; loss = mse_loss(pred, target) ; loss.backward() ; Adam.step()
 * for ind_agg = 0 (the only configuration train.py builds, gnn_dagger.py:43).  Forward, MSE gradient and the
 * parameter backward of 16 agent columns run inside one workgroup (activations never leave LDS); a second launch adds
 * the per-workgroup partials in a fixed order.  X (B,K,F,N), G (B,K,N,N), target (B,1,nA,N), all fp32 contiguous.
 *   flat_grad : dW_0 | db_0 | dW_1 | db_1 | ...  (the order torch enumerates Actor.parameters()), overwritten
 *   loss      : loss[0] = mean squared error (may be NULL)
 *   workspace : mgp_train_workspace(...) floats; zero it once after allocation (its last word is a ticket counter
 *               that mgp_train_step leaves at zero)
 * mgp_train_grads stops at the gradient (data-parallel runs all-reduce it, then call mgp_adam_step*).
 * mgp_train_step also applies torch.optim.Adam (defaults) in the second launch: flat_param holds the parameters in the
 * flat_grad order and is what the forward reads; *step_dev (0-based) is read, then incremented -- HIP-graph replayable.
 * Coverage (mgp_train_supported): F <= 8, F*K <= 64, widths <= 64, B * ceil(N/16) <= 8192, LDS plan (X[b], all parameters, 16-column activations) <= 96 KB;
 * otherwise MGP_EUNSUPPORTED -- compose mgp_actor_fwd / mgp_mse_grad / mgp_actor_bwd / mgp_adam_step instead. */
int  mgp_train_supported(const int* dims, int n_layers, int B, int K, int N);
long mgp_train_workspace(const int* dims, int n_layers, int B, int K, int N);
int  mgp_train_grads(const float* X, const float* G, const float* target,
                     const float* const* W, const float* const* b, const int* dims, int n_layers,
                     float* flat_grad, float* loss, float* workspace, int B, int K, int N, void* stream);
int  mgp_train_step(const float* X, const float* G, const float* target,
                    float* flat_param, float* flat_grad, float* m, float* v, const int* dims, int n_layers,
                    float lr, float beta1, float beta2, float eps, int* step_dev,
                    float* loss, float* workspace, int B, int K, int N, void* stream);

/* mgp_train_step with the minibatch gathered inside the kernel and no per-update host work (reference
 * replay_buffer.py:40 + gnn_dagger.py:83-93: sample, cat, update): batch item b of update u reads row
 * idx[u * B + b] of the replay arrays Xr (cap,K,F,N) / Gr (cap,K,N,N) / Yr (cap,1,nA,N), where u = *cursor is a device
 * counter the call advances together with *step_dev; the loss of update u lands in loss_hist[u % hist_cap].  The host
 * uploads a whole round's index table once, zeroes *cursor and replays one HIP graph per update. */
int  mgp_train_step_indexed(const float* Xr, const float* Gr, const float* Yr, const long* idx, int* cursor,
                            float* loss_hist, int hist_cap, float* flat_param, float* flat_grad, float* m, float* v,
                            const int* dims, int n_layers, float lr, float beta1, float beta2, float eps,
                            int* step_dev, float* workspace, int B, int K, int N, void* stream);

/* ------------------------------------------------------------------------------------
 * Data-parallel DAGGER update: the one-shot gradient exchange
 * The reference trains on one device (train.py:31; gradient_step: gnn_dagger.py:76-96).  Sharding its episodes over the
 * GPUs of a node leaves exactly one exchange per update: the flat gradient (1,730 floats at cfg/dagger.cfg) plus the
 * loss -- 6.9 KB, pure latency.  Instead of a ring all-reduce (2(W-1) dependent hops) every rank pushes its values into a
 * mailbox in every peer's memory (hipIpc-mapped; xGMI is point to point) as 64-bit {sequence number | fp32} packets, one
 * 8-byte system-scope store each, and adds what arrives in rank order 0..W-1, then divides by W: one hop, no fence, and
 * bit-identical results on every rank.  Two slots per source make back-to-back exchanges safe without a barrier.
 *   mgp_p2p_create         allocates the local mailbox (world x 2 x n_floats packets of uncached device memory) -- the one
 *                          entry point of this library that allocates; world <= 8, one communicator per process and device
 *   mgp_p2p_handle         copies out the mailbox's IPC handle (mgp_p2p_handle_bytes() = 64 bytes); the caller gathers the
 *                          handles of all ranks, in rank order, by any means (torch.distributed.all_gather here)
 *   mgp_p2p_connect        opens every peer's mailbox from the gathered handles (world x 64 bytes)
 *   mgp_p2p_allreduce_mean buf[i] <- mean over ranks of buf[i], i < n <= n_floats, in place, one launch on `stream`;
 *                          every rank must make the same sequence of exchange calls (this and mgp_train_step_p2p)
 *   mgp_p2p_status         synchronises `stream`; *status != 0 if a poll gave up (a peer did not publish within the timeout,
 *                          default 5 s: its values counted as 0) -- an error for the caller to raise, never a hung GPU;
 *                          *seq = number of exchanges completed
 *   mgp_p2p_info           *mem_kind: 2 = uncached, 1 = fine-grained, 0 = plain device memory (what the runtime granted)
 * HIP-graph capturable (the sequence number lives on the device).  Ranks may share one device (IPC between processes). */
typedef struct MgpP2P MgpP2P;
int  mgp_p2p_create(int world, int rank, int n_floats, MgpP2P** out);
int  mgp_p2p_handle_bytes(void);
int  mgp_p2p_handle(const MgpP2P* comm, void* handle_out);
int  mgp_p2p_connect(MgpP2P* comm, const void* handles);
int  mgp_p2p_set_timeout_ms(MgpP2P* comm, int ms);
int  mgp_p2p_info(const MgpP2P* comm, int* world, int* rank, int* n_floats, int* mem_kind);
int  mgp_p2p_status(MgpP2P* comm, int* status, int* seq, void* stream);
int  mgp_p2p_destroy(MgpP2P* comm);
int  mgp_p2p_allreduce_mean(MgpP2P* comm, float* buf, int n, void* stream);

/* mgp_train_step / mgp_train_step_indexed for a data-parallel run (gnn_dagger.py:85-93 on a world-times larger minibatch):
 * the same two launches, with the exchange above between the local reduction and Adam inside the second one -- each of its
 * workgroups exchanges the 64 gradient entries it has just reduced.  flat_grad receives the averaged gradient, loss[0] /
 * loss_hist the averaged loss; every rank applies the identical step.  idx, cursor and loss_hist: all NULL (mgp_train_step
 * semantics) or all given (mgp_train_step_indexed semantics).  comm: n_floats >= parameter count + 1.
 * Timeout semantics: a workgroup whose poll gives up sets the communicator's sticky status word and skips Adam for ITS 64
 * entries; workgroups that see the word set skip too; entries whose packets had arrived in time ARE stepped -- so after a
 * timed-out exchange the weights are partially stepped on that rank and fully stepped on a peer that merely arrived late.
 * The caller restores the round's starting point on EVERY rank (DAGGER.end_updates: MAX all-reduce of the status, rollback of
 * weights / moments / step counter, raise) and re-creates the communicator before the next round. */
int  mgp_train_step_p2p(const float* X, const float* G, const float* target, const long* idx, int* cursor,
                        float* loss_hist, int hist_cap, float* flat_param, float* flat_grad, float* m, float* v,
                        const int* dims, int n_layers, float lr, float beta1, float beta2, float eps,
                        int* step_dev, float* loss, float* workspace, int B, int K, int N, MgpP2P* comm, void* stream);
/* mgp_adam_step_dev + the bookkeeping of an indexed round (loss_hist[*cursor % hist_cap] = *loss; *cursor += 1): the tail
 * of a data-parallel update whose gradient went through a library collective after mgp_train_grads. */
int  mgp_adam_step_filed(float* param, const float* grad, float* m, float* v, long n,
                         float lr, float beta1, float beta2, float eps, int* step_dev,
                         const float* loss, float* loss_hist, int hist_cap, int* cursor, void* stream);

/* ------------------------------------------------------------------------------------
 * Factored state for episodes beyond the LDS-resident rollout (N > 256)
 * Same loop as mgp_rollout_steps (test_model.py:38-44), same mathematics -- tap j = x_{t-j} A_t ... A_{t-j+1} evaluated
 * left to right -- with the state in HBM/L2 instead of LDS and the networks as membership BIT ROWS:
 *   bits (B,H,N,NW) u64   rows of the last H = max(K-1, 1) networks, ring over time, NW = mgp_sparse_words(N)
 *   wrow (B,H,N)    f32   row weights (1/deg or 1);  a network that does not exist yet = zero bits, zero weights
 *   feat (B,K,N,8)  f32   features x_t .. x_{t-K+1} as (N,8) rows (6 used), ring over time; missing history = zeros
 * A step never touches a dense N x N operator (the dense path moves 12 MB per episode and step at N = 1000).
 *   mgp_flock_step_sparse   simulator step (FLOCK-SPEC, the arithmetic of mgp_flock_step): x -> x_out, then bit rows, row
 *                           weights and (N,8) features of the new state into the ring slots the caller points at
 *                           (batch strides sBb / sWb / sTb in words / floats / floats); u == NULL: observe x itself
 *   mgp_sparse_policy_image weights -> MFMA fragment image (mgp_sparse_policy_image_floats floats), once per policy
 *   mgp_sparse_policy_step  action (B,1,2,N) from the factored state: K-2 gather launches + one policy launch
 *   mgp_sparse_to_dense     G (B,K,N,N) slices 1..K-1 of the reference contract from the bit rows (on demand)
 * Coverage: dims[0] = 6, dims[n_layers] = 2, widths and 6K <= 32, K <= 5, N <= 4096. */
int  mgp_sparse_words(int N);
int  mgp_flock_step_sparse(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                           unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                           double* reward, float* expert, const MgpFlockParams* p, int B, int N, void* stream);
/* Same step, same outputs, with a cell list instead of the all-pairs sweep (cells no smaller than the radius, the spec's
 * fp64 membership test on every candidate: identical bit rows; N <= 2048, else MGP_EUNSUPPORTED). */
int  mgp_flock_step_cells(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                          unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                          double* reward, float* expert, const MgpFlockParams* p, int B, int N, void* stream);
int  mgp_sparse_policy_supported(const int* dims, int n_layers, int K, int N);
long mgp_sparse_policy_image_floats(const int* dims, int n_layers, int K);
int  mgp_sparse_policy_image(const float* const* W, const float* const* b, const int* dims, int n_layers, int K,
                             float* image, void* stream);
int  mgp_sparse_policy_step(const unsigned long long* bits, const float* wrow, const float* feat, const float* image,
                            const int* dims, int n_layers, float* scratch, float* action,
                            int B, int K, int N, int cur, int hs, void* stream);
int  mgp_sparse_to_dense(const unsigned long long* bits, const float* wrow, float* G, int B, int K, int N, int hs,
                         void* stream);
/* mgp_sparse_policy_step has two forms of its gather / policy launches: source rows staged in the LDS (default wherever an
 * episode's rows fit: N <= ~2400) or gathered straight from global memory.  Test hook (process-wide): mode 1 keeps the second
 * form everywhere, mode 2 the staged form on bit rows only (ignores the neighbour lists of mgp_sparse_rollout), 0 restores the
 * default; returns the previous mode. */
int  mgp_sparse_force_direct(int mode);
/* DAGGER data collection on the factored state (reference gnn_dagger.py:154-178 per lane; the semantics of
 * mgp_rollout_collect for N > 256): mgp_sparse_policy_step that ALSO files the frame of the state the step starts from at
 * ring step `ring_step` of a ring laid out [ring_steps][B] -- features x_t (6,N), bit rows (N x mgp_sparse_words(N) u64) and
 * row weights (N) of its network A_t, the expert's action for it (2,N: the label, :174-176), its age -- and writes into
 * `action` what drives the step: the expert's action where dagger_coin(seed, episode[b], age_now) < floor(beta[b] 2^32)
 * (:157-161; csrc/mgp_device.h, oracle/dagger_vec.py), else the policy's.  `expert` (B,N,2) is the by-product of the simulator
 * kernel that produced the current state (mgp_flock_step_cells / _sparse).  One env step stays K launches. */
typedef struct MgpSparseCollect {
    float* feat;                 /* [ring_steps][B][6][N] */
    unsigned long long* bits;    /* [ring_steps][B][N][NW] */
    float* wrow;                 /* [ring_steps][B][N] */
    float* label;                /* [ring_steps][B][2][N] */
    int* age;                    /* [ring_steps][B] */
    const float* expert;         /* (B,N,2) */
    const float* beta;           /* (B) */
    const unsigned int* episode; /* (B) */
    unsigned int seed;
    int age_now;                 /* steps since the reset (all lanes in lock step) */
    int ring_step;
    int ring_steps;
} MgpSparseCollect;
int  mgp_sparse_policy_collect(const unsigned long long* bits, const float* wrow, const float* feat, const float* image,
                               const int* dims, int n_layers, float* scratch, float* action,
                               int B, int K, int N, int cur, int hs, const MgpSparseCollect* collect, void* stream);
/* T closed-loop steps on the factored state enqueued by ONE call (reference test_model.py:38-44 / gnn_dagger.py:154-178 for
 * every lane): per step mgp_sparse_policy_step (or _collect when `collect` is given), then the simulator (cell list up to
 * N = 2048, all pairs beyond) into the next ring slots.  x_a holds the state on entry; x_a / x_b ping-pong: the final state
 * is in x_b if T is odd, else x_a.  rewards (T,B) fp64 or NULL; expert (B,N,2) or NULL (required with collect: the label /
 * the expert's action of every visited state).  collect->ring_step / age_now describe the FIRST step and advance per step.
 * *cur / *hs: ring slots of x_t / A_t on entry, of the final state on return. */
int  mgp_sparse_rollout(unsigned long long* bits, float* wrow, float* feat, const float* image, const int* dims,
                        int n_layers, float* scratch, float* action, double* x_a, double* x_b, double* rewards,
                        float* expert, const MgpFlockParams* p, int B, int K, int N, int T, int* cur, int* hs,
                        const MgpSparseCollect* collect, unsigned short* nbr, void* stream);
/* Persistent form (csrc/sparse_persist.hip; replaces the per-step launches of gnn_dagger.py:154-165 / test_model.py:38-44 for
 * N > 256): where mgp_sparse_rollout_persistent(...) = 1 -- K = 3, N <= 1024, <= 4 layers -- and the call has
 * neighbour lists (policy rollouts and DAGGER collection alike), the T steps run as ONE launch of workgroups that stay resident, keep the episode's feature
 * rows / row weights / own list rows in LDS and hand each other only what a sibling lacks (write-through stores, one arrival
 * counter per exchange) through the same state buffers: every output is bit-identical to the K-launch form; bit rows, list
 * rows and x are written for the networks / the state that outlive the call.  A launch holds (CUs / workgroups per episode) episodes
 * (more: further launches); a workgroup whose siblings do not arrive within 3 s (CUs held by another process) gives up: the
 * episode's action / state / rewards are NaN and mgp_sparse_rollout_status -- which synchronises `stream` -- returns
 * MGP_ELAUNCH (MGP_OK otherwise; meaningful after a call that ran the persistent form).  The persistent form uses
 * scratch[0 .. B N 8) and B * 16 words behind it.  Environment: MGP_SP_PERSIST=0 keeps the K-launch form;
 * MGP_SP_PERSIST_TIMEOUT_MS (default 3000) bounds every wait; MGP_SP_PERSIST_NEAR=0 keeps write-through stores in the sibling
 * exchanges even where an episode's workgroups found themselves on one XCD (default: plain stores there, served by that XCD's
 * L2 -- same bits, ~4 % faster); MGP_SP_PERSIST_TILES=<n> forces the workgroups per episode (default: ceil(N / 256), more when
 * the call has fewer episodes than the device CUs); MGP_SP_PERSIST_FAULT=<episode> is a test hook (that episode's second
 * workgroup stops arriving). */
int  mgp_sparse_rollout_persistent(const int* dims, int n_layers, int K, int N, const MgpFlockParams* p);
int  mgp_sparse_rollout_status(const float* scratch, int B, int K, int N, void* stream);
/* nbr (B,H,N,16) u16 or NULL: the networks of the bit-row ring once more as compact neighbour LISTS, kept in step with it by
 * the cell-list simulator (mgp_flock_step_cells_nbr; N <= 2048) and read by the gather / policy launches instead of the bit rows
 * (32 instead of 128 bytes per row to request at N = 1000, entries dealt evenly over a column's four lanes).  Row layout: up
 * to 15 neighbour indices, entry e at position (e & 3) * 4 + (e >> 2); position 15 = the count, 0xFFFF = use the bit row.
 * The ring slot of the current network must have been written by mgp_flock_step_cells_nbr (e.g. the reset observation). */
int  mgp_flock_step_cells_nbr(const double* x, double* x_out, const float* u, long su_agent, long su_axis,
                              unsigned long long* bits, long sBb, float* wrow, long sWb, float* featT, long sTb,
                              unsigned short* nbr, long sNb, double* reward, float* expert, const MgpFlockParams* p,
                              int B, int N, void* stream);
/* mgp_replay_gather_many for frames filed by mgp_sparse_policy_collect (N > 256: NW = mgp_sparse_words(N) words per bit
 * row, row weights stored with the frame): same outputs, the products e_i A_t A_{t-1} .. evaluated row by row from HBM. */
int  mgp_replay_gather_rows(const float* feat, const unsigned long long* bits, const float* wrow, const float* label,
                            const int* age, const long* idx, const int* cursor, int Bt, int nb, int lanes, int ring_steps,
                            int K, int N, float* X, float* G, float* Y, void* stream);

/* ---- DAGGER updates without the dense operator slices -------------------------------------------------------------------
 * The reference's update (gnn_dagger.py:76-96) evaluates actor(delay_state, delay_gso): actor.py:64-75 multiplies tap k of the
 * delay line by slice k of delay_gso (K N^2 floats per sample: 12 MB at N = 1000) and concatenates the K results into the
 * (F K, N) input of the first filter layer.  Parameters are the only leaves, so nothing flows back through that product: the
 * update needs only its RESULT, and the frame ring holds every factor of every slice as bit rows.
 *   mgp_replay_aggregate   Z[s, f K + k, n] = (x_{t-k} . A_t A_{t-1} .. A_{t-k+1})[f, n] for nb minibatches of Bt frames (same
 *                          indexing, history rule -- zero for k > age -- and symmetric-membership requirement as
 *                          mgp_replay_gather_many / _rows), evaluated left to right as k sparse products along the bit rows,
 *                          set bits in ascending order; Y = the labels.  Z (nb Bt, 6 K, N), Y (nb Bt, 2, N) fp32.  Rows of Z are
 *                          in the column order of the first layer's weight (conv_layers.0.weight viewed (out, F K)).
 *                          wrow: the row weights stored with the frames of the factored path (required for N > 256, where
 *                          NW = mgp_sparse_words(N) words per bit row); N <= 256: 2 (N <= 128) or 4 words per row, weights
 *                          (float)(1 / max(deg, 1)) from the row populations (mean_pooling) or 1, wrow ignored.  N <= 2048.
 *   mgp_train_grads_agg    mgp_train_grads on Z instead of (X, G)
 *   mgp_train_step_agg     mgp_train_step / _indexed / _p2p on Z: idx, cursor, loss_hist all NULL or all given; comm NULL or
 *                          the exchange; loss may be NULL
 *   mgp_train_agg_supported  shapes the two above cover (the dense forms also hold K F N floats of X per tile in LDS; these
 *                          do not: any N).  Workspace: mgp_train_workspace. */
int  mgp_replay_aggregate(const float* feat, const unsigned long long* bits, const float* wrow, const float* label,
                          const int* age, const long* idx, const int* cursor, int Bt, int nb, int lanes, int ring_steps,
                          int K, int N, int mean_pooling, float* Z, float* Y, void* stream);
int  mgp_train_agg_supported(const int* dims, int n_layers, int B, int K, int N);
int  mgp_train_grads_agg(const float* Z, const float* target, const float* const* W, const float* const* b,
                         const int* dims, int n_layers, float* flat_grad, float* loss, float* workspace, int B, int K, int N,
                         void* stream);
int  mgp_train_step_agg(const float* Z, const float* target, const long* idx, int* cursor, float* loss_hist, int hist_cap,
                        float* flat_param, float* flat_grad, float* m, float* v, const int* dims, int n_layers,
                        float lr, float beta1, float beta2, float eps, int* step_dev, float* loss, float* workspace,
                        int B, int K, int N, MgpP2P* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGP_H */
