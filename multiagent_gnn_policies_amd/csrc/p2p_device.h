// One-shot gradient exchange between the ranks of one node (device side; host side in p2p.hip).
//
// The only exchange of a data-parallel DAGGER run is the 1,730-float gradient (+ the loss): 6.9 KB per update, pure
// latency.  A ring all-reduce pays 2(W-1) dependent hops for it; here every rank PUSHES its values straight into a
// mailbox in every peer's memory (IPC-mapped, xGMI is point to point) and sums what arrives, in rank order:
//
//   packet   = 64-bit word {sequence number : 32 | fp32 payload : 32}, written with ONE 8-byte system-scope store -- the
//              payload can never be seen without its sequence number, so there is no separate flag, fence or second hop
//   mailbox  = [source rank][slot = seq & 1][entry] packets in the RECEIVER's memory; entry i of source q is polled by the
//              receiver's thread i only
//   sum      = payloads added in rank order 0..W-1 (own value in its place), then divided by W: every rank computes
//              bit-identical results
//   slots    = 2: a rank can start exchange s+1 only after every peer published s+1... which a peer's stream does only
//              after its exchange-s kernel retired, so slot (s & 1) is never overwritten while somebody still reads it
// A poll gives up after `timeout` ticks of the 100 MHz wall clock (sets *status, contributes 0): a missing peer is an
// error the host reports, never a hung GPU.
#pragma once
#include <hip/hip_runtime.h>

#define MGP_P2P_MAX_WORLD 8

struct P2PDev {
    unsigned long long* box[MGP_P2P_MAX_WORLD];   // box[q] = mailbox in rank q's memory (box[rank] is local)
    int world, rank, n;                           // n = entries per (source, slot)
    int* ctl;                                     // local control words: [0] sequence number of the last exchange,
                                                  //                      [1] ticket, [2] status (bit 0: a poll timed out)
    long long timeout;                            // wall-clock ticks (100 MHz)
};

// host-side communicator behind the opaque MgpP2P* of include/mgp.h
struct MgpP2P {
    P2PDev dev;
    void* local;                 // [world][2][n] packets, fine-grained / uncached device memory
    int* ctl;                    // 4 ints, ordinary device memory of this rank
    size_t bytes;
    int mem_kind;                // 2 uncached, 1 fine-grained, 0 plain hipMalloc
    int connected;
    void* opened[MGP_P2P_MAX_WORLD];
    hipIpcMemHandle_t handle;
};

// `late` (optional): set to true when this entry's poll gave up -- the caller must not apply the result as a gradient.
__device__ __forceinline__ float p2p_exchange_mean(const P2PDev& X, int i, float mine, unsigned seq, bool* late = nullptr)
{
    const int slot = (int)(seq & 1u);
    const unsigned long long pkt = ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(mine);
    const size_t out = ((size_t)(X.rank * 2 + slot)) * (size_t)X.n + (size_t)i;
#pragma unroll 1
    for (int d = 1; d < X.world; ++d) {                       // start with the next rank: spreads the first packets over the links
        const int q = (X.rank + d) % X.world;
        __hip_atomic_store(X.box[q] + out, pkt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // all W-1 incoming packets are requested together (independent loads: one memory latency per poll round, not W-1)
    const unsigned long long* in = X.box[X.rank] + (size_t)slot * (size_t)X.n + (size_t)i;
    const size_t per_src = (size_t)2 * (size_t)X.n;
    const long long t0 = wall_clock64();
    unsigned long long p[MGP_P2P_MAX_WORLD];
    for (;;) {
        bool all = true;
#pragma unroll
        for (int q = 0; q < MGP_P2P_MAX_WORLD; ++q)          // unconditional loads (a load under a branch is waited for at
            p[q] = __hip_atomic_load(in + (size_t)min(q, X.world - 1) * per_src,   // the join): clamped address, result ignored
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int q = 0; q < MGP_P2P_MAX_WORLD; ++q)
            if (q < X.world && q != X.rank) all = all && ((unsigned)(p[q] >> 32) == seq);
        if (all) break;
        if (wall_clock64() - t0 > X.timeout) {
            atomicOr(X.ctl + 2, 1);
            if (late != nullptr) *late = true;
#pragma unroll
            for (int q = 0; q < MGP_P2P_MAX_WORLD; ++q)
                if ((unsigned)(p[q] >> 32) != seq) p[q] = 0ull;   // late peers contribute 0
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < MGP_P2P_MAX_WORLD; ++q)
        if (q < X.world) sum += (q == X.rank) ? mine : __uint_as_float((unsigned)(p[q] & 0xFFFFFFFFull));
    return sum / (float)X.world;
}
