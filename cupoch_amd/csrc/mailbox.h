// mailbox.h -- the per-iteration exchange of a sharded registration loop (N ranks, one node) without a
// collective launch: 32 doubles per rank through a MAILBOX in host memory that every rank's GPU maps.
//
// Why.  With the source sharded over 8 GPUs an iteration's kernels take ~40 us per rank (search 15,
// reduction 24 at 1.25M points); an ncclAllReduce of 256 bytes between reduction and step adds a
// launch, ~20 us of collective latency and the step kernel behind it -- about as much again.  The
// exchange itself is tiny and all-to-all, so it is done where the sums appear: the reduction's
// finishing block posts its rank's 32 sums, waits for the other ranks' posts, adds all of them in
// RANK ORDER (every rank gets the same bits) and goes straight on to the loop's step.  An iteration
// stays two launches, as on one GPU.
//
// The box lives in POSIX shared memory (one node), registered with HIP by every rank
// (hipHostRegister): host memory is coherent for all GPUs at system scope, needs no peer mapping and no
// IPC handles.  A post is 64 eight-byte words, each one SELF-VALIDATING: {exchange number << 32 | half of
// a sum's bits}.  An aligned 8-byte store lands whole, so a reader that finds this exchange's number in
// every word of a rank's post has that rank's sums -- no flag behind a release, no fence, and ONE read
// round trip over the link per poll for all ranks at once (the first version -- sums, system-scope
// release, flag; readers: poll the flags, acquire, then read the sums rank after rank -- took three
// dependent trips plus one per rank).  Slots alternate with the exchange number's parity, so a rank that
// is already one exchange ahead cannot overwrite what a slower rank still has to read (to get two ahead
// it needs everybody's post of the exchange in between).  Exchange numbers count a rank's exchanges since
// the box was made (32 bits, wrapping past 0 -- a word's zeroed state -- with the parity kept: only the two
// most recent numbers can ever be in a slot); every rank performs the same
// exchanges (the loop's control flow depends only on the all-reduced sums, which are identical everywhere).
// DEVICE INBOXES (MI_ICP_MAILBOX=device; set up through the box above, mi_comm.hip: mailbox_open).  Host memory costs
// every poll a trip over PCIe.  With an inbox per rank in fine-grained DEVICE memory, opened by every other rank
// through HIP IPC, a rank stores its 64 words into all inboxes -- over xGMI between GPUs, posted writes -- and polls
// its OWN inbox, i.e. local memory.  Same words, same slots, same numbers.  Exercised here by several processes on
// one GPU; not the default until it has been measured across GPUs.
// A rank that does not hear from a peer within ~10 s gives up: the loop is marked failed and the host
// call returns MI_ICP_ERR_COMM (bench.py then falls back to the RCCL path).
#pragma once
#include "device_utils.h"

namespace mi {

constexpr int kMailRanks = 16;
constexpr uint32_t kMailSpinLimit = 8u << 20;  // polls of a flag in host memory (~1-2 us each); MI_ICP_MAIL_SPIN_LIMIT overrides

struct MailBox {
    uint32_t ready;                      // set by rank 0 once the box is zeroed
    uint32_t nranks;
    uint32_t attached;                   // ranks > 0 that have mapped and registered the box
    uint32_t go;                         // set by rank 0 when all have: the box is in use (a box found with go != 0 is another job's)
    uint32_t device_mode;                // rank 0's wish: the posts travel GPU to GPU (device inboxes, below) instead of through `words`
    uint32_t device_failed;              // set by any rank that could not set its part of that up: everybody stays with `words`
    uint32_t use_device;                 // rank 0's MI_ICP_MAILBOX=device: every rank USES the inboxes unmeasured (set before `ready`; the peers' own environment does not matter)
    uint32_t pad_[9];
    // device mode: every rank's inbox (fine-grained device memory, [slot][rank][64] words like `words`) as an IPC handle,
    // and the ranks' progress through the set-up (1: handle published, 2: every peer's inbox opened) and the tear-down
    hipIpcMemHandle_t inbox[kMailRanks];
    uint32_t inbox_state[kMailRanks];
    uint32_t inbox_closed[kMailRanks];
    // host-side agreement between the ranks' CPUs (mi_icp_comm_autotune: each rank's measured latencies, all read all;
    // two slots by the gather number's parity, like `words`)
    uint32_t tune_epoch[kMailRanks];
    double tune_val[2][kMailRanks][4];
    unsigned long long words[2][kMailRanks][64];  // [slot][rank]: exchange << 32 | low / high half of sum k's bits at 2k / 2k + 1
};
constexpr size_t kMailInboxWords = (size_t)2 * kMailRanks * 64;

struct MailArgs {
    MailBox* box;       // device address of the registered host mapping; null: no mailbox
    uint32_t* seq_dev;  // this rank's exchange counter (device memory, zeroed with the box)
    int rank, nranks;
    uint32_t spin_limit;
    // device mode (else null): this rank's inbox and the device addresses of all ranks' inboxes (its own included)
    unsigned long long* inbox;
    unsigned long long* const* peers;
};

// One workgroup of whole waves (>= 64 threads).  sys: this rank's 32 sums (global or LDS, written before a
// barrier); on return (all threads past a barrier) it holds the ranks' totals, added in rank order.
// s_tmp: two LDS words.  False: a peer did not post in time.
// pre_seq: the value of *m.seq_dev, if the caller has read it already (saves the exchange a trip to L2).
__device__ __forceinline__ bool mail_allreduce(const MailArgs& m, double* sys, uint32_t* s_tmp,
                                               const uint32_t* pre_seq = nullptr) {
    __shared__ unsigned long long s_words[kMailRanks][64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (int)(blockDim.x >> 6);
    if (tid == 0) {
        uint32_t s = (pre_seq ? *pre_seq : *m.seq_dev) + 1u;
        if (s == 0u) s = 2u;  // (after 2^32 exchanges: 0 is a word's zeroed state, and the slots' parity keeps alternating)
        *m.seq_dev = s;
        s_tmp[0] = s;
        s_tmp[1] = 1u;
    }
    __syncthreads();
    const uint32_t seq = s_tmp[0];
    const int slot = (int)(seq & 1u);
    const unsigned long long tag = (unsigned long long)seq << 32;
    const bool direct = m.peers != nullptr;  // device inboxes: a post is WRITTEN to every rank's inbox, a poll reads local memory
    if (tid < 64) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sys[tid >> 1]);
        const unsigned long long half = (tid & 1) ? (bits >> 32) : (bits & 0xffffffffull);
        if (direct) {
            const size_t at = ((size_t)slot * kMailRanks + (size_t)m.rank) * 64 + (size_t)tid;
            for (int r = 0; r < m.nranks; ++r)
                __hip_atomic_store(m.peers[r] + at, tag | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            __hip_atomic_store(&m.box->words[slot][m.rank][tid], tag | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // wave w waits for the ranks w, w + nw, ... -- up to four at a time, their loads in flight together
    constexpr int kAtOnce = 4;
    for (int r0 = wid; r0 < m.nranks; r0 += nw * kAtOnce) {
        unsigned long long w[kAtOnce];
        uint32_t spins = 0u;
        while (true) {
#pragma unroll
            for (int j = 0; j < kAtOnce; ++j) {
                const int r = r0 + j * nw;
                w[j] = tag;  // (no such rank: valid as it stands)
                if (r < m.nranks) {
                    const unsigned long long* src = direct ? m.inbox + ((size_t)slot * kMailRanks + (size_t)r) * 64 + (size_t)lane
                                                           : &m.box->words[slot][r][lane];
                    w[j] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            bool stale = false;
#pragma unroll
            for (int j = 0; j < kAtOnce; ++j) stale = stale || (uint32_t)(w[j] >> 32) != seq;
            if (__ballot(stale) == 0ull) break;
            if (++spins > m.spin_limit) {
                if (lane == 0) s_tmp[1] = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
#pragma unroll
        for (int j = 0; j < kAtOnce; ++j) {
            const int r = r0 + j * nw;
            if (r < m.nranks) s_words[r][lane] = w[j];
        }
    }
    __syncthreads();
    if (tid < 32) {
        double s = 0.0;
        for (int r = 0; r < m.nranks; ++r) {
            const unsigned long long lo = s_words[r][2 * tid] & 0xffffffffull, hi = s_words[r][2 * tid + 1] & 0xffffffffull;
            s += __longlong_as_double((long long)((hi << 32) | lo));
        }
        sys[tid] = s;
    }
    __syncthreads();
    return s_tmp[1] != 0u;
}

}  // namespace mi
