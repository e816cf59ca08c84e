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
// IPC handles.  A post is {32 sums, then -- behind a system-scope release -- the sequence number};
// slots alternate with the sequence number's parity, so a rank that is already one exchange ahead
// cannot overwrite what a slower rank still has to read (to get two ahead it needs everybody's post
// of the exchange in between).  Sequence numbers count a rank's exchanges since the box was made and
// never repeat; every rank performs the same exchanges (the loop's control flow depends only on the
// all-reduced sums, which are identical everywhere).
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
    uint32_t pad_[14];
    uint32_t seq[2][kMailRanks][16];     // one 64-byte line per flag
    unsigned long long sums[2][kMailRanks][32];  // doubles, as bits
};

struct MailArgs {
    MailBox* box;       // device address of the registered host mapping; null: no mailbox
    uint32_t* seq_dev;  // this rank's exchange counter (device memory, zeroed with the box)
    int rank, nranks;
    uint32_t spin_limit;
};

// One workgroup of >= 32 threads.  sys: this rank's 32 sums (global or LDS, written before a barrier);
// on return (all threads past a barrier) it holds the ranks' totals.  s_tmp: two LDS words.
// False: a peer did not post in time.
__device__ __forceinline__ bool mail_allreduce(const MailArgs& m, double* sys, uint32_t* s_tmp) {
    if (threadIdx.x == 0) {
        const uint32_t s = *m.seq_dev + 1u;
        *m.seq_dev = s;
        s_tmp[0] = s;
        s_tmp[1] = 1u;
    }
    __syncthreads();
    const uint32_t seq = s_tmp[0];
    const int slot = (int)(seq & 1u);
    if (threadIdx.x < 32)
        __hip_atomic_store(&m.box->sums[slot][m.rank][threadIdx.x], (unsigned long long)__double_as_longlong(sys[threadIdx.x]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&m.box->seq[slot][m.rank][0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < m.nranks) {  // thread r waits for rank r
        uint32_t spins = 0u;
        // (relaxed polls -- host memory is not cached on the GPU side -- and one acquire fence behind the barrier)
        while (__hip_atomic_load(&m.box->seq[slot][threadIdx.x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++spins > m.spin_limit) {
                s_tmp[1] = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();
    __threadfence_system();
    if (threadIdx.x < 32) {
        double s = 0.0;
        for (int r = 0; r < m.nranks; ++r)
            s += __longlong_as_double((long long)__hip_atomic_load(&m.box->sums[slot][r][threadIdx.x], __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_SYSTEM));
        sys[threadIdx.x] = s;
    }
    __syncthreads();
    return s_tmp[1] != 0u;
}

// the exchange on its own (one-shot entry points: compute_system / evaluate_registration under a communicator)
__global__ __launch_bounds__(64) void mail_allreduce_kernel(MailArgs m, double* sys, int32_t* error_out) {
    __shared__ uint32_t s_tmp[2];
    const bool ok = mail_allreduce(m, sys, s_tmp);
    if (!ok && threadIdx.x == 0 && error_out) *error_out = 1;
}

}  // namespace mi
