// mailbox_kernels.h -- the kernels that run mailbox.h's exchange on its own: the one-shot exchange of the C ABI's
// one-shot entry points, and the self-tests of mi_icp_comm_autotune.  Included by mi_comm.hip only.
#pragma once
#include "mailbox.h"

namespace mi {

// the exchange on its own (one-shot entry points: compute_system / evaluate_registration under a communicator)
static __global__ __launch_bounds__(64) void mail_allreduce_kernel(MailArgs m, double* sys, int32_t* error_out) {
    __shared__ uint32_t s_tmp[2];
    const bool ok = mail_allreduce(m, sys, s_tmp);
    if (!ok && threadIdx.x == 0 && error_out) *error_out = 1;
}

// The exchange's self-test (mi_icp_comm_autotune): n exchanges of a KNOWN vector back to back in one launch -- what
// the loop's finishing block does once per iteration, same workgroup shape -- every total checked exactly:
// rank r posts (r + 1) (k + 1) f with f = 1, 2, 3, 1, ... changing every exchange (a stale post of the exchange before
// cannot pass), the total must be R (R + 1) / 2 (k + 1) f.  status[0] <- 1: a peer did not post in time; status[1]:
// totals that were wrong.
// (bias: added to what this rank posts -- the test hook that makes a path sum wrongly)
static __global__ __launch_bounds__(256) void mail_selftest_kernel(MailArgs m, int n, double bias, double* __restrict__ out32,
                                                            int32_t* __restrict__ status) {
    __shared__ double s_sys[32];
    __shared__ uint32_t s_tmp[2];
    const int tid = (int)threadIdx.x;
    const double tri = 0.5 * (double)m.nranks * (double)(m.nranks + 1);
    int bad = 0;
    for (int it = 0; it < n; ++it) {
        const double f = (double)(1 + it % 3);
        if (tid < 32) s_sys[tid] = (double)(m.rank + 1) * (double)(tid + 1) * f + bias;
        __syncthreads();
        const bool ok = mail_allreduce(m, s_sys, s_tmp);
        if (!ok) {  // (uniform)
            if (tid == 0) status[0] = 1;
            return;
        }
        if (tid < 32 && s_sys[tid] != tri * (double)(tid + 1) * f) ++bad;
        __syncthreads();
    }
    if (bad) atomicAdd(&status[1], bad);
    if (tid < 32) out32[tid] = s_sys[tid];
}

// ... and the in-library RCCL path's: the vector of exchange `it` into buf (checking what the all-reduce of the exchange
// before has left there), one launch per exchange -- as the loop's step kernel is one behind every all-reduce
static __global__ __launch_bounds__(64) void rccl_selftest_fill(double* __restrict__ buf, int rank, int nranks, int it,
                                                         int32_t* __restrict__ status) {
    const int k = (int)threadIdx.x;
    if (k >= 32) return;
    if (it > 0) {
        const double tri = 0.5 * (double)nranks * (double)(nranks + 1);
        if (buf[k] != tri * (double)(k + 1) * (double)(1 + (it - 1) % 3)) atomicAdd(&status[1], 1);
    }
    buf[k] = (double)(rank + 1) * (double)(k + 1) * (double)(1 + it % 3);
}

}  // namespace mi
