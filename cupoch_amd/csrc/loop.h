// loop.h -- the device-resident state of the registration loop.
//
// registration::RegistrationICP (registration/registration.cu:154-171) alternates a
// device pass (correspondences + reduction) with a tiny host step (6x6 solve, compose
// T, convergence test).  At 10M points the host round trip is 5 % of an iteration; with
// the source sharded over 8 GPUs it would be a third.  Here the tiny step runs on the
// device too (`loop_step_kernel`, one thread, the same __host__ __device__ solver code
// as the one-shot C ABI entry points), the current transform lives in device memory,
// and the search / reduction kernels read it from there.  The host only ENQUEUES
// iterations; once the loop has converged the remaining enqueued kernels see
// `done != 0` and return immediately.
#pragma once
#include "device_utils.h"
#include "host_solver.h"
#include "mailbox.h"

namespace mi {

struct DevLoop {
    Xform X;             // what the source points see (row-major 3x4), read by the kernels
    int32_t done;        // != 0: loop finished (converged or iteration budget spent)
    int32_t est;
    int32_t iterations;  // updates applied (solves executed)
    int32_t passes;      // evaluations done
    int32_t max_iterations;
    int32_t have_prev;
    float det_thresh;
    float rel_fitness, rel_rmse;  // < 0: never converges (stepping API; |d| < negative is false)
    float fitness, rmse, prev_fitness, prev_rmse;
    int64_t n_source_global;
    int32_t ready;       // estimator inputs present (normals / covariances)
    int32_t error;       // != 0: the ranks' exchange failed (mailbox.h); the loop is finished, its result void
    host::Mat4 T;        // reported transformation (column-major)
    host::Mat4 A;        // applied transformation (differs from T only by an ~identity init)
    double sys[32];      // the reduced (and all-reduced) system of the last evaluation
};

__host__ __device__ inline Xform xform_from(const host::Mat4& T) {
    Xform x;
    x.r00 = host::at(T, 0, 0); x.r01 = host::at(T, 0, 1); x.r02 = host::at(T, 0, 2); x.t0 = host::at(T, 0, 3);
    x.r10 = host::at(T, 1, 0); x.r11 = host::at(T, 1, 1); x.r12 = host::at(T, 1, 2); x.t1 = host::at(T, 1, 3);
    x.r20 = host::at(T, 2, 0); x.r21 = host::at(T, 2, 1); x.r22 = host::at(T, 2, 2); x.t2 = host::at(T, 2, 3);
    return x;
}

// ComputeTransformation's host half for the built-in estimators
// (transformation_estimation.cu:137-142,195-222,289-350; generalized_icp.cu:152-183)
__host__ __device__ inline host::Mat4 solve_update(int est, bool ready, const double* sys,
                                                   float det_thresh, int64_t n_model) {
    host::Mat4 update = host::identity4();
    if (!(sys[29] > 0.0) || !ready) return update;
    if (est == 1) return host::kabsch_from_sums(sys, (long long)n_model);
    if (est == 2 || est == 4) {  // point-to-plane, colored ICP (colored_icp.cu:239-243)
        host::solve_system(sys, det_thresh, update);
    } else if (est == 3) {
        host::Mat4 half;
        if (host::solve_system(sys, det_thresh, half)) update = host::square_rotation(half);
    } else if (est == 5) {
        host::solve_system(sys, -1.0f, update);  // no det check (generalized_icp.cu:180)
    }
    return update;
}

// registration.cu:71-78
__host__ __device__ inline void stats_from_system(const double* sys, int64_t n_source, float* fitness,
                                                  float* rmse) {
    const double count = sys[29];
    if (!(count > 0.0) || n_source <= 0) {
        *fitness = 0.0f;
        *rmse = 0.0f;
        return;
    }
    *fitness = (float)count / (float)n_source;
    *rmse = sqrtf((float)sys[28] / (float)count);
}

// Runs after every evaluation (search + reduction [+ all-reduce]) of the loop:
// statistics, the convergence test of registration.cu:165-170 against the previous
// evaluation, and -- unless finished -- the next update (registration.cu:157-160).
// resume > 0 instead re-opens a finished loop for `resume` more updates (stepping API):
// no statistics / test, just the update from the system of the last evaluation.
// One workgroup (any size >= 32): the state is staged through LDS -- one coalesced read, one thread
// of scalar work at LDS latency, one coalesced write (a thread poking at global memory field by
// field took 13 us; this takes ~3).  sys_in may have been written by this very block just before
// (behind a __syncthreads()), or by an earlier kernel.
__device__ __forceinline__ void loop_step_block(DevLoop* st_g, const double* sys_in, int resume, DevLoop& st_s) {
    constexpr int kWords = (int)(sizeof(DevLoop) / 4);
    static_assert(sizeof(DevLoop) % 4 == 0, "DevLoop is copied word by word");
    const int nth = (int)blockDim.x;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&st_s);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(st_g);
    for (int i = (int)threadIdx.x; i < kWords; i += nth) dst[i] = src[i];
    __syncthreads();
    if (resume <= 0 && st_s.done) return;  // uniform: every thread reads the same flag
    if (threadIdx.x < 32 && resume <= 0) st_s.sys[threadIdx.x] = sys_in[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        DevLoop* st = &st_s;
        bool update_now = true;
        if (resume > 0) {
            st->max_iterations = st->iterations + resume;
            st->done = 0;
        } else {
            float fit, rmse;
            stats_from_system(st->sys, st->n_source_global, &fit, &rmse);
            st->fitness = fit;
            st->rmse = rmse;
            st->passes += 1;
            bool finished = false;
            if (st->have_prev && fabsf(st->prev_fitness - fit) < st->rel_fitness &&
                fabsf(st->prev_rmse - rmse) < st->rel_rmse)
                finished = true;
            if (st->iterations >= st->max_iterations || st->error) finished = true;
            if (finished) {
                st->done = 1;
                update_now = false;
            }
        }
        if (update_now) {
            st->prev_fitness = st->fitness;
            st->prev_rmse = st->rmse;
            st->have_prev = 1;
            const host::Mat4 update =
                    solve_update(st->est, st->ready != 0, st->sys, st->det_thresh, st->n_source_global);
            st->T = host::mul4(update, st->T);
            st->A = host::mul4(update, st->A);
            st->X = xform_from(st->A);
            st->iterations += 1;
        }
    }
    __syncthreads();
    uint32_t* out = reinterpret_cast<uint32_t*>(st_g);
    for (int i = (int)threadIdx.x; i < kWords; i += nth) out[i] = dst[i];
}

// The ranks' exchange (mailbox.h) in front of the step, for a block that already holds this rank's
// sums in sys (global memory, written before a barrier or by an earlier kernel).  A finished loop
// exchanges nothing -- on every rank alike, the flag derives from the all-reduced sums.
__device__ __forceinline__ void loop_exchange(DevLoop* st_g, const MailArgs& mail, double* sys) {
    __shared__ uint32_t s_mail[2];
    if (mail.box == nullptr || st_g->done) return;  // (uniform)
    if (!mail_allreduce(mail, sys, s_mail) && threadIdx.x == 0) st_g->error = 1;
    __syncthreads();
}

__global__ __launch_bounds__(64) void loop_step_kernel(DevLoop* st_g, double* sys_in, int resume, MailArgs mail) {
    __shared__ DevLoop st_s;
    if (resume <= 0) loop_exchange(st_g, mail, sys_in);
    loop_step_block(st_g, sys_in, resume, st_s);
}

}  // namespace mi
