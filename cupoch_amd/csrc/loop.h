// loop.h -- the device-resident state of the registration loop.
//
// registration::RegistrationICP (registration/registration.cu:154-171) alternates a
// device pass (correspondences + reduction) with a tiny host step (6x6 solve, compose
// T, convergence test).  At 10M points the host round trip is 5 % of an iteration; with
// the source sharded over 8 GPUs it would be a third.  Here the tiny step runs on the
// device too (`loop_step_kernel`: one workgroup; the same __host__ __device__ solver code
// as the one-shot C ABI entry points, or wave_solver.h's bit-identical wave-wide form of it), the current transform lives in device memory,
// and the search / reduction kernels read it from there.  The host only ENQUEUES
// iterations; once the loop has converged the remaining enqueued kernels see
// `done != 0` and return immediately.
#pragma once
#include <cstddef>
#include "device_utils.h"
#include "host_solver.h"
#include "mailbox.h"
#include "wave_solver.h"

namespace mi {

constexpr int kSysSize = 32;  // the reduced system: 21 + 6 sums, r^2, d^2, count, two spare (reduce.h)
constexpr int kEstP2P = 1, kEstPt2Pl = 2, kEstSym = 3, kEstColored = 4, kEstGICP = 5;  // MI_ICP_EST_* (include/mi_icp.h)

constexpr float kRelocateNears = 1.5f;  // a step beyond this many near radii (kd_build.h tree_scale) re-locates the next search's seeds
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
    uint64_t history;    // device address of float2[kLoopHistory] or 0: (fitness, rmse) an update started from, by iteration
    uint64_t stamps;     // device address of uint64[kStampWords] or 0: where an iteration's time goes (mi_icp_debug_set_step_stamps)
    // RE-LOCATION (nn_search.h locate_by_planes).  The step sizes what it does to the source: the largest displacement of
    // the 8 corners of the source's box under the update just solved.  Beyond about a LEAF'S WIDTH (kRelocateNears x the
    // tree's near radius: ~1.9 point spacings) the next search's seeds are stale: `relocate` is set and the gated locate
    // launch in front of that search (when the host has armed one) replaces every seed by the leaf the moved query
    // falls into.  (The bound was a quarter spacing until late in round 5.  A step that short leaves a query in its
    // seed's leaf or the one beside it, where a located seed is no better than the stale one -- and a registration
    // that keeps sliding re-located at EVERY iteration: the reference's own benchmark call, 113k points turned by 30
    // degrees against themselves, spent 0.2 of its 2.04 ms on 26 descents that changed nothing.)
    uint64_t near2_ptr;       // device address of the tree's squared "near" radius (~1.25 spacings; kd_build.h tree_scale) or 0: never
    uint64_t src_bounds_ptr;  // device address of the staged source's min[3], max[3]
    int32_t ready;       // estimator inputs present (normals / covariances)
    int32_t error;       // != 0: the ranks' exchange failed (mailbox.h); the loop is finished, its result void
    int32_t relocate;    // the step just taken moved the source by more than about a leaf's width
    int32_t relocations; // steps of this loop that did
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

// element i of a register-held array (compile-time indexing only)
__device__ __forceinline__ float select16(const float* m, int i) {
    float v = m[0];
#pragma unroll
    for (int t = 1; t < 16; ++t) v = (i == t) ? m[t] : v;
    return v;
}

// WHERE AN ITERATION'S TIME GOES (include/mi_icp_debug.h mi_icp_debug_set_step_stamps; kernels instantiated with STAMP
// only): s_memrealtime stamps -- one clock for the whole device, 100 MHz -- of
//   [0] the search's earliest wave start (atomic min)   [1] its latest wave end (atomic max)
//   [2] the reduction's earliest block start (min)      [3] the last block has taken the ticket
//   [4] rows totalled   [5] ranks' exchange done   [6] solve + compose done   [7] state written
// The finishing block then adds the eight spans between them -- [0]-[7 of the iteration before], [1]-[0], ... [7]-[6]
// -- to [16 .. 23], counts the iteration in [24], keeps [7] in [8] and re-arms [0 .. 2].
constexpr int kStampWords = 32;
__device__ __forceinline__ unsigned long long stamp_now() { return (unsigned long long)wall_clock64(); }
__device__ __forceinline__ void stamps_close_iteration(unsigned long long* s) {  // one thread, behind the last stamp
    const unsigned long long prev7 = s[8];
    unsigned long long t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = __hip_atomic_load(s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t[0] != ~0ull && t[2] != ~0ull && t[1] != 0ull) {  // (an iteration whose search took the stamping kernel)
        if (prev7 != 0ull && t[0] >= prev7) s[16] += t[0] - prev7;
#pragma unroll
        for (int k = 1; k < 8; ++k) s[16 + k] += (t[k] >= t[k - 1]) ? t[k] - t[k - 1] : 0ull;
        s[24] += 1ull;
    }
    s[8] = t[7];
    __hip_atomic_store(s + 0, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s + 2, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kLoopHistory = 4096;  // entries of the per-iteration record (a ring: iteration i at i % kLoopHistory)
constexpr int kStepThreads = 192;  // the least a block that steps may have: three waves with a role each

// What a block that is about to step may already hold in registers: word threadIdx.x of the loop state
// (nothing else writes the state while the kernel runs; reduce.h's finishing block asks for it together
// with the rows it totals) and, for its first 32 threads, this evaluation's sums.  Saves the step two
// dependent trips to L2.
struct StepPre {
    bool have;
    uint32_t word;
    double sum;
    uint32_t mail_seq;  // (with a mailbox) this rank's exchange counter
};

__device__ __forceinline__ uint32_t loop_state_word(const DevLoop* st_g) {
    constexpr int kWords = (int)(sizeof(DevLoop) / 4);
    return (threadIdx.x < (unsigned)kWords) ? reinterpret_cast<const uint32_t*>(st_g)[threadIdx.x] : 0u;
}


// Runs after every evaluation (search + reduction [+ all-reduce]) of the loop:
// statistics, the convergence test of registration.cu:165-170 against the previous
// evaluation, and -- unless finished -- the next update (registration.cu:157-160).
// resume > 0 instead re-opens a finished loop for `resume` more updates (stepping API):
// no statistics / test, just the update from the system of the last evaluation.
// One workgroup of >= kStepThreads threads.  The state is staged through LDS (one coalesced read, one
// coalesced write; a thread poking at global memory field by field took 13 us), and what used to be one
// thread's ~2000 dependent instructions (3.6-4.2 us: two 6x6 eliminations, two 4x4 products) is spread
// over three waves and their lanes:
//   wave 0  the update: LDL^T solve with a matrix row per lane (wave_solver.h; the estimators that end in
//           SolveJacobianSystemAndObtainExtrinsicMatrix -- the serial routines remain for the others),
//           then, behind the barrier, update * T and update * A with an output element per lane;
//   wave 1  the determinant check of the same system (partial-pivot LU, a row per lane);
//   wave 2  statistics and the convergence test.
// sys_in may have been written by this very block just before (behind a __syncthreads()), or by an
// earlier kernel.
// mail.box != nullptr (N ranks, one node): the ranks' exchange (mailbox.h) sits between the staging and the
// step and works on the staged sums in LDS -- this rank's sums in, every rank's total out; a finished loop
// exchanges nothing (on every rank alike: the flag derives from the all-reduced sums), a failed exchange
// finishes the loop with its error flag set.
__device__ __forceinline__ void loop_step_block(DevLoop* st_g, const double* sys_in, int resume, DevLoop& st_s,
                                                const StepPre pre = StepPre{false, 0u, 0.0, 0u},
                                                const MailArgs mail = MailArgs{nullptr, nullptr, 0, 1, 0u, nullptr, nullptr},
                                                unsigned long long* stamps = nullptr) {
    constexpr int kWords = (int)(sizeof(DevLoop) / 4);
    constexpr int kSysWord0 = (int)(offsetof(DevLoop, sys) / 4);
    static_assert(sizeof(DevLoop) % 4 == 0 && kWords <= kStepThreads, "DevLoop is copied a word per thread");
    static_assert(kSysWord0 + 64 == kWords, "sys closes the state");
    const int tid = (int)threadIdx.x, wid = tid >> 6, lane = tid & 63;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&st_s);
    // the state's words and -- over its sys field, unless resuming -- this evaluation's sums: one round of loads
    if (tid < ((resume <= 0) ? kSysWord0 : kWords)) dst[tid] = pre.have ? pre.word : reinterpret_cast<const uint32_t*>(st_g)[tid];
    if (tid < 32 && resume <= 0) st_s.sys[tid] = pre.have ? pre.sum : sys_in[tid];
    __syncthreads();
    if (resume <= 0 && st_s.done) return;  // uniform: every thread reads the same flag
    if (mail.box != nullptr && resume <= 0) {
        __shared__ uint32_t s_mail[2];
        const bool ok = mail_allreduce(mail, st_s.sys, s_mail, pre.have ? &pre.mail_seq : nullptr);
        if (!ok && tid == 0) st_s.error = 1;
        __syncthreads();
    }
    if (stamps && tid == 0) stamps[5] = stamp_now();
    __shared__ host::Mat4 s_update;
    __shared__ int s_det_ok, s_update_now;
    DevLoop* st = &st_s;
    const int est = st->est;
    const bool wave_case = (est == 2 || est == 4 || est == 5) && st->ready != 0 && st->sys[29] > 0.0;
    if (wid == 0) {
#ifdef MI_AB_NO_SOLVE
        if (wave_case) {
            if (lane < 16) s_update.m[lane] = (lane % 5 == 0) ? 1.0f : 0.0f;
        } else
#endif
        if (wave_case) {
            const host::Mat4 U = wave_solve_update(st->sys);
            if (lane < 16) s_update.m[lane] = select16(U.m, lane);
        } else if (lane == 0) {
            s_update = solve_update(est, st->ready != 0, st->sys, st->det_thresh, st->n_source_global);
        }
    } else if (wid == 1) {
        // est 5: no det check (generalized_icp.cu:180)
        const bool ok = !(wave_case && est != 5 && st->det_thresh > 0.0f) || wave_det_passes(st->sys, st->det_thresh);
        if (lane == 0) s_det_ok = ok ? 1 : 0;
    }
    // (wave 2, lanes 0..7: the corners of the source's box as the searches saw them -- A is stable until the barrier)
    float cx = 0.0f, cy = 0.0f, cz = 0.0f, far2 = INFINITY;  // far2: (a leaf's width)^2 = (kRelocateNears * ~1.25 spacings)^2
#ifdef MI_AB_NO_SIZING
    const bool sized = false;
#else
    const bool sized = st->near2_ptr != 0ull && st->src_bounds_ptr != 0ull;
#endif
    // (everything this needs from memory is asked for BEFORE the barrier, beside the solve: a load behind it was 2 us on
    // every step's critical path -- 6 % of an 8-way shard's)
    if (wid == 2 && lane == 0 && sized) far2 = (kRelocateNears * kRelocateNears) * *reinterpret_cast<const float*>(st->near2_ptr);
    if (wid == 2 && lane < 8 && sized) {
        const float* sb = reinterpret_cast<const float*>(st->src_bounds_ptr);
        const float px = sb[(lane & 1) ? 3 : 0], py = sb[(lane & 2) ? 4 : 1], pz = sb[(lane & 4) ? 5 : 2];
        cx = host::at(st->A, 0, 0) * px + host::at(st->A, 0, 1) * py + host::at(st->A, 0, 2) * pz + host::at(st->A, 0, 3);
        cy = host::at(st->A, 1, 0) * px + host::at(st->A, 1, 1) * py + host::at(st->A, 1, 2) * pz + host::at(st->A, 1, 3);
        cz = host::at(st->A, 2, 0) * px + host::at(st->A, 2, 1) * py + host::at(st->A, 2, 2) * pz + host::at(st->A, 2, 3);
    }
    if (wid == 2 && lane == 0) {
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
        // what registration.cu:155-156 logs: the evaluation an update starts from, by iteration number
        if (update_now && st->history != 0ull)
            reinterpret_cast<float2*>(st->history)[st->iterations & (kLoopHistory - 1)] = make_float2(st->fitness, st->rmse);
        s_update_now = update_now ? 1 : 0;
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[6] = stamp_now();  // (the solve; the compose below is a handful of instructions)
    if (wid == 2 && sized) {  // how far does the update move the corners?  (s_update is final; a failed determinant check: identity)
        float d2 = 0.0f;
        if (lane < 8 && s_update_now && s_det_ok) {
            const host::Mat4& U = s_update;
            const float dx = host::at(U, 0, 0) * cx + host::at(U, 0, 1) * cy + host::at(U, 0, 2) * cz + host::at(U, 0, 3) - cx;
            const float dy = host::at(U, 1, 0) * cx + host::at(U, 1, 1) * cy + host::at(U, 1, 2) * cz + host::at(U, 1, 3) - cy;
            const float dz = host::at(U, 2, 0) * cx + host::at(U, 2, 1) * cy + host::at(U, 2, 2) * cz + host::at(U, 2, 3) - cz;
            d2 = dx * dx + dy * dy + dz * dz;
        }
        d2 = fmaxf(d2, __shfl_xor(d2, 1, 64));
        d2 = fmaxf(d2, __shfl_xor(d2, 2, 64));
        d2 = fmaxf(d2, __shfl_xor(d2, 4, 64));
        if (lane == 0) {
            const int far = (d2 > far2) ? 1 : 0;  // (NaN / inf radius: never)
            st->relocate = far;
            st->relocations += far;
        }
    }
    if (wid == 0 && s_update_now) {
        // host::mul4(update, T) and (update, A): lane = 16 * matrix + 4 * column + row; a failed determinant
        // check leaves the identity as the update (solve_system)
        const int r = lane & 3, c = (lane >> 2) & 3;
        const host::Mat4& B = (lane & 16) ? st->A : st->T;
        const bool ok = s_det_ok != 0;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sum += (ok ? host::at(s_update, r, k) : ((r == k) ? 1.0f : 0.0f)) * host::at(B, k, c);
        __builtin_amdgcn_wave_barrier();  // all of T, A read before any of it is written
        if (lane < 16) st->T.m[lane] = sum;
        else if (lane < 32) st->A.m[lane - 16] = sum;
        __builtin_amdgcn_wave_barrier();
        if (lane < 12) reinterpret_cast<float*>(&st->X)[lane] = host::at(st->A, lane >> 2, lane & 3);
        if (lane == 0) {
            st->prev_fitness = st->fitness;
            st->prev_rmse = st->rmse;
            st->have_prev = 1;
            st->iterations += 1;
        }
    }
    __syncthreads();
    if (tid < kWords) reinterpret_cast<uint32_t*>(st_g)[tid] = dst[tid];
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            stamps[7] = stamp_now();
            stamps_close_iteration(stamps);
        }
    }
}

}  // namespace mi
