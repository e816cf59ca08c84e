// micro-benchmark: issue cost of the VALU / SALU instruction mixes the traversal uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 256
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x1, x2}, p3 = {x3, x0};
    f2 pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
            if (MODE == 1) { p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb); p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb); }
            if (MODE == 2) { p0 = p0 - pa; p1 = p1 - pb; p2 = p2 - pa; p3 = p3 - pb; }
            if (MODE == 3) { x0 = fmaxf(fmaxf(x0, a), x1 * 0.5f); x1 = fmaxf(fmaxf(x1, b), x2); x2 = fmaxf(fmaxf(x2, a), x3); x3 = fmaxf(fmaxf(x3, b), x0); }
            if (MODE == 4) { x0 = fmaxf(x0, a); x1 = fmaxf(x1, b); x2 = fmaxf(x2, a); x3 = fmaxf(x3, b); x0 += 1.0f; x1 += 1.0f; x2 += 1.0f; x3 += 1.0f; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, int instr_per_rep) {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    const int blocks = 256 * 8, iters = 64;   // 8 blocks/CU = 32 waves/CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, 1.0001f, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * REP * instr_per_rep;   // wave-instructions
    double per_simd_cycle = winstr / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-28s %8.3f ms  %.3f wave-instr/SIMD/cycle(@2.4GHz)  => %.2f cycles per wave-instr\n", name, ms, per_simd_cycle, 1.0 / per_simd_cycle);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32 x4", 4);
    run<1>("v_pk_fma_f32 x4", 4);
    run<2>("v_pk_add_f32 x4", 4);
    run<3>("v_max3_f32 x4 (+1 mul)", 5);
    run<4>("v_max_f32 x4 + v_add x4", 8);
    return 0;
}
