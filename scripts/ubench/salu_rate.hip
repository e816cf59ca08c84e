// micro-benchmark: SALU issue rate per CU, alone and interleaved with VALU, at several
// occupancies.  Answers: is the scalar ALU a per-CU resource (1 instr/cycle shared by the
// 4 SIMDs) or per-SIMD, and do SALU and VALU streams of one wave overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
// MODE 0: 8 independent s_add per rep   MODE 1: 8 v_fma per rep   MODE 2: 8 s_add + 8 v_fma interleaved
// MODE 3: dependent s_add chain (latency)  MODE 4: s_cmp/s_addc/s_mov_b64 triple x4 (the box-test tail)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters, int s_in) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int s0 = s_in, s1 = s_in + 1, s2 = s_in + 2, s3 = s_in + 3, s4 = s_in + 4, s5 = s_in + 5, s6 = s_in + 6, s7 = s_in + 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (MODE == 0 || MODE == 2) {
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)::"scc");
            }
            if (MODE == 1 || MODE == 2) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
            if (MODE == 3) {
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                             : "+s"(s0)::"scc");
            }
            if (MODE == 4) {
                asm volatile("s_mov_b64 s[40:41], exec\n"
                             "s_cmp_lg_u64 exec, 0\n s_mov_b64 exec, s[40:41]\n s_addc_u32 %0, %0, %0\n"
                             "s_cmp_lg_u64 exec, 0\n s_mov_b64 exec, s[40:41]\n s_addc_u32 %0, %0, %0\n"
                             "s_cmp_lg_u64 exec, 0\n s_mov_b64 exec, s[40:41]\n s_addc_u32 %0, %0, %0\n"
                             "s_cmp_lg_u64 exec, 0\n s_mov_b64 exec, s[40:41]\n s_addc_u32 %0, %0, %0\n"
                             : "+s"(s0)::"scc", "s40", "s41");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
}
template <int MODE>
void run(const char* name, int salu_per_rep, int valu_per_rep, int blocks_per_cu, int threads) {
    float* d; hipMalloc(&d, 256 * 64 * 256 * 4);
    const int blocks = 256 * blocks_per_cu, iters = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(d, 1.0001f, 0.5f, 2, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, 1.0001f, 0.5f, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_cu = (double)blocks_per_cu * threads / 64;
    const double cyc = ms * 1e-3 * 2.4e9;
    const double s_per_cu = waves_per_cu * iters * REP * salu_per_rep, v_per_cu = waves_per_cu * iters * REP * valu_per_rep;
    printf("%-34s waves/CU %4.0f  %8.3f ms  SALU/CU/cycle %.3f  VALU(wave-instr)/CU/cycle %.3f\n", name, waves_per_cu, ms,
           s_per_cu / cyc, v_per_cu / cyc);
    hipFree(d);
}
int main() {
    const int occ[][2] = {{1, 64}, {1, 256}, {2, 256}, {4, 256}, {8, 256}};
    for (auto& o : occ) run<0>("s_add x8 (independent)", 8, 0, o[0], o[1]);
    for (auto& o : occ) run<1>("v_fma x8", 0, 8, o[0], o[1]);
    for (auto& o : occ) run<2>("s_add x8 + v_fma x8 interleaved", 8, 8, o[0], o[1]);
    for (auto& o : occ) run<3>("s_add dependent chain", 8, 0, o[0], o[1]);
    for (auto& o : occ) run<4>("box tail: cmp/mov exec/addc x4", 13, 0, o[0], o[1]);
    return 0;
}
