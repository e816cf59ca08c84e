// fetch_calib.hip -- FETCH_SIZE / WRITE_SIZE calibration on KNOWN byte counts, in the access
// shapes of the search kernel (nn_search.h) -- MI355X_MICROARCH.md, section HBM: on gfx950
// FETCH_SIZE under-reports wide coalesced reads by 2x and "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern".
//   run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
//               rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ./fetch_calib
// Every kernel reads (or writes) each byte of a 1 GiB buffer exactly once (4x the 256 MB
// Infinity Cache, so nothing is served from a previous pass) and prints the byte count the
// counters should show.  Shapes:
//   soa4      4 B per lane, consecutive lanes consecutive words        (sx/sy/sz/nn_idx loads)
//   leaf16    16 B per lane, 8 consecutive lanes share one 128-B line and each reads its 6 x 16 B
//             (x, y, z of the 8 slots)                                  (seed-leaf evaluation)
//   pair32    2 x 16 B per lane, consecutive lanes consecutive 32-B records (leaf regions, list chunks)
//   scalar64  3 x s_load_dwordx16 per wave = 192 B of a 256-B record     (tree records)
//   stream16  16 B per lane, consecutive lanes consecutive               (the guide's 2x case)
//   write4    4 B per lane stores                                        (nn_idx / nn_d2 output)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f16v* cf16_p;

__global__ __launch_bounds__(256) void calib_soa4(const float* __restrict__ a, size_t n, float* __restrict__ sink) {
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
    if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(64) void calib_leaf16(const float4* __restrict__ a, size_t nlines, float* __restrict__ sink) {
    // one wave = 8 lines of 128 B: lane l reads the 6 first float4 of line (wave * 8 + l / 8)
    const size_t line = (size_t)blockIdx.x * 8 + (threadIdx.x >> 3);
    if (line >= nlines) return;
    const float4* p = a + line * 8;
    const float4 x0 = p[0], x1 = p[1], y0 = p[2], y1 = p[3], z0 = p[4], z1 = p[5];
    const float s = x0.x + x1.y + y0.z + y1.w + z0.x + z1.y;
    if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_pair32(const float4* __restrict__ a, size_t nrec, float* __restrict__ sink) {
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nrec; i += (size_t)gridDim.x * 256) {
        const float4 g0 = a[2 * i], g1 = a[2 * i + 1];
        s += g0.x + g1.w;
    }
    if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(64) void calib_scalar64(const float* __restrict__ a, size_t nrec, float* __restrict__ sink) {
    const size_t r = blockIdx.x;
    if (r >= nrec) return;
    const cf16_p rec = (cf16_p)(uintptr_t)(a + r * 64);
    const f16v r0 = rec[0], r1 = rec[1], r2 = rec[2];
    const float s = r0[0] + r1[5] + r2[15];
    if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_stream16(const float4* __restrict__ a, size_t n, float* __restrict__ sink) {
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i].x;
    if (s == 1.2345f) sink[0] = s;
}
__global__ __launch_bounds__(256) void calib_write4(float* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = (float)i;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    float *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const size_t nf = bytes / 4;
    for (int rep = 0; rep < 3; ++rep) {
        calib_soa4<<<8192, 256>>>(buf, nf, sink);
        calib_leaf16<<<(unsigned)(bytes / 128 / 8), 64>>>((const float4*)buf, bytes / 128, sink);
        calib_pair32<<<8192, 256>>>((const float4*)buf, bytes / 32, sink);
        calib_scalar64<<<(unsigned)(bytes / 256), 64>>>(buf, bytes / 256, sink);
        calib_stream16<<<8192, 256>>>((const float4*)buf, bytes / 16, sink);
        calib_write4<<<8192, 256>>>(buf, nf);
        hipDeviceSynchronize();
    }
    // bytes each kernel touches: whole 128-B lines for leaf16 (96 of every 128 B requested), 192 of every
    // 256 B for scalar64 (three 64-B scalar-cache lines)
    printf("known_bytes calib_soa4 %zu\n", bytes);
    printf("known_bytes calib_leaf16 %zu (requested %zu)\n", bytes, bytes / 128 * 96);
    printf("known_bytes calib_pair32 %zu\n", bytes);
    printf("known_bytes calib_scalar64 %zu\n", bytes / 256 * 192);
    printf("known_bytes calib_stream16 %zu\n", bytes);
    printf("known_bytes calib_write4 %zu\n", bytes);
    return 0;
}
