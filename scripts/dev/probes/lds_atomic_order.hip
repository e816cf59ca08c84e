// Probe: in ONE ds_add_rtn_u32 instruction of a wave64, are lanes that add to the same LDS word served in ascending lane
// order?  (voxel_dense.h ranks a tile's points inside their bucket with such adds on wave-private counters; the rank must
// be the input order.)  Random bin patterns with many conflicts, plain and packed (two 16-bit counters per word) forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const unsigned* __restrict__ bins, unsigned* __restrict__ out, int rounds, int nbins, int packed) {
    __shared__ unsigned cnt[4][2048];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = lane; k < 2048; k += 64) cnt[w][k] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < rounds; ++r) {
        const unsigned b = bins[((size_t)blockIdx.x * 4 + w) * rounds * 64 + (size_t)r * 64 + lane] % (unsigned)nbins;
        unsigned got;
        if (packed) {
            const unsigned v = atomicAdd(&cnt[w][b >> 1], (b & 1u) ? 0x10000u : 1u);
            got = (b & 1u) ? (v >> 16) : (v & 0xffffu);
        } else {
            got = atomicAdd(&cnt[w][b], 1u);
        }
        out[((size_t)blockIdx.x * 4 + w) * rounds * 64 + (size_t)r * 64 + lane] = got;
    }
}

int main() {
    const int blocks = 512, rounds = 64;
    const size_t n = (size_t)blocks * 4 * rounds * 64;
    std::vector<unsigned> h(n), o(n);
    unsigned *d_in, *d_out;
    hipMalloc(&d_in, n * 4);
    hipMalloc(&d_out, n * 4);
    long bad_total = 0;
    for (int packed = 0; packed < 2; ++packed)
        for (int nbins : {1, 2, 7, 32, 33, 64, 257, 1024, 2048}) {
            srand(nbins * 2 + packed);
            for (size_t i = 0; i < n; ++i) h[i] = (unsigned)rand();
            hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
            probe<<<blocks, 256>>>(d_in, d_out, rounds, nbins, packed);
            hipMemcpy(o.data(), d_out, n * 4, hipMemcpyDeviceToHost);
            long bad = 0;
            for (size_t wv = 0; wv < (size_t)blocks * 4; ++wv) {
                std::vector<unsigned> c(2048, 0u);
                for (int r = 0; r < rounds; ++r)
                    for (int l = 0; l < 64; ++l) {
                        const size_t i = wv * rounds * 64 + (size_t)r * 64 + l;
                        const unsigned b = h[i] % (unsigned)nbins;
                        if (o[i] != c[b]) ++bad;
                        ++c[b];
                    }
            }
            printf("packed %d nbins %4d: %ld of %zu ranks out of lane order\n", packed, nbins, bad, n);
            bad_total += bad;
        }
    printf(bad_total ? "ORDER NOT LANE-ASCENDING\n" : "lane-ascending in every case\n");
    return bad_total ? 1 : 0;
}
