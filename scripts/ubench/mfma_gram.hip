#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
// each lane p holds a 16-vector V_p (floats).  G = sum_p V_p V_p^T (16x16, fp64) via v_mfma_f64_16x16x4f64
__global__ void gram(const float* V /*[64][16]*/, double* out /*[16][16]*/, double* raw /*[64][4]*/) {
    __shared__ float s[64][17];
    const int l = threadIdx.x;
    for (int c = 0; c < 16; ++c) s[l][c] = V[l * 16 + c];
    __syncthreads();
    d4 acc = {0, 0, 0, 0};
    for (int c = 0; c < 16; ++c) {           // chunk of 4 points
        const double a = (double)s[4 * c + (l >> 4)][l & 15];   // A[i = l%16][k = l/16] = B[k][j = l%16]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) raw[l * 4 + r] = acc[r];
    // layout (verified on MI355X): acc[r] = D[4*r + l/16][l%16]
    for (int r = 0; r < 4; ++r) out[(4 * r + (l >> 4)) * 16 + (l & 15)] = acc[r];
}
int main() {
    std::vector<float> V(64 * 16);
    for (int i = 0; i < 64 * 16; ++i) V[i] = (float)((i * 7919 % 1000) - 500) / 137.0f;
    float* dV; double *dO, *dR;
    hipMalloc(&dV, V.size() * 4); hipMalloc(&dO, 256 * 8); hipMalloc(&dR, 256 * 8);
    hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice);
    gram<<<1, 64>>>(dV, dO, dR);
    std::vector<double> O(256);
    hipMemcpy(O.data(), dO, 256 * 8, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0; for (int p = 0; p < 64; ++p) ref += (double)V[p * 16 + i] * (double)V[p * 16 + j];
        maxerr = fmax(maxerr, fabs(ref - O[i * 16 + j]));
    }
    printf("max |gram - ref| = %g  (G[0][0]=%f G[3][5]=%f)\n", maxerr, O[0], O[3 * 16 + 5]);
    // empirical layout: which (i, j) does register r of lane l hold?
    std::vector<double> R(256), G(256);
    hipMemcpy(R.data(), dR, 256 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0; for (int p = 0; p < 64; ++p) ref += (double)V[p * 16 + i] * (double)V[p * 16 + j];
        G[i * 16 + j] = ref;
    }
    for (int l = 0; l < 64; l += 5) for (int r = 0; r < 4; ++r) {
        int found = 0;
        for (int i = 0; i < 16 && !found; ++i) for (int j = 0; j < 16; ++j)
            if (fabs(G[i * 16 + j] - R[l * 4 + r]) < 1e-9 * (1 + fabs(R[l * 4 + r]))) { printf("lane %2d reg %d -> D[%2d][%2d]\n", l, r, i, j); found = 1; break; }
        if (!found) printf("lane %2d reg %d -> no match (%f)\n", l, r, R[l * 4 + r]);
    }
    return 0;
}
