// Pins the lane layout of v_mfma_f64_16x16x4_f64 on gfx950: D = A(16x4) * B(4x16) + C with asymmetric operands.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* out) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];   // A[i = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l>>4][j = l&15]
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    double hA[64], hB[64], ref[256], hout[256];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = 1 + i * 0.37 + kk * 1.9;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 2 - j * 0.11 + kk * kk * 0.7;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dO;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dO, sizeof hout);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    hipMemcpy(hout, dO, sizeof hout, hipMemcpyDeviceToHost);
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int col = l & 15;
        const int rowA = (l >> 4) + 4 * r;      // candidate A
        const int rowB = 4 * (l >> 4) + r;      // candidate B
        if (fabs(hout[l * 4 + r] - ref[rowA * 16 + col]) > 1e-12) okA = 0;
        if (fabs(hout[l * 4 + r] - ref[rowB * 16 + col]) > 1e-12) okB = 0;
    }
    printf("layout row=(lane>>4)+4*reg: %s ; row=4*(lane>>4)+reg: %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
    return 0;
}
