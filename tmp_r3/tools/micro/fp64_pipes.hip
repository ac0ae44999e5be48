// Micro-benchmark: are the FP64 vector pipe (v_fma_f64) and the FP64 matrix pipe (v_mfma_f64_16x16x4_f64) of a gfx950
// SIMD separate resources?  Both have the same peak (78.6 TFLOP/s on MI355X); if they run side by side a kernel that
// mixes rotation-style vector work with tile GEMMs could exceed either alone.
//   variant 0: FMA only        (16 independent accumulators per lane)
//   variant 1: MFMA only       (4 independent accumulator tiles)
//   variant 2: both interleaved in the same wavefront
//   variant 3: even wavefronts FMA only, odd wavefronts MFMA only (two wavefronts per SIMD)
// Prints TFLOP/s for each (FMA = 2 flop x 64 lanes, MFMA 16x16x4 = 2048 flop).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/fp64_pipes tools/micro/fp64_pipes.hip && tools/micro/fp64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(512) void k(double* out, int iters) {
    const int wave = threadIdx.x / 64;
    double a[16];
    v4d c[4];
    const double x = 1.0 + 1e-9 * threadIdx.x, y = 0.999999;
    for (int i = 0; i < 16; ++i) a[i] = i * 1e-3;
    for (int i = 0; i < 4; ++i) c[i] = {0.0, 0.0, 0.0, 0.0};
    const bool do_fma = (VAR == 0) || (VAR == 2) || (VAR == 3 && (wave & 1) == 0);
    const bool do_mfma = (VAR == 1) || (VAR == 2) || (VAR == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_fma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_fma(a[i], y, x);
        }
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i], 0, 0, 0);
        }
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR>
static void run(const char* name, double* d, int grid) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(512), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * 8;
    double fma_waves = (VAR == 0 || VAR == 2) ? waves : (VAR == 3 ? waves / 2 : 0);
    double mfma_waves = (VAR == 1 || VAR == 2) ? waves : (VAR == 3 ? waves / 2 : 0);
    const double flop = (double)iters * (fma_waves * 64.0 * 64 * 2 + mfma_waves * 4.0 * 2048);
    printf("%-46s %8.2f ms  %7.2f TFLOP/s (vector part %.2f, matrix part %.2f)\n", name, ms, flop / ms * 1e-9,
           iters * fma_waves * 64.0 * 64 * 2 / ms * 1e-9, iters * mfma_waves * 4.0 * 2048 / ms * 1e-9);
}

int main() {
    const int grid = 256 * 4;   // 2 wavefronts per SIMD x 4 rounds
    double* d;
    hipMalloc(&d, sizeof(double) * grid * 512);
    run<0>("0 v_fma_f64 only", d, grid);
    run<1>("1 v_mfma_f64_16x16x4 only", d, grid);
    run<2>("2 both in every wavefront", d, grid);
    run<3>("3 even waves FMA, odd waves MFMA", d, grid);
    return 0;
}
