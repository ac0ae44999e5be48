// Micro-benchmark: what the LDS traffic of one Jacobi step costs by itself on gfx950 -- per wavefront three ds_read2_b64
// (six rows of a column per lane, eight lanes per column, LD = 56 doubles) and, optionally, three ds_write2_b64 of the
// same addresses -- at 21 wavefronts per CU (seven workgroups of three), with no arithmetic beside one add per value.
// Prints LDS-pipe clocks per wavefront step = 2.4 GHz x 256 CUs x time / (wavefronts x steps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/lds_rate tools/micro/lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WRITE, int LD>
__global__ __launch_bounds__(192) void k(double* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t / 64;
    for (int i = t; i < 49 * LD; i += 192) lds[i] = 1e-3 * i;
    __syncthreads();
    const int slot = lane / 8, sub = lane % 8;
    double acc = 0.0;
    int col = 16 * wave + 8 + slot;
    for (int s = 0; s < steps; ++s) {
        double* c = lds + col * LD + sub;
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) y[i] = c[8 * i];
        if (WRITE) {
#pragma unroll
            for (int i = 0; i < 6; ++i) c[8 * i] = y[i] + 1.0;
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc += y[i];
        }
        col = col + 1; if (col >= 16 * wave + 16) col -= 8;
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 192 + t] = acc + lds[t];
}

template <int WRITE, int LD>
static void run(const char* name, double* d) {
    const int steps = 4000, grid = 256 * 7 * 8;
    const size_t lds = 22 * 1024;
    hipFuncSetAttribute((const void*)k<WRITE, LD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<WRITE, LD>), dim3(grid), dim3(192), lds, 0, d, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WRITE, LD>), dim3(grid), dim3(192), lds, 0, d, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.3f ms   %7.1f CU clk per wavefront step\n", name, ms, 256.0 * 2.4e6 * ms / ((double)grid * 3 * steps));
}

int main() {
    double* d; hipMalloc(&d, sizeof(double) * 192 * 256 * 7 * 8);
    run<0, 56>("3 x ds_read2_b64, LD 56", d);
    run<1, 56>("3 x ds_read2_b64 + 3 x ds_write2_b64, LD 56", d);
    run<0, 57>("3 x ds_read2_b64, LD 57 (bank conflicts)", d);
    run<1, 72>("3 x ds_read2_b64 + 3 x ds_write2_b64, LD 72", d);
    return 0;
}
