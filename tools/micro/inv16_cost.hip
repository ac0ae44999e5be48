// Micro-benchmark: the in-register 16 x 16 eliminations of dort_finish_reg.hpp (rg::inv16_step chain, the look-ahead variant,
// the two-column variant) on their own: shader cycles per inversion, 1 / 2 / 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I smrt_amd/csrc -o tools/micro/inv16_cost tools/micro/inv16_cost.hip && tools/micro/inv16_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dort_device.hpp"
#include "dort_finish_reg.hpp"
using namespace smrt;
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
    const rg::LaneId L = rg::lane_id();
    double d[4];
    for (int r = 0; r < 4; ++r) d[r] = ((4 * r + L.g == L.c) ? 4.0 : 0.0) + seed * (1 + ((7 * L.lane + 3 * r) % 13)) * 0.01;
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 32; ++it) {
        if (MODE == 0) { rg::inv16_step<0>(d, L); rg::inv16_step<1>(d, L); rg::inv16_step<2>(d, L); rg::inv16_step<3>(d, L); rg::inv16_step<4>(d, L); rg::inv16_step<5>(d, L); rg::inv16_step<6>(d, L); rg::inv16_step<7>(d, L);
                         rg::inv16_step<8>(d, L); rg::inv16_step<9>(d, L); rg::inv16_step<10>(d, L); rg::inv16_step<11>(d, L); rg::inv16_step<12>(d, L); rg::inv16_step<13>(d, L); rg::inv16_step<14>(d, L); rg::inv16_step<15>(d, L); }
        if (MODE == 1) rg::inv16_la(d, L);
        if (MODE == 2) { rg::inv16_step2<0>(d, L); rg::inv16_step2<2>(d, L); rg::inv16_step2<4>(d, L); rg::inv16_step2<6>(d, L); rg::inv16_step2<8>(d, L); rg::inv16_step2<10>(d, L); rg::inv16_step2<12>(d, L); rg::inv16_step2<14>(d, L); }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d[0] + d[1] + d[2] + d[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* what, int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-44s %4d threads per CU: %8.0f cycles per inversion\n", what, threads, s / 256 / 32);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {64, 256, 512, 1024}) {
        run<0>("inv16_step chain (one column per step)", th);
        run<1>("inv16_la (next pivot row one step ahead)", th);
        run<2>("inv16_step2 (two columns per step)", th);
    }
    return 0;
}
