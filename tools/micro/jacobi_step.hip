// Micro-benchmark: what one inner step of the Jacobi kernel (dort_jacobi_kernel.hpp: rotate_resident, eight lanes per column
// pair, six rows per lane = a 48-column item) costs on gfx950, by path and by occupancy -- the product code itself, on an
// LDS matrix of random columns, with the data-dependent decisions pinned by the thresholds:
//   never rotate    skip2 = 1e300                  (loads, dot product, lane-group sum, threshold: the skipped step)
//   always, series  skip2 = -1, small angle        (the columns are orthogonal after the first passes: |g| ~ 1e-17)
//   always, full    skip2 = -1, built with -DSMRT_JACOBI_SMALL_ANGLE=-1   (rsq / rcp / rsq and their Newton steps)
// for at most 1 .. 7 workgroups of three wavefronts per CU (the LDS request pads the rest away), sixteen rounds of them:
// `SIMD clk per wavefront step` is the throughput figure, 2.4 GHz x 1024 SIMDs x time / (wavefronts x steps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I smrt_amd/csrc -o tools/micro/jacobi_step tools/micro/jacobi_step.hip
//   hipcc ... -DSMRT_JACOBI_SMALL_ANGLE=-1 -o tools/micro/jacobi_step_full tools/micro/jacobi_step.hip
//   hipcc ... -DUSE_CARRIED: the experiment of round 4 -- the J column's LDS address and its norm's address carried from
//   step to step (add, compare, two selects, subtract) instead of recomputed from (lane group, step): 11 vector
//   instructions fewer per step, one of them a quarter-rate integer multiply, but the address now hangs on the previous
//   step (profiles/r4_jacobi_step_micro.txt: slower at every occupancy; in the kernel 38.82 against 38.97 ms, dropped).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "dort_device.hpp"

using namespace smrt;

constexpr int GS = 8, RPL = 6, M = 8, LDJ = 56, NCOLS = 49;

#ifdef USE_CARRIED
typedef unsigned lds_addr_t;
typedef __attribute__((address_space(3))) double lds_double_t;
__device__ __forceinline__ lds_addr_t lds_address(const void* p) { return (lds_addr_t)(size_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ lds_double_t* lds_double_at(lds_addr_t a) { return (lds_double_t*)(size_t)a; }
template <int GS, int RPL>
__device__ __forceinline__ void rotate_cross(double* Bm, int LD, int CP, int m, int i0, int j0, int slot, int sub, double* nrm,
                                             int* flag, double skip2, double exit2) {
    const int ps = slot;
    const bool valid = ps < m;
    const int pc = valid ? i0 + ps : CP;
    double* cp = Bm + pc * LD;
    double x[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) x[i] = cp[sub + i * GS];
    double a = nrm[pc];
    const unsigned q0 = valid ? (unsigned)(j0 + ps) : (unsigned)CP;
    const lds_addr_t base = lds_address(Bm);
    lds_addr_t addr = base + (q0 * (unsigned)LD + (unsigned)sub) * 8u, naddr = lds_address(nrm) + q0 * 8u;
    const lds_addr_t dstep = valid ? (unsigned)LD * 8u : 0u, nstep = valid ? 8u : 0u;
    const lds_addr_t hi = valid ? base + (unsigned)((j0 + m) * LD) * 8u : ~(lds_addr_t)0;
    const lds_addr_t span = valid ? (unsigned)(m * LD) * 8u : 0u, nspan = valid ? (unsigned)m * 8u : 0u;
    for (int j = 0; j < m; ++j) {
        auto* cq = lds_double_at(addr);
        auto* nq = lds_double_at(naddr);
        double y[RPL];
        double gg = 0.0, gg2 = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            y[i] = cq[i * GS];
            if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
        }
        const double bb = *nq;
        gg = group_sum<GS>(gg + gg2);
        const double g2 = gg * gg, ab = a * bb;
        if (g2 > skip2 * ab) {
            const double dd = bb - a;
            double tt, c, sn;
            jacobi_rotation(gg, g2, dd, tt, c, sn);
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const double xn = c * x[i] - sn * y[i];
                cq[i * GS] = sn * x[i] + c * y[i];
                x[i] = xn;
            }
            if (sub == 0) {
                *nq = bb + tt * gg;
                if (g2 > exit2 * ab) lds_or(flag, 1);
            }
            a -= tt * gg;
        }
        const lds_addr_t a1 = addr + dstep, n1 = naddr + nstep;
        const bool wrap = a1 >= hi;
        addr = wrap ? a1 - span : a1;
        naddr = wrap ? n1 - nspan : n1;
        wave_sync_lds();
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) cp[sub + i * GS] = x[i];
    if (sub == 0) nrm[pc] = a;
    wave_sync_lds();
}
#endif

__global__ __launch_bounds__(192) void k(double* out, int passes, double skip2, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Bm = lds;
    double* nrm = lds + NCOLS * LDJ;
    int* flag = (int*)(nrm + 64);
    const int t = threadIdx.x, lane = t & 63, wave = t / 64;
    unsigned s = 12345u + 977u * t + 31u * blockIdx.x;
    for (int i = t; i < NCOLS * LDJ; i += 192) {
        s = s * 1664525u + 1013904223u;
        Bm[i] = (i / LDJ < 48 && i % LDJ < 48) ? ((double)(s >> 8) / 16777216.0 - 0.5) + ((i / LDJ == i % LDJ) ? 2.0 + 0.1 * (i / LDJ) : 0.0) : 0.0;
    }
    __syncthreads();
    for (int c = t; c < NCOLS; c += 192) {
        double a = 0.0;
        for (int r = 0; r < 48; ++r) a += Bm[c * LDJ + r] * Bm[c * LDJ + r];
        nrm[c] = a;
    }
    if (t == 0) *flag = 0;
    __syncthreads();
    const long long c0 = clock64();
    for (int p = 0; p < passes; ++p)
#ifdef USE_CARRIED
        rotate_cross<GS, RPL>(Bm, LDJ, 48, M, 16 * wave, 16 * wave + 8, lane / GS, lane % GS, nrm, flag, skip2, 1e-15);
#else   // the kernel's step: cross pairs of two blocks, the I column resident in registers
        rotate_resident<GS, RPL>(Bm, LDJ, 48, M, M, lane / GS, lane % GS, nrm, flag, skip2, 1e-15,
            [&](int ps) { return 16 * wave + ps; },
            [&](int ps, int j) { int bq = ps + j; if (bq >= M) bq -= M; return 16 * wave + 8 + bq; });
#endif
    const long long c1 = clock64();
    __syncthreads();
    double acc = 0.0;
    for (int i = t; i < 48 * LDJ; i += 192) acc += Bm[i];
    out[blockIdx.x * 192 + t] = acc + *flag;
    if (t == 0 && blockIdx.x == 0) *cycles = c1 - c0;
}

int main(int argc, char** argv) {
    const int passes = 200;
    double* d; long long* dc;
    hipMalloc(&d, sizeof(double) * 192 * 256 * 8 * 16);
    hipMalloc(&dc, 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char* names[2] = {"never rotate (skipped step)", "always rotate"};
    printf("# one inner step of rotate_resident<8, 6> (48-column item, three wavefronts per workgroup), small-angle threshold %g\n", (double)SMRT_JACOBI_SMALL_ANGLE);
    printf("# %-28s %9s %12s %22s %30s\n", "path", "wg / CU", "ms", "clk / step of a wave", "SIMD clk per wavefront step");
    for (int mode = 0; mode < 2; ++mode)
        for (int per_cu = 1; per_cu <= 7; ++per_cu) {
            const size_t lds = (160 * 1024 / per_cu) & ~(size_t)1023;   // so that exactly per_cu workgroups fit a CU
            const int grid = 256 * per_cu * 16;                         // sixteen resident rounds: the steady state
            const double skip2 = mode == 0 ? 1e300 : -1.0;
            hipLaunchKernelGGL(k, dim3(grid), dim3(192), lds, 0, d, 20, skip2, dc);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(192), lds, 0, d, passes, skip2, dc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            long long cyc = 0; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
            const double steps = (double)passes * M;
            // throughput: 1024 SIMDs x 2.4 GHz x time / (wavefronts x steps)
            printf("  %-28s %9d %12.3f %22.1f %30.1f\n", names[mode], per_cu, ms, (double)cyc / steps,
                   1024.0 * 2.4e6 * ms / ((double)grid * 3 * steps));
        }
    return 0;
}
