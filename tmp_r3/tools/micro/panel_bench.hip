// Micro-benchmark: cost per column of the wavefront-serial Gauss-Jordan panel (lane = row, 16 columns), with parts
// of it switched off, to see where the cycles go.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I smrt_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "dort_device.hpp"
using namespace smrt;

// full-wavefront max with DPP only (row_bcast15 / row_bcast31 carry the 16-lane row maxima across), one readlane
__device__ __forceinline__ unsigned wave_max_u32_bcast(unsigned k) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0xB1, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x4E, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x141, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x140, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x142, 0xA, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x143, 0xC, 0xF, false); k = o > k ? o : k;
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}

template <int VAR>
__device__ __forceinline__ void panel_var(double* A, int N, int LD, int lane, int* perm) {
    double a[16], u[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { a[j] = A[j * LD + lane]; u[j] = 0.0; }
    bool used = false;
    int pj_store = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int p;
        if (VAR & 1) {  // pivot search
            unsigned key = 0u;
            if (!used) {
                const float xr = (float)fabs(a[j]);
                memcpy(&key, &xr, 4);
                key = (key & ~0x7Fu) | (unsigned)(127 - lane);
            }
            key = (VAR & 16) ? wave_max_u32_bcast(key) : wave_max_u32(key);
            p = 127 - (int)(key & 0x7Fu);
        } else p = j;
        if (lane == j) pj_store = p;
        const bool isp = (lane == p);
        const double pvv = wave_bcast(a[j], p);
        const double rpv = (VAR & 8) ? fast_rcp(pvv) : __builtin_amdgcn_rcp(pvv);
        if (isp) used = true;
        const double uj = isp ? rpv - 1.0 : -(a[j] * rpv);
        if ((VAR & 2) && (VAR & 32)) {   // all broadcasts first, then all FMAs
            double pa[16], ti[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) { if (jj > j) pa[jj] = wave_bcast(a[jj], p); if (jj < j) ti[jj] = wave_bcast(u[jj], p); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) { if (jj > j) a[jj] = __builtin_fma(uj, pa[jj], a[jj]); if (jj < j) u[jj] = __builtin_fma(uj, ti[jj], u[jj]); }
            __builtin_amdgcn_sched_barrier(0);
        } else if (VAR & 2) {
#pragma unroll
            for (int jj = 0; jj < 16; ++jj)
                if (jj > j) { const double pa = wave_bcast(a[jj], p); a[jj] = __builtin_fma(uj, pa, a[jj]); }
        } else if (j < 15) { const double pa = wave_bcast(a[j + 1], p); a[j + 1] = __builtin_fma(uj, pa, a[j + 1]); }
        if ((VAR & 4) && !(VAR & 32)) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < j) { const double ti = wave_bcast(u[i], p); u[i] = __builtin_fma(uj, ti, u[i]); }
        }
        u[j] = uj;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) A[j * LD + lane] = u[j] + a[j];
    if (lane < 16) perm[lane] = pj_store;
}

template <int VAR>
__global__ __launch_bounds__(64) void k(const double* src, double* out, long long* cyc, int reps) {
    __shared__ double A[16 * 65];
    __shared__ int perm[16];
    const int lane = threadIdx.x;
    long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int j = 0; j < 16; ++j) A[j * 65 + lane] = src[j * 64 + lane] + 1e-3 * r;
        __syncthreads();
        const long long t0 = clock64();
        panel_var<VAR>(A, 64, 65, lane, perm);
        __syncthreads();
        tot += clock64() - t0;
    }
    if (blockIdx.x == 0) for (int j = 0; j < 16; ++j) out[j * 64 + lane] = A[j * 65 + lane];
    if (lane == 0 && blockIdx.x == 0) cyc[0] = tot / reps;
}

// the product's own panel routine (noinline device function, generic pointers)
template <bool TR>
__global__ __launch_bounds__(64) void kreal(const double* src, double* out, long long* cyc, int reps) {
    __shared__ double A[64 * 65];
    __shared__ int perm[96];
    __shared__ int rowblk[64];
    const int lane = threadIdx.x;
    long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int j = 0; j < 16; ++j) { if (TR) A[lane * 65 + j] = src[j * 64 + lane] + 1e-3 * r; else A[j * 65 + lane] = src[j * 64 + lane] + 1e-3 * r; }
        rowblk[lane] = -1;
        __syncthreads();
        const long long t0 = clock64();
        gj_panel16<TR, 1>(A, 64, 65, 0, lane, perm, rowblk);
        __syncthreads();
        tot += clock64() - t0;
    }
    if (blockIdx.x == 0) for (int j = 0; j < 16; ++j) out[j * 64 + lane] = A[j * 65 + lane];
    if (lane == 0 && blockIdx.x == 0) cyc[0] = tot / reps;
}
static int g_grid = 1;
template <bool TR>
void run_real(const char* name, const double* dsrc, double* dout, long long* dcyc) {
    hipLaunchKernelGGL(kreal<TR>, dim3(g_grid), dim3(64), 0, 0, dsrc, dout, dcyc, 50);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s %6lld cycles / panel  = %5.0f / column\n", name, c, c / 16.0);
}
template <int VAR>
void run(const char* name, const double* dsrc, double* dout, long long* dcyc) {
    hipLaunchKernelGGL(k<VAR>, dim3(g_grid), dim3(64), 0, 0, dsrc, dout, dcyc, 50);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s %6lld cycles / panel  = %5.0f / column\n", name, c, c / 16.0);
}

int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    printf("grid = %d workgroups of one wavefront\n", g_grid);
    double h[16 * 64];
    for (int j = 0; j < 16; ++j)
        for (int i = 0; i < 64; ++i) h[j * 64 + i] = (i == j ? 4.0 : 0.0) + sin(1.0 + i * 0.37 + j * 1.91);
    double *dsrc, *dout; long long* dcyc;
    hipMalloc(&dsrc, sizeof(h)); hipMalloc(&dout, sizeof(h)); hipMalloc(&dcyc, 8);
    hipMemcpy(dsrc, h, sizeof(h), hipMemcpyHostToDevice);
    run_real<false>("product gj_panel16<TR=false>", dsrc, dout, dcyc);
    run_real<true>("product gj_panel16<TR=true>", dsrc, dout, dcyc);
    run<15>("full (search + a-updates + u-tracking + NR rcp)", dsrc, dout, dcyc);
    run<31>("full, DPP-only wavefront max", dsrc, dout, dcyc);
    run<47>("full, broadcasts batched before the FMAs", dsrc, dout, dcyc);
    run<63>("full, both", dsrc, dout, dcyc);
    run<7>("raw v_rcp instead of Newton-refined", dsrc, dout, dcyc);
    run<14>("no pivot search (p = j)", dsrc, dout, dcyc);
    run<11>("no u-tracking", dsrc, dout, dcyc);
    run<13>("a-update of the next column only", dsrc, dout, dcyc);
    run<9>("search + rcp + next column only", dsrc, dout, dcyc);
    run<8>("rcp + next column only (p = j)", dsrc, dout, dcyc);
    return 0;
}
