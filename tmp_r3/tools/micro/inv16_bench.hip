// Micro-benchmark for the next Gauss-Jordan panel design: in-place inversion of a 16 x 16
// block with partial pivoting by ONE wavefront, two register layouts:
//   A  lane = row (16 lanes busy), 16 entries per lane, pivot row broadcast by v_readlane (the style of gj_panel16)
//   B  lane = (row, column group): 16 x 4 lanes, 4 entries per lane, row arg-max by DPP inside the 16-lane rows,
//      pivot row / multiplier column moved by ds_bpermute (__shfl)
// Prints cycles per inversion (clock64 around the routine, mean over repetitions) for a grid of one-wavefront
// workgroups, and checks P * P^-1 = I on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/inv16_bench tools/micro/inv16_bench.hip && tools/micro/inv16_bench [grid]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ double bcast(double v, int lane) {   // uniform lane index
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// max over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ unsigned row_max_u32(unsigned k) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0xB1, 0xF, 0xF, false); k = o > k ? o : k;   // quad_perm [1,0,3,2]
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x4E, 0xF, 0xF, false); k = o > k ? o : k;   // quad_perm [2,3,0,1]
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x141, 0xF, 0xF, false); k = o > k ? o : k;  // row_half_mirror
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x140, 0xF, 0xF, false); k = o > k ? o : k;  // row_mirror
    return k;
}

// ---- layout A: lane = row r (< 16), a[c] = P[r][c]; implicit pivoting; Z[p_k][:] ends up as row k of (QP)^-1
__device__ __forceinline__ void inv16_rows(double (&a)[16], int lane, int (&piv)[16]) {
    bool used = lane >= 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        unsigned key = 0u;
        if (!used) { const float f = (float)fabs(a[k]); memcpy(&key, &f, 4); key = (key & ~0xFu) | (unsigned)(15 - lane) | 0x10u; }
        key = row_max_u32(key);
        const int p = 15 - (int)((unsigned)__builtin_amdgcn_readlane((int)key, 0) & 0xFu);
        piv[k] = p;
        const double rpv = rcp_nr(bcast(a[k], p));
        const bool isp = lane == p;
        if (isp) used = true;
        const double f = a[k];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double pr = (c == k) ? rpv : bcast(a[c], p) * rpv;
            const double base = (c == k) ? 0.0 : a[c];
            a[c] = isp ? pr : __builtin_fma(-f, pr, base);
        }
    }
}

// ---- layout B: lane = (r = lane & 15, g = lane >> 4), x[s] = P[r][4 g + s]
__device__ __forceinline__ void inv16_grid(double (&x)[4], int lane, int (&piv)[16]) {
    const int r = lane & 15, g = lane >> 4;
    bool used = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int gk = k >> 2, q = k & 3;
        unsigned key = 0u;
        if (!used && g == gk) { const float f = (float)fabs(x[q]); memcpy(&key, &f, 4); key = (key & ~0xFu) | (unsigned)(15 - r) | 0x10u; }
        key = row_max_u32(key);
        const int p = 15 - (int)((unsigned)__builtin_amdgcn_readlane((int)key, 16 * gk) & 0xFu);
        piv[k] = p;
        const double rpv = rcp_nr(bcast(x[q], 16 * gk + p));
        const bool isp = r == p;
        if (isp) used = true;
        const double f = __shfl(x[q], r + 16 * gk, 64);      // multiplier of this lane's row
        double pr[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) pr[s] = __shfl(x[s], p + 16 * g, 64) * rpv;   // scaled pivot row, own columns
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool pc = (g == gk) && (s == q);
            const double prs = pc ? rpv : pr[s];
            const double base = pc ? 0.0 : x[s];
            x[s] = isp ? prs : __builtin_fma(-f, prs, base);
        }
    }
}

template <int VAR>
__global__ __launch_bounds__(64) void k(const double* src, double* out, int* pout, long long* cyc, int reps) {
    const int lane = threadIdx.x;
    long long tot = 0;
    double a[16];
    double x[4];
    int piv[16];
    for (int rep = 0; rep < reps; ++rep) {
        if (VAR == 0) { for (int c = 0; c < 16; ++c) a[c] = src[(lane & 15) * 16 + c] + 1e-3 * rep * (c == (lane & 15)); }
        else { for (int s = 0; s < 4; ++s) { const int c = 4 * (lane >> 4) + s; x[s] = src[(lane & 15) * 16 + c] + 1e-3 * rep * (c == (lane & 15)); } }
        __builtin_amdgcn_s_waitcnt(0);
        const long long t0 = clock64();
        if (VAR == 0) inv16_rows(a, lane, piv); else inv16_grid(x, lane, piv);
        __builtin_amdgcn_sched_barrier(0);
        tot += clock64() - t0;
        if (rep == 0 && blockIdx.x == 0) {
            if (VAR == 0) { if (lane < 16) for (int c = 0; c < 16; ++c) out[lane * 16 + c] = a[c]; }
            else for (int s = 0; s < 4; ++s) out[(lane & 15) * 16 + 4 * (lane >> 4) + s] = x[s];
            if (lane == 0) for (int c = 0; c < 16; ++c) pout[c] = piv[c];
        }
    }
    if (lane == 0 && blockIdx.x == 0) cyc[0] = tot / reps;
}

static double check(const double* P, const double* Z, const int* piv) {
    // W[k][:] = Z[p_k][:] = (P^-1 Q^T)[k][:]  ->  P^-1[k][p_j] = W[k][j]
    double inv[256];
    for (int kk = 0; kk < 16; ++kk)
        for (int j = 0; j < 16; ++j) inv[kk * 16 + piv[j]] = Z[piv[kk] * 16 + j];
    double worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int m = 0; m < 16; ++m) s += P[i * 16 + m] * inv[m * 16 + j];
            worst = fmax(worst, fabs(s - (i == j)));
        }
    return worst;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1;
    double h[256];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) h[i * 16 + j] = ((i + 5) % 16 == j ? 3.0 : 0.0) + sin(1.0 + i * 0.37 + j * 1.91);   // needs pivoting
    double *dsrc, *dout; int* dp; long long* dcyc;
    hipMalloc(&dsrc, sizeof(h)); hipMalloc(&dout, sizeof(h)); hipMalloc(&dp, 64); hipMalloc(&dcyc, 8);
    hipMemcpy(dsrc, h, sizeof(h), hipMemcpyHostToDevice);
    printf("grid = %d workgroups of one wavefront\n", grid);
    for (int var = 0; var < 2; ++var) {
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, dsrc, dout, dp, dcyc, 50);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, dsrc, dout, dp, dcyc, 50);
        hipDeviceSynchronize();
        double z[256]; int piv[16]; long long c;
        hipMemcpy(z, dout, sizeof(z), hipMemcpyDeviceToHost); hipMemcpy(piv, dp, 64, hipMemcpyDeviceToHost); hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %6lld cycles / inversion = %4.0f / column   |P P^-1 - I| = %.1e\n",
               var == 0 ? "A  lane = row, readlane broadcasts" : "B  16 x 4 lanes, ds_bpermute", c, c / 16.0, check(h, z, piv));
    }
    return 0;
}
