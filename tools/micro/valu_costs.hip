// Micro-benchmark: cost in shader cycles of the cross-lane and f64 instructions the 16 x 16 eliminations are made of, one
// wavefront per SIMD (grid = CUs, 256 threads) and ONE wavefront alone (64 threads), dependent chains and independent streams.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/micro/valu_costs tools/micro/valu_costs.hip && tools/micro/valu_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 256
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
    double x = seed + threadIdx.x * 1e-3, y = seed * 0.5 + threadIdx.x * 1e-4, z = 1.0 + threadIdx.x * 1e-5, w = 2.0;
    union U { double d; unsigned i[2]; };
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 16; ++r) {
            if (MODE == 0) { x = __builtin_fma(x, 1.0000001, 1e-9); }                                   // dependent f64 fma chain
            if (MODE == 1) { x = __builtin_fma(x, 1.0000001, 1e-9); y = __builtin_fma(y, 1.0000001, 1e-9); z = __builtin_fma(z, 1.0000001, 1e-9); w = __builtin_fma(w, 1.0000001, 1e-9); }   // 4 independent
            if (MODE == 2) { U a; a.d = x; a.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x150 + 3, 0xF, 0xF, false); a.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x150 + 3, 0xF, 0xF, false); x = a.d + 1e-9; }   // dpp bcast + add, dependent
            if (MODE == 3) { U a; a.d = x; auto s0 = __builtin_amdgcn_permlane16_swap(a.i[0], a.i[0], false, false); auto s1 = __builtin_amdgcn_permlane16_swap(a.i[1], a.i[1], false, false); a.i[0] = s0[0]; a.i[1] = s1[0]; x = a.d + 1e-9; }   // permlane16 swap x2 + add, dependent
            if (MODE == 4) { U a; a.d = x; auto s0 = __builtin_amdgcn_permlane16_swap(a.i[0], a.i[0], false, false); auto s1 = __builtin_amdgcn_permlane16_swap(a.i[1], a.i[1], false, false);
                             auto q0 = __builtin_amdgcn_permlane32_swap(s0[0], s0[0], false, false); auto q1 = __builtin_amdgcn_permlane32_swap(s1[0], s1[0], false, false); a.i[0] = q0[0]; a.i[1] = q1[0]; x = a.d + 1e-9; }   // full row broadcast, dependent
            if (MODE == 5) { x = __builtin_amdgcn_rcp(x) + 1.5; }                                           // rcp + add, dependent
            if (MODE == 6) { double yv = __builtin_amdgcn_rcp(x); double e = __builtin_fma(-x, yv, 1.0); yv = __builtin_fma(yv, e, yv); e = __builtin_fma(-x, yv, 1.0); x = __builtin_fma(yv, e, yv) + 1.5; }   // fast_rcp + add
            if (MODE == 7) { U a; a.d = y; auto s0 = __builtin_amdgcn_permlane16_swap(a.i[0], a.i[0], false, false); auto s1 = __builtin_amdgcn_permlane16_swap(a.i[1], a.i[1], false, false);
                             auto q0 = __builtin_amdgcn_permlane32_swap(s0[0], s0[0], false, false); auto q1 = __builtin_amdgcn_permlane32_swap(s1[0], s1[0], false, false); a.i[0] = q0[0]; a.i[1] = q1[0]; z += a.d;
                             x = __builtin_fma(x, 1.0000001, 1e-9); }   // row broadcast of an INDEPENDENT value beside a dependent fma chain
            if (MODE == 8) { union { double d; int i[2]; } a, r2; a.d = x; r2.i[0] = __builtin_amdgcn_ds_bpermute(((threadIdx.x & 15) | 16) << 2, a.i[0]); r2.i[1] = __builtin_amdgcn_ds_bpermute(((threadIdx.x & 15) | 16) << 2, a.i[1]); x = r2.d + 1e-9; }   // ds_bpermute row bcast, dependent
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* what, int threads) {
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 8);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-78s %3d threads: %7.1f cycles per repetition\n", what, threads, s / 256 / REP);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {64, 256}) {
        run<0>("f64 fma, dependent chain", th);
        run<1>("4 independent f64 fma (per group of four)", th);
        run<2>("DPP row_newbcast of a double (2 v_mov_dpp) + add, dependent", th);
        run<3>("v_permlane16_swap x 2 + add, dependent", th);
        run<4>("row to all lane rows (permlane16_swap x 2, permlane32_swap x 2) + add, dependent", th);
        run<5>("v_rcp_f64 + add, dependent", th);
        run<6>("v_rcp_f64 + two Newton steps + add, dependent", th);
        run<7>("row broadcast of an independent value beside a dependent fma", th);
        run<8>("ds_bpermute x 2 + add, dependent", th);
    }
    return 0;
}
