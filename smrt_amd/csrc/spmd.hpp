// SPMD primitives used by the DORT device code.
//
// On the GPU (hipcc, gfx950) these map 1:1 onto the CDNA4 execution model: a workgroup of NT threads made of
// 64-lane wavefronts, LDS, s_barrier, and DPP/bpermute cross-lane moves.
//
// With -DSMRT_HOST_EMU (g++, tests only) the same source runs under a deterministic fiber emulator
// (tests/hostemu/emu_runtime.hpp): every "thread" is a ucontext fiber, barriers and shuffles are the only yield
// points.  That build exists so that kernel logic can be unit-tested and race-checked (fibers scheduled in
// forward and reverse order must give identical results) in a container without a GPU.  The product never
// loads it.
#pragma once

#if defined(SMRT_HOST_EMU)

#include <cmath>
#include <cstring>
#include "emu_runtime.hpp"
#define SMRT_DEV inline
#define SMRT_DEV_NOINLINE inline
#define SMRT_LANES 64
namespace smrt {
SMRT_DEV int tid() { return emu::tid(); }
SMRT_DEV void block_sync() { emu::block_barrier(); }
SMRT_DEV void wave_sync() { emu::wave_barrier(); }
SMRT_DEV void wave_sync_lds() { emu::wave_barrier(); }
SMRT_DEV double fast_rcp1(double x) { return 1.0 / x; }
SMRT_DEV double fast_rsqrt1(double x) { return 1.0 / std::sqrt(x); }
SMRT_DEV double shfl_xor(double v, int mask) { return emu::shfl_xor(v, mask); }
SMRT_DEV int shfl_xor(int v, int mask) { return (int)emu::shfl_xor((double)v, mask); }
SMRT_DEV long long cycle_counter() { return 0; }
// max of a 64-bit key over the 64 lanes of the wavefront, result in every lane
SMRT_DEV unsigned long long wave_max_u64(unsigned long long k) {
    for (int m = 32; m >= 1; m >>= 1) {
        double d; std::memcpy(&d, &k, 8);
        d = emu::shfl_xor(d, m);
        unsigned long long o; std::memcpy(&o, &d, 8);
        if (o > k) k = o;
    }
    return k;
}
SMRT_DEV unsigned wave_max_u32(unsigned k) {
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned o = (unsigned)emu::shfl_xor((double)k, m);
        if (o > k) k = o;
    }
    return k;
}
SMRT_DEV double wave_bcast(double v, int src_lane) { return emu::wave_bcast(v, src_lane); }
// value of lane src_lane (any lane index, different per lane) of the wavefront
SMRT_DEV double wave_shfl(double v, int src_lane) { return emu::wave_bcast(v, src_lane); }
SMRT_DEV unsigned wave_bcast_u32(unsigned v, int src_lane) { return (unsigned)emu::wave_bcast((double)v, src_lane); }
// max over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane of the row
SMRT_DEV unsigned row16_max_u32(unsigned k) {
    for (int m = 8; m >= 1; m >>= 1) {
        unsigned o = (unsigned)emu::shfl_xor((double)k, m);
        if (o > k) k = o;
    }
    return k;
}
SMRT_DEV void mfma_f64_16x16x4(double a, double b, double (&c)[4]) { emu::mfma_f64_16x16x4(a, b, c); }
// value of lane K of the caller's 16-lane row (lanes 16 g .. 16 g + 15), K a compile-time constant
template <int K>
SMRT_DEV double row_bcast16(double v) { return emu::wave_bcast(v, (emu::tid() & 48) | K); }
// value of lane 16 G0 + c in every lane 16 g + c (the 16-lane row G0 copied to all four rows), G0 a compile-time constant
template <int G0>
SMRT_DEV double rows_bcast(double v) { return emu::wave_bcast(v, 16 * G0 + (emu::tid() & 15)); }
// a 16 x 16 tile in MFMA accumulator layout: four doubles per lane (one register tuple on the GPU)
struct tile4 {
    double x[4];
    double& operator[](int i) { return x[i]; }
    const double& operator[](int i) const { return x[i]; }
};
SMRT_DEV tile4 tile_zero() { tile4 t; t.x[0] = t.x[1] = t.x[2] = t.x[3] = 0.0; return t; }
SMRT_DEV void mfma_tile(double a, double b, tile4& c) { emu::mfma_f64_16x16x4(a, b, c.x); }
SMRT_DEV double fast_rcp(double x) { return 1.0 / x; }
SMRT_DEV double fast_rsqrt(double x) { return 1.0 / std::sqrt(x); }
// sum over aligned groups of GS consecutive lanes (GS power of two <= 64); every lane gets the group total
template <int GS>
SMRT_DEV double group_sum(double v) {
    for (int m = GS / 2; m >= 1; m >>= 1) v += emu::shfl_xor(v, m);
    return v;
}
SMRT_DEV void lds_or(int* p, int v) { *p |= v; }
SMRT_DEV void lds_max(int* p, int v) { if (v > *p) *p = v; }
SMRT_DEV void gmem_max(int* p, int v) { if (v > *p) *p = v; }
SMRT_DEV void gmem_add(double* p, double v) { *p += v; }
}  // namespace smrt

#else

#include <hip/hip_runtime.h>
#define SMRT_DEV __device__ __forceinline__
#define SMRT_DEV_NOINLINE __device__ __noinline__
#define SMRT_LANES 64
namespace smrt {
SMRT_DEV int tid() { return threadIdx.x; }
SMRT_DEV void block_sync() { __syncthreads(); }
// All lanes of a wavefront execute in lockstep and their LDS operations are issued in program order; this only
// stops the compiler from moving LDS traffic across the point (and is a real rendezvous in the emulator).
SMRT_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
SMRT_DEV double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
SMRT_DEV int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
SMRT_DEV long long cycle_counter() { return (long long)clock64(); }
// Same-wavefront LDS hand-off: the LDS unit processes the DS instructions of one wavefront in issue order, so a
// load issued after a store of the same wavefront sees it without draining lgkmcnt; only the compiler must not
// move DS traffic across this point.
SMRT_DEV void wave_sync_lds() { __builtin_amdgcn_wave_barrier(); }
// one Newton step (about 1e-13 relative): enough wherever only a rotation ANGLE depends on the value
SMRT_DEV double fast_rcp1(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
SMRT_DEV double fast_rsqrt1(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-0.5 * x * y, y, 0.5);
    return __builtin_fma(y, e, y);
}
// v_rcp_f64 / v_rsq_f64 seeds refined by two Newton steps (full double accuracy for normal operands; no
// denormal / inf fix-up, which the callers do not need).  Replaces the ~30-instruction IEEE division sequences
// that sat on the critical path of every factorisation step.
SMRT_DEV double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
SMRT_DEV double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    return __builtin_fma(y, e, y);
}
// DPP cross-lane move of a double (two 32-bit DPP movs); CTRL is a DPP control word.  mov_dpp, not update_dpp(0, ...):
// with full row / bank masks every lane is written, and an `old` operand of 0 costs a v_mov_b32 per DPP move (six
// instructions per 8-lane group sum in the Jacobi step)
template <int CTRL>
SMRT_DEV double dpp_move(double v) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], CTRL, 0xF, 0xF, false);
    r.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], CTRL, 0xF, 0xF, false);
    return r.d;
}
// value of lane K of the caller's 16-lane row, K a compile-time constant: DPP row_newbcast (gfx90a and later), two
// v_mov_b32_dpp, no LDS traffic
// (ONE v_mov_b64_dpp: gfx90a and later take 64-bit operands under this DPP control -- and only this one --, which halves the
// cross-lane instructions of the in-register eliminations; a wavefront issues about one vector instruction every 4.5
// cycles whatever it is, tools/micro/inv16_cost.hip)
template <int K>
SMRT_DEV double row_bcast16(double v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xF, 0xF, false); }
// value of lane 16 G0 + c in every lane 16 g + c (the 16-lane row G0 copied to all four rows), G0 a compile-time constant:
// v_permlane16_swap / v_permlane32_swap (gfx950) -- the first makes rows {0, 1} and {2, 3} equal, the second the halves
template <int G0>
SMRT_DEV double rows_bcast(double v) {
    union { double d; unsigned i[2]; } a, r;
    a.d = v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        auto s16 = __builtin_amdgcn_permlane16_swap(a.i[k], a.i[k], false, false);   // [x0 x0 x2 x2], [x1 x1 x3 x3]
        const unsigned y = (G0 & 1) ? s16[1] : s16[0];
        auto s32 = __builtin_amdgcn_permlane32_swap(y, y, false, false);             // [y0 y1 y0 y1], [y2 y3 y2 y3]
        r.i[k] = (G0 >> 1) ? s32[1] : s32[0];
    }
    return r.d;
}
template <int CTRL>
SMRT_DEV unsigned long long dpp_move_u64(unsigned long long v) {
    union { unsigned long long u; int i[2]; } a, r;
    a.u = v;
    r.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], CTRL, 0xF, 0xF, false);
    r.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], CTRL, 0xF, 0xF, false);
    return r.u;
}
SMRT_DEV unsigned long long readlane_u64(unsigned long long v, int lane) {
    union { unsigned long long u; int i[2]; } a, r;
    a.u = v;
    r.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return r.u;
}
// max of a 64-bit key over the 64 lanes of the wavefront, result in every lane: four DPP steps give every
// 16-lane row its maximum, four readlanes combine the rows.
SMRT_DEV unsigned long long wave_max_u64(unsigned long long k) {
    unsigned long long o;
    o = dpp_move_u64<0xB1>(k); k = o > k ? o : k;
    o = dpp_move_u64<0x4E>(k); k = o > k ? o : k;
    o = dpp_move_u64<0x141>(k); k = o > k ? o : k;
    o = dpp_move_u64<0x140>(k); k = o > k ? o : k;
    const unsigned long long r0 = readlane_u64(k, 0), r1 = readlane_u64(k, 16), r2 = readlane_u64(k, 32),
                             r3 = readlane_u64(k, 48);
    const unsigned long long a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}
// value of lane src_lane (wavefront-uniform index) in every lane: two v_readlane_b32
SMRT_DEV double wave_bcast(double v, int src_lane) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_readlane(a.i[0], src_lane);
    r.i[1] = __builtin_amdgcn_readlane(a.i[1], src_lane);
    return r.d;
}
// value of lane src_lane of the wavefront, src_lane different per lane: ds_bpermute_b32 x 2
SMRT_DEV double wave_shfl(double v, int src_lane) { return __shfl(v, src_lane, 64); }
SMRT_DEV unsigned wave_bcast_u32(unsigned v, int src_lane) { return (unsigned)__builtin_amdgcn_readlane((int)v, src_lane); }
// max over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane of the row: four DPP steps
SMRT_DEV unsigned row16_max_u32(unsigned k) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0xB1, 0xF, 0xF, false); k = o > k ? o : k;   // quad_perm [1,0,3,2]
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x4E, 0xF, 0xF, false); k = o > k ? o : k;   // quad_perm [2,3,0,1]
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x141, 0xF, 0xF, false); k = o > k ? o : k;  // row_half_mirror
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x140, 0xF, 0xF, false); k = o > k ? o : k;  // row_mirror
    return k;
}
// D = A(16x4) B(4x16) + C on the matrix core, v_mfma_f64_16x16x4_f64.  Lane layout (pinned on gfx950 by
// tools/micro/mfma_f64_layout.hip): a = A[i = l&15][k = l>>4], b = B[k = l>>4][j = l&15],
// c[reg] = C[row = (l>>4) + 4*reg][col = l&15].
typedef double smrt_v4d __attribute__((ext_vector_type(4)));
SMRT_DEV void mfma_f64_16x16x4(double a, double b, double (&c)[4]) {
    smrt_v4d cv = {c[0], c[1], c[2], c[3]};
    cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cv, 0, 0, 0);
    c[0] = cv[0]; c[1] = cv[1]; c[2] = cv[2]; c[3] = cv[3];
}
// a 16 x 16 tile in MFMA accumulator layout: four doubles per lane, kept as ONE 256-bit register tuple (the accumulator
// operand of v_mfma_f64_16x16x4_f64 as it is; its elements are directly addressable 64-bit sub-registers)
typedef smrt_v4d tile4;
SMRT_DEV tile4 tile_zero() { tile4 t = {0.0, 0.0, 0.0, 0.0}; return t; }
SMRT_DEV void mfma_tile(double a, double b, tile4& c) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// max of a 32-bit key over the wavefront with DPP only: four steps inside the 16-lane rows, then row_bcast15 /
// row_bcast31 carry the row maxima across (lane 63 ends up with the maximum of all 64), one readlane
SMRT_DEV unsigned wave_max_u32(unsigned k) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0xB1, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x4E, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x141, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x140, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x142, 0xA, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x143, 0xC, 0xF, false); k = o > k ? o : k;
    return (unsigned)__builtin_amdgcn_readlane((int)k, 63);
}
SMRT_DEV unsigned wave_max_u32_rl(unsigned k) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0xB1, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x4E, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x141, 0xF, 0xF, false); k = o > k ? o : k;
    o = (unsigned)__builtin_amdgcn_mov_dpp((int)k, 0x140, 0xF, 0xF, false); k = o > k ? o : k;
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)k, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)k, 16),
                   r2 = (unsigned)__builtin_amdgcn_readlane((int)k, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)k, 48);
    const unsigned a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}
// Sum over aligned groups of GS consecutive lanes; every lane gets the group total.  quad_perm swaps inside a
// quad, row_half_mirror / row_mirror reach the other quad / the other half of a 16-lane row (the source lane
// already holds its own partial total, so a mirror works as well as a butterfly); 32 and 64 fall back to bpermute.
template <int GS>
SMRT_DEV double group_sum(double v) {
    if (GS >= 2) v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    if (GS >= 4) v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    if (GS >= 8) v += dpp_move<0x141>(v);  // row_half_mirror
    if (GS >= 16) v += dpp_move<0x140>(v); // row_mirror
    if (GS >= 32) v += __shfl_xor(v, 16, 64);
    if (GS >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
SMRT_DEV void lds_or(int* p, int v) { atomicOr(p, v); }
SMRT_DEV void lds_max(int* p, int v) { atomicMax(p, v); }
SMRT_DEV void gmem_max(int* p, int v) { atomicMax(p, v); }
SMRT_DEV void gmem_add(double* p, double v) { atomicAdd(p, v); }   // (profiling builds only)
}  // namespace smrt

#endif
