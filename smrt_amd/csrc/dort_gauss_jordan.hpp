// 16-wide blocked Gauss-Jordan solves of the layer recursion.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include <type_traits>
#include "dort_dense.hpp"

namespace smrt {

// ---- 16-wide blocked Gauss-Jordan (N <= 128): ONE workgroup barrier per 16 columns ---------------------------------
// Gauss-Jordan with implicit partial pivoting: per block one wavefront factorises the panel (lane = row, arg-max over
// the rows not used yet by DPP on a 32-bit key; rows are never swapped, the permutation is undone once at the end) and
// tracks the columns u_j = T[:, p_j] - e_pj of the accumulated row transformation T, so that the whole block update is
// C <- C + U R_P with the ORIGINAL pivot rows R_P (no inverse to form); the pivot rows come out normalised, so after
// the last block row perm[k] of B is row k of the solution.  (An earlier version used 4-column blocks with a side
// buffer for U and a copy of the pivot rows, two barriers per block.)  Design points of this one:
//   * block width 16 = one MFMA tile column = four chained v_mfma_f64_16x16x4 per tile (the C tile is loaded and
//     stored once per 16 eliminated columns instead of once per 4);
//   * the multipliers u_j are written into the panel's own, now dead, columns of A -- no side buffer;
//   * the pivot rows of the running block are NOT touched by the tile updates (stores to them are masked), so they
//     can be read in place as the B operand by every wavefront; their own new values R_P + U_P R_P are computed as
//     one extra "virtual" tile per column tile, kept in registers across the block barrier and stored after it.
// Every wavefront owns fixed absolute column tiles of [A | B] for the whole solve, so the only cross-wavefront
// traffic per block is the panel (u columns, permutation, row states), published by the one barrier.  The panel of
// block k+1 is factorised by the owner of that column tile right after it has updated the tile (look-ahead).
// RPLN = rows per lane: 1 for N <= 64 (lane = row), 2 for N <= 128 (lane holds rows lane and lane + 64).
template <bool TR, int RPLN>
SMRT_DEV bool gj_panel16_impl(double* A, int N, int LD, int k, int lane, int* perm, int* rowblk) {
    // x[r][s] holds panel column s of row (lane + 64 r) until the column has been a pivot column, its multiplier u_s
    // afterwards: both kinds of slot receive the same update x[s] += u_j * x[s][pivot row], so a step treats all
    // slots but the pivot one alike.  The loop is unrolled by four only, with the slots rotated by four after every
    // group (the pivot slot index stays a compile-time constant): a fully unrolled panel is ~18 KB of straight-line
    // code that is executed once per call and does not live in the instruction cache next to the rest of the kernel.
    const int k0 = 16 * k;
    const int nbk = (N - k0 < 16) ? N - k0 : 16;
    double x[RPLN][16];
    bool used[RPLN], mine[RPLN];
#pragma unroll
    for (int r = 0; r < RPLN; ++r) {
        const int row = lane + 64 * r;
        const int rc = row < N ? row : N - 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cc = (k0 + j < N) ? k0 + j : N - 1;
            const double v = at<TR>(A, rc, cc, LD);
            x[r][j] = (row < N && j < nbk) ? v : 0.0;
        }
        used[r] = (row < N) ? (rowblk[rc] >= 0) : true;
#ifdef SMRT_GJ_DIAG_PIVOT
        // numerical experiment: pivots only from the 16 rows of the diagonal block -- what a panel
        // built from a 16 x 16 inverse and MFMA products would do
        if (row < k0 || row >= k0 + 16) used[r] = true;
#endif
        mine[r] = false;
    }
    bool ok = true;
    int pj_store = 0;
    int grp = 0;
    for (; grp * 4 < nbk; ++grp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = grp * 4 + q;
            if (j < nbk) {   // uniform
                // arg-max over the unused rows: float magnitude bits with (511 - row) in the 9 low mantissa bits (N <= 384)
                unsigned key = 0u;
#pragma unroll
                for (int r = 0; r < RPLN; ++r) {
                    if (!used[r]) {
                        const float xr = (float)fabs(x[r][q]);
                        unsigned kr;
                        memcpy(&kr, &xr, 4);
                        kr = (kr & ~0x1FFu) | (unsigned)(511 - (lane + 64 * r));
                        key = kr > key ? kr : key;
                    }
                }
                key = wave_max_u32(key);
                if (key < 512u) ok = false;
                const int p = ok ? 511 - (int)(key & 0x1FFu) : 0;
                if (lane == j) pj_store = p;
                const int pl = p & 63, ps = p >> 6;   // lane and slot of the pivot row (uniform)
                double pvq = x[0][q];
#pragma unroll
                for (int r2 = 1; r2 < RPLN; ++r2) pvq = (ps == r2) ? x[r2][q] : pvq;
                const double rpv = fast_rcp(ok ? wave_bcast(pvq, pl) : 1.0);
                double pr[16];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) {
                    if (s2 != q) {
                        double src = x[0][s2];
#pragma unroll
                        for (int r2 = 1; r2 < RPLN; ++r2) src = (ps == r2) ? x[r2][s2] : src;
                        pr[s2] = wave_bcast(src, pl);
                    }
                }
#if !defined(SMRT_HOST_EMU)
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int r = 0; r < RPLN; ++r) {
                    const bool isp = (lane == pl) && (r == ps);
                    if (isp) { used[r] = true; mine[r] = true; }
                    // the pivot row itself is scaled by 1/pivot: a - (1 - 1/pv) a = a / pv, i.e. the same update as
                    // every other row with the multiplier 1 - 1/pv
                    const double uj = isp ? rpv - 1.0 : -(x[r][q] * rpv);
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2)
                        if (s2 != q) x[r][s2] = __builtin_fma(uj, pr[s2], x[r][s2]);
                    x[r][q] = uj;
                }
            }
        }
        // rotate the slots left by four: slot s now holds what slot s + 4 held
#pragma unroll
        for (int r = 0; r < RPLN; ++r) {
            const double t0 = x[r][0], t1 = x[r][1], t2 = x[r][2], t3 = x[r][3];
#pragma unroll
            for (int s2 = 0; s2 < 12; ++s2) x[r][s2] = x[r][s2 + 4];
            x[r][12] = t0; x[r][13] = t1; x[r][14] = t2; x[r][15] = t3;
        }
    }
    // after grp rotations slot s holds column (s + 4 grp) mod 16
#pragma unroll
    for (int r = 0; r < RPLN; ++r) {
        const int row = lane + 64 * r;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int j = (s2 + 4 * grp) & 15;
            if (row < N && j < nbk) at<TR>(A, row, k0 + j, LD) = ok ? x[r][s2] : 0.0;
        }
        if (mine[r] && ok) rowblk[row] = k;
    }
    if (lane < nbk) perm[k0 + lane] = pj_store;
    return ok;
}
// Inlined into the kernels (it used to be a separate function to save code size): a callable function is compiled
// without the kernel's register cap, and the compiler then parks values in accumulation registers -- 32 AGPRs in the
// panel were enough to push the two-slot finish kernel from 256 to 288 registers, i.e. from two workgroups per CU to
// one (50 ms instead of 29 ms per step, profiles/r2_gj_fast_panel.txt).  The build records the compiler's resource
// remarks per kernel and tests/test_host_logic.py::test_kernel_occupancy_as_designed checks them.
template <bool TR, int RPLN>
SMRT_DEV bool gj_panel16(double* A, int N, int LD, int k, int lane, int* perm, int* rowblk) {
    return gj_panel16_impl<TR, RPLN>(A, N, LD, k, lane, perm, rowblk);
}

// ---- the fast panel: pivots from the 16 x 16 DIAGONAL block only ----------------------------------------------------
// With the eigenpairs of a layer sorted by their singular value (the Jacobi kernel does that on the way out), column c
// of the matrices of the layer recursion "belongs" to row c -- in the no-scattering limit they are diagonal -- and
// elimination with pivots restricted to the diagonal block has a growth factor of a few units
// (tests/studies/nopivot_elimination.py).  Then no search over all the rows and no 16 dependent rank-one updates of a
// lane-per-row panel are needed: with P = A[blk, blk] the accumulated transformation of the block is
//     U[rows of blk] = P^-1 - I,      U[other rows] = -A[:, blk] P^-1,
// i.e. one 16 x 16 inversion (partial pivoting INSIDE the block; the wavefront as a 16 x 4 grid of lanes, four entries
// of a row per lane, row arg-max by DPP inside the 16-lane rows, pivot row and multiplier column moved by ds_bpermute:
// tools/micro/inv16_bench.hip, 350-425 cycles per column against 1140 for a column of gj_panel16 in situ) and four
// chained MFMAs per row tile.  The unknowns of the block stay in its own rows: perm is the identity there.
// Acceptance test: every entry of U (and of P^-1) must be at most `growth_max` in magnitude; otherwise -- a block that
// needs a pivot row from outside (clusters of equal eigenvalues, whose eigenvectors are arbitrary rotations) -- nothing
// has been written, 0 is returned and the caller runs the full-pivot panel instead (from then on for every block of
// the solve: its pivot rows are no longer aligned with the blocks).
// Part one (a separate function, vector instructions only: the matrix-core part stays in the calling kernel, whose
// register budget it shares): invert the diagonal block into the LDS scratch `pinv` ([16][16], element (i, j) at
// pinv[16 j + i]); A is not touched.  Returns the largest magnitude of the inverse (1e301 for NaN / inf / a busy row).
template <bool TR>
SMRT_DEV_NOINLINE double gj_inv16_block(const double* A, int N, int LD, int k, int lane, const int* rowblk, double* pinv) {
    const int k0 = 16 * k;
    const int nbk = (N - k0 < 16) ? N - k0 : 16;
    const int r = lane & 15, g = lane >> 4;
    // the rows of the block must still be free
    if (wave_max_u32((r < nbk && rowblk[k0 + r] >= 0) ? 1u : 0u) != 0u) return 1e301;
    // P in the 16 x 4 grid layout, identity-padded for a ragged last block: x[s] = P[r][4 g + s]
    double x[4];
    const int rc = r < nbk ? r : 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * g + s, cc = c < nbk ? c : 0;
        const double v = at<TR>(const_cast<double*>(A), k0 + rc, k0 + cc, LD);
        x[s] = (r < nbk && c < nbk) ? v : ((r == c) ? 1.0 : 0.0);
    }
#ifdef SMRT_ABLATE_INV16
    {   // timing experiment only: "inverse" = reciprocal diagonal
#pragma unroll
        for (int s = 0; s < 4; ++s) pinv[16 * (4 * g + s) + r] = (r == 4 * g + s) ? 1.0 / x[s] : 0.0;
        wave_sync_lds();
        return 1.0;
    }
#endif
    // in-place Gauss-Jordan inversion with row pivoting inside the block: row piv[j] of the result is row j of the
    // inverse of the row-permuted block, P^-1[kk][piv[j]] = Z[piv[kk]][j]
    bool used = false;
    int myinv = r;          // the step at which this lane's row was the pivot row
    int piv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int gj = j >> 2, q = j & 3;
        unsigned key = 0u;
        if (!used && g == gj) {
            const float f = (float)fabs(x[q]);
            memcpy(&key, &f, 4);
            key = (key & ~0xFu) | (unsigned)(15 - r) | 0x10u;
        }
        key = row16_max_u32(key);
        const int p = 15 - (int)(wave_bcast_u32(key, 16 * gj) & 0xFu);   // uniform
        piv[j] = p;
        const double rpv = fast_rcp(wave_bcast(x[q], 16 * gj + p));
        const bool isp = (r == p);
        if (isp) { used = true; myinv = j; }
        const double f = wave_shfl(x[q], r + 16 * gj);          // multiplier of this lane's row
        double pr[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) pr[s] = wave_shfl(x[s], p + 16 * g) * rpv;   // scaled pivot row, own columns
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool pc = (g == gj) && (s == q);
            const double prs = pc ? rpv : pr[s];
            const double base = pc ? 0.0 : x[s];
            x[s] = isp ? prs : __builtin_fma(-f, prs, base);
        }
    }
    double amax = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pcol = (g == 0) ? piv[s] : (g == 1) ? piv[4 + s] : (g == 2) ? piv[8 + s] : piv[12 + s];
        pinv[16 * pcol + myinv] = x[s];
        const double m = fabs(x[s]);
        amax = (m <= 1e300) ? (m > amax ? m : amax) : 1e301;
    }
    wave_sync_lds();
    return amax;
}

// Part two (inline in the caller): the multiplier block of every other row tile, -A[:, blk] P^-1, by MFMA with the
// inverse as the B operand; acceptance test; U into the panel columns.  Returns 1 if the block was taken.
template <bool TR, int MAXRT>
SMRT_DEV int gj_panel16_fast(double* A, int N, int LD, int k, int lane, int* perm, int* rowblk, double* pinv,
                             double growth_max) {
    const int k0 = 16 * k;
    const int nbk = (N - k0 < 16) ? N - k0 : 16;
    const int RT = (N + 15) >> 4;
    double amax = gj_inv16_block<TR>(A, N, LD, k, lane, rowblk, pinv);
    const int lr = lane & 15, lk = lane >> 4;
    double c[MAXRT][4];
    {   // first vote: a busy row or a singular block needs no multiplier block at all
        const float fm = amax > 3e38 ? 3e38f : (float)amax;
        unsigned key;
        memcpy(&key, &fm, 4);
        key = wave_max_u32(key);
        float gm;
        memcpy(&gm, &key, 4);
        if (!((double)gm <= growth_max)) return 0;
        amax = (double)gm;
    }
#pragma unroll
    for (int ti = 0; ti < MAXRT; ++ti) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) c[ti][reg] = 0.0;
        if (ti < RT && ti != k) {   // uniform
            const int arow = ti * 16 + lr, arowc = arow < N ? arow : 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + lk, jc = j < nbk ? j : 0;
                const double av = at<TR>(A, arowc, k0 + jc, LD), bv = pinv[16 * lr + j];
                mfma_f64_16x16x4((arow < N && j < nbk) ? av : 0.0, (j < nbk && lr < nbk) ? bv : 0.0, c[ti]);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const double m = fabs(c[ti][reg]);
                amax = (m <= 1e300) ? (m > amax ? m : amax) : 1e301;
            }
        }
    }
    {   // second vote: the largest multiplier over the wavefront (float keys order like the values)
        const float fm = amax > 3e38 ? 3e38f : (float)amax;
        unsigned key;
        memcpy(&key, &fm, 4);
        key = wave_max_u32(key);
        float gm;
        memcpy(&gm, &key, 4);
        if (!((double)gm <= growth_max)) return 0;   // nothing has been written to A
    }
    // U into the panel columns: -A[:, blk] P^-1 outside the block, P^-1 - I inside
#pragma unroll
    for (int ti = 0; ti < MAXRT; ++ti) {
        if (ti < RT && ti != k) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = ti * 16 + lk + 4 * reg;
                if (row < N && lr < nbk) at<TR>(A, row, k0 + lr, LD) = -c[ti][reg];
            }
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int i = lk + 4 * reg;   // element (i, lr) of the block
        if (i < nbk && lr < nbk) at<TR>(A, k0 + i, k0 + lr, LD) = pinv[16 * lr + i] - (i == lr ? 1.0 : 0.0);
    }
    if (lane < nbk) {
        perm[k0 + lane] = k0 + lane;
        rowblk[k0 + lane] = k;
    }
    wave_sync_lds();
    return 1;
}

#if defined(SMRT_HOST_EMU)
inline long smrt_emu_panels[2] = {0, 0};   // emulator builds count [0] fast and [1] full-pivot panels (tests)
#define SMRT_COUNT_PANEL(i) do { if (lane == 0) ++smrt_emu_panels[i]; } while (0)
#else
#define SMRT_COUNT_PANEL(i) do {} while (0)
#endif

#ifndef SMRT_GJ_GROWTH_MAX
#define SMRT_GJ_GROWTH_MAX 64.0   // acceptance threshold of the fast panel on |U| (growth of block-diagonal pivoting)
#endif

// result_in_A: leave the solution in A (one pass and one barrier less than copying it back over Bm), optionally scaled
// X[k][c] * rs[k] * cs[c] on the way (the t Q t scaling of the recursion).
// BIG: the instantiation may meet N > 64 (two rows per lane in the full-pivot panel, up to eight row tiles in the fast
// one).  The LDS-resident kernels (N <= 64) pass false: the panels are separate (noinline) functions whose register
// footprint counts against the kernel's even when they are never called, and the two-workgroups-per-CU finish kernel
// has none to spare (<= 256 VGPRs).
#ifndef SMRT_GJ_TILES_IN_FLIGHT
#define SMRT_GJ_TILES_IN_FLIGHT 4
#endif
#ifndef SMRT_GJ_WIDE_TILES_IN_FLIGHT
#define SMRT_GJ_WIDE_TILES_IN_FLIGHT 4   // row tiles in flight in a grouped pass (2 | 3 | 4 measured: 2297 / 2291 / 2261 ms)
#endif
#ifndef SMRT_GJ_WIDE_BLOCKS
#define SMRT_GJ_WIDE_BLOCKS 2   // blocks of 16 columns whose updates are applied together when N > 128 (1: one by one; 2 | 3 | 4 measured)
#endif
template <int NT, bool TR, int CHN = 2>   // CHN = the kernel's CH: N <= 64 CHN
SMRT_DEV bool gj_solve_b16(double* A, double* Bm, double* v, const Lds& s, int N, int LD, bool result_in_A = false,
                           const double* rs = nullptr, const double* cs = nullptr, bool allow_fast = false) {
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const int NMX = s.gj_nmax;
    int* perm = (int*)s.gj;                     // [NMX + 16] pivot row of every column
    int* rowblk = perm + NMX + 16;              // [NMX] block in which the row was a pivot row, -1 before
    int* fail = rowblk + NMX;
    int* fast = fail + 1;                       // 1 while the blocks so far took their pivots from the diagonal block
    double* pinv = s.gj + (2 * NMX + 18 + 1) / 2 + 1;   // [256] inverse of the running diagonal block (fast panel)
    const bool has_v = (v != nullptr);
    const int RT = (N + 15) >> 4;
    const int lr = lane & 15, lk = lane >> 4;
    for (int r = t; r < NMX; r += NT) rowblk[r] = -1;
    // Measured on MI355X (profiles/r2_gj_fast_panel.txt): inside the finish kernel the diagonal-block panel costs as
    // much as the full-pivot one (its ds_bpermute traffic queues behind the LDS reads of the other wavefronts' tile
    // updates: ~1300 cycles per column in situ against 350-425 alone), so the default build keeps the full-pivot panel,
    // whose numerics do not depend on the order of the eigenpairs.  -DSMRT_GJ_FAST_PANEL builds take the fast one.
    // (Round 5, N = 384 with the eigenpairs sorted on the way in: a group of four blocks took 1.17 M cycles with the
    // fast panel against 0.93 M with the full-pivot one, profiles/r5_cfg3_gauss_jordan.txt.)
#ifdef SMRT_GJ_FAST_PANEL
    constexpr bool kFastPanel = true;
#else
    constexpr bool kFastPanel = false;
#endif
    if (!kFastPanel) allow_fast = false;
    if (t == 0) { *fail = 0; *fast = allow_fast ? 1 : 0; }
    block_sync();
#ifdef SMRT_STAGE_TIMING
    long long tg0 = cycle_counter();
#define SMRT_GSUB(k) do { const long long n_ = cycle_counter(); if (t == 0 && s.sub_acc) s.sub_acc[k] += (double)(n_ - tg0); tg0 = n_; } while (0)
#else
#define SMRT_GSUB(k) do {} while (0)
#endif
    auto panel = [&](int kb) -> bool {   // one wavefront; the flag was published by the previous block's barrier
#ifdef SMRT_ABLATE_PANEL
        {   // timing experiment only: identity pivots, no arithmetic
            const int k0 = 16 * kb;
            if (lane < 16 && k0 + lane < N) { perm[k0 + lane] = k0 + lane; rowblk[k0 + lane] = kb; }
            wave_sync_lds();
            return true;
        }
#endif
        if constexpr (kFastPanel) {
        if (*fast) {
            int took;
            if constexpr (CHN >= 2) took = (N > 64) ? gj_panel16_fast<TR, 4 * CHN>(A, N, LD, kb, lane, perm, rowblk, pinv, SMRT_GJ_GROWTH_MAX)
                                                    : gj_panel16_fast<TR, 4>(A, N, LD, kb, lane, perm, rowblk, pinv, SMRT_GJ_GROWTH_MAX);
            else took = gj_panel16_fast<TR, 4>(A, N, LD, kb, lane, perm, rowblk, pinv, SMRT_GJ_GROWTH_MAX);
            if (took) { SMRT_COUNT_PANEL(0); return true; }
            if (lane == 0) *fast = 0;
            wave_sync_lds();
        }
        }
        SMRT_COUNT_PANEL(1);
        if constexpr (CHN > 2) {          // rows lane + 64 r, r < CHN, of the panel in registers (inlined, see gj_panel16)
            if (N > 128) return gj_panel16_impl<TR, CHN>(A, N, LD, kb, lane, perm, rowblk);
            return (N > 64) ? gj_panel16<TR, 2>(A, N, LD, kb, lane, perm, rowblk) : gj_panel16<TR, 1>(A, N, LD, kb, lane, perm, rowblk);
        } else if constexpr (CHN == 2) {
            return (N > 64) ? gj_panel16<TR, 2>(A, N, LD, kb, lane, perm, rowblk) : gj_panel16<TR, 1>(A, N, LD, kb, lane, perm, rowblk);
        } else {
            return gj_panel16<TR, 1>(A, N, LD, kb, lane, perm, rowblk);
        }
    };
    if constexpr (CHN > 2 && SMRT_GJ_WIDE_BLOCKS > 1) {
    // ---- N > 128, matrices in the global workspace: the updates of SB = SMRT_GJ_WIDE_BLOCKS consecutive blocks are
    // applied to a column tile in ONE pass (rank 16 SB instead of SB passes of rank 16: the tile is read and written once
    // per 16 SB eliminated columns -- at N = 384 the rank-16 passes moved ~85 MB per solve through HBM and ran at its
    // bandwidth).  The blocks of a group are factorised one after the other by one wavefront, each applied at once to
    // the group's OTHER column tiles only -- the live ones to its right and the dead ones to its left, whose multiplier
    // columns U_q thereby become T_k U_q: with T_k = I + U_k E_k^T (E_k^T = the pivot rows of block k),
    //     T_2 T_1 M = M + (T_2 U_1) (E_1^T M) + U_2 (E_2^T M),
    // so after the group its columns hold U_tot and every other tile is M + U_tot (E^T M) with the pivot rows of M as
    // they were BEFORE the group.  The group after the running one is factorised (look-ahead) by the wavefront that owns
    // it, after it has brought that group's tiles up to date.
    constexpr int SB = SMRT_GJ_WIDE_BLOCKS;
    constexpr int PF1 = SMRT_GJ_TILES_IN_FLIGHT, PFW = SMRT_GJ_WIDE_TILES_IN_FLIGHT;
    // blocks [kf, kf + nb) applied to the absolute column tile g of [A | B] (NB = capacity of the operand arrays)
    auto apply = [&](auto nbcap, int g, int kf, int nb) {
        constexpr int NB = decltype(nbcap)::value;
        constexpr int PF = (NB == 1) ? PF1 : PFW;
        double* Mat = (g < RT) ? A : Bm;
        const int col = ((g < RT) ? g : g - RT) * 16 + lr;
        const bool cin = col < N;
        const int colc = cin ? col : 0;
        double bop[NB][4];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int k0q = 16 * (kf + q);
            const int nbq = (q < nb) ? ((N - k0q < 16) ? N - k0q : 16) : 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + lk;
                const int pr = (j < nbq) ? perm[k0q + j] : 0;
                const double x = at<TR>(Mat, pr, colc, LD);
                bop[q][kk] = (cin && j < nbq) ? x : 0.0;
            }
        }
        for (int t0 = 0; t0 < RT; t0 += PF) {
            double c[PF][4], av[PF][NB][4];
            bool keep[PF][4];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int ti = t0 + u;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = ti * 16 + lk + 4 * reg;
                    const int rowc = row < N ? row : 0;
                    const double x = at<TR>(Mat, rowc, colc, LD);
                    const int rb = rowblk[rowc];
                    keep[u][reg] = cin && row < N && !(rb >= kf && rb < kf + nb);
                    c[u][reg] = keep[u][reg] ? x : 0.0;
                }
                const int arow = ti * 16 + lr;
                const int arowc = arow < N ? arow : 0;
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int k0q = 16 * (kf + q);
                    const int nbq = (q < nb) ? ((N - k0q < 16) ? N - k0q : 16) : 0;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int j = 4 * kk + lk;
                        const int jc = (j < nbq) ? j : 0;
                        const double x = at<TR>(A, arowc, (nbq > 0 ? k0q : 0) + jc, LD);
                        av[u][q][kk] = (arow < N && j < nbq) ? x : 0.0;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) mfma_f64_16x16x4(av[u][q][kk], bop[q][kk], c[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = (t0 + u) * 16 + lk + 4 * reg;
                    if (keep[u][reg]) at<TR>(Mat, row, col, LD) = c[u][reg];
                }
            }
        }
        // the new pivot rows of every block of the group: R_P' + U_tot[P', :] R_P
        double pvt[NB][4];
#pragma unroll
        for (int q2 = 0; q2 < NB; ++q2) {
            const int k0p = 16 * (kf + q2);
            const int nbp = (q2 < nb) ? ((N - k0p < 16) ? N - k0p : 16) : 0;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lk + 4 * reg;
                const int pr = (j < nbp) ? perm[k0p + j] : 0;
                const double x = at<TR>(Mat, pr, colc, LD);
                pvt[q2][reg] = (cin && j < nbp) ? x : 0.0;
            }
            const int prl = (lr < nbp) ? perm[k0p + lr] : 0;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int k0q = 16 * (kf + q);
                const int nbq = (q < nb) ? ((N - k0q < 16) ? N - k0q : 16) : 0;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int j = 4 * kk + lk;
                    const int jc = (j < nbq) ? j : 0;
                    const double x = at<TR>(A, prl, (nbq > 0 ? k0q : 0) + jc, LD);
                    mfma_f64_16x16x4((lr < nbp && j < nbq) ? x : 0.0, bop[q][kk], pvt[q2]);
                }
            }
        }
        wave_sync_lds();
#pragma unroll
        for (int q2 = 0; q2 < NB; ++q2) {
            const int k0p = 16 * (kf + q2);
            const int nbp = (q2 < nb) ? ((N - k0p < 16) ? N - k0p : 16) : 0;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lk + 4 * reg;
                if (col < N && j < nbp) at<TR>(Mat, perm[k0p + j], col, LD) = pvt[q2][reg];
            }
        }
        wave_sync_lds();
    };
    using One = std::integral_constant<int, 1>;
    using Cap = std::integral_constant<int, SB>;
    // the blocks of group gi, one after the other, by the calling wavefront
    auto factor_group = [&](int gi) -> bool {
        const int kf = SB * gi, kl = (kf + SB < RT) ? kf + SB : RT;
        for (int kb = kf; kb < kl; ++kb) {
            if (!panel(kb)) return false;
            wave_sync_lds();
            for (int g = kf; g < kl; ++g)
                if (g != kb) apply(One{}, g, kb, 1);
        }
        return true;
    };
    const int NG = (RT + SB - 1) / SB;
    if (wave == 0) { if (!factor_group(0) && lane == 0) *fail = 1; }
    SMRT_GSUB(0);
    block_sync();
    if (*fail) return false;  // uniform
    for (int gi = 0; gi < NG; ++gi) {
        const int kf = SB * gi, kl = (kf + SB < RT) ? kf + SB : RT, nb = kl - kf;
        const bool has_next = gi + 1 < NG;
        const int owner = has_next ? ((gi + 1) % NW) : -1;
        const int kl2 = has_next ? ((kl + SB < RT) ? kl + SB : RT) : RT;
        if (has_next && wave == owner) {
            for (int g = kl; g < kl2; ++g) apply(Cap{}, g, kf, nb);
            if (!factor_group(gi + 1) && lane == 0) *fail = 1;
        }
        const bool worker = (NW == 1) || !has_next || wave != owner;
        const int nworkers = (NW == 1 || !has_next) ? NW : NW - 1;
        const int widx = (NW == 1 || !has_next) ? wave : (wave - owner - 1 + NW) % NW;
        if (worker) {
            int idx = 0;
            for (int g = kl2; g < 2 * RT; ++g, ++idx) {
                if (idx % nworkers != widx) continue;
                apply(Cap{}, g, kf, nb);
            }
            if (has_v && widx == 0) {  // extra right-hand side: v + U_tot v[P], rows lane + 64 r2
                double acc[CHN];
#pragma unroll
                for (int r2 = 0; r2 < CHN; ++r2) {
                    const int row = lane + 64 * r2;
                    acc[r2] = 0.0;
                    if (row < N) {
                        acc[r2] = v[row];
                        for (int c2 = 16 * kf; c2 < 16 * kl && c2 < N; ++c2) acc[r2] += at<TR>(A, row, c2, LD) * v[perm[c2]];
                    }
                }
                wave_sync_lds();
#pragma unroll
                for (int r2 = 0; r2 < CHN; ++r2)
                    if (lane + 64 * r2 < N) v[lane + 64 * r2] = acc[r2];
            }
        }
#ifdef SMRT_GJ_TIMING_SPLIT
        SMRT_GSUB(1);   // (experiment: [1] = wavefront 0's own work, [2] = its wait at the group barrier)
        block_sync();
        SMRT_GSUB(2);
#else
        block_sync();
#endif
        if (*fail) return false;  // uniform
    }
    } else {
    if (wave == 0) { if (!panel(0) && lane == 0) *fail = 1; }
    SMRT_GSUB(0);
    block_sync();
    if (*fail) return false;  // uniform

    for (int k = 0; k < RT; ++k) {
        const int k0 = 16 * k;
        const int nbk = (N - k0 < 16) ? N - k0 : 16;
        // one absolute column tile g of [A | B]: all row tiles (pivot rows masked) + the virtual pivot-row tile -> pvt
        auto do_tile = [&](int g, double (&pvt)[4]) {
            double* Mat = (g < RT) ? A : Bm;
            const int col = ((g < RT) ? g : g - RT) * 16 + lr;
            const bool cin = col < N;
            const int colc = cin ? col : 0;
            double bop[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + lk;
                const int pr = (j < nbk) ? perm[k0 + j] : 0;
                const double x = at<TR>(Mat, pr, colc, LD);
                bop[kk] = (cin && j < nbk) ? x : 0.0;
            }
            // PF row tiles in flight: with the matrices in the global workspace (CHN > 2) a tile's loads take a couple
            // of thousand cycles under load, and the compiler cannot move the next tile's loads above this tile's stores
            // (same array).  The loads of PF tiles are issued together, then their matrix-core passes, then the stores.
            constexpr int PF = (CHN > 2) ? SMRT_GJ_TILES_IN_FLIGHT : 1;
            for (int t0 = 0; t0 < RT; t0 += PF) {
                double c[PF][4], av[PF][4];
                bool keep[PF][4];
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int ti = t0 + u;   // (past the last row tile: nothing kept, zero operands)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = ti * 16 + lk + 4 * reg;
                        const int rowc = row < N ? row : 0;
                        const double x = at<TR>(Mat, rowc, colc, LD);
                        keep[u][reg] = cin && row < N && rowblk[rowc] != k;
                        c[u][reg] = keep[u][reg] ? x : 0.0;
                    }
                    const int arow = ti * 16 + lr;
                    const int arowc = arow < N ? arow : 0;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int j = 4 * kk + lk;
                        const int jc = (j < nbk) ? j : 0;
                        const double x = at<TR>(A, arowc, k0 + jc, LD);
                        av[u][kk] = (arow < N && j < nbk) ? x : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) mfma_f64_16x16x4(av[u][kk], bop[kk], c[u]);
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = (t0 + u) * 16 + lk + 4 * reg;
                        if (keep[u][reg]) at<TR>(Mat, row, col, LD) = c[u][reg];
                    }
                }
            }
            // new pivot rows: R_P + U_P R_P with U_P[j][i] = u_i[p_j]
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lk + 4 * reg;
                const int pr = (j < nbk) ? perm[k0 + j] : 0;
                const double x = at<TR>(Mat, pr, colc, LD);
                pvt[reg] = (cin && j < nbk) ? x : 0.0;
            }
            const int prl = (lr < nbk) ? perm[k0 + lr] : 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int j = 4 * kk + lk;
                const int jc = (j < nbk) ? j : 0;
                const double x = at<TR>(A, prl, k0 + jc, LD);
                mfma_f64_16x16x4((lr < nbk && j < nbk) ? x : 0.0, bop[kk], pvt);
            }
        };
        auto store_pivot_rows = [&](int g, const double (&pvt)[4]) {
            double* Mat = (g < RT) ? A : Bm;
            const int col = ((g < RT) ? g : g - RT) * 16 + lr;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int j = lk + 4 * reg;
                if (col < N && j < nbk) at<TR>(Mat, perm[k0 + j], col, LD) = pvt[reg];
            }
        };

        // Work distribution of block k: the owner of the next panel (wavefront gnext mod NW) takes only that column
        // tile and then factorises the panel; the other live column tiles ([A | B] minus the dead A tiles) go round
        // robin to the remaining wavefronts.  A column tile is read and written by exactly one wavefront per block
        // (pivot rows included: only the tile's own pivot-row entries serve as its B operand), so the new pivot rows
        // are stored right away and the block barrier is the only synchronisation.
        const int gnext = k + 1;
        const bool has_next = gnext < RT;
        const int owner = has_next ? (gnext % NW) : -1;
        if (has_next && wave == owner) {
            double tmp[4];
            do_tile(gnext, tmp);
            wave_sync_lds();
            store_pivot_rows(gnext, tmp);
            wave_sync_lds();
            if (!panel(k + 1) && lane == 0) *fail = 1;
        }
        const bool worker = (NW == 1) || !has_next || wave != owner;
        const int nworkers = (NW == 1 || !has_next) ? NW : NW - 1;
        const int widx = (NW == 1 || !has_next) ? wave : (wave - owner - 1 + NW) % NW;
        if (worker) {
            int idx = 0;
            for (int g = (has_next ? gnext + 1 : RT); g < 2 * RT; ++g, ++idx) {
                if (idx % nworkers != widx) continue;
#ifdef SMRT_ABLATE_GJ_WORKERS
                continue;   // timing experiment only: wrong results
#endif
                double tmp[4];
                do_tile(g, tmp);
                wave_sync_lds();
                store_pivot_rows(g, tmp);
                wave_sync_lds();
            }
            if (has_v && widx == 0) {  // extra right-hand side: same transformation, rows lane + 64 r2
                constexpr int VR = CHN < 2 ? 2 : CHN;
                double acc[VR];
#pragma unroll
                for (int r2 = 0; r2 < VR; ++r2) acc[r2] = 0.0;
#pragma unroll
                for (int r2 = 0; r2 < VR; ++r2) {
                    const int row = lane + 64 * r2;
                    if (row < N) {
                        acc[r2] = v[row];
                        for (int j = 0; j < nbk; ++j) acc[r2] += at<TR>(A, row, k0 + j, LD) * v[perm[k0 + j]];
                    }
                }
                wave_sync_lds();
#pragma unroll
                for (int r2 = 0; r2 < VR; ++r2)
                    if (lane + 64 * r2 < N) v[lane + 64 * r2] = acc[r2];
            }
        }
        block_sync();
        if (*fail) return false;  // uniform
    }
    }
    block_sync();
    SMRT_GSUB(1);
    // ---- undo the implicit row permutation: row perm[k] of B is row k of the solution (A is free scratch now)
    if (rs) for_2d<NT>(N, N, [&](int k, int c) { at<TR>(A, k, c, LD) = at<TR>(Bm, perm[k], c, LD) * (rs[k] * cs[c]); });
    else for_2d<NT>(N, N, [&](int k, int c) { at<TR>(A, k, c, LD) = at<TR>(Bm, perm[k], c, LD); });
    constexpr int VK = (64 * (CHN < 1 ? 1 : CHN) + NT - 1) / NT < 2 ? 2 : (64 * CHN + NT - 1) / NT;   // N <= 64 CHN <= VK NT
    double vk[VK];
#pragma unroll
    for (int q = 0; q < VK; ++q) vk[q] = 0.0;
    if (has_v) {
#pragma unroll
        for (int q = 0; q < VK; ++q)
            if (t + q * NT < N) vk[q] = v[perm[t + q * NT]];
    }
    block_sync();
    if (!result_in_A) for_2d<NT>(N, N, [&](int k, int c) { at<TR>(Bm, k, c, LD) = at<TR>(A, k, c, LD); });
    if (has_v) {
#pragma unroll
        for (int q = 0; q < VK; ++q)
            if (t + q * NT < N) v[t + q * NT] = vk[q];
    }
    block_sync();
    SMRT_GSUB(2);
    return true;
}

// the Gauss-Jordan entry point of the drivers (solution copied back over Bm)
template <int NT, bool TR, int CHN = 2>
SMRT_DEV bool gj_solve(double* A, double* Bm, double* v, const Lds& s, int N, int LD, bool allow_fast = false) {
    return gj_solve_b16<NT, TR, CHN>(A, Bm, v, s, N, LD, false, nullptr, nullptr, allow_fast);
}

}  // namespace smrt
