// Dense kernels on LDS-resident matrices: tile helpers, Cholesky, triangular products / solves, the one-sided Jacobi
// of the fused kernels, and the row-block GEMM passes of the layer recursion (matrix-core and scalar variants).
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_physics.hpp"

namespace smrt {

// ------------------------------------------------------------------------------------------------------------
// dense kernels in LDS.  Element (r, c) of every matrix lives at [c * LD + r].
// ------------------------------------------------------------------------------------------------------------
// Power-of-two 2-D tiling of an (R rows) x (C columns) index space over the workgroup without integer division:
// a wavefront covers RW = pow2 >= min(R, 64) rows and 64 / RW columns at a time (consecutive lanes -> consecutive
// rows -> consecutive LDS addresses).  body(r, c) is called for every r < R, c < C exactly once.
struct Tile2D { int rmask, cshift, cols_per_wave; };
SMRT_DEV Tile2D make_tile(int R) {
    Tile2D t;
    int sh = 6;                       // RW = 64
    if (R <= 32) sh = 5;
    if (R <= 16) sh = 4;
    if (R <= 8) sh = 3;
    if (R <= 4) sh = 2;
    t.rmask = (1 << sh) - 1; t.cshift = sh; t.cols_per_wave = SMRT_LANES >> sh;
    return t;
}
template <int NT, class Body>
SMRT_DEV void for_2d(int R, int C, Body body) {
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const Tile2D tl = make_tile(R);
    const int rl = lane & tl.rmask, cs = lane >> tl.cshift;
    for (int r0 = 0; r0 < R; r0 += SMRT_LANES) {
        const int r = r0 + rl;
        for (int c = wave * tl.cols_per_wave + cs; c < C; c += NW * tl.cols_per_wave)
            if (r < R) body(r, c);
    }
}

template <int NT>
SMRT_DEV bool chol2(double* A, double* Bm, int N, int LD) {
    // Two right-looking Cholesky factorisations side by side (lower triangles, in place), ONE barrier per column:
    // during step k column k stays unscaled (read-only) and the trailing update carries the 1/pivot factor; the
    // columns are scaled by 1/sqrt(pivot) in a final pass.
    const int t = tid();
    for (int k = 0; k < N; ++k) {
        const double akk = A[k * LD + k], bkk = Bm[k * LD + k];
        if (!(akk > 0.0) || !(bkk > 0.0)) return false;  // uniform: every thread reads the same words
        const int m = N - k - 1;
        if (m > 0) {
            const double ra = fast_rcp(akk), rb = fast_rcp(bkk);
            for_2d<NT>(m, m, [&](int ri, int ci) {
                const int i = k + 1 + ri, j = k + 1 + ci;
                if (i >= j) {
                    A[j * LD + i] -= A[k * LD + i] * (A[k * LD + j] * ra);
                    Bm[j * LD + i] -= Bm[k * LD + i] * (Bm[k * LD + j] * rb);
                }
            });
            block_sync();
        }
    }
    for_2d<NT>(N, N, [&](int i, int k) {
        if (i > k) {
            A[k * LD + i] *= fast_rsqrt(A[k * LD + k]);
            Bm[k * LD + i] *= fast_rsqrt(Bm[k * LD + k]);
        }
    });
    block_sync();
    for (int k = t; k < N; k += NT) {
        const double akk = A[k * LD + k], bkk = Bm[k * LD + k];
        A[k * LD + k] = akk * fast_rsqrt(akk);
        Bm[k * LD + k] = bkk * fast_rsqrt(bkk);
    }
    block_sync();
    return true;
}

// C = Lp^T Lm for lower-triangular Lp, Lm
template <int NT>
SMRT_DEV void lt_times_l(const double* Lp, const double* Lm, double* C, int N, int LD) {
    for_2d<NT>(N, N, [&](int i, int j) {
        double acc = 0.0;
        for (int k = (i > j ? i : j); k < N; ++k) acc += Lp[i * LD + k] * Lm[j * LD + k];
        C[j * LD + i] = acc;
    });
    block_sync();
}

// One-sided (Hestenes) Jacobi, two-level ordering.
//
// The columns are cut into NB = 2 * (number of wavefronts) blocks of m columns.  An outer round-robin over the
// blocks gives every wavefront one pair of blocks (I, J) per outer step; inside the step the wavefront rotates all
// m*m cross pairs (m inner steps of m disjoint pairs) -- and, at the first outer step of a sweep, the pairs inside
// its two blocks -- touching only its own 2m columns, so the inner steps need a wavefront-level sync only.  One
// workgroup barrier per OUTER step (NB-1 per sweep instead of N-1).
// GS lanes own one column pair and keep their RPL rows of both columns in registers between the three dot products
// and the rotation.  A sweep in which no pair had cos^2 > 1e-15 before its rotation is the last one (the residual
// non-orthogonality is second order).  On exit sigma[c] = |column c|, rsig[c] = 1/sigma[c].
#ifndef SMRT_JACOBI_EXIT_COS2
#define SMRT_JACOBI_EXIT_COS2 1e-15
#endif
// rotations between columns whose cosine is already below 1e-13 are skipped in the split-pipeline kernel: they
// cannot change the result at the 1e-9 relative level of the parity requirement (1e-6 K), and in the last sweeps
// most pairs are in that state (saves the update + store half of the step)
#ifndef SMRT_JACOBI_SKIP_COS2
#define SMRT_JACOBI_SKIP_COS2 1e-26
#endif
template <int GS, int RPL>
SMRT_DEV void rotate_pair(double* Bm, int LD, int N, int p, int q, bool valid, int sub, int slot, int* flag) {
    // Branch-free: every lane always loads and stores its RPL rows.  Rows >= N of a column are padding inside the
    // N_max x LD buffer (never read by any other stage), loads from them are masked to zero with a select instead
    // of being predicated (per-element exec-mask branches were costing more than the arithmetic).
    double x[RPL], y[RPL];
    double a = 0.0, bb = 0.0, gg = 0.0, a2 = 0.0, bb2 = 0.0, gg2 = 0.0;  // two accumulators: half the FMA chain
    double* cp = Bm + p * LD;
    double* cq = Bm + q * LD;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r0 = sub + i * GS;
        const int r = r0 < LD - 1 ? r0 : LD - 1;  // row LD-1 is always padding (LD = N_max + 1)
        const bool in = valid && (r0 < N);
        const double xv = cp[r], yv = cq[r];
        x[i] = in ? xv : 0.0;
        y[i] = in ? yv : 0.0;
        if (i & 1) { a2 += x[i] * x[i]; bb2 += y[i] * y[i]; gg2 += x[i] * y[i]; }
        else { a += x[i] * x[i]; bb += y[i] * y[i]; gg += x[i] * y[i]; }
    }
    a += a2; bb += bb2; gg += gg2;
    a = group_sum<GS>(a); bb = group_sum<GS>(bb); gg = group_sum<GS>(gg);
    const double g2 = gg * gg, ab = a * bb;
    if (valid && g2 > 1e-30 * ab) {
        // tan of the rotation angle: t = 2 g sign(d) / (|d| + sqrt(d^2 + 4 g^2)), d = b - a
        const double dd = bb - a;
        const double hh = dd * dd + 4.0 * g2;
        const double h = hh * fast_rsqrt1(hh);
        const double tt = (dd >= 0.0 ? 2.0 : -2.0) * gg * fast_rcp1(fabs(dd) + h);  // angle only: 1 Newton step
        const double c = fast_rsqrt(1.0 + tt * tt), sn = c * tt;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r0 = sub + i * GS;
            const int r = r0 < LD - 1 ? r0 : LD - 1;
            cp[r] = c * x[i] - sn * y[i];
            cq[r] = sn * x[i] + c * y[i];
        }
        if (sub == 0 && g2 > SMRT_JACOBI_EXIT_COS2 * ab) lds_or(flag, 1);
    }
}

template <int NT, int JW, int GS, int RPL>
SMRT_DEV bool jacobi_onesided(double* Bm, int N, int LD, double* sigma, double* rsig, int* flag, int* n_sweeps,
                              double* sub_acc = nullptr) {
    // JW = wavefronts that take part (the others only meet the workgroup barriers): with few lanes per pair and many
    // rows per lane the fixed per-rotation cost (index math, reductions, rotation parameters) is amortised better
    // than by spreading every pair over more lanes of more wavefronts.
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    constexpr int NB = 2 * JW;               // column blocks
    constexpr int SLOTS = SMRT_LANES / GS;   // column pairs a wavefront rotates at once
    const int slot = lane / GS, sub = lane % GS;
    const int m = (N + NB - 1) / NB;         // columns per block
    const int me = m + (m & 1);              // even player count of the in-block tournament
    bool converged = false;
#ifdef SMRT_STAGE_TIMING
    long long tj0 = cycle_counter();
#define SMRT_JSUB(k) do { const long long n_ = cycle_counter(); if (t == 0 && sub_acc) sub_acc[k] += (double)(n_ - tj0); tj0 = n_; } while (0)
#else
#define SMRT_JSUB(k) do {} while (0)
#endif
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        block_sync();  // everyone has read the previous flag
        if (t == 0) *flag = 0;
        block_sync();
        SMRT_JSUB(5);
        for (int s = 0; s < NB - 1; ++s) {
            if (wave >= JW) { block_sync(); continue; }
            int I, J;
            if (wave == 0) { I = NB - 1; J = s; }
            else {
                I = s + wave; if (I >= NB - 1) I -= NB - 1;
                J = s - wave; if (J < 0) J += NB - 1;
            }
            const int i0 = I * m, j0 = J * m;
            if (s == 0 && m > 1) {
                // pairs inside block I and inside block J: (me - 1) steps of me/2 pairs per block
                const int half = me / 2;
                for (int u = 0; u < me - 1; ++u) {
                    for (int ps0 = 0; ps0 < 2 * half; ps0 += SLOTS) {  // uniform trip count over the wavefront
                        const int ps = ps0 + slot;
                        const int base = (ps < half) ? i0 : j0;
                        const int k = (ps < half) ? ps : ps - half;
                        int a, b;
                        if (k == 0) { a = me - 1; b = u; }
                        else {
                            a = u + k; if (a >= me - 1) a -= me - 1;
                            b = u - k; if (b < 0) b += me - 1;
                        }
                        const int p = base + a, q = base + b;
                        const bool valid = (ps < 2 * half) && (a < m) && (b < m) && (p < N) && (q < N);
                        rotate_pair<GS, RPL>(Bm, LD, N, valid ? p : 0, valid ? q : 0, valid, sub, slot, flag);
                    }
                    wave_sync_lds();
                }
                SMRT_JSUB(3);
            }
            for (int j = 0; j < m; ++j) {
                for (int ps0 = 0; ps0 < m; ps0 += SLOTS) {  // uniform trip count over the wavefront
                    const int ps = ps0 + slot;
                    int bq = ps + j; if (bq >= m) bq -= m;
                    const int p = i0 + ps, q = j0 + bq;
                    const bool valid = (ps < m) && (p < N) && (q < N);
                    rotate_pair<GS, RPL>(Bm, LD, N, valid ? p : 0, valid ? q : 0, valid, sub, slot, flag);
                }
                wave_sync_lds();
            }
            SMRT_JSUB(3);
            block_sync();
            SMRT_JSUB(4);
        }
        converged = (*flag == 0);
        ++*n_sweeps;
    }
    block_sync();
    // column norms
    {
        constexpr int NG = NT / GS;
        const int grp = t / GS;
        const int rounds2 = (N + NG - 1) / NG;
        for (int rd = 0; rd < rounds2; ++rd) {
            const int c = grp + rd * NG;
            double a = 0.0;
            if (c < N)
                for (int r = sub; r < N; r += GS) { const double xx = Bm[c * LD + r]; a += xx * xx; }
            a = group_sum<GS>(a);
            if (c < N && sub == 0) { const double rs = fast_rsqrt(a); sigma[c] = a * rs; rsig[c] = rs; }
        }
    }
    block_sync();
    return converged;
}

// C = Lp * Bm (Lp lower triangular)
template <int NT>
SMRT_DEV void l_times_m(const double* Lp, const double* Bm, double* C, int N, int LD) {
    for_2d<NT>(N, N, [&](int i, int c) {
        double acc = 0.0;
        for (int k = 0; k <= i; ++k) acc += Lp[k * LD + i] * Bm[c * LD + k];
        C[c * LD + i] = acc;
    });
    block_sync();
}

// Bm <- Lp^-T Bm (back substitution with the upper-triangular Lp^T, all columns at once)
template <int NT>
SMRT_DEV void lt_solve(const double* Lp, double* Bm, int N, int LD) {
    for (int i = N - 1; i >= 1; --i) {
        const double rd = fast_rcp(Lp[i * LD + i]);
        for_2d<NT>(i, N, [&](int r, int c) { Bm[c * LD + r] -= Lp[r * LD + i] * (Bm[c * LD + i] * rd); });
        block_sync();
    }
    for_2d<NT>(N, N, [&](int i, int c) { Bm[c * LD + i] *= fast_rcp(Lp[i * LD + i]); });
    block_sync();
}

// Solve A X = Bm (+ one extra right-hand-side vector v, may be null) by LU with partial pivoting; X overwrites
// Bm / v, A is destroyed.  TR selects the storage view: element (r, c) at [c*LD + r] (false) or [r*LD + c]
// (true, i.e. the routine then solves A^T X^T = Bm^T on the same buffers).
// Every thread scans the pivot column itself (LDS broadcast reads): no cross-lane reduction and no barrier
// between the search and the row swap.
template <bool TR>
SMRT_DEV double& at(double* M, int r, int c, int LD) { return TR ? M[r * LD + c] : M[c * LD + r]; }

template <int NT, bool TR>
SMRT_DEV bool lu_solve(double* A, double* Bm, double* v, double* udiag, int N, int LD) {
    // Gauss-Jordan elimination with partial pivoting: every step eliminates column k from ALL other rows, so there
    // is no back-substitution phase (one third fewer barriers than LU + back substitution; the path is latency
    // bound, not flop bound).  Column k is never written once step k starts: the row swap skips it (the multipliers
    // are taken from the unswapped column) and the pivot goes to udiag[k]; that makes the per-wavefront pivot
    // search race-free against the swap of faster wavefronts without an extra barrier.
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1);
    const int nv = (v != nullptr) ? 1 : 0;
    for (int k = 0; k < N; ++k) {
        // pivot row: every wavefront finds it on its own (one LDS load per lane, DPP arg-max on a key made of the
        // magnitude bits with the row index in the 8 low mantissa bits: exactness of the choice is irrelevant)
        unsigned long long key = 0ull;
        for (int r = k + lane; r < N; r += SMRT_LANES) {
            const double xr = fabs(at<TR>(A, r, k, LD));
            unsigned long long bits;
            memcpy(&bits, &xr, 8);
            bits = (bits & ~0xFFull) | (unsigned long long)(255 - (r - k < 255 ? r - k : 255));
            if (bits > key) key = bits;
        }
        key = wave_max_u64(key);
        if (key < 256ull) return false;  // zero column: singular, uniform exit
        const int p = k + 255 - (int)(key & 0xFFull);
        const double pv = at<TR>(A, p, k, LD);
        const double akk = at<TR>(A, k, k, LD);
        if (!(fabs(pv) > 0.0 && fabs(pv) < 1e300)) return false;  // uniform
        if (p != k) {
            const int na = N - k - 1;
            for (int idx = t; idx < na + N + nv; idx += NT) {
                if (idx < na) {
                    const int c = k + 1 + idx;
                    const double xx = at<TR>(A, k, c, LD);
                    at<TR>(A, k, c, LD) = at<TR>(A, p, c, LD);
                    at<TR>(A, p, c, LD) = xx;
                } else if (idx < na + N) {
                    const int c = idx - na;
                    const double xx = at<TR>(Bm, k, c, LD);
                    at<TR>(Bm, k, c, LD) = at<TR>(Bm, p, c, LD);
                    at<TR>(Bm, p, c, LD) = xx;
                } else {
                    const double xx = v[k]; v[k] = v[p]; v[p] = xx;
                }
            }
            block_sync();
        }
        if (t == 0) udiag[k] = pv;
        const double rp = fast_rcp(pv);
        const int m = N - k - 1;
        for_2d<NT>(N - 1, m + N + nv, [&](int ri, int cc) {
            const int r = ri + (ri >= k ? 1 : 0);  // every row but k
            const double l = ((r == p) ? akk : at<TR>(A, r, k, LD)) * rp;
            if (cc < m) {
                const int c = k + 1 + cc;
                at<TR>(A, r, c, LD) -= l * at<TR>(A, k, c, LD);
            } else if (cc < m + N) {
                const int c = cc - m;
                at<TR>(Bm, r, c, LD) -= l * at<TR>(Bm, k, c, LD);
            } else {
                v[r] -= l * v[k];
            }
        });
        block_sync();
    }
    for_2d<NT>(N, N + nv, [&](int i, int c) {
        const double rd = fast_rcp(udiag[i]);
        if (c < N) at<TR>(Bm, i, c, LD) *= rd;
        else v[i] *= rd;
    });
    block_sync();
    return true;
}

// ---- 16x16 tile GEMM on the FP64 matrix core ---------------------------------------------------------------
// c (the 16x16 tile at tile-row ti, tile-column tj, in MFMA accumulator layout) += sum_k A[i][k] B[k][j];
// fa(i, k) / fb(k, j) fetch operands (they must return 0 outside the matrix); K is rounded up to 4.
template <class FA, class FB>
SMRT_DEV void gemm_tile(double (&c)[4], int K, int ti, int tj, FA fa, FB fb) {
    const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) mfma_f64_16x16x4(fa(ti * 16 + lr, k0 + lk), fb(k0 + lk, tj * 16 + lr), c);
}
// store / load an accumulator tile to a column-major matrix (element (r, c) at [c*LD + r]), rows/cols < N only
// Element (r, c) of a symmetric / lower-triangular N x N matrix.  PK = false: the usual column-major layout with leading
// dimension LD.  PK = true: only the lower triangle is stored, column after column (column c holds rows c .. LD - 1, LD =
// the padded order): half the LDS, which is what lets THREE prep workgroups share a CU.  An address above the diagonal
// maps onto its mirror image (valid memory: reads of it are masked or symmetric, writes must be guarded by the caller).
template <bool PK>
SMRT_DEV int sidx(int r, int c, int LD) {
    if (!PK) return c * LD + r;
    const int rr = r >= c ? r : c, cc = r >= c ? c : r;
    return cc * LD - ((cc * (cc - 1)) >> 1) + (rr - cc);
}
SMRT_HD int packed_lower_doubles(int n) { return n * (n + 1) / 2; }

template <class F>
SMRT_DEV void tile_foreach(int ti, int tj, int N, F f) {
    const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
    const int col = tj * 16 + lr;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = ti * 16 + lk + 4 * reg;
        if (row < N && col < N) f(reg, row, col);
    }
}

// C = Lp^T Lm (both lower triangular; whatever sits above their diagonals is ignored)
// reverse_cols: column c of the product is stored as column N-1-c.  The column norms of B = L+^T L- grow with the
// column index (beta ~ ke / mu, mu descending); the one-sided Jacobi converges in fewer sweeps when the large
// columns come first (de Rijk), and nothing downstream depends on the order of the eigenpairs.
template <int NT, bool PK = false>   // PK: Lp and Lm in packed lower storage (C is always a full matrix)
SMRT_DEV void lt_times_l_mfma(const double* Lp, const double* Lm, double* C, int N, int LD, bool reverse_cols = false) {
    const int wave = tid() / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    for (int tix = wave; tix < RT * RT; tix += NW) {
        const int ti = tix % RT, tj = tix / RT;
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        const int kmin = 16 * (ti > tj ? ti : tj);  // k >= max(i, j)
        const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
        const int i = ti * 16 + lr, j = tj * 16 + lr;
        const int ic = i < N ? i : N - 1, jc = j < N ? j : N - 1;
        // column bases hoisted out of the k loop: element (k, col) of a lower triangle sits at base(col) + k for k >= col
        // (rows above the diagonal are masked, their address only has to be valid: k is clamped to the diagonal)
        const int abase = sidx<PK>(ic, ic, LD) - ic, bbase = sidx<PK>(jc, jc, LD) - jc;
        for (int k0 = kmin; k0 < N; k0 += 4) {
            const int k = k0 + lk, kc = k < N ? k : N - 1;
            const double av = Lp[abase + (kc > ic ? kc : ic)], bv = Lm[bbase + (kc > jc ? kc : jc)];
            const bool kin = k < N;
            mfma_f64_16x16x4((kin && k >= i) ? av : 0.0, (kin && k >= j) ? bv : 0.0, c);   // (k >= i, k < N imply i < N)
        }
        tile_foreach(ti, tj, N, [&](int reg, int row, int col) { C[(reverse_cols ? N - 1 - col : col) * LD + row] = c[reg]; });
    }
    block_sync();
}

// C = Lp * Bm (Lp lower triangular).  A wavefront carries TPW output tiles of one tile row together: the operand of
// Lp is loaded once for the three, and the loads of a chunk of 16 columns (4 + 4 TPW) are all in flight before its
// matrix-core passes (operands from global memory for N > 64: two dependent loads per pass was the latency of L2 every
// four columns).
template <int NT>
SMRT_DEV void l_times_m_mfma(const double* Lp, const double* Bm, double* C, int N, int LD) {
    const int wave = tid() / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    constexpr int TPW = 3;
    const int RT = (N + 15) >> 4;
    const int NG = (RT + TPW - 1) / TPW;
    const int lane = tid() & (SMRT_LANES - 1), lr = lane & 15, lk = lane >> 4;
    for (int u = wave; u < RT * NG; u += NW) {
        const int ti = u % RT, tg = u / RT;
        double c[TPW][4];
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) c[q][reg] = 0.0;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const int kend = (ti * 16 + 16 < N) ? ti * 16 + 16 : N;  // k <= i
        const int nq = (RT - tg * TPW < TPW) ? RT - tg * TPW : TPW;   // tiles of this group (uniform)
        for (int k0 = 0; k0 < kend; k0 += 16) {
            double av[4], bv[TPW][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = k0 + 4 * kk + lk, kc = k < N ? k : N - 1;
                const double x = Lp[kc * LD + ic];
                av[kk] = (i < N && k <= i && k < N) ? x : 0.0;
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    bv[q][kk] = 0.0;
                    if (q < nq) {
                        const int j = (tg * TPW + q) * 16 + lr, jc = j < N ? j : N - 1;
                        const double y = Bm[jc * LD + kc];
                        bv[q][kk] = (j < N && k < N) ? y : 0.0;
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int q = 0; q < TPW; ++q)
                    if (q < nq) mfma_f64_16x16x4(av[kk], bv[q][kk], c[q]);
        }
#pragma unroll
        for (int q = 0; q < TPW; ++q)
            if (tg * TPW + q < RT) tile_foreach(ti, tg * TPW + q, N, [&](int reg, int row, int col) { C[col * LD + row] = c[q][reg]; });
    }
    block_sync();
}

// ---- the two "row block times matrix" passes of the layer recursion on the matrix core (N <= 64) -------------
// Every wavefront owns one (or, for small workgroups, a few) 16-row tile(s): it first pulls the A operands of its
// rows -- the whole 16 x N row block, 16 registers per lane -- into registers, the workgroup synchronises, and only
// then are results written; that is what makes the in-place updates (rows of Rt, rows of F) safe.
template <int NT>
struct RowTiles {
    static constexpr int NW = NT / SMRT_LANES;
    static constexpr int RPW = (4 + NW - 1) / NW;               // row tiles per wavefront (RT <= 4)
    static constexpr int CS = (NW >= 4) ? NW / 4 : 1;           // wavefronts sharing one row tile (column split)
};

// Wk = F - Rt G ; Rt <- Rt F - G (in place) ; cvec = (Rt 1) B - B + svec
template <int NT>
SMRT_DEV void r1_mfma(const double* F, const double* G, double* Rt, double* Wk, double* cvec, const double* svec,
                      double Bl, int N, int LD) {
    using RTc = RowTiles<NT>;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double a[RTc::RPW][16];
    int tis[RTc::RPW];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        tis[o] = ti;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        double rs = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double x = Rt[kc * LD + ic];
            a[o][kk] = (ti < RT && i < N && k < N) ? x : 0.0;
            rs += a[o][kk];
        }
        rs += shfl_xor(rs, 16);
        rs += shfl_xor(rs, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) cvec[i] = rs * Bl - Bl + svec[i];
    }
    block_sync();
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = tis[o];
        if (ti >= RT) continue;
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
        for (int tj = cs; tj < RT; tj += RTc::CS) {
            double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
            const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (4 * kk < N) {
                    const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                    const bool in = (j < N && k < N);
                    const double gv = G[jc * LD + kc], fv = F[jc * LD + kc];
                    mfma_f64_16x16x4(a[o][kk], in ? gv : 0.0, c1);
                    mfma_f64_16x16x4(a[o][kk], in ? fv : 0.0, c2);
                }
            }
            tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                Wk[col * LD + row] = F[col * LD + row] - c1[reg];
                Rt[col * LD + row] = c2[reg] - G[col * LD + row];
            });
        }
    }
    block_sync();
}

// Y = F tQt + G -> Wk ; W = (G - Rtop F) tQt + (F - Rtop G) -> over F (in place)
// upb = F tq + B ; g = (G - Rtop F) tq + (1 - Rtop) B
// r1_mfma in two halves for the two-slot finish kernel.  r1_load pulls the A operands (the rows of R~ of this
// wavefront's row tile) into registers and writes cvec -- after it slot R is free.  r1_compute runs the MFMA loops
// with both B operands in LDS (Gl in slot X, Fl in slot R: staged there by the F/G formation), keeps the results in
// registers until every wavefront is done reading, and then writes Wk = F - R~ G over Gl and R~ F - G over Fl.
template <int NT>
SMRT_DEV void r1_load(const double* Rt, double (&a)[RowTiles<NT>::RPW][16], double* cvec, const double* svec, double Bl,
                      int N, int LD, const double* colsign = nullptr /* active mode: R~ D, D = +-1 per column */) {
    using RTc = RowTiles<NT>;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        double rs = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double x = Rt[kc * LD + ic] * (colsign ? colsign[kc] : 1.0);
            a[o][kk] = (ti < RT && i < N && k < N) ? x : 0.0;
            rs += a[o][kk];
        }
        rs += shfl_xor(rs, 16);
        rs += shfl_xor(rs, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) cvec[i] = rs * Bl - Bl + svec[i];
    }
    block_sync();
}

template <int NT>
SMRT_DEV void r1_compute(double* Fl /* slot R */, double* Gl /* slot X */, const double (&a)[RowTiles<NT>::RPW][16],
                         int N, int LD) {
    using RTc = RowTiles<NT>;
    constexpr int MAXTJ = (4 + RTc::CS - 1) / RTc::CS;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double r1[RTc::RPW][MAXTJ][4], r2[RTc::RPW][MAXTJ][4];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
            if (ti < RT && tj < RT) {
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const bool in = (j < N && k < N);
                        const double gv = Gl[jc * LD + kc], fv = Fl[jc * LD + kc];
                        mfma_f64_16x16x4(a[o][kk], in ? gv : 0.0, c1);
                        mfma_f64_16x16x4(a[o][kk], in ? fv : 0.0, c2);
                    }
                }
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    c1[reg] = Fl[col * LD + row] - c1[reg];   // Wk
                    c2[reg] = c2[reg] - Gl[col * LD + row];   // R~ F - G
                });
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { r1[o][q][reg] = c1[reg]; r2[o][q][reg] = c2[reg]; }
        }
    }
    block_sync();  // every wavefront has finished reading F and G
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            if (ti < RT && tj < RT)
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    Gl[col * LD + row] = r1[o][q][reg];
                    Fl[col * LD + row] = r2[o][q][reg];
                });
        }
    }
    block_sync();
}

// SIGNED (azimuth modes m >= 1, three polarisations): the down-going eigenvectors carry the row signs
// dsg = (+1, +1, -1) per (V, H, U) (dort.py:951-953), i.e. W = (D G - Rtop F) tQt + (D F - Rtop G).
template <int NT, bool SIGNED = false>
SMRT_DEV void r45_mfma(double* F, const double* G, const double* Q, double* Wk, const double* Rtop, const double* tq,
                       double* upb, double* gvec, double Bl, int N, int LD, const double* dsg = nullptr) {
    using RTc = RowTiles<NT>;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double af[RTc::RPW][16], aw[RTc::RPW][16];
    int tis[RTc::RPW];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        tis[o] = ti;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const double rt = Rtop[ic];
        const double sg = SIGNED ? dsg[ic] : 1.0;
        double vy = 0.0, vg = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double fv = F[kc * LD + ic], gv = G[kc * LD + ic], tk = tq[kc];
            const bool in = (ti < RT && i < N && k < N);
            af[o][kk] = in ? fv : 0.0;
            aw[o][kk] = in ? (SIGNED ? sg * gv : gv) - rt * fv : 0.0;
            vy += af[o][kk] * tk;
            vg += aw[o][kk] * tk;
        }
        vy += shfl_xor(vy, 16); vy += shfl_xor(vy, 32);
        vg += shfl_xor(vg, 16); vg += shfl_xor(vg, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) { upb[i] = vy + Bl; gvec[i] = vg + (1.0 - rt) * Bl; }
    }
    block_sync();
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = tis[o];
        if (ti >= RT) continue;
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
        for (int tj = cs; tj < RT; tj += RTc::CS) {
            double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
            const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (4 * kk < N) {
                    const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                    const double qv = Q[jc * LD + kc];
                    const double bop = (j < N && k < N) ? qv : 0.0;
                    mfma_f64_16x16x4(af[o][kk], bop, cy);
                    mfma_f64_16x16x4(aw[o][kk], bop, cw);
                }
            }
            tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                const double fic = F[col * LD + row], gic = G[col * LD + row];
                Wk[col * LD + row] = cy[reg] + gic;
                F[col * LD + row] = cw[reg] + (SIGNED ? dsg[row] * fic : fic) - Rtop[row] * gic;
            });
        }
    }
    block_sync();
}

// Two-slot variant for the finish kernel whose F and G live in global memory: Y and W are held in registers until
// every wavefront has finished reading Q, then Y goes to Yout and W OVER Q (Wout == Q is allowed).
template <int NT, bool SIGNED = false>
SMRT_DEV void r45_mfma2(const double* F, const double* G, const double* Q, double* Yout, double* Wout,
                        const double* Rtop, const double* tq, double* upb, double* gvec, double Bl, int N, int LD,
                        const double* dsg = nullptr) {
    using RTc = RowTiles<NT>;
    constexpr int MAXTJ = (4 + RTc::CS - 1) / RTc::CS;
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    double ys[RTc::RPW][MAXTJ][4], ws[RTc::RPW][MAXTJ][4];
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const double rt = Rtop[ic];
        const double sg = SIGNED ? dsg[ic] : 1.0;   // row sign of the down-going eigenvectors (+-1)
        double af[16], aw[16];
        double vy = 0.0, vg = 0.0;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double fv = F[kc * LD + ic], gv = G[kc * LD + ic], tk = tq[kc];
            const bool in = (ti < RT && i < N && k < N);
            af[kk] = in ? fv : 0.0;
            aw[kk] = in ? (SIGNED ? sg * gv : gv) - rt * fv : 0.0;
            vy += af[kk] * tk;
            vg += aw[kk] * tk;
        }
        vy += shfl_xor(vy, 16); vy += shfl_xor(vy, 32);
        vg += shfl_xor(vg, 16); vg += shfl_xor(vg, 32);
        const bool owner = (RTc::NW >= 4) ? (wave < 4) : true;
        if (owner && ti < RT && lk == 0 && i < N) { upb[i] = vy + Bl; gvec[i] = vg + (1.0 - rt) * Bl; }
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
            if (ti < RT && tj < RT) {
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const double qv = Q[jc * LD + kc];
                        const double bop = (j < N && k < N) ? qv : 0.0;
                        mfma_f64_16x16x4(af[kk], bop, cy);
                        mfma_f64_16x16x4(aw[kk], bop, cw);
                    }
                }
                // the elementwise terms + G (for Y) and + F - Rtop G (for W) of this tile: the row block of F and G is
                // already in registers in A-operand layout, so they are added as four more k-steps against an identity
                // B operand instead of being re-read from global memory in accumulator layout (uncoalesced)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int kk = 4 * tj + q4;                      // k = 4 kk + lk runs over the columns of tile tj
                    const double idb = (4 * q4 + lk == lr) ? 1.0 : 0.0;
                    double gk = 0.0, fk = 0.0;
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2)
                        if (k2 == kk) { fk = af[k2]; gk = (SIGNED ? sg : 1.0) * (aw[k2] + rt * af[k2]); }
                    mfma_f64_16x16x4(gk, idb, cy);
                    mfma_f64_16x16x4((SIGNED ? sg * fk : fk) - rt * gk, idb, cw);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { ys[o][q][reg] = cy[reg]; ws[o][q][reg] = cw[reg]; }
        }
    }
    block_sync();  // every wavefront has read Q
#pragma unroll
    for (int o = 0; o < RTc::RPW; ++o) {
        const int ti = (RTc::NW >= 4) ? (wave & 3) : (wave + o * RTc::NW);
        const int cs = (RTc::NW >= 4) ? (wave >> 2) : 0;
#pragma unroll
        for (int q = 0; q < MAXTJ; ++q) {
            const int tj = cs + q * RTc::CS;
            if (ti < RT && tj < RT)
                tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                    Yout[col * LD + row] = ys[o][q][reg];
                    Wout[col * LD + row] = ws[o][q][reg];
                });
        }
    }
    block_sync();
}

// ---- the two row-block passes for 64 < N <= 128 (global-workspace kernels): same algorithm, 32 k-groups per row,
// up to eight row tiles, one row tile per wavefront at a time (its A operands in registers), operands from wherever
// the matrices live (all pointers are generic).
// KG = k-groups of four columns a row block spans = 16 per 64 rows (N <= 4 KG): the A operands of one 16-row tile, KG
// doubles per lane, are pulled into registers before anything of that tile is written (in-place row updates).
// PASS 0: both products in one sweep (operands straight from memory).  PASS 1: Wk = F - Rt G and cvec only; PASS 2:
// Rt <- Rt F - G only -- the N > 128 finish kernels run 1 then 2, each with its ONE operand matrix (G, then F) staged
// through LDS like r45_mfma_big below (stage / stage_bufs / stage_stride: see there).
template <int NT, int KG = 32, int PASS = 0>
SMRT_DEV void r1_mfma_big(const double* F, const double* G, double* Rt, double* Wk, double* cvec, const double* svec,
                          double Bl, int N, int LD, double* stage = nullptr, int stage_bufs = 2, int stage_stride = 64 * KG) {
    constexpr int NW = NT / SMRT_LANES;
    constexpr int MAXRT = KG / 4;
    constexpr int RPW = (NW >= MAXRT) ? 1 : (MAXRT + NW - 1) / NW;
    constexpr int CS = (NW > MAXRT) ? NW / MAXRT : 1;
    constexpr bool DO_W = (PASS != 2), DO_R = (PASS != 1);
    constexpr int QPT = (KG + NW - 1) / NW;   // k-groups of a staged tile each wavefront copies
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    const bool staged = (stage != nullptr) && CS == 1 && PASS != 0;
    const double* Bop = (PASS == 2) ? F : G;   // the staged operand matrix
    double qn[QPT];
    auto fetch = [&](int tj) {
        const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int kk = wave + NW * i;
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double qv = Bop[jc * LD + kc];
            qn[i] = (j < N && k < N) ? qv : 0.0;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int kk = wave + NW * i;
            if (kk < KG && 4 * kk < N) stage[buf * stage_stride + kk * 64 + lane] = qn[i];
        }
    };
    for (int o = 0; o < RPW; ++o) {
        const int ti = (NW >= MAXRT) ? (wave % MAXRT) : (wave + o * NW);
        const int cs = (NW >= MAXRT) ? (wave / MAXRT) : 0;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        double a[KG];
        double rs = 0.0;
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double x = Rt[kc * LD + ic];
            a[kk] = (ti < RT && i < N && k < N) ? x : 0.0;
            rs += a[kk];
        }
        if (DO_W) {
            rs += shfl_xor(rs, 16);
            rs += shfl_xor(rs, 32);
            if (cs == 0 && ti < RT && lk == 0 && i < N) cvec[i] = rs * Bl - Bl + svec[i];
        }
        if (staged) { fetch(0); put(0); }
        block_sync();  // column-split wavefronts share a row tile: everybody has its A operands before anybody writes
        auto epilogue = [&](int tj, const double (&c1)[4], const double (&c2)[4]) {
            tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                if (DO_W) Wk[col * LD + row] = F[col * LD + row] - c1[reg];
                if (DO_R) Rt[col * LD + row] = c2[reg] - G[col * LD + row];
            });
        };
        if (staged) {
            for (int tj = 0; tj < RT; ++tj) {
                const int nxt = (stage_bufs == 2) ? ((tj + 1) & 1) : 0;
                const double* cur = stage + ((stage_bufs == 2) ? (tj & 1) : 0) * stage_stride;
                if (tj + 1 < RT) fetch(tj + 1);
                if (ti < RT) {
                    double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk) {
                        if (4 * kk < N) {
                            const double bop = cur[kk * 64 + lane];
                            if (DO_W) mfma_f64_16x16x4(a[kk], bop, c1); else mfma_f64_16x16x4(a[kk], bop, c2);
                        }
                    }
                    epilogue(tj, c1, c2);
                }
                if (stage_bufs != 2) block_sync();
                if (tj + 1 < RT) put(nxt);
                block_sync();
            }
        } else if (ti < RT) {
            for (int tj = cs; tj < RT; tj += CS) {
                double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < KG; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const bool in = (j < N && k < N);
                        if (DO_W) { const double gv = G[jc * LD + kc]; mfma_f64_16x16x4(a[kk], in ? gv : 0.0, c1); }
                        if (DO_R) { const double fv = F[jc * LD + kc]; mfma_f64_16x16x4(a[kk], in ? fv : 0.0, c2); }
                    }
                }
                epilogue(tj, c1, c2);
            }
        }
    }
    block_sync();
}

// PASS 0: Y and W together (two A-operand arrays per lane); PASS 1: Y = F tQt + G -> Wk and upb only; PASS 2:
// W -> over F and gvec only.  For N > 128 the caller runs PASS 1 then PASS 2: one array of KG doubles per lane instead
// of two keeps the kernel within 256 VGPRs, i.e. two workgroups per CU.
// stage (N > 128 only, or nullptr): stage_bufs = two LDS buffers of 64 KG doubles (one if two do not fit).  Every wavefront needs the whole operand tile
// Q[:, 16 tj .. 16 tj + 15] for its row tile, and a tile (48 KB at N = 384) does not fit the vector L1: read straight from
// global memory it crossed L2 -> L1 once per wavefront.  With the buffers the workgroup copies each tile once, in
// matrix-core lane order (stage[64 kk + lane] = the B operand of lane `lane` for k-group kk), double-buffered: the
// loads of tile tj + 1 are in flight during the matrix-core passes of tile tj, one workgroup barrier per tile (two
// with a single buffer).
template <int NT, bool SIGNED, int KG = 32, int PASS = 0>
SMRT_DEV void r45_mfma_big(double* F, const double* G, const double* Q, double* Wk, const double* Rtop, const double* tq,
                           double* upb, double* gvec, double Bl, int N, int LD, const double* dsg, double* stage = nullptr,
                           int stage_bufs = 2, int stage_stride = 64 * KG /* doubles per buffer (LdsPlan) */) {
    constexpr int NW = NT / SMRT_LANES;
    constexpr int MAXRT = KG / 4;
    constexpr int RPW = (NW >= MAXRT) ? 1 : (MAXRT + NW - 1) / NW;
    constexpr int CS = (NW > MAXRT) ? NW / MAXRT : 1;
    constexpr bool DO_Y = (PASS != 2), DO_W = (PASS != 1);
    constexpr int QPT = (KG + NW - 1) / NW;   // k-groups of a staged tile each wavefront copies
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    const int RT = (N + 15) >> 4;
    const bool staged = (stage != nullptr) && CS == 1;
    double qn[QPT];
    auto fetch = [&](int tj) {   // this wavefront's share of operand tile tj into registers
        const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int kk = wave + NW * i;
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double qv = Q[jc * LD + kc];
            qn[i] = (j < N && k < N) ? qv : 0.0;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            const int kk = wave + NW * i;
            if (kk < KG && 4 * kk < N) stage[buf * stage_stride + kk * 64 + lane] = qn[i];
        }
    };
    for (int o = 0; o < RPW; ++o) {
        const int ti = (NW >= MAXRT) ? (wave % MAXRT) : (wave + o * NW);
        const int cs = (NW >= MAXRT) ? (wave / MAXRT) : 0;
        const int i = ti * 16 + lr, ic = i < N ? i : N - 1;
        const double rt = Rtop[ic];
        const double sg = SIGNED ? dsg[ic] : 1.0;
        double af[DO_Y ? KG : 1], aw[DO_W ? KG : 1];
        double vy = 0.0, vg = 0.0;
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
            const double fv = F[kc * LD + ic], tk = tq[kc];
            const bool in = (ti < RT && i < N && k < N);
            if (DO_Y) { af[kk] = in ? fv : 0.0; vy += af[kk] * tk; }
            if (DO_W) {
                const double gv = G[kc * LD + ic];
                aw[kk] = in ? (SIGNED ? sg * gv : gv) - rt * fv : 0.0;
                vg += aw[kk] * tk;
            }
        }
        if (DO_Y) { vy += shfl_xor(vy, 16); vy += shfl_xor(vy, 32); }
        if (DO_W) { vg += shfl_xor(vg, 16); vg += shfl_xor(vg, 32); }
        if (cs == 0 && ti < RT && lk == 0 && i < N) {
            if (DO_Y) upb[i] = vy + Bl;
            if (DO_W) gvec[i] = vg + (1.0 - rt) * Bl;
        }
        if (staged) { fetch(0); put(0); }
        block_sync();
        auto epilogue = [&](int tj, const double (&cy)[4], const double (&cw)[4]) {
            tile_foreach(ti, tj, N, [&](int reg, int row, int col) {
                const double gic = G[col * LD + row];
                if (DO_Y) Wk[col * LD + row] = cy[reg] + gic;
                if (DO_W) {
                    const double fic = F[col * LD + row];
                    F[col * LD + row] = cw[reg] + (SIGNED ? dsg[row] * fic : fic) - Rtop[row] * gic;
                }
            });
        };
        if (staged) {
            for (int tj = 0; tj < RT; ++tj) {
                const int nxt = (stage_bufs == 2) ? ((tj + 1) & 1) : 0;
                const double* cur = stage + ((stage_bufs == 2) ? (tj & 1) : 0) * stage_stride;
                if (tj + 1 < RT) fetch(tj + 1);
                if (ti < RT) {
                    double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk) {
                        if (4 * kk < N) {
                            const double bop = cur[kk * 64 + lane];
                            if (DO_Y) mfma_f64_16x16x4(af[kk], bop, cy);
                            if (DO_W) mfma_f64_16x16x4(aw[kk], bop, cw);
                        }
                    }
                    epilogue(tj, cy, cw);
                }
                if (stage_bufs != 2) block_sync();   // one buffer: everybody has read the tile before it is overwritten
                if (tj + 1 < RT) put(nxt);
                block_sync();
            }
        } else if (ti < RT) {
            for (int tj = cs; tj < RT; tj += CS) {
                double cy[4] = {0.0, 0.0, 0.0, 0.0}, cw[4] = {0.0, 0.0, 0.0, 0.0};
                const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
                for (int kk = 0; kk < KG; ++kk) {
                    if (4 * kk < N) {
                        const int k = 4 * kk + lk, kc = k < N ? k : N - 1;
                        const double qv = Q[jc * LD + kc];
                        const double bop = (j < N && k < N) ? qv : 0.0;
                        if (DO_Y) mfma_f64_16x16x4(af[kk], bop, cy);
                        if (DO_W) mfma_f64_16x16x4(aw[kk], bop, cw);
                    }
                }
                epilogue(tj, cy, cw);
            }
        }
    }
    block_sync();
}

// ---- the same two passes without the matrix core (N > 64: CH column chunks of 64 per lane) --------------------
// rows-per-wavefront register blocking
constexpr int RB = 2;

// Wk = F - Rt G ; Rt <- Rt F - G (row-wise in place) ; cvec = (Rt 1) B - B + svec
template <int NT, int CH>
SMRT_DEV void r1_rows(const double* F, const double* G, double* Rt, double* Wk, double* cvec, const double* svec,
                      double Bl, int N, int LD) {
    const int t = tid(), lane = t % SMRT_LANES, wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    for (int i0 = wave * RB; i0 < N; i0 += NW * RB) {
        double a1[RB][CH], a2[RB][CH], rsum[RB];
        for (int bb = 0; bb < RB; ++bb) { rsum[bb] = 0.0; for (int ch = 0; ch < CH; ++ch) { a1[bb][ch] = 0.0; a2[bb][ch] = 0.0; } }
        for (int k = 0; k < N; ++k) {
            double fk[CH], gk[CH];
            for (int ch = 0; ch < CH; ++ch) {
                const int c = ch * SMRT_LANES + lane;
                fk[ch] = (c < N) ? F[c * LD + k] : 0.0;
                gk[ch] = (c < N) ? G[c * LD + k] : 0.0;
            }
            for (int bb = 0; bb < RB; ++bb) {
                const int i = i0 + bb;
                const double r = (i < N) ? Rt[k * LD + i] : 0.0;
                rsum[bb] += r;
                for (int ch = 0; ch < CH; ++ch) { a1[bb][ch] += r * gk[ch]; a2[bb][ch] += r * fk[ch]; }
            }
        }
        wave_sync();  // every lane has read rows i0.. of Rt before they are overwritten
        for (int bb = 0; bb < RB; ++bb) {
            const int i = i0 + bb;
            if (i < N) {
                for (int ch = 0; ch < CH; ++ch) {
                    const int c = ch * SMRT_LANES + lane;
                    if (c < N) {
                        Wk[c * LD + i] = F[c * LD + i] - a1[bb][ch];
                        Rt[c * LD + i] = a2[bb][ch] - G[c * LD + i];
                    }
                }
                if (lane == 0) cvec[i] = rsum[bb] * Bl - Bl + svec[i];
            }
        }
    }
    block_sync();
}

// Y = F tQt + G -> Wk ; W = (D G - Rtop F) tQt + (D F - Rtop G) -> over F (row-wise in place; D = 1 unless SIGNED)
// upb = F tq + B ; g = (D G - Rtop F) tq + (1 - Rtop) B
template <int NT, int CH, bool SIGNED>
SMRT_DEV void r45_rows(double* F, const double* G, const double* Q, double* Wk, const double* Rtop, const double* tq,
                       double* upb, double* gvec, double Bl, int N, int LD, const double* dsg) {
    const int t = tid(), lane = t % SMRT_LANES, wave = t / SMRT_LANES;
    constexpr int NW = NT / SMRT_LANES;
    for (int i0 = wave * RB; i0 < N; i0 += NW * RB) {
        double ay[RB][CH], aw[RB][CH], vy[RB], vg[RB];
        for (int bb = 0; bb < RB; ++bb) { vy[bb] = 0.0; vg[bb] = 0.0; for (int ch = 0; ch < CH; ++ch) { ay[bb][ch] = 0.0; aw[bb][ch] = 0.0; } }
        for (int k = 0; k < N; ++k) {
            double tk[CH];
            for (int ch = 0; ch < CH; ++ch) {
                const int c = ch * SMRT_LANES + lane;
                tk[ch] = (c < N) ? Q[c * LD + k] : 0.0;
            }
            const double tqk = tq[k];
            for (int bb = 0; bb < RB; ++bb) {
                const int i = i0 + bb;
                double fik = 0.0, wik = 0.0;
                if (i < N) {
                    fik = F[k * LD + i];
                    const double gik = G[k * LD + i];
                    wik = (SIGNED ? dsg[i] * gik : gik) - Rtop[i] * fik;
                }
                vy[bb] += fik * tqk; vg[bb] += wik * tqk;
                for (int ch = 0; ch < CH; ++ch) { ay[bb][ch] += fik * tk[ch]; aw[bb][ch] += wik * tk[ch]; }
            }
        }
        wave_sync();  // every lane has read rows i0.. of F before they are overwritten
        for (int bb = 0; bb < RB; ++bb) {
            const int i = i0 + bb;
            if (i < N) {
                const double rt = Rtop[i];
                for (int ch = 0; ch < CH; ++ch) {
                    const int c = ch * SMRT_LANES + lane;
                    if (c < N) {
                        const double fic = F[c * LD + i], gic = G[c * LD + i];
                        Wk[c * LD + i] = ay[bb][ch] + gic;
                        F[c * LD + i] = aw[bb][ch] + (SIGNED ? dsg[i] * fic : fic) - rt * gic;
                    }
                }
                if (lane == 0) { upb[i] = vy[bb] + Bl; gvec[i] = vg[bb] + (1.0 - rt) * Bl; }
            }
        }
    }
    block_sync();
}

// ---- Cholesky of one 16 x 16 block in registers ---------------------------------------------------------------
// a: the symmetric positive definite block (both triangles) in MFMA accumulator layout -- lane 16 g + c, register r holds
// element (4 r + g, c); w: the identity on entry.  On return the lower triangle of a is L (a = L L^T) and w = L^-1 (lower
// triangular); whatever sits above the diagonals is meaningless.  Right-looking, one column per step: row K reaches the
// four lane rows through v_permlane16/32_swap, column K and the pivot through DPP row broadcasts -- no LDS, no
// v_readlane; the row operations of the step (scale row K by 1 / L_KK, subtract L_iK times it from the rows below) are
// applied to w at once, so the inverse of the factor comes out of the same 16 steps (two independent dependency chains
// in one instruction stream).  ok goes false (uniformly) on a pivot that is not positive.
template <int K>
SMRT_DEV void chol16_step(double (&a)[4], double (&w)[4], bool& ok, int g, int c) {
    constexpr int r0 = K >> 2, g0 = K & 3;
    const double rowk = rows_bcast<g0>(a[r0]);      // A[K][c] = A[c][K] (the trailing matrix is kept symmetric)
    const double wrow = rows_bcast<g0>(w[r0]);      // W[K][c]
    const double akk = row_bcast16<K>(rowk);
    if (!(akk > 0.0)) ok = false;
    const double rk = fast_rsqrt(ok ? akk : 1.0);
    const double ljk = (c > K) ? rowk * rk : 0.0;   // L[c][K] for the columns still to be eliminated (finished columns stay)
    const double wk = wrow * rk;                    // row K of the inverse: final
    // rows 4 r + g: registers r < r0 hold finished rows only (nothing to do, known at compile time), registers r > r0 rows
    // below K; only register r0 mixes the three cases (lane rows g < g0, g == g0, g > g0)
#pragma unroll
    for (int r = r0; r < 4; ++r) {
        const double lik = row_bcast16<K>(a[r]) * rk;   // L[i][K] for i >= K
        a[r] = (c == K) ? lik : a[r] - lik * ljk;
        if (r > r0) w[r] -= lik * wk;
        else {
            // one fused multiply-add for the three cases: the multiplier of row K itself is L_KK - 1, which turns
            // W[K][c] into W[K][c] - (L_KK - 1) W[K][c] / L_KK = W[K][c] / L_KK
            const double m = (g > g0) ? lik : ((g == g0) ? lik - 1.0 : 0.0);
            w[r] -= m * wk;
        }
    }
}
SMRT_DEV void chol16_reg(double (&a)[4], double (&w)[4], bool& ok, int g, int c) {
    chol16_step<0>(a, w, ok, g, c); chol16_step<1>(a, w, ok, g, c); chol16_step<2>(a, w, ok, g, c); chol16_step<3>(a, w, ok, g, c);
    chol16_step<4>(a, w, ok, g, c); chol16_step<5>(a, w, ok, g, c); chol16_step<6>(a, w, ok, g, c); chol16_step<7>(a, w, ok, g, c);
    chol16_step<8>(a, w, ok, g, c); chol16_step<9>(a, w, ok, g, c); chol16_step<10>(a, w, ok, g, c); chol16_step<11>(a, w, ok, g, c);
    chol16_step<12>(a, w, ok, g, c); chol16_step<13>(a, w, ok, g, c); chol16_step<14>(a, w, ok, g, c); chol16_step<15>(a, w, ok, g, c);
}

// ---- blocked Cholesky of two SPD matrices side by side on the matrix core (N <= 64) --------------------------
// Right-looking with 16-column blocks: the 16x16 diagonal block is factorised (and the inverse of its factor formed) by
// one wavefront per matrix entirely in registers (chol16_reg); the panel below (L_IJ = A_IJ inv(L_JJ)^T) and the
// trailing update (A_IK -= L_IJ L_KJ^T) are MFMA tile GEMMs.  3 workgroup barriers per block column (12 for N = 64)
// instead of one per column, and the O(N^3) part runs on the matrix core.
template <int NT, bool PK = false>
SMRT_DEV bool chol2_mfma(double* A0, double* A1, double* inv /* [2][256] */, int* fail, int N, int LD,
                         double* inv_out = nullptr /* [4][256]: inverses of the diagonal blocks of the first factor */) {
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    if (t == 0) *fail = 0;
    block_sync();
    // (a) diagonal block J of matrix mi: L_JJ and its inverse, one wavefront, in registers
    auto diag = [&](int J, int mi) {
        const int b0 = J * 16;
        double* A = mi ? A1 : A0;
        // the tile in registers in MFMA accumulator layout (lane 16 g + c, register r: element (4 r + g, c)), identity
        // padding; both triangles are filled from the stored lower one (chol16_reg keeps the trailing part symmetric)
        double a[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + lk, gi = b0 + i, gj = b0 + lr;
            const int hi = gi > gj ? gi : gj, lo = gi > gj ? gj : gi;
            const double v = A[sidx<PK>(hi < N ? hi : N - 1, lo < N ? lo : N - 1, LD)];
            a[r] = (hi < N) ? v : ((i == lr) ? 1.0 : 0.0);
            w[r] = (i == lr) ? 1.0 : 0.0;
        }
        bool ok = true;
        chol16_reg(a, w, ok, lk, lr);
        if (!ok && lane == 0) *fail = 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + lk, gi = b0 + i, gj = b0 + lr;
            if (gi < N && gj < N && lr <= i) A[sidx<PK>(gi, gj, LD)] = a[r];
            const double wv = (lr <= i) ? w[r] : 0.0;     // (L^-1)[i][j = lr]
            inv[mi * 256 + lr * 16 + i] = wv;
            if (mi == 0 && inv_out) inv_out[J * 256 + lr * 16 + i] = wv;
        }
    };
    // (c) one tile of the trailing update A_IK -= L_IJ L_KJ^T, I >= K > J
    auto trailing_tile = [&](int J, int mi, int I, int K) {
        const int b0 = J * 16;
        double* A = mi ? A1 : A0;
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        const int i = I * 16 + lr, ic = i < N ? i : N - 1;
        const int j = K * 16 + lr, jc = j < N ? j : N - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int gk = b0 + 4 * kk + lk, gkc = gk < N ? gk : N - 1;
            const double av = A[sidx<PK>(ic, gkc, LD)], bv = A[sidx<PK>(jc, gkc, LD)];
            mfma_f64_16x16x4((i < N && gk < N) ? av : 0.0, (j < N && gk < N) ? bv : 0.0, c);
        }
        tile_foreach(I, K, N, [&](int reg, int row_, int col) { if (!PK || row_ >= col) A[sidx<PK>(row_, col, LD)] -= c[reg]; });
    };
    // Look-ahead (four or more wavefronts): the pair of wavefronts that factorises the diagonal blocks takes tile
    // (J + 1, J + 1) of the trailing update first and goes straight on to block J + 1, while the other wavefronts do the
    // rest of the update -- the long dependent chain of the next diagonal block no longer waits behind a barrier with
    // half the workgroup idle.
    constexpr bool AHEAD = NW >= 4;
    for (int J = 0; J < RT; ++J) {
        const int b0 = J * 16;
        if (!AHEAD || J == 0)
            for (int mi = wave; mi < 2; mi += NW) diag(J, mi);
        block_sync();             // (with look-ahead and J > 0: the barrier behind the trailing update of block J - 1)
        if (*fail) return false;  // uniform
        // (b) panel below the diagonal block: L_IJ = A_IJ inv(L_JJ)^T
        {
            const int nt_ = 2 * (RT - 1 - J);
            for (int tix = wave; tix < nt_; tix += NW) {
                const int mi = tix & 1, I = J + 1 + (tix >> 1);
                double* A = mi ? A1 : A0;
                double c[4] = {0.0, 0.0, 0.0, 0.0};
                const int i = I * 16 + lr, ic = i < N ? i : N - 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = 4 * kk + lk, gk = b0 + k, gkc = gk < N ? gk : N - 1;
                    const double av = A[sidx<PK>(ic, gkc, LD)];
                    const double bv = inv[mi * 256 + k * 16 + lr];          // (L^-1)[lr][k] = invT[k][lr]
                    mfma_f64_16x16x4((i < N && gk < N) ? av : 0.0, bv, c);
                }
                tile_foreach(I, J, N, [&](int reg, int row_, int col) { A[sidx<PK>(row_, col, LD)] = c[reg]; });
            }
        }
        block_sync();
        // (c) trailing update
        {
            const int nb = RT - 1 - J;
            const int ntri = nb * (nb + 1) / 2;
            if (AHEAD && nb > 0) {
                if (wave < 2) {
                    trailing_tile(J, wave, J + 1, J + 1);
                    wave_sync_lds();       // the block is read back in another lane mapping
                    diag(J + 1, wave);
                } else {
                    for (int tix = wave - 2; tix < 2 * (ntri - 1); tix += NW - 2) {
                        const int mi = tix & 1;
                        int q = (tix >> 1) + 1, Ir = 0;      // (q = 0 is the tile the other two wavefronts took)
                        while ((Ir + 1) * (Ir + 2) / 2 <= q) ++Ir;
                        const int Kr = q - Ir * (Ir + 1) / 2;
                        trailing_tile(J, mi, J + 1 + Ir, J + 1 + Kr);
                    }
                }
            } else {
                for (int tix = wave; tix < 2 * ntri; tix += NW) {
                    const int mi = tix & 1;
                    int q = tix >> 1, Ir = 0;
                    while ((Ir + 1) * (Ir + 2) / 2 <= q) ++Ir;
                    const int Kr = q - Ir * (Ir + 1) / 2;
                    trailing_tile(J, mi, J + 1 + Ir, J + 1 + Kr);
                }
            }
        }
        if (!AHEAD) block_sync();   // (with look-ahead the barrier at the top of the next iteration does it)
    }
    if (AHEAD) block_sync();
    return true;
}

// ---- blocked triangular solve on the matrix core: Bm <- Lp^-T Bm  (N <= 64) ---------------------------------
// The (up to four) 16x16 diagonal blocks of Lp are inverted once (one wavefront per block, lane = column of the
// inverse, forward substitution); then, block row by block row from the bottom,
//   X_I = inv(L_II)^T (B_I - sum_{J>I} L_JI^T X_J)
// is two MFMA GEMMs per 16x16 tile -- the accumulator layout of the first is exactly the B-operand layout of the
// second (c[reg] = R[lk + 4 reg][lr] = B[k = 4 kk + lk][j = lr] for kk = reg), so nothing moves between them.
template <int NT>
SMRT_DEV void lt_solve_mfma(const double* Lp, double* Bm, double* inv /* [4][16*16] */, int N, int LD,
                            bool have_inv = false /* inv already holds the block inverses (from chol2_mfma) */) {
    const int t = tid(), lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES, lr = lane & 15, lk = lane >> 4;
    constexpr int NW = NT / SMRT_LANES;
    const int RT = (N + 15) >> 4;
    // inverse of the diagonal blocks: inv[I][j*16 + i] = (L_II^-1)[i][j]  (identity padding beyond N).
    // lane (mod 16) = row i of the block with the row in registers; entries of other rows come by wave_bcast
    // (loading the block through broadcast LDS reads made the compiler hoist all 136 loads into registers).
    for (int I = wave; I < RT && !have_inv; I += NW) {
        const int b0 = I * 16, gi = b0 + lr;
        double row[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int gk = b0 + k;
            const int gic = gi < N ? gi : N - 1, gkc = gk < N ? gk : N - 1;
            const double v = Lp[gkc * LD + gic];
            row[k] = (gi < N && gk < N && k <= lr) ? v : ((k == lr) ? 1.0 : 0.0);
        }
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double acc = (i == lr) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < i) { const double lik = wave_bcast(row[k], i); acc -= lik * ((k >= lr) ? x[k] : 0.0); }
            const double dii = wave_bcast(row[i], i);
            x[i] = (i >= lr) ? acc * fast_rcp(dii) : 0.0;
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) inv[I * 256 + lr * 16 + i] = x[i];
        }
    }
    if (!have_inv) block_sync();
    // A column tile of the solution depends on nothing but itself: a wavefront carries TPW of them from the bottom block
    // row to the top without a workgroup barrier, the operand of Lp loaded once for the three and the loads of a chunk
    // of 16 rows all in flight before its matrix-core passes.
    constexpr int TPW = 3;
    for (int tb = wave; tb < RT; tb += NW * TPW) {
        for (int I = RT - 1; I >= 0; --I) {
            double c[TPW][4];
#pragma unroll
            for (int q = 0; q < TPW; ++q)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) c[q][reg] = 0.0;
            const int i = I * 16 + lr, ic = i < N ? i : N - 1;
            const int nq = (RT - tb + NW - 1) / NW < TPW ? (RT - tb + NW - 1) / NW : TPW;   // tiles carried (uniform)
            // acc = sum_{k in later blocks} L[k][i] X[k][j]
            for (int k0 = (I + 1) * 16; k0 < N; k0 += 16) {
                double av[4], bv[TPW][4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = k0 + 4 * kk + lk, kc = k < N ? k : N - 1;
                    const double a = Lp[ic * LD + kc];
                    av[kk] = (i < N && k < N) ? a : 0.0;
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        bv[q][kk] = 0.0;
                        if (q < nq) {
                            const int j = (tb + q * NW) * 16 + lr, jc = j < N ? j : N - 1;
                            const double y = Bm[jc * LD + kc];
                            bv[q][kk] = (j < N && k < N) ? y : 0.0;
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int q = 0; q < TPW; ++q)
                        if (q < nq) mfma_f64_16x16x4(av[kk], bv[q][kk], c[q]);
            }
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int tj = tb + q * NW;
                if (tj < RT) {   // uniform
                    const int j = tj * 16 + lr, jc = j < N ? j : N - 1;
                    // R = B_I - acc in accumulator layout
                    double r[4];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = I * 16 + lk + 4 * reg;
                        const int rowc = row < N ? row : N - 1;
                        const double bvr = Bm[jc * LD + rowc];
                        r[reg] = ((row < N && j < N) ? bvr : 0.0) - c[q][reg];
                    }
                    // X = inv(L_II)^T R : A[i][k] = inv[k][i]
                    double x[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = 4 * kk + lk;                          // row of inv(L_II)
                        const double a = inv[I * 256 + lr * 16 + k];       // (L_II^-1)[k][lr]
                        mfma_f64_16x16x4(a, r[kk], x);
                    }
                    tile_foreach(I, tj, N, [&](int reg, int row, int col) { Bm[col * LD + row] = x[reg]; });
                }
            }
            wave_sync();   // the rows just written are operands of this wavefront's next block row
        }
    }
    block_sync();
}

}  // namespace smrt
