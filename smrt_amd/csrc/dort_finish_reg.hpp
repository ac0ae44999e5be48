// The register-resident finish kernel of the LDS pipeline (passive mode, streams x polarisations N <= 64): ONE
// wavefront per (snowpack, frequency) pair, every N x N matrix of the layer recursion held in registers in the
// accumulator layout of v_mfma_f64_16x16x4_f64, no workgroup barrier and no LDS round trip for a matrix.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
//
// What is solved is the boundary system of smrt/rtsolver/dort.py:263-488 (the same linear system as the other finish
// kernels, eliminated in a different order); tests/studies/admittance_recursion.py is the NumPy statement of the
// algebra below, checked against the oracle.
//
// State carried bottom-up instead of the reflection matrix:  delta = -C s + c  at a level, s = I_up + I_dn,
// delta = I_up - I_dn ("admittance" C, N x N; source c).  With the symmetric reduction
//   E+ = D A+,  A+ = L+^-T B',     E- = D A-,  A- = -L+ B' Sigma^-1,     A+^T A- = -Sigma,
// so the inverses of the eigenvector matrices are transposes, and a layer takes (hats: C^ = D^-1 C D)
//   H = A+^T C^ A+,   P = (H + Sigma)^-1,   M3 = Sigma (1 - t^2) + 2 (Sigma t) P (t Sigma),   Theta = 2 M3^-1 - Sigma^-1,
//   C^' = A- Theta A-^T                                     (t = exp(-sigma thickness))
// and a Flat interface (diagonal r1, t1, r2, t2)  Y = a - b C',  C_u = -t2^-1 (c - d C') Y^-1 t2  with diagonal a..d.
// Every matrix that is inverted is "positive diagonal + (nearly) symmetric positive definite": Gauss-Jordan WITHOUT
// pivoting (growth <= 62 on the headline batch), which is what makes a register-resident elimination possible at all.
//
// Register layout of a matrix X (padded to 64 x 64): tile (ti, tj), register r, lane l = 16 g + c holds
// X[16 ti + 4 r + g][16 tj + c].  That is the MFMA accumulator layout AND its B-operand layout for the k-slab r,
// and the A-operand layout of X^T: the native product of two matrices in registers is  X^T Y  (gemm_tn), and every
// product of the recursion is arranged to be of that form (the chain runs on the transposes H^T, P^T, M3^T, Theta^T).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_passive.hpp"

namespace smrt {
namespace rg {

constexpr int TM = 4;   // 16 x 16 tiles per side
struct Mat { double v[TM][TM][4]; };

struct LaneId { int lane, g, c; };
SMRT_DEV LaneId lane_id() { LaneId L; L.lane = tid() & (SMRT_LANES - 1); L.g = L.lane >> 4; L.c = L.lane & 15; return L; }

SMRT_DEV void zero(Mat& M) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) M.v[i][j][r] = 0.0;
}

// t = X^T Y for two 16 x 16 tiles in register layout (4 MFMAs)
SMRT_DEV void tile_tn(double (&z)[4], const double (&x)[4], const double (&y)[4]) {
    z[0] = z[1] = z[2] = z[3] = 0.0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mfma_f64_16x16x4(x[kk], y[kk], z);
}
SMRT_DEV void tile_tn_acc(double (&z)[4], const double (&x)[4], const double (&y)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mfma_f64_16x16x4(x[kk], y[kk], z);
}
// transpose of a tile: X^T I on the matrix core
SMRT_DEV void tile_transpose(double (&z)[4], const double (&x)[4], const LaneId& L) {
    z[0] = z[1] = z[2] = z[3] = 0.0;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mfma_f64_16x16x4(x[kk], (4 * kk + L.g == L.c) ? 1.0 : 0.0, z);
}

// Z = X^T Y on the leading nt x nt tiles (Z must not alias X or Y)
SMRT_DEV void gemm_tn(Mat& Z, const Mat& X, const Mat& Y, int nt) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TM; ++tj) {
            if (ti < nt && tj < nt) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int tk = 0; tk < TM; ++tk)
                    if (tk < nt) tile_tn_acc(acc, X.v[tk][ti], Y.v[tk][tj]);
#pragma unroll
                for (int r = 0; r < 4; ++r) Z.v[ti][tj][r] = acc[r];
            }
        }
}

// ---- in-register inverse of a 16 x 16 tile, Gauss-Jordan without pivoting ------------------------------------------
// Step k: row k reaches every lane group through two lane-row swaps (v_permlane16_swap / v_permlane32_swap), column k
// through DPP row broadcasts; the in-place update (Gauss-Jordan inversion in place) is one fused multiply-add per register,
// uniform over the tile, column k included.
template <int K>
SMRT_DEV void inv16_step(double (&d)[4], const LaneId& L) {
    constexpr int r0 = K >> 2, g0 = K & 3;
#ifdef SMRT_INV16_MFMA_BCAST
    double rk4[4] = {0.0, 0.0, 0.0, 0.0};
    mfma_f64_16x16x4((L.g == g0) ? 1.0 : 0.0, d[r0], rk4);   // every register / lane group: D[K][c]
    double rk = rk4[0];
#else
    double rk = rows_bcast<g0>(d[r0]);                       // every lane group: D[K][c]
#endif
    const double piv = row_bcast16<K>(rk);
    const double pinv = fast_rcp(piv);
    // column K itself takes part in the same fused multiply-add as every other column: with piv + 1 in its place in the
    // pivot row, D[i][K] - m_i (piv + 1) = -m_i (m_i = D[i][K] / piv), and for the pivot row, whose multiplier is
    // 1 - 1 / piv, piv - (1 - 1 / piv)(piv + 1) = 1 / piv: no select per register.  (The cancellation costs |piv| ulps
    // of relative accuracy in that column: 1e-13 for the largest pivots of the recursion.)
    rk = (L.c == K) ? piv + 1.0 : rk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double f = row_bcast16<K>(d[r]);           // D[4 r + g][K]
        double m = f * pinv;
        if (r == r0) m = (L.g == g0) ? (1.0 - pinv) : m;  // row K itself: new = old - (1 - pinv) old = old pinv
        d[r] -= m * rk;
    }
}
// Two columns per step (K even): the 2 x 2 pivot block P = D[K..K+1][K..K+1] is inverted in closed form and rows / columns
// K, K + 1 are eliminated together -- the same in-place Gauss-Jordan with the unit-vector trick, with HALF the dependent
// chains (row broadcast -> pivot -> reciprocal -> multipliers -> update).  Measured: no gain (see inv16), so not the default.  det P > 0 for the matrices of the recursion (positive definite symmetric
// part), like the 1 x 1 pivots.
template <int K>
SMRT_DEV void inv16_step2(double (&d)[4], const LaneId& L) {
    constexpr int r0 = K >> 2, g0 = K & 3;   // rows K, K + 1: register r0, lane rows g0 and g0 + 1
    double rk0 = rows_bcast<g0>(d[r0]);      // D[K][c]
    double rk1 = rows_bcast<g0 + 1>(d[r0]);  // D[K + 1][c]
    const double pa = row_bcast16<K>(rk0), pb = row_bcast16<K + 1>(rk0);
    const double pc = row_bcast16<K>(rk1), pd = row_bcast16<K + 1>(rk1);
    const double idet = fast_rcp(pa * pd - pb * pc);
    const double i00 = pd * idet, i01 = -pb * idet, i10 = -pc * idet, i11 = pa * idet;   // P^-1
    const bool c0 = (L.c == K), c1 = (L.c == K + 1);
    rk0 = c0 ? 1.0 : (c1 ? 0.0 : rk0);       // rows K, K + 1 with the columns K, K + 1 replaced by the identity
    rk1 = c0 ? 0.0 : (c1 ? 1.0 : rk1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double f0 = row_bcast16<K>(d[r]), f1 = row_bcast16<K + 1>(d[r]);   // D[4 r + g][K], [K + 1]
        double m0 = f0 * i00 + f1 * i10, m1 = f0 * i01 + f1 * i11;                 // [f0 f1] P^-1
        double e = 0.0;                                                             // entry of the unit columns in this row
        if (r == r0) {                                                              // rows K, K + 1: I - P^-1
            if (L.g == g0) { m0 = 1.0 - i00; m1 = -i01; e = c0 ? 1.0 : 0.0; }
            if (L.g == g0 + 1) { m0 = -i10; m1 = 1.0 - i11; e = c1 ? 1.0 : 0.0; }
        }
        const double old = (c0 || c1) ? e : d[r];
        d[r] = old - m0 * rk0 - m1 * rk1;
    }
}
// The same elimination with the NEXT pivot row carried one step ahead.  In inv16_step every step is one dependent chain:
// update of the tile -> row K to all lane rows (two lane-row swaps) -> pivot -> reciprocal (seed + two Newton steps) ->
// multipliers -> update, about 290 cycles of latency for some 45 instructions.  Here row K + 1 is fetched as it stands at
// the beginning of step K (that fetch hangs on step K - 1 only), eliminated in its broadcast copy with the same two
// operations the tile applies to it (rn' = rn - m rk': bitwise the row the tile will hold), and the reciprocal of ITS
// pivot is started at once -- behind it runs the update of the tile, which no longer sits between two reciprocals.  What
// is left of the recurrence from one reciprocal to the next: multiplier, update of the copy, pivot broadcast, reciprocal.
// State between the steps: rk (row K in every lane row), piv, pinv.  Same results as inv16_step, bit for bit.
SMRT_DEV void inv16_la_begin(const double (&d)[4], double& rk, double& piv, double& pinv) {
    rk = rows_bcast<0>(d[0]);
    piv = row_bcast16<0>(rk);
    pinv = fast_rcp(piv);
}
// (the two halves of a step separately, for callers that issue matrix-core work between them: state of the half step in s)
struct Inv16Half { double rn, pn, pinvn, rkp; };
template <int K>
SMRT_DEV void inv16_la_pivot(const double (&d)[4], const double& rk, const double& piv, const double& pinv, Inv16Half& s, const LaneId& L) {
    constexpr int K1 = (K < 15) ? K + 1 : 15;
    s.rn = 0.0; s.pn = 1.0; s.pinvn = 1.0;
    if (K < 15) s.rn = rows_bcast<(K1 & 3)>(d[K1 >> 2]);   // row K + 1 before this step, every lane row
    s.rkp = (L.c == K) ? piv + 1.0 : rk;                    // (the unit-vector trick of inv16_step)
    if (K < 15) {
        const double mn = row_bcast16<K>(s.rn) * pinv;
        s.rn -= mn * s.rkp;                                 // row K + 1 after this step
        s.pn = row_bcast16<K1>(s.rn);
        s.pinvn = fast_rcp(s.pn);
    }
}
template <int K>
SMRT_DEV void inv16_la_update(double (&d)[4], double& rk, double& piv, double& pinv, const Inv16Half& s, const LaneId& L) {
    constexpr int r0 = K >> 2, g0 = K & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double f = row_bcast16<K>(d[r]);              // D[4 r + g][K]
        double m = f * pinv;
        if (r == r0) m = (L.g == g0) ? (1.0 - pinv) : m;
        d[r] -= m * s.rkp;
    }
    rk = s.rn; piv = s.pn; pinv = s.pinvn;
}
template <int K>
SMRT_DEV void inv16_la_step(double (&d)[4], double& rk, double& piv, double& pinv, const LaneId& L) {
    Inv16Half s;
    inv16_la_pivot<K>(d, rk, piv, pinv, s, L);
    inv16_la_update<K>(d, rk, piv, pinv, s, L);
}
SMRT_DEV void inv16_la(double (&d)[4], const LaneId& L) {
    double rk, piv, pinv;
    inv16_la_begin(d, rk, piv, pinv);
    inv16_la_step<0>(d, rk, piv, pinv, L); inv16_la_step<1>(d, rk, piv, pinv, L); inv16_la_step<2>(d, rk, piv, pinv, L);
    inv16_la_step<3>(d, rk, piv, pinv, L); inv16_la_step<4>(d, rk, piv, pinv, L); inv16_la_step<5>(d, rk, piv, pinv, L);
    inv16_la_step<6>(d, rk, piv, pinv, L); inv16_la_step<7>(d, rk, piv, pinv, L); inv16_la_step<8>(d, rk, piv, pinv, L);
    inv16_la_step<9>(d, rk, piv, pinv, L); inv16_la_step<10>(d, rk, piv, pinv, L); inv16_la_step<11>(d, rk, piv, pinv, L);
    inv16_la_step<12>(d, rk, piv, pinv, L); inv16_la_step<13>(d, rk, piv, pinv, L); inv16_la_step<14>(d, rk, piv, pinv, L);
    inv16_la_step<15>(d, rk, piv, pinv, L);
}
SMRT_DEV void inv16(double (&d)[4], const LaneId& L) {
#if !defined(SMRT_INV16_NO_LOOKAHEAD) && !defined(SMRT_INV16_TWO_COLUMNS)
    inv16_la(d, L);
    return;
#endif
#ifndef SMRT_INV16_TWO_COLUMNS   // (two columns per step: same instruction count, same time -- 42.5 vs 42.3 ms per step: the
                                  // elimination is issue bound, not latency bound; kept as an opt-in build)
    inv16_step<0>(d, L); inv16_step<1>(d, L); inv16_step<2>(d, L); inv16_step<3>(d, L);
    inv16_step<4>(d, L); inv16_step<5>(d, L); inv16_step<6>(d, L); inv16_step<7>(d, L);
    inv16_step<8>(d, L); inv16_step<9>(d, L); inv16_step<10>(d, L); inv16_step<11>(d, L);
    inv16_step<12>(d, L); inv16_step<13>(d, L); inv16_step<14>(d, L); inv16_step<15>(d, L);
#else
    inv16_step2<0>(d, L); inv16_step2<2>(d, L); inv16_step2<4>(d, L); inv16_step2<6>(d, L);
    inv16_step2<8>(d, L); inv16_step2<10>(d, L); inv16_step2<12>(d, L); inv16_step2<14>(d, L);
#endif
}

// cyclic shift of the leading NTT x NTT tiles: new[i][j] = old[(i + 1) % NTT][(j + 1) % NTT]
template <int NTT>
SMRT_DEV void shift_tiles(Mat& M) {
    Mat T;
#pragma unroll
    for (int i = 0; i < NTT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) T.v[i][j][r] = M.v[i][j][r];
#pragma unroll
    for (int i = 0; i < NTT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) M.v[i][j][r] = T.v[(i + 1) % NTT][(j + 1) % NTT][r];
}

// M <- M^-1 on the leading nt x nt tiles (identity padding inside the last tile), block Gauss-Jordan in place without
// pivoting; the tiles are rotated after every block step so that the running diagonal block is always tile (0, 0)
// (one copy of the step code for every block).  Look-ahead: tile (1, 1) -- the next diagonal block -- is updated first
// and its 16 x 16 elimination (a long dependent chain on the vector unit) is issued behind the matrix-core work of the
// rest of the step, which runs in its shadow.
SMRT_DEV void invert(Mat& M, int nt, const LaneId& L) {
    double D[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) D[r] = M.v[0][0][r];
    inv16(D, L);
#if !defined(SMRT_HOST_EMU)
#pragma nounroll
#endif
    for (int step = 0; step < nt; ++step) {
        double DT[4], Dn[4] = {0.0, 0.0, 0.0, 0.0};
        tile_transpose(DT, D, L);
        double R[TM][4];
#pragma unroll
        for (int j = 1; j < TM; ++j)
            if (j < nt) tile_tn(R[j], DT, M.v[0][j]);               // D M[0][j]
#pragma unroll
        for (int i = 1; i < TM; ++i)
            if (i < nt) {
                double LT[4], n0[4];
                tile_transpose(LT, M.v[i][0], L);
                tile_tn(n0, LT, D);                                   // M[i][0] D
#pragma unroll
                for (int r = 0; r < 4; ++r) M.v[i][0][r] = -n0[r];
#pragma unroll
                for (int j = 1; j < TM; ++j)
                    if (j < nt) {
                        double u[4];
                        tile_tn(u, LT, R[j]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) M.v[i][j][r] -= u[r];
                        if (i == 1 && j == 1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) Dn[r] = M.v[1][1][r];
                        }
                    }
            }
        if (step + 1 < nt) inv16(Dn, L);   // (its inputs are ready after the first update above)
#pragma unroll
        for (int j = 1; j < TM; ++j)
            if (j < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) M.v[0][j][r] = R[j][r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) M.v[0][0][r] = D[r];
        if (nt == 4) shift_tiles<4>(M);
        else if (nt == 3) shift_tiles<3>(M);
        else if (nt == 2) shift_tiles<2>(M);
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = Dn[r];
    }
}

// ---- products with one operand streamed, results in place (two matrices in registers at most) ----------------------
// Y <- X^T Y on the leading nt x nt tiles, tile column by tile column (16 doubles of temporaries)
SMRT_DEV void gemm_tn_inplace(Mat& Y, const Mat& X, int nt) {
#pragma unroll
    for (int tj = 0; tj < TM; ++tj) {
        if (tj < nt) {
            double acc[TM][4];
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
                acc[ti][0] = acc[ti][1] = acc[ti][2] = acc[ti][3] = 0.0;
                if (ti < nt) {
#pragma unroll
                    for (int tk = 0; tk < TM; ++tk)
                        if (tk < nt) tile_tn_acc(acc[ti], X.v[tk][ti], Y.v[tk][tj]);
                }
            }
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
                if (ti < nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Y.v[ti][tj][r] = acc[ti][r];
                }
        }
    }
}
// the wavefront's matrix slot in LDS: element (tile ti, tj; register r; lane) at ((4 ti + tj) 4 + r) 64 + lane
SMRT_DEV void slot_store_tile(double* slot, int ti, int tj, const double (&t)[4], const LaneId& L) {
#pragma unroll
    for (int r = 0; r < 4; ++r) slot[((4 * ti + tj) * 4 + r) * 64 + L.lane] = t[r];
}
SMRT_DEV void slot_load_tile(double (&t)[4], const double* slot, int ti, int tj, const LaneId& L) {
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = slot[((4 * ti + tj) * 4 + r) * 64 + L.lane];
}
SMRT_DEV void slot_load(Mat& M, const double* slot, int nt, const LaneId& L) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TM; ++tj)
            if (ti < nt && tj < nt) slot_load_tile(M.v[ti][tj], slot, ti, tj, L);
}
// Z = X^T Y with X in the LDS slot (one tile column of X in registers at a time)
SMRT_DEV void gemm_tn_slot(Mat& Z, const double* slot, const Mat& Y, int nt, const LaneId& L) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
        if (ti < nt) {
            double xc[TM][4];
#pragma unroll
            for (int tk = 0; tk < TM; ++tk) {
                xc[tk][0] = xc[tk][1] = xc[tk][2] = xc[tk][3] = 0.0;
                if (tk < nt) slot_load_tile(xc[tk], slot, tk, ti, L);
            }
#pragma unroll
            for (int tj = 0; tj < TM; ++tj)
                if (tj < nt) {
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int tk = 0; tk < TM; ++tk)
                        if (tk < nt) tile_tn_acc(acc, xc[tk], Y.v[tk][tj]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) Z.v[ti][tj][r] = acc[r];
                }
        }
    }
}
// M <- M^T on the leading nt x nt tiles
SMRT_DEV void transpose_inplace(Mat& M, int nt, const LaneId& L) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = ti; tj < TM; ++tj)
            if (tj < nt) {
                double a[4], bq[4];
                tile_transpose(a, M.v[ti][tj], L);
                if (tj != ti) {
                    tile_transpose(bq, M.v[tj][ti], L);
#pragma unroll
                    for (int r = 0; r < 4; ++r) M.v[ti][tj][r] = bq[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) M.v[tj][ti][r] = a[r];
            }
}

// ---- vectors: one element per lane in a register for elementwise work; a matrix-vector product or a scaling reads its
// operand from an LDS exchange vector in "row form" (element 16 tj + c) or "column form" (element 16 ti + 4 r + g) ----
SMRT_DEV void put(double* x, double v, const LaneId& L) { x[L.lane] = v; }
// y[i] = sum_j X[i][j] w[j]  -> out (LDS)
SMRT_DEV void matvec(const Mat& X, const double* w, double* out, int nt, const LaneId& L) {
    double wv[TM];
#pragma unroll
    for (int tj = 0; tj < TM; ++tj) wv[tj] = w[16 * tj + L.c];
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
        if (ti < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int tj = 0; tj < TM; ++tj)
                    if (tj < nt) acc += X.v[ti][tj][r] * wv[tj];
                acc = group_sum<16>(acc);
                if (L.c == 0) out[16 * ti + 4 * r + L.g] = acc;
            }
        }
    }
    wave_sync();
}
// y[j] = sum_i X[i][j] v[i]  -> out (LDS)
SMRT_DEV void matvec_t(const Mat& X, const double* v, double* out, int nt, const LaneId& L) {
#pragma unroll
    for (int tj = 0; tj < TM; ++tj) {
        if (tj < nt) {
            double acc = 0.0;
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
                if (ti < nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc += X.v[ti][tj][r] * v[16 * ti + 4 * r + L.g];
                }
            acc += shfl_xor(acc, 16);
            acc += shfl_xor(acc, 32);
            if (L.g == 0) out[16 * tj + L.c] = acc;
        }
    }
    wave_sync();
}
// X[i][j] <- factor rowf[i] X[i][j] colf[j] + (i == j) diag[i] on the leading nt x nt tiles (null pointers: factor 1 / nothing)
SMRT_DEV void scale_add_diag(Mat& X, const double* rowf, const double* colf, const double* diag, double factor, int nt, const LaneId& L) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TM; ++tj)
            if (ti < nt && tj < nt) {
                const double cf = (colf ? colf[16 * tj + L.c] : 1.0) * factor;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + L.g;
                    double x = X.v[ti][tj][r] * cf;
                    if (rowf) x *= rowf[row];
                    if (diag && ti == tj && 4 * r + L.g == L.c) x += diag[row];
                    X.v[ti][tj][r] = x;
                }
            }
}

// Staged matrices are column-major with leading dimension LD (element (r, c) at p[c LD + r]).  A tile load is "uniform
// base + one 32-bit lane offset": the lane offsets are the same for every tile of a layer (LaneOffsets), the tile offset
// is scalar arithmetic -- no per-tile 64-bit address registers.  Everything outside N x N reads as zero.
struct LaneOffsets {
    unsigned direct;   // c LD + g      lane part of element (16 ti + 4 r + g, 16 tj + c)
    unsigned transp;   // g LD + c      the same element of the transpose
    unsigned blk;      // 16 c + g      ... of a 16 x 16 column-major block (diagonal-block inverses)
    int rl, cl;        // N - g, N - c  row / column bounds of this lane
};
SMRT_DEV LaneOffsets lane_offsets(int LD, int N, const LaneId& L) {
    LaneOffsets o;
    o.direct = (unsigned)(L.c * LD + L.g); o.transp = (unsigned)(L.g * LD + L.c); o.blk = (unsigned)(16 * L.c + L.g);
    o.rl = N - L.g; o.cl = N - L.c;
    return o;
}
SMRT_DEV void load_tile(double (&t)[4], const double* p, int LD, int ti, int tj, const LaneOffsets& o) {
    const bool cin = 16 * tj < o.cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double* q = p + ((16 * tj) * LD + 16 * ti + 4 * r);
        const bool in = cin && 16 * ti + 4 * r < o.rl;
        const double v = q[in ? o.direct : 0u];   // (the tile origin is always inside the matrix)
        t[r] = in ? v : 0.0;
    }
}
// tile (ti, tj) of the TRANSPOSE of p
SMRT_DEV void load_tile_t(double (&t)[4], const double* p, int LD, int ti, int tj, const LaneOffsets& o) {
    const bool cin = 16 * tj < o.cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double* q = p + ((16 * ti + 4 * r) * LD + 16 * tj);
        const bool in = cin && 16 * ti + 4 * r < o.rl;
        const double v = q[in ? o.transp : 0u];
        t[r] = in ? v : 0.0;
    }
}
// The same in two passes -- raw loads first (no branch, no use of the data: every request of a phase is in flight before
// the first wait), masks afterwards.  `have`: the tile exists (uniform); a tile that does not is read at the origin.
SMRT_DEV void load_tile_raw(double (&t)[4], const double* p, int LD, int ti, int tj, bool have, const LaneOffsets& o) {
    const bool cin = 16 * tj < o.cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double* q = p + (have ? ((16 * tj) * LD + 16 * ti + 4 * r) : 0);
        const bool in = have && cin && 16 * ti + 4 * r < o.rl;
        t[r] = q[in ? o.direct : 0u];
    }
}
SMRT_DEV void load_tile_t_raw(double (&t)[4], const double* p, int LD, int ti, int tj, bool have, const LaneOffsets& o) {
    const bool cin = 16 * tj < o.cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double* q = p + (have ? ((16 * ti + 4 * r) * LD + 16 * tj) : 0);
        const bool in = have && cin && 16 * ti + 4 * r < o.rl;
        t[r] = q[in ? o.transp : 0u];
    }
}
SMRT_DEV void mask_tile(double (&t)[4], int ti, int tj, bool have, const LaneOffsets& o) {
    const bool cin = have && 16 * tj < o.cl;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = (cin && 16 * ti + 4 * r < o.rl) ? t[r] : 0.0;
}
// block ti of the [4][256] diagonal-block inverses (identity padded by the prep kernel)
SMRT_DEV void load_block(double (&t)[4], const double* p, int ti, const LaneOffsets& o) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const double* q = p + (ti * 256 + 4 * r); t[r] = q[o.blk]; }
}

constexpr int kRegVectors = 5;                   // 64-double LDS exchange vectors of the register-resident finish kernel
constexpr int kSlotDoubles = TM * TM * 4 * 64;   // one 64 x 64 matrix in register layout
}  // namespace rg

// LDS layout of the register-resident finish kernel: the matrix slot, the exchange vectors, then the tables pair_setup fills
// (stream tables 3 x n_max_stream, layer tables 15 x Lmax, 8 doubles of flags)
SMRT_HD int finish_reg_lds_doubles(int n_max_stream, int Lmax) {
    return rg::kSlotDoubles + rg::kRegVectors * 64 + 3 * n_max_stream + 15 * Lmax + 8;
}

// reflectivities (V, H) of the Flat interface from medium 1 into medium 2 for the stream whose sine in the most
// refringent layer is gsin, ri = relative index of medium 1 (streams.py:195-206); T = 1 - R (core/fresnel.py:446-474)
SMRT_DEV double flat_R(cplx e1, cplx e2, double ri_sin, int pol) {
    double Rv, Rh;
    fresnel_RvRh(e1, e2, sqrt(1.0 - ri_sin * ri_sin), &Rv, &Rh);
    return pol ? Rh : Rv;
}

// Optional phase timing (profiling builds, -DSMRT_REG_TIMING): shader-clock deltas per phase, summed per pair into stage_out
#ifdef SMRT_REG_TIMING
#define SMRT_RT(k) do { const long long now_ = cycle_counter(); rt_acc[rt_cur] += (double)(now_ - rt_t0); rt_t0 = now_; rt_cur = (k); } while (0)
#else
#define SMRT_RT(k) do {} while (0)
#endif
enum { RT_SETUP = 0, RT_VEC, RT_AT, RT_APLUS, RT_HT, RT_INV, RT_ST1, RT_T2, RT_CP, RT_IFACE, RT_POST, RT_SURF, RT_COUNT };

// ------------------------------------------------------------------------------------------------------------
// the per-pair driver: one wavefront (NT = 64)
// ------------------------------------------------------------------------------------------------------------
// Supported: passive mode, N <= 64, Flat interfaces, no / Flat / Reflector substrate, atmosphere, prune_deep_snowpack.
// The host routes batches with process_coherent_layers (T != 1 - R) or a host-evaluated dense substrate to the two-slot
// finish kernel (dort_hip.hip).
SMRT_DEV void dort_pair_passive_reg(const DevBatch& b, long long p, double* lds_base, const DevStage& stg) {
    using namespace rg;
    constexpr int NT = 64, P = 2;
    const LaneId Ln = lane_id();
    const int t = Ln.lane;
    const int nmax = b.n_max_stream;
    const int out_stride = P * b.n_theta;

    // ---- LDS: matrix slot, exchange vectors, tables
    double* const slot = lds_base;
    double* const E0 = lds_base + kSlotDoubles;
    double* const E1 = E0 + 64;
    double* const E2 = E0 + 2 * 64;
    double* const E3 = E0 + 3 * 64;
    double* const E4 = E0 + 4 * 64;
    Lds s;
    {
        double* v = E0 + kRegVectors * 64;
        s.gmu = v; s.gsin = v + nmax; s.outmu = v + 2 * nmax; s.mu = s.w = s.muu = nullptr;
        v += 3 * nmax;
        const int Lm = b.Lmax;
        s.eps_re = v; s.eps_im = v + Lm; s.ks = v + 2 * Lm; s.ka = v + 3 * Lm; s.pa = v + 4 * Lm; s.pb = v + 5 * Lm;
        s.pc = v + 6 * Lm; s.BT = v + 7 * Lm; s.thick = v + 8 * Lm; s.ri = v + 9 * Lm; s.nl = v + 10 * Lm;
        s.slab_re = v + 11 * Lm; s.slab_im = v + 12 * Lm; s.slab_th = v + 13 * Lm; s.lo = v + 14 * Lm;
        s.ints = (int*)(v + 15 * Lm);
    }

    const long long gp = global_pair(b, p);
    const int fi = (int)(gp / b.S), si = (int)(gp % b.S);
    const double frequency = b.frequency[fi];
    int L = b.n_layers[si];
    const double* thickness = b.thickness + (long long)si * b.Lmax;
    const double* fracvol = b.frac_volume + (long long)si * b.Lmax;
    const double* temperature = b.temperature + (long long)si * b.Lmax;
    const double* mp1 = b.p1 + (long long)si * b.Lmax;
    const double* mp2 = b.p2 + (long long)si * b.Lmax;

#ifdef SMRT_REG_TIMING
    double rt_acc[RT_COUNT];
    for (int k = 0; k < RT_COUNT; ++k) rt_acc[k] = 0.0;
    long long rt_t0 = cycle_counter();
    int rt_cur = RT_SETUP;
#endif
    if (t < 8) s.ints[t] = (t == 7) ? ((nmax * P + 1) | 1) : 0;   // [7]: leading dimension of the staged matrices (make_plan)
    block_sync();
    {
        const int prev = b.status[p];
        if (prev != ST_OK) { fail_pair<NT>(b, p, prev, out_stride); return; }
    }
    {
        const int st = pair_setup<NT>(b, s, frequency, L, thickness, fracvol, temperature, mp1, mp2,
                                      b.layer_kind ? b.layer_kind + (long long)si * b.Lmax : nullptr, gp);
        if (st != ST_OK) { fail_pair<NT>(b, p, st, out_stride); return; }
        L = s.ints[6];
    }
    const int n_air = s.ints[5];
    if (b.want_layer_out) {
        double* lo = b.layer_out + p * (long long)b.Lmax * 5;
        for (int l = t; l < b.Lmax; l += NT) {
            const bool in = l < L;
            lo[l * 5 + 0] = in ? s.eps_re[l] : 0.0; lo[l * 5 + 1] = in ? s.eps_im[l] : 0.0;
            lo[l * 5 + 2] = in ? s.ks[l] : 0.0; lo[l * 5 + 3] = in ? s.ka[l] : 0.0;
            lo[l * 5 + 4] = in ? s.nl[l] : 0.0;
        }
    }
    if (b.want_stream_out) {
        double* so = b.stream_out + p * (long long)(1 + nmax);
        if (t == 0) so[0] = (double)n_air;
        for (int j = t; j < nmax; j += NT) so[1 + j] = (j < n_air) ? s.outmu[j] : 0.0;
    }
    int Lk = L;
    if (b.prune_tau > 0.0) Lk = pruned_layer_count<NT>(stg, p * (long long)b.Lmax, L, s.thick, s.pa, b.prune_tau);
    {
        const int bad = first_failed_layer(stg, p * (long long)b.Lmax, Lk);
        if (bad != ST_OK) { fail_pair<NT>(b, p, bad, out_stride); return; }
    }

    double n3 = 0.0;
    // element t of the vectors carried from layer to layer: source c of the relation (physical coordinates of the layer it
    // is used in) and u = C^ 1^ (C^ itself lies in the LDS slot)
    double c_e = 0.0, u_e = 0.0;
    double tb_e = 0.0;
    double* const ws = stg.ws + p * (long long)kSlotDoubles;   // this pair's matrix in global memory (At between its phases)

    for (int l = Lk - 1; l >= 0; --l) {
        const int n = (int)s.nl[l];
        const int N = n * P;
        const int nt = (N + 15) >> 4;
        n3 += (double)N * N * N;
        const cplx el = cmk(s.eps_re[l], s.eps_im[l]);
        const double Bl = s.BT[l];
        const long long item = p * (long long)b.Lmax + l;
        const double* gL = stg.L + item * stg.mat_stride;
        const double* gB = stg.B + item * stg.mat_stride;
        const double* gI = stg.Linv + item * stg.linv_stride;
        const bool in_e = t < N;
        SMRT_RT(RT_VEC);
        // ---- element t of the vectors of this layer (padding: d = sigma = 1, t = 0)
        const double d_e = in_e ? stg.d[item * stg.vec_stride + (in_e ? t : 0)] : 1.0;
        const double sg_e = in_e ? stg.sigma[item * stg.vec_stride + (in_e ? t : 0)] : 1.0;
        const double di_e = fast_rcp(d_e);
        const double nrs_e = -fast_rcp(sg_e);                      // -1 / sigma
        const double tt_e = in_e ? exp(-sg_e * s.thick[l]) : 0.0;
        const double st_e = sg_e * tt_e;                           // sigma t
        const double m3_e = in_e ? sg_e * (1.0 - tt_e * tt_e) : 1.0;

        if (l == Lk - 1) {
            // what the last layer sees below (rtsolver_utils.py:544-551,579-584,601-603; dort.py:429-441,446-452):
            // I_up = R I_dn + src  ->  C = (1 - R) / (1 + R) (diagonal: C^ = C), c = (C + 1) src
            double Rs = 0.0, src = 0.0;
            if (in_e) {
                const double rs = s.ri[l] * s.gsin[t >> 1];
                if (Lk < L) Rs = flat_R(el, cmk(s.eps_re[l + 1], s.eps_im[l + 1]), rs, t & 1);
                else if (b.sub_kind != SUB_NONE) {
                    const double q1 = b.sub_p1[gp], q2 = b.sub_p2[gp];
                    Rs = (b.sub_kind == SUB_FLAT) ? flat_R(el, cmk(q1, q2), rs, t & 1) : ((t & 1) ? q2 : q1);
                    const double Ts = b.sub_T[si];
                    if (Ts > 0.0) src = (1.0 - Rs) * (b.rayleigh_jeans ? Ts : planck_radiance(frequency, Ts));
                }
            }
            const double cd = in_e ? (1.0 - Rs) * fast_rcp(1.0 + Rs) : 0.0;
            c_e = in_e ? (cd + 1.0) * src : 0.0;
            u_e = cd * di_e;
            put(E0, cd, Ln);
            wave_sync();
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < TM; ++tj)
                    if (ti < nt && tj < nt) {
                        double z4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) z4[r] = (ti == tj && 4 * r + Ln.g == Ln.c) ? E0[16 * ti + Ln.c] : 0.0;
                        slot_store_tile(slot, ti, tj, z4, Ln);
                    }
            wave_sync();
        }

        const int LD = s.ints[7];   // (read per layer from LDS on purpose: keeps the tile offsets out of loop-invariant registers)
        const LaneOffsets lo = lane_offsets(LD, N, Ln);
        Mat X2;   // the matrix that is inverted (H^T + Sigma, M3^T, Y / S); the only one in registers meanwhile
        {
            SMRT_RT(RT_AT);
            Mat X1;
            zero(X1);
            // ---- B' (X1) and L+^T (tile by tile, requested up front: one wavefront per SIMD, nothing else hides the latency)
            double Lt[TM][TM][4], Li[TM][4];   // Lt[tk][ti], tk > ti: tiles of L+ and the diagonal-block inverses for A+ below
            {
                double LtT[TM][TM][4];   // (L+^T)[tk][tj], tk <= tj
                // every request of the two phases first ...
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TM; ++tj) load_tile_raw(X1.v[ti][tj], gB, LD, ti, tj, ti < nt && tj < nt, lo);
#pragma unroll
                for (int tj = 0; tj < TM; ++tj)
#pragma unroll
                    for (int tk = 0; tk <= tj; ++tk) load_tile_t_raw(LtT[tk][tj], gL, LD, tk, tj, tj < nt, lo);
#pragma unroll
                for (int ti = 0; ti < TM; ++ti) {
                    load_block(Li[ti], gI, ti < nt ? ti : 0, lo);
#pragma unroll
                    for (int tk = ti + 1; tk < TM; ++tk) load_tile_raw(Lt[tk][ti], gL, LD, tk, ti, tk < nt, lo);
                }
                // ... then the masks
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TM; ++tj) mask_tile(X1.v[ti][tj], ti, tj, ti < nt && tj < nt, lo);
#pragma unroll
                for (int tj = 0; tj < TM; ++tj)
#pragma unroll
                    for (int tk = 0; tk <= tj; ++tk) mask_tile(LtT[tk][tj], tk, tj, tj < nt, lo);
                // ---- At = A-^T = -Sigma^-1 B'^T L+^T, column by column, to this pair's matrix in global memory
                put(E2, nrs_e, Ln);
                wave_sync();
#pragma unroll
                for (int tj = 0; tj < TM; ++tj) {
                    if (tj < nt) {
                        double acc[TM][4];
#pragma unroll
                        for (int ti = 0; ti < TM; ++ti) acc[ti][0] = acc[ti][1] = acc[ti][2] = acc[ti][3] = 0.0;
#pragma unroll
                        for (int tk = 0; tk <= tj; ++tk) {
#pragma unroll
                            for (int ti = 0; ti < TM; ++ti)
                                if (ti < nt) tile_tn_acc(acc[ti], X1.v[tk][ti], LtT[tk][tj]);
                        }
#pragma unroll
                        for (int ti = 0; ti < TM; ++ti)
                            if (ti < nt) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[ti][r] *= E2[16 * ti + 4 * r + Ln.g];
                                slot_store_tile(ws, ti, tj, acc[ti], Ln);
                            }
                    }
                }
            }
            SMRT_RT(RT_APLUS);
            // ---- A+ = L+^-T B' (in place, X1): blocked back substitution with the diagonal-block inverses of the prep kernel
            {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                    for (int tk = ti + 1; tk < TM; ++tk) mask_tile(Lt[tk][ti], tk, ti, tk < nt, lo);
#pragma unroll
                for (int ti = TM - 1; ti >= 0; --ti) {
                    if (ti < nt) {
#pragma unroll
                        for (int tk = ti + 1; tk < TM; ++tk)
                            if (tk < nt) {
#pragma unroll
                                for (int tj = 0; tj < TM; ++tj)
                                    if (tj < nt) {
                                        double u[4];
                                        tile_tn(u, Lt[tk][ti], X1.v[tk][tj]);
#pragma unroll
                                        for (int r = 0; r < 4; ++r) X1.v[ti][tj][r] -= u[r];
                                    }
                            }
#pragma unroll
                        for (int tj = 0; tj < TM; ++tj)
                            if (tj < nt) {
                                double u[4];
                                tile_tn(u, Li[ti], X1.v[ti][tj]);
#pragma unroll
                                for (int r = 0; r < 4; ++r) X1.v[ti][tj][r] = u[r];
                            }
                    }
                }
            }
            SMRT_RT(RT_HT);
            // ---- H^T = A+^T (C^^T A+), column by column (C^ streamed from the slot); r = A+^T z, z = c^ - 2 B C^ 1^
            zero(X2);
#pragma unroll
            for (int tj = 0; tj < TM; ++tj) {
                if (tj < nt) {
                    double t1[TM][4];
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti) {
                        t1[ti][0] = t1[ti][1] = t1[ti][2] = t1[ti][3] = 0.0;
                        if (ti < nt) {
#pragma unroll
                            for (int tk = 0; tk < TM; ++tk)
                                if (tk < nt) {
                                    double ct[4];
                                    slot_load_tile(ct, slot, tk, ti, Ln);
                                    tile_tn_acc(t1[ti], ct, X1.v[tk][tj]);
                                }
                        }
                    }
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti)
                        if (ti < nt) {
                            double h[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int tk = 0; tk < TM; ++tk)
                                if (tk < nt) tile_tn_acc(h, X1.v[tk][ti], t1[tk]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) X2.v[ti][tj][r] = h[r];
                        }
                }
            }
            put(E0, c_e * di_e - 2.0 * Bl * u_e, Ln);
            wave_sync();
            matvec_t(X1, E0, E1, nt, Ln);                                  // r in E1 (natural order)
        }

        // ---- three inversions of X2, one copy of the elimination code
        double q_e = 0.0;
        bool last = false;
        double cb_e = 0.0, cd_e = 0.0, t2_e = 0.0, it2_e = 0.0, extra_e = 0.0;   // interface coefficients (stage 2 -> post)
        double t1s_e = 0.0, Rair_e = 0.0, Idn = 0.0;                            // surface (l == 0)
        int Nu = 0, nc = 0;
#if !defined(SMRT_HOST_EMU)
#pragma nounroll
#endif
        for (int stage = 0; stage < 3; ++stage) {
            if (stage == 0) {
                SMRT_RT(RT_ST1);
                put(E2, sg_e, Ln);
                wave_sync();
                scale_add_diag(X2, nullptr, nullptr, E2, 1.0, nt, Ln);      // H^T + Sigma
            } else if (stage == 1) {
                SMRT_RT(RT_ST1);
                matvec_t(X2, E1, E0, nt, Ln);                               // q = P r
                q_e = E0[t];
                put(E2, st_e, Ln); put(E3, m3_e, Ln);
                wave_sync();
                scale_add_diag(X2, E2, E2, E3, 2.0, nt, Ln);                // M3^T
            } else {
                SMRT_RT(RT_T2);
                put(E0, st_e * q_e, Ln);
                wave_sync();
                matvec_t(X2, E0, E1, nt, Ln);                               // y = M3^-1 (Sigma t q), natural order in E1
                put(E2, nrs_e, Ln);
                wave_sync();
                scale_add_diag(X2, nullptr, nullptr, E2, 2.0, nt, Ln);      // Theta^T = 2 M3^-T - Sigma^-1
                // T2 = Theta At, column by column from global memory into the slot; A- y = At^T y on the way
#pragma unroll
                for (int tj = 0; tj < TM; ++tj) {
                    if (tj < nt) {
                        double ac[TM][4];
                        double amy = 0.0;
#pragma unroll
                        for (int tk = 0; tk < TM; ++tk) {
                            ac[tk][0] = ac[tk][1] = ac[tk][2] = ac[tk][3] = 0.0;
                            if (tk < nt) {
                                slot_load_tile(ac[tk], ws, tk, tj, Ln);
#pragma unroll
                                for (int r = 0; r < 4; ++r) amy += ac[tk][r] * E1[16 * tk + 4 * r + Ln.g];
                            }
                        }
                        amy += shfl_xor(amy, 16);
                        amy += shfl_xor(amy, 32);
                        if (Ln.g == 0) E0[16 * tj + Ln.c] = amy;
#pragma unroll
                        for (int ti = 0; ti < TM; ++ti)
                            if (ti < nt) {
                                double h[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                                for (int tk = 0; tk < TM; ++tk)
                                    if (tk < nt) tile_tn_acc(h, X2.v[tk][ti], ac[tk]);
                                slot_store_tile(slot, ti, tj, h, Ln);
                            }
                    }
                }
                wave_sync();
                const double amy_e = E0[t];
                put(E2, di_e, Ln); put(E3, d_e, Ln);
                wave_sync();
                SMRT_RT(RT_CP);
                // C^' = At^T T2 with At in registers (X1), T2 column by column from the slot; C^' 1^ on the way; the
                // finished column goes back to the slot as C' = D C^' D^-1
                {
                    Mat X1;
                    zero(X1);
                    slot_load(X1, ws, nt, Ln);
                    double urow[TM][4];
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti) urow[ti][0] = urow[ti][1] = urow[ti][2] = urow[ti][3] = 0.0;
#pragma unroll
                    for (int tj = 0; tj < TM; ++tj) {
                        if (tj < nt) {
                            double tc[TM][4];
#pragma unroll
                            for (int tk = 0; tk < TM; ++tk) {
                                tc[tk][0] = tc[tk][1] = tc[tk][2] = tc[tk][3] = 0.0;
                                if (tk < nt) slot_load_tile(tc[tk], slot, tk, tj, Ln);
                            }
                            const double dic = E2[16 * tj + Ln.c];
#pragma unroll
                            for (int ti = 0; ti < TM; ++ti)
                                if (ti < nt) {
                                    double h[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                                    for (int tk = 0; tk < TM; ++tk)
                                        if (tk < nt) tile_tn_acc(h, X1.v[tk][ti], tc[tk]);
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        urow[ti][r] += h[r] * dic;
                                        h[r] *= E3[16 * ti + 4 * r + Ln.g] * dic;
                                    }
                                    slot_store_tile(slot, ti, tj, h, Ln);
                                }
                        }
                    }
                    wave_sync();
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti)
                        if (ti < nt) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const double a = group_sum<16>(urow[ti][r]);
                                if (Ln.c == 0) E0[16 * ti + 4 * r + Ln.g] = a;
                            }
                        }
                    wave_sync();
                }
                c_e = d_e * (2.0 * Bl * E0[t] - 2.0 * amy_e);               // c' (physical coordinates)
                wave_sync();
                SMRT_RT(RT_IFACE);
                zero(X2);
                if (l == 0) {
                    // surface (dort.py:391-395,484): I_dn = r2 I_up + t2 I_sky just below it;
                    // S I_up = c' + (I - C') t2 I_sky,  S = (1 - r2) + C' (1 + r2);  I0 = R_air I_sky + t1 I_up
                    last = true;
                    const bool atm = (b.atm_down != nullptr);
                    Idn = atm ? (b.rayleigh_jeans ? b.atm_down[fi] : planck_radiance(frequency, b.atm_down[fi])) : 0.0;
                    double Tair = 0.0, r2s = 0.0;
                    const cplx one = cmk(1.0, 0.0);
                    if (in_e) { r2s = flat_R(el, one, s.ri[0] * s.gsin[t >> 1], t & 1); t1s_e = 1.0 - r2s; }
                    if (t < n_air * P) {
                        double Rv, Rh;
                        fresnel_RvRh(one, el, s.outmu[t >> 1], &Rv, &Rh);
                        Rair_e = (t & 1) ? Rh : Rv; Tair = 1.0 - Rair_e;
                    }
                    put(E0, Tair * Idn, Ln);                                // t2 I_sky (0 beyond the air streams)
                    wave_sync();
                    slot_load(X2, slot, nt, Ln);                            // C'
                    matvec(X2, E0, E1, nt, Ln);
                    const double rhs = in_e ? c_e + E0[t] - E1[t] : 0.0;
                    wave_sync();
                    put(E4, rhs, Ln);
                    put(E2, in_e ? 1.0 + r2s : 0.0, Ln); put(E3, in_e ? 1.0 - r2s : 1.0, Ln);
                    wave_sync();
                    scale_add_diag(X2, nullptr, E2, E3, 1.0, nt, Ln);
                } else {
                    // interface with the layer above: diagonal coefficients per element (streams paired by index)
                    Nu = (int)s.nl[l - 1] * P;
                    nc = (N < Nu) ? N : Nu;
                    const cplx eup = cmk(s.eps_re[l - 1], s.eps_im[l - 1]);
                    double r1 = 1.0, t2 = 0.0, r2 = 0.0, t1 = 0.0;
                    extra_e = 0.0;
                    if (in_e) {   // from this layer upwards
                        r2 = flat_R(el, eup, s.ri[l] * s.gsin[t >> 1], t & 1);
                        t1 = (t < nc) ? 1.0 - r2 : 0.0;
                    }
                    if (t < Nu) {  // from the upper layer downwards
                        const double rb = flat_R(eup, el, s.ri[l - 1] * s.gsin[t >> 1], t & 1);
                        if (t < nc) { r1 = rb; t2 = 1.0 - rb; }
                        else extra_e = (1.0 - rb) * fast_rcp(1.0 + rb);   // a stream that does not exist below: I_up = R I_dn
                    }
                    const double tt2 = t1 * t2;
                    const double ca = 0.5 * (tt2 + (1.0 + r1) * (1.0 - r2));
                    const double cc = 0.5 * (tt2 - (1.0 - r1) * (1.0 - r2));
                    cb_e = 0.5 * (tt2 - (1.0 + r1) * (1.0 + r2));
                    cd_e = 0.5 * (tt2 + (1.0 - r1) * (1.0 + r2));
                    t2_e = (t < nc) ? t2 : 0.0;
                    it2_e = (t < nc) ? fast_rcp(t2) : 0.0;
                    // X2 = Y = a - b C';  Nn = c - d C' goes back to the slot transposed (it waits there while Y is inverted)
                    put(E0, in_e ? -cb_e : 0.0, Ln); put(E1, in_e ? ca : 1.0, Ln);
                    put(E2, in_e ? -cd_e : 0.0, Ln); put(E3, in_e ? cc : 0.0, Ln);
                    wave_sync();
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                        for (int tj = ti; tj < TM; ++tj)
                            if (tj < nt) {
                                double a[4], bq[4], na[4], nb[4], ta[4], tb4[4];
                                slot_load_tile(a, slot, ti, tj, Ln);
                                slot_load_tile(bq, slot, tj, ti, Ln);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int ra = 16 * ti + 4 * r + Ln.g, rb_ = 16 * tj + 4 * r + Ln.g;
                                    const bool dg = (4 * r + Ln.g == Ln.c);
                                    X2.v[ti][tj][r] = a[r] * E0[ra] + ((ti == tj && dg) ? E1[ra] : 0.0);
                                    na[r] = a[r] * E2[ra] + ((ti == tj && dg) ? E3[ra] : 0.0);
                                    if (ti != tj) {
                                        X2.v[tj][ti][r] = bq[r] * E0[rb_];
                                        nb[r] = bq[r] * E2[rb_];
                                    }
                                }
                                tile_transpose(ta, na, Ln);
                                if (ti != tj) {
                                    tile_transpose(tb4, nb, Ln);
                                    slot_store_tile(slot, ti, tj, tb4, Ln);
                                }
                                slot_store_tile(slot, tj, ti, ta, Ln);
                            }
                }
                wave_sync();
            }
            SMRT_RT(RT_INV);
            invert(X2, nt, Ln);
        }
        if (last) {
            SMRT_RT(RT_SURF);
            matvec(X2, E4, E0, nt, Ln);    // I_up just below the surface
            if (t < n_air * P) {
                const bool atm = (b.atm_down != nullptr);
                double I0 = Rair_e * Idn + t1s_e * E0[t];
                if (atm) I0 = (b.rayleigh_jeans ? b.atm_up[fi] : planck_radiance(frequency, b.atm_up[fi])) + b.atm_trans[fi] * I0;
                tb_e = b.rayleigh_jeans ? I0 : planck_inverse(frequency, I0);
            }
            break;
        }
        SMRT_RT(RT_POST);
        // ---- Z = Nn Y^-1, column by column: Nn^T moves from the slot into registers (X1) while Y^-1 takes its place there;
        //      C_u = -t2^-1 Z t2 on the common streams, (1 - R) / (1 + R) on the diagonal of the upper layer's extra streams;
        //      c_u = (d c' - Z b c') / t2;  the column goes back to the slot in the hats of the layer above, C^ = D^-1 C D
        const int ntu = (Nu + 15) >> 4;
        const long long item_u = item - 1;
        const bool in_u = t < Nu;
        const double du_e = in_u ? stg.d[item_u * stg.vec_stride + (in_u ? t : 0)] : 1.0;
        const double dui_e = fast_rcp(du_e);
        wave_sync();
        put(E0, in_e ? cb_e * c_e : 0.0, Ln);
        put(E2, -it2_e * dui_e, Ln); put(E3, t2_e * du_e, Ln); put(E4, extra_e, Ln); put(E1, dui_e, Ln);
        wave_sync();
        {
            Mat X1;
            zero(X1);
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < TM; ++tj)
                    if (ti < nt && tj < nt) {
                        slot_load_tile(X1.v[ti][tj], slot, ti, tj, Ln);
                        slot_store_tile(slot, ti, tj, X2.v[ti][tj], Ln);
                    }
            wave_sync();
            double zb[TM][4], urow[TM][4];
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
                zb[ti][0] = zb[ti][1] = zb[ti][2] = zb[ti][3] = 0.0;
                urow[ti][0] = urow[ti][1] = urow[ti][2] = urow[ti][3] = 0.0;
            }
#pragma unroll
            for (int tj = 0; tj < TM; ++tj) {
                if (tj < nt || tj < ntu) {
                    double yc[TM][4];
#pragma unroll
                    for (int tk = 0; tk < TM; ++tk) {
                        yc[tk][0] = yc[tk][1] = yc[tk][2] = yc[tk][3] = 0.0;
                        if (tk < nt && tj < nt) slot_load_tile(yc[tk], slot, tk, tj, Ln);
                    }
                    const int col = 16 * tj + Ln.c;
                    const double wb = E0[col], cf = E3[col], wu = E1[col];
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti)
                        if (ti < nt || ti < ntu) {
                            double h[4] = {0.0, 0.0, 0.0, 0.0};
                            if (ti < nt && tj < nt) {
#pragma unroll
                                for (int tk = 0; tk < TM; ++tk)
                                    if (tk < nt) tile_tn_acc(h, X1.v[tk][ti], yc[tk]);
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 16 * ti + 4 * r + Ln.g;
                                zb[ti][r] += h[r] * wb;
                                double x = (row < nc && col < nc) ? h[r] * E2[row] * cf : 0.0;
                                if (row == col) x += E4[row];
                                urow[ti][r] += x * wu;
                                h[r] = x;
                            }
                            if (ti < ntu && tj < ntu) slot_store_tile(slot, ti, tj, h, Ln);
                        }
                }
            }
            wave_sync();
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double a = group_sum<16>(zb[ti][r]);
                    const double u2 = group_sum<16>(urow[ti][r]);
                    if (Ln.c == 0) { E0[16 * ti + 4 * r + Ln.g] = a; E1[16 * ti + 4 * r + Ln.g] = u2; }
                }
            wave_sync();
        }
        c_e = (t < nc) ? (cd_e * c_e - E0[t]) * it2_e : 0.0;
        u_e = in_u ? E1[t] : 0.0;
        wave_sync();
    }

    put(E0, tb_e, Ln);
    block_sync();
    bool bad = false;
    for (int i = t; i < n_air * P; i += NT) bad = bad || !(fabs(E0[i]) < 1e300);   // NaN / inf: a vanishing pivot
    if (bad) lds_max(&s.ints[0], ST_SINGULAR);
    block_sync();
    if (s.ints[0] != ST_OK) { fail_pair<NT>(b, p, s.ints[0], out_stride); return; }
    for (int idx = t; idx < P * b.n_theta; idx += NT) {
        const int pol = idx / b.n_theta, it = idx % b.n_theta;
        const double um = cos(b.theta[it]);
        // (rtsolver_utils.py:191-198, see dort_pair_passive)
        double x0, x1, y0, y1;
        const double top = 0.5 * (E0[0] + E0[1]);
        if (um > s.outmu[0] || n_air == 1) { x0 = 1.0; y0 = top; x1 = s.outmu[0]; y1 = E0[pol]; }
        else {
            int k = 0;
            while (k < n_air - 2 && um < s.outmu[k + 1]) ++k;
            x0 = s.outmu[k]; y0 = E0[2 * k + pol]; x1 = s.outmu[k + 1]; y1 = E0[2 * (k + 1) + pol];
        }
        b.out[p * out_stride + idx] = y0 + (y1 - y0) * ((um - x0) / (x1 - x0));
    }
    if (t == 0) { b.status[p] = ST_OK; if (b.n3_out) b.n3_out[p] = n3; }
#ifdef SMRT_REG_TIMING
    SMRT_RT(RT_SETUP);
    if (t == 0 && b.stage_out) for (int k = 0; k < 16; ++k) b.stage_out[p * 16 + k] = (k < RT_COUNT) ? rt_acc[k] : 0.0;
#endif
}

}  // namespace smrt
