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
struct Tile { double r[4]; };

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
// Step k: row k reaches every lane through the matrix core (a selector matrix picks it out of register k >> 2), column
// k through DPP row broadcasts; column k is replaced by the unit vector e_k before the rank-one update, which makes the
// in-place update uniform over the tile (Gauss-Jordan inversion in place).
template <int K>
SMRT_DEV void inv16_step(double (&d)[4], const LaneId& L) {
    constexpr int r0 = K >> 2, g0 = K & 3;
    double rk4[4] = {0.0, 0.0, 0.0, 0.0};
    mfma_f64_16x16x4((L.g == g0) ? 1.0 : 0.0, d[r0], rk4);   // every register / lane group: D[K][c]
    double rk = rk4[0];
    const double piv = row_bcast16<K>(rk);
    const double pinv = fast_rcp(piv);
    const bool colk = (L.c == K);
    rk = colk ? 1.0 : rk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double f = row_bcast16<K>(d[r]);           // D[4 r + g][K]
        double m = f * pinv;
        if (r == r0) m = (L.g == g0) ? (1.0 - pinv) : m;  // row K itself: new = old - (1 - pinv) old = old pinv
        const double old = colk ? ((r == r0 && L.g == g0) ? 1.0 : 0.0) : d[r];
        d[r] = old - m * rk;
    }
}
SMRT_DEV void inv16(double (&d)[4], const LaneId& L) {
    inv16_step<0>(d, L); inv16_step<1>(d, L); inv16_step<2>(d, L); inv16_step<3>(d, L);
    inv16_step<4>(d, L); inv16_step<5>(d, L); inv16_step<6>(d, L); inv16_step<7>(d, L);
    inv16_step<8>(d, L); inv16_step<9>(d, L); inv16_step<10>(d, L); inv16_step<11>(d, L);
    inv16_step<12>(d, L); inv16_step<13>(d, L); inv16_step<14>(d, L); inv16_step<15>(d, L);
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
// (one copy of the step code for every block)
SMRT_DEV void invert(Mat& M, int nt, const LaneId& L) {
#if !defined(SMRT_HOST_EMU)
#pragma nounroll
#endif
    for (int step = 0; step < nt; ++step) {
        double D[4], DT[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = M.v[0][0][r];
        inv16(D, L);
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
                    }
            }
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
    }
}

// ---- vectors live in LDS in natural order (one element per lane for elementwise work); a matrix-vector product reads
// its operand in "row form" (element 16 tj + c) or "column form" (element 16 ti + 4 r + g) --------------------------
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
// X[i][j] <- rowf[i] X[i][j] colf[j] + (i == j) diag[i]     (all TM x TM tiles; null pointers: factor 1 / nothing)
SMRT_DEV void scale_add_diag(Mat& X, const double* rowf, const double* colf, const double* diag, double factor, const LaneId& L) {
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TM; ++tj) {
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

// tile (ti, tj) of a column-major matrix p (element (r, c) at p[c LD + r]), zero outside N x N
SMRT_DEV void load_tile(double (&t)[4], const double* p, int LD, int N, int ti, int tj, const LaneId& L) {
    const int col = 16 * tj + L.c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * r + L.g;
        const bool in = row < N && col < N;
        t[r] = in ? p[(in ? col : 0) * LD + (in ? row : 0)] : 0.0;
    }
}
// tile (ti, tj) of the TRANSPOSE of p
SMRT_DEV void load_tile_t(double (&t)[4], const double* p, int LD, int N, int ti, int tj, const LaneId& L) {
    const int col = 16 * tj + L.c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * r + L.g;
        const bool in = row < N && col < N;
        t[r] = in ? p[(in ? row : 0) * LD + (in ? col : 0)] : 0.0;
    }
}

constexpr int kRegVectors = 15;   // 64-double LDS vectors of the register-resident finish kernel
}  // namespace rg
// LDS doubles of the register-resident finish kernel
SMRT_HD int finish_reg_lds_doubles(int n_max_stream, int Lmax, int ntheta) {
    return make_plan(n_max_stream, 2, Lmax, ntheta, 9, 0, 0, 2, 0).total + rg::kRegVectors * 64;
}

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
    const LdsPlan plan = make_plan(b.n_max_stream, P, b.Lmax, b.n_theta, 9, 0, 0, 2, 0);
    Lds s = carve(lds_base, lds_base, plan);
    const int nmax = b.n_max_stream;
    const int out_stride = P * b.n_theta;
    const int LD = plan.LD;

    const long long gp = global_pair(b, p);
    const int fi = (int)(gp / b.S), si = (int)(gp % b.S);
    const double frequency = b.frequency[fi];
    int L = b.n_layers[si];
    const double* thickness = b.thickness + (long long)si * b.Lmax;
    const double* fracvol = b.frac_volume + (long long)si * b.Lmax;
    const double* temperature = b.temperature + (long long)si * b.Lmax;
    const double* mp1 = b.p1 + (long long)si * b.Lmax;
    const double* mp2 = b.p2 + (long long)si * b.Lmax;

    if (t < 8) s.ints[t] = 0;
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

    // LDS vectors (64 doubles each whatever n_max_stream is, natural order, lane e = element e), behind the tables of the plan
    double* const vec = lds_base + plan.total;
    double* const v_d = vec;              // row scaling d of the layer (E = D A)
    double* const v_di = vec + 64;        // 1 / d
    double* const v_sg = vec + 2 * 64;    // singular values (padding: 1)
    double* const v_st = vec + 3 * 64;    // sigma t (padding 0)
    double* const v_c = vec + 4 * 64;     // source c of the relation, physical coordinates
    double* const v_a = vec + 5 * 64;     // per-layer temporaries
    double* const v_b = vec + 6 * 64;
    double* const v_e = vec + 7 * 64;
    double* const v_f = vec + 8 * 64;
    double* const v_g = vec + 9 * 64;
    double* const v_h = vec + 10 * 64;
    double* const v_i = vec + 11 * 64;
    double* const v_j = vec + 12 * 64;
    double* const v_k = vec + 13 * 64;
    double* const v_tb = vec + 14 * 64;   // brightness temperatures at the air streams

    Mat C, X1, X2;      // C: the relation; X1, X2: work matrices
    zero(C); zero(X1); zero(X2);
    double n3 = 0.0;

    for (int l = Lk - 1; l >= 0; --l) {
        const int n = (int)s.nl[l];
        const int N = n * P;
        const int nt = (N + 15) >> 4;
        n3 += (double)N * N * N;
        const cplx el = cmk(s.eps_re[l], s.eps_im[l]);
        const double Bl = s.BT[l];
        const int nu = (l > 0) ? (int)s.nl[l - 1] : n_air;
        const int Nu = nu * P;
        const int nc = (N < Nu) ? N : Nu;
        const long long item = p * (long long)b.Lmax + l;
        const double* gL = stg.L + item * stg.mat_stride;
        const double* gB = stg.B + item * stg.mat_stride;
        const double* gI = stg.Linv + item * 1024;
        const double thick_l = s.thick[l];

        // ---- element e of the vectors of this layer
        {
            const int e = t;
            const bool in = e < N;
            const double dd = in ? stg.d[item * stg.vec_stride + (in ? e : 0)] : 1.0;
            const double sg = in ? stg.sigma[item * stg.vec_stride + (in ? e : 0)] : 1.0;
            const double tt = in ? exp(-sg * thick_l) : 0.0;
            v_d[e] = dd; v_di[e] = 1.0 / dd; v_sg[e] = sg; v_st[e] = sg * tt;
            v_a[e] = in ? sg * (1.0 - tt * tt) : 1.0;    // diagonal of M3
            v_b[e] = in ? -1.0 / sg : -1.0;              // -1 / sigma
        }
        if (l == Lk - 1) {
            // what the last layer sees below (rtsolver_utils.py:544-551,579-584,601-603; dort.py:429-441,446-452):
            // I_up = R I_dn + src  ->  C = (1 - R) / (1 + R), c = (C + 1) src
            const int r = t;
            double Rs = 0.0, src = 0.0;
            if (r < N) {
                const double rs = s.ri[l] * s.gsin[r >> 1];
                const double mu_r = sqrt(1.0 - rs * rs);
                if (Lk < L) {
                    double Rv, Rh, Tv, Th;
                    interface_RT(frequency, el, cmk(s.eps_re[l + 1], s.eps_im[l + 1]), mu_r,
                                 cmk(s.slab_re[l + 1], s.slab_im[l + 1]), s.slab_th[l + 1], &Rv, &Rh, &Tv, &Th);
                    Rs = (r & 1) ? Rh : Rv;
                } else if (b.sub_kind != SUB_NONE) {
                    const double q1 = b.sub_p1[gp], q2 = b.sub_p2[gp];
                    if (b.sub_kind == SUB_FLAT) {
                        double Rv, Rh;
                        fresnel_RvRh(el, cmk(q1, q2), mu_r, &Rv, &Rh);
                        Rs = (r & 1) ? Rh : Rv;
                    } else Rs = (r & 1) ? q2 : q1;
                    const double Ts = b.sub_T[si];
                    if (Ts > 0.0) src = (1.0 - Rs) * (b.rayleigh_jeans ? Ts : planck_radiance(frequency, Ts));
                }
            }
            const double cd = (1.0 - Rs) / (1.0 + Rs);
            v_k[r] = (r < N) ? cd : 0.0;
            v_c[r] = (r < N) ? (cd + 1.0) * src : 0.0;
            wave_sync();
            zero(C);
            scale_add_diag(C, nullptr, nullptr, v_k, 1.0, Ln);
        }
        wave_sync();

        // ---- A+ = L+^-T B' (in X1), blocked back substitution with the diagonal-block inverses of the prep kernel
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TM; ++tj)
                if (ti < nt && tj < nt) load_tile(X1.v[ti][tj], gB, LD, N, ti, tj, Ln);
#pragma unroll
        for (int ti = TM - 1; ti >= 0; --ti) {
            if (ti < nt) {
#pragma unroll
                for (int tk = ti + 1; tk < TM; ++tk)
                    if (tk < nt) {
                        double Lt[4];
                        load_tile(Lt, gL, LD, N, tk, ti, Ln);
#pragma unroll
                        for (int tj = 0; tj < TM; ++tj)
                            if (tj < nt) {
                                double u[4];
                                tile_tn(u, Lt, X1.v[tk][tj]);
#pragma unroll
                                for (int r = 0; r < 4; ++r) X1.v[ti][tj][r] -= u[r];
                            }
                    }
                double Li[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) Li[r] = gI[ti * 256 + Ln.c * 16 + 4 * r + Ln.g];
#pragma unroll
                for (int tj = 0; tj < TM; ++tj)
                    if (tj < nt) {
                        double u[4];
                        tile_tn(u, Li, X1.v[ti][tj]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) X1.v[ti][tj][r] = u[r];
                    }
            }
        }
        // ---- C^ = D^-1 C D;  z = c^ - 2 B C^ 1^  (column form, v_e);  T1 = C^^T A+ (X2);  H^T = A+^T T1 (C)
        scale_add_diag(C, v_di, v_d, nullptr, 1.0, Ln);
        matvec(C, v_di, v_e, nt, Ln);                                 // u = C^ 1^
        v_e[t] = v_c[t] * v_di[t] - 2.0 * Bl * v_e[t];                // z
        wave_sync();
        gemm_tn(X2, C, X1, nt);
        matvec_t(X1, v_e, v_f, nt, Ln);                               // r = A+^T z   (row form, v_f)
        gemm_tn(C, X1, X2, nt);
        scale_add_diag(C, nullptr, nullptr, v_sg, 1.0, Ln);           // H^T + Sigma
        invert(C, nt, Ln);                                            // P^T
        matvec_t(C, v_f, v_e, nt, Ln);                                // q = P r
        // ---- M3^T = Sigma (1 - t^2) + 2 (Sigma t) P^T (t Sigma), inverse, y = M3^-1 (Sigma t q), Theta^T = 2 M3^-T - Sigma^-1
        scale_add_diag(C, v_st, v_st, v_a, 2.0, Ln);
        invert(C, nt, Ln);
        v_f[t] = v_st[t] * v_e[t];
        wave_sync();
        matvec_t(C, v_f, v_g, nt, Ln);                                // y (v_g)
        scale_add_diag(C, nullptr, nullptr, v_b, 2.0, Ln);            // Theta^T
        // ---- At = A-^T = -Sigma^-1 B'^T L+^T (X1): B' again (X2), L+^T tile by tile
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TM; ++tj)
                if (ti < nt && tj < nt) load_tile(X2.v[ti][tj], gB, LD, N, ti, tj, Ln);
#pragma unroll
        for (int tj = 0; tj < TM; ++tj) {
            if (tj < nt) {
                double acc[TM][4];
#pragma unroll
                for (int ti = 0; ti < TM; ++ti) acc[ti][0] = acc[ti][1] = acc[ti][2] = acc[ti][3] = 0.0;
#pragma unroll
                for (int tk = 0; tk <= tj; ++tk) {
                    double Lt[4];
                    load_tile_t(Lt, gL, LD, N, tk, tj, Ln);           // (L+^T)[tk][tj]
#pragma unroll
                    for (int ti = 0; ti < TM; ++ti)
                        if (ti < nt) tile_tn_acc(acc[ti], X2.v[tk][ti], Lt);
                }
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X1.v[ti][tj][r] = (ti < nt) ? acc[ti][r] : 0.0;
            }
        }
        scale_add_diag(X1, v_b, nullptr, nullptr, 1.0, Ln);           // rows times -1 / sigma
        // ---- C^' = At^T (Theta At): T2 = (Theta^T)^T At (X2), C^' = At^T T2 (C)
        gemm_tn(X2, C, X1, nt);
        gemm_tn(C, X1, X2, nt);
        // c^' = 2 B C^' 1^ - 2 A- y
        matvec(C, v_di, v_e, nt, Ln);                                 // C^' 1^  (v_e)
        matvec_t(X1, v_g, v_f, nt, Ln);                               // A- y = At^T y  (v_f)
        v_c[t] = v_d[t] * (2.0 * Bl * v_e[t] - 2.0 * v_f[t]);         // c' (physical)
        wave_sync();
        scale_add_diag(C, v_d, v_di, nullptr, 1.0, Ln);               // C' = D C^' D^-1

        if (l == 0) break;
        // ---- interface with the layer above: diagonal coefficients per element e (streams paired by index)
        {
            const int e = t;
            double r1 = 1.0, t2 = 0.0, r2 = 0.0, t1 = 0.0, extra = 0.0;
            const cplx eup = cmk(s.eps_re[l - 1], s.eps_im[l - 1]);
            const cplx slab = cmk(s.slab_re[l], s.slab_im[l]);
            if (e < N) {   // from this layer upwards
                const double rs = s.ri[l] * s.gsin[e >> 1];
                double Rv, Rh, Tv, Th;
                interface_RT(frequency, el, eup, sqrt(1.0 - rs * rs), slab, s.slab_th[l], &Rv, &Rh, &Tv, &Th);
                r2 = (e & 1) ? Rh : Rv;
                t1 = (e < nc) ? ((e & 1) ? Th : Tv) : 0.0;
            }
            if (e < Nu) {  // from the upper layer downwards
                const double rs = s.ri[l - 1] * s.gsin[e >> 1];
                double Rv, Rh, Tv, Th;
                interface_RT(frequency, eup, el, sqrt(1.0 - rs * rs), slab, s.slab_th[l], &Rv, &Rh, &Tv, &Th);
                const double rb = (e & 1) ? Rh : Rv;
                if (e < nc) { r1 = rb; t2 = (e & 1) ? Th : Tv; }
                else extra = (1.0 - rb) / (1.0 + rb);   // a stream that does not exist below: I_up = R I_dn
            }
            const bool in = e < N;
            const double tt2 = t1 * t2;
            const double ca = 0.5 * (tt2 + (1.0 + r1) * (1.0 - r2));
            const double cb = 0.5 * (tt2 - (1.0 + r1) * (1.0 + r2));
            const double cc = 0.5 * (tt2 - (1.0 - r1) * (1.0 - r2));
            const double cd = 0.5 * (tt2 + (1.0 - r1) * (1.0 + r2));
            v_a[e] = in ? ca : 1.0;          // Y = a - b C'
            v_b[e] = in ? -cb : 0.0;
            v_e[e] = in ? cc : 0.0;          // Nn = c - d C'
            v_f[e] = in ? -cd : 0.0;
            v_g[e] = (e < nc) ? t2 : 0.0;
            v_h[e] = (e < nc) ? -1.0 / t2 : 0.0;
            v_i[e] = extra;
            v_j[e] = in ? cb * v_c[e] : 0.0;     // b c'
            v_k[e] = (e < nc) ? cd * v_c[e] / t2 : 0.0;
        }
        wave_sync();
        // X1 = Y, X2 = Nn
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TM; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) { X1.v[ti][tj][r] = C.v[ti][tj][r]; X2.v[ti][tj][r] = C.v[ti][tj][r]; }
        scale_add_diag(X1, v_b, nullptr, v_a, 1.0, Ln);
        scale_add_diag(X2, v_f, nullptr, v_e, 1.0, Ln);
        // C <- Nn^T (tile transposes)
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TM; ++tj)
                if (ti < nt && tj < nt) tile_transpose(C.v[tj][ti], X2.v[ti][tj], Ln);
        invert(X1, nt, Ln);                                           // Y^-1
        gemm_tn(X2, C, X1, nt);                                       // Z = Nn Y^-1
        matvec(X2, v_j, v_e, nt, Ln);                                 // Z (b c')
        // C_u = -t2^-1 Z t2 on the common streams, (1 - R) / (1 + R) on the diagonal of the upper layer's extra streams
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TM; ++tj) {
                const int col = 16 * tj + Ln.c;
                const double cf = v_g[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + Ln.g;
                    const bool in = (ti < nt && tj < nt) && row < nc && col < nc;
                    double x = in ? X2.v[ti][tj][r] * v_h[row] * cf : 0.0;
                    if (row == col) x += v_i[row];
                    C.v[ti][tj][r] = x;
                }
            }
        // c_u = (d c' - Z b c') / t2
        v_c[t] = (t < nc) ? v_k[t] + v_e[t] * v_h[t] : 0.0;
        wave_sync();
    }

    // ---- surface (dort.py:391-395,484): I_dn = r2 I_up + t2 I_sky below it, S I_up = c' + (I - C') t2 I_sky with
    //      S = (1 - r2) + C' (1 + r2); emerging I0 = R_air I_sky + t1 I_up on the air streams
    {
        const int N0 = (int)s.nl[0] * P;
        const int nt = (N0 + 15) >> 4;
        const bool atm = (b.atm_down != nullptr);
        const double Idn = atm ? (b.rayleigh_jeans ? b.atm_down[fi] : planck_radiance(frequency, b.atm_down[fi])) : 0.0;
        const double Iup_atm = atm ? (b.rayleigh_jeans ? b.atm_up[fi] : planck_radiance(frequency, b.atm_up[fi])) : 0.0;
        const double trans = atm ? b.atm_trans[fi] : 1.0;
        const cplx e0 = cmk(s.eps_re[0], s.eps_im[0]);
        const cplx slab0 = cmk(s.slab_re[0], s.slab_im[0]);
        const int e = t;
        double r2 = 0.0, t1 = 0.0, Rair = 0.0, Tair = 0.0;
        if (e < N0) {
            const double rs = s.ri[0] * s.gsin[e >> 1];
            double Rv, Rh, Tv, Th;
            interface_RT(frequency, e0, cmk(1.0, 0.0), sqrt(1.0 - rs * rs), slab0, s.slab_th[0], &Rv, &Rh, &Tv, &Th);
            r2 = (e & 1) ? Rh : Rv; t1 = (e & 1) ? Th : Tv;
        }
        if (e < n_air * P) {
            double Rv, Rh, Tv, Th;
            interface_RT(frequency, cmk(1.0, 0.0), e0, s.outmu[e >> 1], slab0, s.slab_th[0], &Rv, &Rh, &Tv, &Th);
            Rair = (e & 1) ? Rh : Rv; Tair = (e & 1) ? Th : Tv;
        }
        v_a[e] = (e < N0) ? 1.0 - r2 : 1.0;
        v_b[e] = (e < N0) ? 1.0 + r2 : 0.0;
        v_e[e] = Tair * Idn;            // t2 I_sky (0 beyond the air streams)
        wave_sync();
        matvec(C, v_e, v_f, nt, Ln);    // C' (t2 I_sky)
        v_g[e] = (e < N0) ? v_c[e] + v_e[e] - v_f[e] : 0.0;
        wave_sync();
        scale_add_diag(C, nullptr, v_b, v_a, 1.0, Ln);
        invert(C, nt, Ln);
        matvec(C, v_g, v_f, nt, Ln);    // I_up just below the surface
        if (e < n_air * P) {
            double I0 = Rair * Idn + t1 * v_f[e];
            if (atm) I0 = Iup_atm + trans * I0;
            v_tb[e] = b.rayleigh_jeans ? I0 : planck_inverse(frequency, I0);
        }
    }
    block_sync();
    bool bad = false;
    for (int i = t; i < n_air * P; i += NT) bad = bad || !(fabs(v_tb[i]) < 1e300);   // NaN / inf: a vanishing pivot
    if (bad) lds_max(&s.ints[0], ST_SINGULAR);
    block_sync();
    if (s.ints[0] != ST_OK) { fail_pair<NT>(b, p, s.ints[0], out_stride); return; }
    for (int idx = t; idx < P * b.n_theta; idx += NT) {
        const int pol = idx / b.n_theta, it = idx % b.n_theta;
        const double um = cos(b.theta[it]);
        // (rtsolver_utils.py:191-198, see dort_pair_passive)
        double x0, x1, y0, y1;
        const double top = 0.5 * (v_tb[0] + v_tb[1]);
        if (um > s.outmu[0] || n_air == 1) { x0 = 1.0; y0 = top; x1 = s.outmu[0]; y1 = v_tb[pol]; }
        else {
            int k = 0;
            while (k < n_air - 2 && um < s.outmu[k + 1]) ++k;
            x0 = s.outmu[k]; y0 = v_tb[2 * k + pol]; x1 = s.outmu[k + 1]; y1 = v_tb[2 * (k + 1) + pol];
        }
        b.out[p * out_stride + idx] = y0 + (y1 - y0) * ((um - x0) / (x1 - x0));
    }
    if (t == 0) { b.status[p] = ST_OK; if (b.n3_out) b.n3_out[p] = n3; }
}

}  // namespace smrt
