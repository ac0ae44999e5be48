// The strip finish kernels (passive mode, Flat interfaces): the pivot-free admittance recursion of dort_finish_reg.hpp (same
// algebra) with every N x N matrix spread over the wavefronts of ONE workgroup per (snowpack, frequency)
// pair -- wavefront w keeps tile column w (16 columns, up to NTT tiles of 16 x 16, 4 NTT doubles per lane and matrix) in the
// accumulator layout of v_mfma_f64_16x16x4_f64 -- and ONE matrix at a time broadcast through LDS as the A operand of the
// products.  Two instances: eight wavefronts for 64 < N <= 128 (one workgroup per CU), four for N <= 64 (three per CU).
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
//
// What is solved is the boundary system of smrt/rtsolver/dort.py:263-488 (the same linear system as the other finish
// kernels, eliminated in the order of dort_finish_reg.hpp; tests/studies/admittance_recursion.py is the NumPy statement).
//
// Why strips: a product Z = op(A) Y needs, for the tile column w of Z, the tile column w of Y (registers: B operands as
// they are) and ALL of A -- 128 x 128 doubles are 128 KB, which is what the LDS of a CU holds (tiles of 16 rows x 17
// doubles: 136 KB).  A tile in LDS can be read as the A operand of itself (lane (g, c), k-slab r: element (c, 4 r + g))
// or of its transpose (element (4 r + g, c), which is also the accumulator layout): both walks touch 31 of the 32
// eight-byte banks with the row stride of 17, so every product of the chain is written in its natural orientation and
// nothing is transposed on the matrix core.  The matrix carried from layer to layer is C^^T: every matrix-vector product of
// the recursion is then a sum down the columns of a wavefront's own tile column (no row sums, no cross-wavefront
// reductions).  The three inversions of a layer are block Gauss-Jordan WITHOUT pivoting on the tile columns: the owner of
// block column k broadcasts the inverse of the diagonal block (16 x 16, in registers: rg::inv16_la) and its column through
// LDS, every other wavefront updates its own column; the owner of column k + 1 updates its diagonal tile first and
// eliminates it while the others update (look-ahead), double-buffered: one workgroup barrier per block step.
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_passive.hpp"
#include "dort_finish_reg.hpp"

#ifndef SMRT_STRIP_WOODBURY
#define SMRT_STRIP_WOODBURY 1          // one inversion inside a layer where it is safe (see the layer loop)
#endif
#ifndef SMRT_STRIP_WOODBURY_TAU
#define SMRT_STRIP_WOODBURY_TAU 1e-3   // ... i.e. where min over the streams of sigma x thickness is at least this
#endif

namespace smrt {

#ifdef SMRT_STRIP_TIMING_INV
#define SMRT_SI_PTR (b.stage_out + p * 16)
#else
#define SMRT_SI_PTR nullptr
#endif
// Optional phase timing (profiling builds, -DSMRT_STRIP_TIMING): shader-clock deltas per phase as thread 0 sees them (the
// barriers make every wavefront's view the same to a few hundred cycles), summed per pair into stage_out
#ifdef SMRT_STRIP_TIMING
#define SMRT_ST(k) do { const long long now_ = cycle_counter(); st_acc[st_cur] += (double)(now_ - st_t0); st_t0 = now_; st_cur = (k); } while (0)
#else
#define SMRT_ST(k) do {} while (0)
#endif
enum { STP_SETUP = 0, STP_LOAD, STP_TRI, STP_AT, STP_T1, STP_H, STP_INV1, STP_INV2, STP_T2, STP_CP, STP_IFACE, STP_INV3, STP_Z, STP_SURF, STP_COUNT };

// NTT_: 16 x 16 tiles per side = wavefronts of the workgroup.  8: the 64 < N <= 128 pipeline (one workgroup per CU);
// 4: N <= 64 (three workgroups per CU).
template <int NTT_>
struct StripFinish {

static constexpr int NTT = NTT_;             // 16 x 16 tiles per side: N <= 16 NTT
static constexpr int NW = NTT_;              // wavefronts of the workgroup: one tile column each
static constexpr int NTH = NW * SMRT_LANES;
static constexpr int TROW = 17;              // doubles per tile row in LDS
static constexpr int TS = 16 * TROW;         // doubles per tile in LDS
static constexpr int kBig = NTT * NTT * TS;  // the matrix region
static constexpr int kVecLen = 16 * NTT;     // elements of a vector (128 / 64)
static constexpr int kVectors = 12;          // exchange vectors
static constexpr int kWsDoubles = NTT * NTT * 256;   // one matrix per pair in global memory (DevStage.ws): At between its phases
static constexpr bool kPark = NTT > 4;               // (four wavefronts: a tile column is 16 doubles per lane, At stays in registers -- the
                                                     //  round trip was half of the kernel's 13.5 GB of HBM traffic on the headline batch)
using LaneId = rg::LaneId;

// tile column of a matrix: tile ti, register r, lane 16 g + c of wavefront w holds X[16 ti + 4 r + g][16 w + c]
struct Strip { double v[NTT][4]; };

static SMRT_DEV void zero(Strip& S) {
#pragma unroll
    for (int i = 0; i < NTT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) S.v[i][r] = 0.0;
}

// LDS addressing.  A DS instruction takes one address register and a 16-bit byte offset, and the matrix region spans
// 136 KB: left to itself the compiler builds one address register per tile, hoists them all out of the layer loop and
// spills them (600 registers in the first version).  Here the lane part of an address exists THREE times -- once per
// window of 30 tiles (65 280 bytes) -- and a tile is reached by a compile-time offset inside its window; the values are
// laundered once per layer so that nothing derived from them is loop-invariant.
static constexpr int kWinTiles = 30;
struct Lane {
    LaneId id;
    double* d[3];   // window base + (g * 17 + c): the walk of the accumulator layout
    double* t[3];   // window base + (c * 17 + g): the walk of the transposed tile
    double* col[3]; // the tiles (3 k + i, w) of this wavefront's own column, accumulator walk (windows of three tile rows)
    double* rowt;   // the tiles (w, j) of this wavefront's tile row, transposed walk
};
static SMRT_DEV int launder(int v) {
#if !defined(SMRT_HOST_EMU)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
static SMRT_DEV Lane make_lane(double* big, const LaneId& id, int w) {
    Lane L;
    L.id = id;
    const int od = launder(id.g * TROW + id.c), ot = launder(id.c * TROW + id.g);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        L.d[k] = big + k * kWinTiles * TS + od; L.t[k] = big + k * kWinTiles * TS + ot;
        L.col[k] = big + (3 * k * NTT + w) * TS + od;
    }
    L.rowt = big + w * NTT * TS + ot;
    return L;
}
static SMRT_DEV int tile_index(int ti, int tj) { return ti * NTT + tj; }
// accumulator layout <-> LDS tile T (= tile_index(ti, tj), a compile-time constant after unrolling)
static SMRT_DEV void put_tile(int T, const double (&x)[4], const Lane& L) {
    double* p = L.d[T / kWinTiles] + (T % kWinTiles) * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[4 * r * TROW] = x[r];
}
// the tile in accumulator layout (= the A operand of its transpose)
static SMRT_DEV void get_tile(double (&x)[4], int T, const Lane& L) {
    const double* p = L.d[T / kWinTiles] + (T % kWinTiles) * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = p[4 * r * TROW];
}
// the transpose of the tile in accumulator layout (= the A operand of the tile itself)
static SMRT_DEV void get_tile_t(double (&x)[4], int T, const Lane& L) {
    const double* p = L.t[T / kWinTiles] + (T % kWinTiles) * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = p[4 * r];
}
// tile (ti, w) of this wavefront's own column
static SMRT_DEV void put_own(int ti, const double (&x)[4], const Lane& L) {
    double* p = L.col[ti / 3] + (ti % 3) * NTT * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[4 * r * TROW] = x[r];
}
// the transpose of tile (w, tj) of this wavefront's tile row
static SMRT_DEV void get_row_t(double (&x)[4], int tj, const Lane& L) {
    const double* p = L.rowt + tj * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = p[4 * r];
}
// the same with a run-time tile index inside the first window (the broadcast buffers of the elimination)
static SMRT_DEV void put_tile_rt(int T, const double (&x)[4], const Lane& L) {
    double* p = L.d[0] + T * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[4 * r * TROW] = x[r];
}
static SMRT_DEV void get_tile_t_rt(double (&x)[4], int T, const Lane& L) {
    const double* p = L.t[0] + T * TS;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = p[4 * r];
}

// Z = op(A) Y on the leading nt x nt tiles: A in the LDS matrix region, Y / Z tile columns of this wavefront.
// TRANS: op(A) = A^T.  LOWER: A is lower triangular by tiles (tiles above the diagonal are not read).
template <bool TRANS, bool LOWER = false>
static SMRT_DEV void strip_gemm(Strip& Z, const Strip& Y, int nt, const Lane& L) {
#pragma unroll
    for (int ti = 0; ti < NTT; ++ti) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (ti < nt) {
#pragma unroll
            for (int tk = 0; tk < NTT; ++tk) {
                // (A^T)[ti][tk] = A[tk][ti]^T: lower-triangular A has tk >= ti there, tk <= ti for A itself
                const bool have = tk < nt && (!LOWER || (TRANS ? tk >= ti : tk <= ti));
                if (have) {
                    double a[4];
                    if (TRANS) get_tile(a, tile_index(tk, ti), L);
                    else get_tile_t(a, tile_index(ti, tk), L);
                    rg::tile_tn_acc(acc, a, Y.v[tk]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Z.v[ti][r] = acc[r];
    }
}

// the tile column of this wavefront into the LDS matrix
static SMRT_DEV void put_strip(const Strip& S, int nt, const Lane& L) {
#pragma unroll
    for (int ti = 0; ti < NTT; ++ti)
        if (ti < nt) put_own(ti, S.v[ti], L);
}

// y[j] = sum_i X[i][j] v[i] for the 16 columns of this wavefront -> out[16 w + c] (LDS); v: LDS vector
static SMRT_DEV void strip_matvec_t(const Strip& X, const double* v, double* out, int nt, int w, const LaneId& L) {
    double acc = 0.0;
#pragma unroll
    for (int ti = 0; ti < NTT; ++ti)
        if (ti < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += X.v[ti][r] * v[16 * ti + 4 * r + L.g];
        }
    acc += shfl_xor(acc, 16);
    acc += shfl_xor(acc, 32);
    if (L.g == 0) out[16 * w + L.c] = acc;
}
// X[i][j] <- factor rowf[i] X[i][j] colf[j] + (i == j) diag[i] on this wavefront's columns (null pointers: factor 1 / nothing)
static SMRT_DEV void strip_scale_add_diag(Strip& X, const double* rowf, const double* colf, const double* diag, double factor, int nt,
                                   int w, const LaneId& L) {
    const double cf = (colf ? colf[16 * w + L.c] : 1.0) * factor;
#pragma unroll
    for (int ti = 0; ti < NTT; ++ti)
        if (ti < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + 4 * r + L.g;
                double x = X.v[ti][r] * cf;
                if (rowf) x *= rowf[row];
                if (diag && ti == w && 4 * r + L.g == L.c) x += diag[row];
                X.v[ti][r] = x;
            }
        }
}

// cyclic shift of the leading nt tiles of the column: new[i] = old[(i + 1) % nt]
static SMRT_DEV void rotate_strip(Strip& M, int nt) {
    double t0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t0[r] = M.v[0][r];
#pragma unroll
    for (int i = 0; i < NTT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double nx = M.v[(i + 1 < NTT) ? i + 1 : i][r];
            M.v[i][r] = (i == nt - 1) ? t0[r] : ((i < nt - 1) ? nx : M.v[i][r]);   // (tiles beyond nt stay as they are)
        }
    }
}

// M <- M^-1 on the leading nt x nt tiles (identity padding inside the last tile): block Gauss-Jordan in place without
// pivoting on the tile columns of the wavefronts w < nt.  Broadcast buffers in the LDS matrix region (which must be free):
// the column of the running block step in tiles (buf, 1 .. nt - 1), the inverse of its diagonal block in tile (2 + buf, 0).
// After every step the tiles of a column are rotated so that the running block row is always tile 0 (one copy of the
// step code in a run-time loop); nt rotations restore the order.  The caller provides the barrier in front (the region is
// free, the matrix complete) and must put one behind before the region is reused.
#ifdef SMRT_STRIP_TIMING_INV
#define SMRT_SI(slot) do { if (L.id.lane == 0) { const long long n_ = cycle_counter(); gmem_add(&si_dbg[slot], (double)(n_ - si_t)); si_t = n_; } } while (0)
#define SMRT_SI0() long long si_t = cycle_counter()
#else
#define SMRT_SI(slot) do {} while (0)
#define SMRT_SI0() do {} while (0)
#endif
static SMRT_DEV void strip_invert(Strip& M, int nt, int w, const Lane& L, double* si_dbg = nullptr) {
    double D[4] = {0.0, 0.0, 0.0, 0.0};
    SMRT_SI0();
    if (w == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = M.v[0][r];
        rg::inv16(D, L.id);
        put_tile(tile_index(2, 0), D, L);
#pragma unroll
        for (int i = 1; i < NTT; ++i)
            if (i < nt) put_tile(tile_index(0, i), M.v[i], L);
    }
#if !defined(SMRT_HOST_EMU)
#pragma nounroll
#endif
    for (int k = 0; k < nt; ++k) {
        const int buf = k & 1;
        const int colb = buf * NTT, nxtb = (buf ^ 1) * NTT;        // tiles (buf, i) / (buf ^ 1, i): run-time index inside the first window
        if (w == 0) SMRT_SI(k == 0 ? 0 : 1);     // wavefront 0: prologue / its step work
        block_sync();
        if (w == 0) SMRT_SI(2);                  // wavefront 0: waiting at the barrier
        if (w < nt) {
            if (w != k) {
                double a[4], R[4] = {0.0, 0.0, 0.0, 0.0};
                get_tile_t_rt(a, (2 + buf) * NTT, L);
                rg::tile_tn_acc(R, a, M.v[0]);                                   // row of the block step: D M[k][w]
                const bool next = (w == k + 1);
#ifdef SMRT_STRIP_TIMING_INV
                if (next) si_t = cycle_counter();
#endif
                double Dn[4] = {0.0, 0.0, 0.0, 0.0};
                if (next) {
                    // The owner of the next block column: its diagonal tile first, then the 16 x 16 elimination -- one long
                    // dependent chain on the vector unit -- with the updates of its other tiles issued INTO that chain, one tile
                    // every other elimination step (operands requested a step ahead): the matrix core works in the shadow
                    // of the chain instead of behind it.  No branches in the sequence: a tile beyond nt is read from tile 1's
                    // place and its result is never used (the tiles of a column beyond nt are dead storage).
                    double u[4] = {0.0, 0.0, 0.0, 0.0};
                    get_tile_t_rt(a, colb + 1, L);
                    rg::tile_tn_acc(u, a, R);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { M.v[1][r] -= u[r]; Dn[r] = M.v[1][r]; }
#ifdef SMRT_STRIP_TIMING_INV
                    if (w != 0) SMRT_SI(4);      // next owner: diagonal tile update (from the end of R)
#endif
                    // tile i = 2 + j / 4, k-slab j % 4: ONE matrix-core instruction (64 cycles of the pipe) per half step, so
                    // that the wavefront never waits for the pipe in the middle of the chain; the operands of a tile are
                    // requested two half steps before its first slab
                    double ta[2][4], tv[2][4];
                    auto want = [&](int i) { if (i < NTT) get_tile_t_rt(ta[i & 1], colb + (i < nt ? i : 1), L); };
                    auto slab = [&](int j) {
                        const int i = 2 + j / 4, kk = j % 4;
                        if (i < NTT) {
                            if (kk == 0) { tv[i & 1][0] = tv[i & 1][1] = tv[i & 1][2] = tv[i & 1][3] = 0.0; }
                            mfma_f64_16x16x4(ta[i & 1][kk], R[kk], tv[i & 1]);
                            if (kk == 3) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) M.v[i < NTT ? i : 0][r] -= tv[i & 1][r];
                            }
                        }
                    };
                    double rk, piv, pinv;
                    rg::Inv16Half hs;
                    want(2);
                    rg::inv16_la_begin(Dn, rk, piv, pinv);
                    // half steps 2 K (pivot look-ahead) and 2 K + 1 (tile update): slab j after half step j + 1; tile i wanted
                    // before half step 4 (i - 2) - 1
#define SMRT_LA(K) rg::inv16_la_pivot<K>(Dn, rk, piv, pinv, hs, L.id); if (2 * (K) >= 1) slab(2 * (K) - 1); \
                   if ((2 * (K) + 2) % 4 == 0) want(2 + (2 * (K) + 2) / 4); \
                   rg::inv16_la_update<K>(Dn, rk, piv, pinv, hs, L.id); slab(2 * (K));
                    SMRT_LA(0) SMRT_LA(1) SMRT_LA(2) SMRT_LA(3) SMRT_LA(4) SMRT_LA(5) SMRT_LA(6) SMRT_LA(7)
                    SMRT_LA(8) SMRT_LA(9) SMRT_LA(10) SMRT_LA(11) SMRT_LA(12) SMRT_LA(13) SMRT_LA(14) SMRT_LA(15)
#undef SMRT_LA
                    slab(31);
#ifdef SMRT_STRIP_TIMING_INV
                    if (w != 0) SMRT_SI(5);      // next owner: elimination + interleaved updates
#endif
                } else {
#pragma unroll
                    for (int i = 1; i < NTT; ++i)
                        if (i < nt) {
                            double u[4] = {0.0, 0.0, 0.0, 0.0};
                            get_tile_t_rt(a, colb + i, L);
                            rg::tile_tn_acc(u, a, R);
#pragma unroll
                            for (int r = 0; r < 4; ++r) M.v[i][r] -= u[r];
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) M.v[0][r] = R[r];
                if (next) {   // broadcast for step k + 1, in its (rotated) order: tile i of then is tile i + 1 of now, the last one tile 0
                    put_tile_rt((2 + (buf ^ 1)) * NTT, Dn, L);
#pragma unroll
                    for (int i = 2; i < NTT; ++i)
                        if (i < nt) put_tile_rt(nxtb + i - 1, M.v[i], L);
                    if (nt > 1) put_tile_rt(nxtb + nt - 1, M.v[0], L);
#pragma unroll
                    for (int r = 0; r < 4; ++r) D[r] = Dn[r];
#ifdef SMRT_STRIP_TIMING_INV
                    if (w != 0) SMRT_SI(6);      // next owner: broadcast stores
#endif
                }
            } else {   // the column of the block step itself: M[i][k] = -M[i][k] D, M[k][k] = D
#pragma unroll
                for (int i = 1; i < NTT; ++i)
                    if (i < nt) {
                        double a[4], u[4] = {0.0, 0.0, 0.0, 0.0};
                        get_tile_t_rt(a, colb + i, L);
                        rg::tile_tn_acc(u, a, D);
#pragma unroll
                        for (int r = 0; r < 4; ++r) M.v[i][r] = -u[r];
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) M.v[0][r] = D[r];
            }
            rotate_strip(M, nt);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// the per-pair driver: one workgroup of NTT wavefronts
// ------------------------------------------------------------------------------------------------------------
// Supported: passive mode, N <= 16 NTT, Flat interfaces, no / Flat / Reflector substrate, atmosphere, prune_deep_snowpack.
// The host routes batches with process_coherent_layers (T != 1 - R), a host-evaluated dense substrate or rough interfaces
// to the pivoted finish kernels (dort_hip.hip).
template <bool MAY_DIRECT>
static SMRT_DEV void run(const DevBatch& b, long long p, double* lds_base, const DevStage& stg) {
    constexpr int NT = NTH, P = 2;
    const LaneId Ln0 = rg::lane_id();
    const int t = tid();
    const int w = t >> 6;
    const int nmax = b.n_max_stream;
    const int out_stride = P * b.n_theta;

    // ---- LDS: matrix region, exchange vectors, tables
    double* const big = lds_base;
    double* const V = lds_base + kBig;
    double* const E0 = V;   // (the layer loop addresses the twelve vectors through its own laundered base)
    Lds s;
    {
        double* v = V + kVectors * kVecLen;
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

    if (t < 16) s.ints[t] = 0;   // ([8], [9]: the thin-layer flags of the layer loop, by layer parity)
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
    const int LD = (nmax * P + 1) | 1;   // leading dimension of the staged matrices (make_plan)

#ifdef SMRT_STRIP_TIMING
    double st_acc[STP_COUNT];
    for (int k = 0; k < STP_COUNT; ++k) st_acc[k] = 0.0;
    long long st_t0 = cycle_counter();
    int st_cur = STP_SETUP;
#endif
    double n3 = 0.0;
    // element t (< 16 NTT: the first wavefronts) of the source c of the relation delta = -C s + c carried from layer to layer
    // (physical coordinates of the layer it is used in).  The matrix is carried TRANSPOSED, Ct = C^^T as tile columns: every
    // matrix-vector product of the recursion is then a sum down the columns of a wavefront's own tile column (strip_matvec_t),
    // and the only use of C^ as a matrix, T1 = C^^T A+, reads it from LDS in either orientation anyway.
    double c_e = 0.0;
    double tb_e = 0.0;
    Strip C;   // C^^T of the layer at hand, tile column of this wavefront (carried in registers between the layers)
    zero(C);

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
        // a layer the Rayleigh kernel diagonalised (dort_rayleigh_kernel.hpp): the matrix slot holds A+ = D V itself and the
        // vector slot 1 / D^2; A- = -D^-2 A+ Sigma, i.e. W = D^-2 A+ Sigma^2 element by element -- no L+, no triangular stage
        // (MAY_DIRECT = false: the instance for batches without such layers -- DevBatch::rayleigh_direct = 0 -- has the other
        //  form of the stage only: 188 -> 160 B of scratch and 10.27 -> 10.0 ms on the headline batch for the four-wavefront kernel)
        const bool direct = MAY_DIRECT && stage_direct(stg.n[item]);   // (uniform)
        const bool in_e = t < N;
        SMRT_ST(STP_LOAD);
        // (LDS addresses: laundered once per layer, see make_lane; the lane coordinates too -- the row indices and diagonal
        // masks derived from them are cheap to recompute and were hoisted out of the layer loop by the dozen, and spilled)
        LaneId Ln = Ln0;
        Ln.g = launder(Ln0.g); Ln.c = launder(Ln0.c);
        const Lane Lw = make_lane(big, Ln, w);
        double* const lb = big + launder(0);   // for the element-wise walks below
        double* const V = lb + kBig;           // (the exchange vectors through the laundered base too)
        double* const E0 = V, * const E1 = V + kVecLen, * const E2 = V + 2 * kVecLen, * const E3 = V + 3 * kVecLen;
        double* const E4 = V + 4 * kVecLen, * const E5 = V + 5 * kVecLen, * const E6 = V + 6 * kVecLen, * const E7 = V + 7 * kVecLen;
        double* const E8 = V + 8 * kVecLen, * const E9 = V + 9 * kVecLen, * const E10 = V + 10 * kVecLen, * const E11 = V + 11 * kVecLen;
        // ---- element t of the vectors of this layer (padding: d = sigma = 1, t = 0)
        const double d_e = in_e ? stg.d[item * stg.vec_stride + (in_e ? t : 0)] : 1.0;
        const double sg_e = in_e ? stg.sigma[item * stg.vec_stride + (in_e ? t : 0)] : 1.0;
        // ---- L+ -> LDS (lower tiles, zero above the diagonal and beyond N); the inverses of its diagonal blocks into
        //      free tiles above the diagonal: block i in tile (i, i + 1), the last one in tile (0, 2).  Eight requests in
        //      flight per thread before the first store (a load per iteration waited for its own latency: 45 k cycles per layer).
        if (!direct) {
            const int NP = 16 * nt;
            constexpr int CPP = NTH / kVecLen, U = 8;   // columns per pass of the workgroup, passes per batch
            const int r = t % kVecLen, cq = t / kVecLen;
            for (int c0 = 0; c0 < NP; c0 += CPP * U) {
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc = c0 + u * CPP + cq;
                    const bool in = r < N && cc < N && r >= cc;
                    v[u] = gL[in ? cc * LD + r : 0];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cc = c0 + u * CPP + cq;
                    const bool in = r < N && cc < N && r >= cc;
                    if (r < NP && cc < NP && (r >> 4) >= (cc >> 4))
                        lb[((r >> 4) * NTT + (cc >> 4)) * TS + (r & 15) * TROW + (cc & 15)] = in ? v[u] : 0.0;
                }
            }
            constexpr int UI = (256 * NTT + NT - 1) / NT;   // passes over the block inverses (4)
            double vi[UI];
#pragma unroll
            for (int u = 0; u < UI; ++u) { const int e = t + u * NT; vi[u] = gI[e < 256 * nt ? e : 0]; }
#pragma unroll
            for (int u = 0; u < UI; ++u) {
                const int e = t + u * NT;
                if (e < 256 * nt) {   // element e = blk 256 + j 16 + i of the prep kernel's table: (L_blk^-1)[i][j]
                    const int blk = e >> 8, j = (e >> 4) & 15, i = e & 15;
                    const int T = (blk < NTT - 1) ? blk * NTT + blk + 1 : 2;
                    lb[T * TS + i * TROW + j] = vi[u];
                }
            }
        }
        // ---- B' (tile column of this wavefront) requested behind the copy -- with it in flight during the copy the batch
        //      above spills -- and in front of the vector arithmetic, which covers most of its latency
        Strip X;
        zero(X);
        const rg::LaneOffsets lo = rg::lane_offsets(LD, N, Ln);
        if (w < nt) {
#pragma unroll
            for (int ti = 0; ti < NTT; ++ti) rg::load_tile_raw(X.v[ti], gB, LD, ti, w, ti < nt, lo);
        }
        const double di_e = fast_rcp(d_e);
        const double nrs_e = -fast_rcp(sg_e);                      // -1 / sigma
        const double tt_e = in_e ? exp(-sg_e * s.thick[l]) : 0.0;
        // One inversion inside the layer instead of two where every stream of the layer is optically thick enough (Woodbury on
        // M3: Theta = diag((1 + t^2) / (Sigma (1 - t^2))) - 4 G (H + Sigma (1 + t^2) / (1 - t^2))^-1 G, G = t / (1 - t^2) -- a
        // difference of O(1 / (sigma d)) terms that loses digits in proportion to 1 / (sigma d)^2: tests/studies/
        // woodbury_per_layer.py -- unchanged at 1e-9 K on the hard inputs with the shortcut for min sigma d >= 1e-4, 6e-6 K
        // with it everywhere; taken from 1e-3 on).  Decided per layer through an LDS flag: uniform after the barrier below.
        const double sd_e = sg_e * s.thick[l];
        if (in_e && !(sd_e >= SMRT_STRIP_WOODBURY_TAU)) lds_or(&s.ints[8 + (l & 1)], 1);
        if (t == 0) s.ints[8 + ((l + 1) & 1)] = 0;                 // (the flag of the next layer; last read many barriers ago)
        const double omt2_e = in_e ? -expm1(-2.0 * sd_e) : 1.0;    // 1 - t^2
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
            if (t < kVecLen) E7[t] = cd;
        }
        if (t < kVecLen) {
            E0[t] = nrs_e; E1[t] = sg_e; E4[t] = di_e; E5[t] = d_e;   // (E2, E3, E7: set where the layer's matrices are formed)
            E6[t] = c_e * di_e;                                    // c^ = D^-1 c
            if (direct) E11[t] = in_e ? gI[t] : 0.0;               // 1 / D^2 (E11 is free until the first inversion is done)
        }
        block_sync();
        const bool wood = SMRT_STRIP_WOODBURY && s.ints[8 + (l & 1)] == 0;   // (uniform)
        if (l == Lk - 1) {
#pragma unroll
            for (int ti = 0; ti < NTT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) C.v[ti][r] = (ti == w && ti < nt && 4 * r + Ln.g == Ln.c) ? E7[16 * ti + Ln.c] : 0.0;
        }
        SMRT_ST(STP_TRI);
        Strip At;   // W = L+ B' -> At = A-^T = -Sigma^-1 W^T
        zero(At);
        if (w < nt) {
#pragma unroll
            for (int ti = 0; ti < NTT; ++ti) rg::mask_tile(X.v[ti], ti, w, ti < nt, lo);
            if (direct) {
                // X is A+ already; W = D^-2 A+ Sigma^2: element (16 ti + 4 r + g, 16 w + c) of this wavefront's tile column
                const double sc = E1[16 * w + Ln.c];
                const double s2c = sc * sc;
#pragma unroll
                for (int ti = 0; ti < NTT; ++ti)
                    if (ti < nt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) At.v[ti][r] = X.v[ti][r] * (E11[16 * ti + 4 * r + Ln.g] * s2c);
                    }
            } else {
            strip_gemm<false, true>(At, X, nt, Lw);                         // W = L+ B'
            // ---- A+ = L+^-T B' in place: blocked back substitution with the diagonal-block inverses of the prep kernel
#pragma unroll
            for (int ti = NTT - 1; ti >= 0; --ti) {
                if (ti < nt) {
#pragma unroll
                    for (int tk = ti + 1; tk < NTT; ++tk)
                        if (tk < nt) {
                            double a[4], u[4] = {0.0, 0.0, 0.0, 0.0};
                            get_tile(a, tile_index(tk, ti), Lw);            // L+[tk][ti]^T X[tk]
                            rg::tile_tn_acc(u, a, X.v[tk]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) X.v[ti][r] -= u[r];
                        }
                    double a[4], u[4] = {0.0, 0.0, 0.0, 0.0};
                    get_tile(a, (ti < NTT - 1) ? tile_index(ti, ti + 1) : tile_index(0, 2), Lw);
                    rg::tile_tn_acc(u, a, X.v[ti]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) X.v[ti][r] = u[r];
                }
            }
            }
            strip_matvec_t(X, E6, E8, nt, w, Ln);                           // A+^T c^ -> E8
            strip_matvec_t(At, E4, E9, nt, w, Ln);                          // W^T D^-1 1 -> E9  (x1 = A-^T D^-1 1 = -Sigma^-1 W^T D^-1 1)
        }
        block_sync();                                                       // L+ is dead
        SMRT_ST(STP_AT);
        if (w < nt) put_strip(At, nt, Lw);                                  // W, to be read back transposed
        block_sync();
        // At waits in this pair's matrix in global memory (lane-contiguous, the wavefront's own part) until Theta exists:
        // fewer tile columns in registers through the two inversions
        // (uniform base + a laundered 32-bit lane offset: as 64-bit addresses the 32 slots were hoisted out of the layer loop and spilled)
        double* const wsb = stg.ws + p * (long long)kWsDoubles;
        const int wso = launder((w * NTT * 4) * SMRT_LANES + Ln.lane);
        if (w < nt) {
#pragma unroll
            for (int tj = 0; tj < NTT; ++tj)
                if (tj < nt) {
                    get_row_t(At.v[tj], tj, Lw);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        At.v[tj][r] *= E0[16 * tj + 4 * r + Ln.g];
                        if (kPark) wsb[wso + (tj * 4 + r) * SMRT_LANES] = At.v[tj][r];
                    }
                }
        }
        block_sync();
        if (w < nt) put_strip(C, nt, Lw);                                   // C^^T
        block_sync();
        SMRT_ST(STP_T1);
        Strip X2;   // T1 = C^^T A+, then the work column of the products
        zero(X2);
        if (w < nt) {
            strip_gemm<false>(X2, X, nt, Lw);                               // (the region holds C^^T as it is)
            strip_matvec_t(X2, E4, E10, nt, w, Ln);                         // A+^T C^ D^-1 1 = T1^T D^-1 1 -> E10
        }
        block_sync();
        SMRT_ST(STP_H);
        if (t < kVecLen) {
            const double r_e = E8[t] - 2.0 * Bl * E10[t];                   // r = A+^T (c^ - 2 B C^ 1^)
            // The matrix inverted next, H + K with K = Sigma (two inversions) or Sigma (1 + t^2) / (1 - t^2) (one), is inverted
            // as I + s H s, s = K^-1/2: pivots of O(1).  The 16 x 16 elimination loses |pivot|^2 ulps on the diagonal of the
            // inverse (inv16_step) -- 1e-13 with Sigma, but K is Sigma / (sigma d) for a thin layer and the one-inversion
            // form amplifies by 1 / (sigma d)^2 on top: 2.5e-4 K on the hard inputs without the scaling, 1e-9 K with it.
            // With two inversions the second matrix takes the same s: M3 = Sigma (1 - t^2) + 2 Sigma t P t Sigma
            // = s^-1 (1 - t^2 + 2 t (I + s H s)^-1 t) s^-1.
            double kk = sg_e, g = tt_e, dg = omt2_e;                        // (padding: 1, 0, 1 in both forms)
            if (wood) {
                const double io = fast_rcp(omt2_e), q = (1.0 + tt_e * tt_e) * io;
                kk = sg_e * q; g = tt_e * io; dg = -nrs_e * q;              // K, G, (1 + t^2) / (Sigma (1 - t^2))
            }
            const double ks = fast_rsqrt(kk);
            E8[t] = r_e * ks;                                               // s r
            E2[t] = wood ? g * ks : g; E3[t] = dg; E7[t] = ks; E11[t] = 1.0;   // (E11 is free: 1 / D^2 of a direct layer was used above)
        }
        if (w < nt) put_strip(X, nt, Lw);                                   // A+
        block_sync();
        if (w < nt) {
            strip_gemm<true>(C, X2, nt, Lw);                                // H^T = A+^T (C^^T A+)   (C is free: the work column)
            strip_scale_add_diag(C, E7, E7, E11, 1.0, nt, w, Ln);           // I + s H^T s
        }
        block_sync();
        SMRT_ST(STP_INV1);
        strip_invert(C, nt, w, Lw, SMRT_SI_PTR);                                         // (I + s H s)^-T
        if (w < nt) strip_matvec_t(C, E8, E10, nt, w, Ln);                  // (I + s H s)^-1 s r -> E10
        block_sync();
        // (one flow for both forms: the second inversion and what hangs on it are skipped with one inversion)
        if (t < kVecLen) {
            const double yv = E2[t] * E10[t];                               // y = G (H + K)^-1 r, or s Sigma t q = t (I + s H s)^-1 s r
            if (wood) E10[t] = yv; else E11[t] = yv;
            E9[t] *= nrs_e;                                                 // x1
        }
        // Theta^T = diag - 4 G (H + K)^-T G, or s M3^T s = 1 - t^2 + 2 t (I + s H s)^-T t
        if (w < nt) strip_scale_add_diag(C, E2, E2, E3, wood ? -4.0 : 2.0, nt, w, Ln);
        block_sync();
        SMRT_ST(STP_INV2);
        if (!wood) {
            strip_invert(C, nt, w, Lw, SMRT_SI_PTR);                                     // (s M3 s)^-T
            if (w < nt) {
                strip_scale_add_diag(C, nullptr, E7, nullptr, 1.0, nt, w, Ln);  // (s M3 s)^-T s
                strip_matvec_t(C, E11, E10, nt, w, Ln);                     // y = M3^-1 (Sigma t q) = s (s M3 s)^-1 (s Sigma t q) -> E10
                strip_scale_add_diag(C, E7, nullptr, E0, 2.0, nt, w, Ln);   // Theta^T = 2 M3^-T - Sigma^-1
            }
        }
        if (w < nt) strip_matvec_t(C, E9, E6, nt, w, Ln);                   // x2 = Theta x1 -> E6 (free since the first phase)
        block_sync();
        SMRT_ST(STP_T2);
        if (kPark) zero(At);
        if (w < nt) {
            put_strip(C, nt, Lw);                                           // Theta^T
            if (kPark) {
#pragma unroll
                for (int ti = 0; ti < NTT; ++ti)
                    if (ti < nt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) At.v[ti][r] = wsb[wso + (ti * 4 + r) * SMRT_LANES];
                    }
            }
            strip_matvec_t(At, E10, E8, nt, w, Ln);                         // A- y = At^T y -> E8
            strip_matvec_t(At, E6, E9, nt, w, Ln);                          // C^' D^-1 1 = A- Theta A-^T D^-1 1 = At^T x2 -> E9
        }
        block_sync();
        if (t < kVecLen) c_e = d_e * (2.0 * Bl * E9[t] - 2.0 * E8[t]);      // c' (physical coordinates)
        if (w < nt) strip_gemm<false>(X2, At, nt, Lw);                      // T2' = Theta^T At
        block_sync();
        SMRT_ST(STP_CP);
        if (w < nt) put_strip(At, nt, Lw);
        block_sync();
        if (w < nt) {
            strip_gemm<true>(C, X2, nt, Lw);                                // C^'^T = At^T Theta^T At
            strip_scale_add_diag(C, E4, E5, nullptr, 1.0, nt, w, Ln);       // C'^T = D^-1 C^'^T D
        }
        SMRT_ST(l == 0 ? STP_SURF : STP_IFACE);
        if (l == 0) {
            // surface (dort.py:391-395,484): I_dn = r2 I_up + t2 I_sky just below it;
            // S I_up = c' + (I - C') t2 I_sky,  S = (1 - r2) + C' (1 + r2);  I0 = R_air I_sky + t1 I_up
            const bool atm = (b.atm_down != nullptr);
            const double Idn = atm ? (b.rayleigh_jeans ? b.atm_down[fi] : planck_radiance(frequency, b.atm_down[fi])) : 0.0;
            double Tair = 0.0, r2s = 0.0, t1s_e = 0.0, Rair_e = 0.0;
            const cplx one = cmk(1.0, 0.0);
            block_sync();                                                   // (E8, E9 are read)
            if (t < kVecLen) {
                if (in_e) { r2s = flat_R(el, one, s.ri[0] * s.gsin[t >> 1], t & 1); t1s_e = 1.0 - r2s; }
                if (t < n_air * P) {
                    double Rv, Rh;
                    fresnel_RvRh(one, el, s.outmu[t >> 1], &Rv, &Rh);
                    Rair_e = (t & 1) ? Rh : Rv; Tair = 1.0 - Rair_e;
                }
                E6[t] = Tair * Idn;                                         // t2 I_sky (0 beyond the air streams)
                E7[t] = in_e ? 1.0 + r2s : 0.0; E8[t] = in_e ? 1.0 - r2s : 1.0;
            }
            block_sync();
            if (w < nt) {
                strip_matvec_t(C, E6, E9, nt, w, Ln);                       // C' (t2 I_sky) -> E9
                strip_scale_add_diag(C, E7, nullptr, E8, 1.0, nt, w, Ln);   // S^T = (1 - r2) + (1 + r2) C'^T
            }
            block_sync();
            if (t < kVecLen) E10[t] = in_e ? c_e + E6[t] - E9[t] : 0.0;     // right-hand side
            strip_invert(C, nt, w, Lw, SMRT_SI_PTR);
            if (w < nt) strip_matvec_t(C, E10, E9, nt, w, Ln);              // I_up just below the surface
            block_sync();
            if (t < n_air * P) {
                double I0 = Rair_e * Idn + t1s_e * E9[t];
                if (atm) I0 = (b.rayleigh_jeans ? b.atm_up[fi] : planck_radiance(frequency, b.atm_up[fi])) + b.atm_trans[fi] * I0;
                tb_e = b.rayleigh_jeans ? I0 : planck_inverse(frequency, I0);
            }
            break;
        }
        // ---- interface with the layer above: diagonal coefficients per element (streams paired by index)
        const int Nu = (int)s.nl[l - 1] * P;
        const int nc = (N < Nu) ? N : Nu;
        const int ntu = (Nu + 15) >> 4;
        const int ntm = nt > ntu ? nt : ntu;
        const long long item_u = item - 1;
        const bool in_u = t < Nu;
        double cd_e = 0.0, it2_e = 0.0;
        block_sync();                                                       // (E0 ... E5 of the layer are read)
        if (t < kVecLen) {
            const cplx eup = cmk(s.eps_re[l - 1], s.eps_im[l - 1]);
            double r1 = 1.0, t2 = 0.0, r2 = 0.0, t1 = 0.0, extra_e = 0.0;
            if (in_e) {   // from this layer upwards
                r2 = flat_R(el, eup, s.ri[l] * s.gsin[t >> 1], t & 1);
                t1 = (t < nc) ? 1.0 - r2 : 0.0;
            }
            if (in_u) {  // from the upper layer downwards
                const double rb = flat_R(eup, el, s.ri[l - 1] * s.gsin[t >> 1], t & 1);
                if (t < nc) { r1 = rb; t2 = 1.0 - rb; }
                else extra_e = (1.0 - rb) * fast_rcp(1.0 + rb);   // a stream that does not exist below: I_up = R I_dn
            }
            const double tt2 = t1 * t2;
            const double ca = 0.5 * (tt2 + (1.0 + r1) * (1.0 - r2));
            const double cc = 0.5 * (tt2 - (1.0 - r1) * (1.0 - r2));
            const double cb_e = 0.5 * (tt2 - (1.0 + r1) * (1.0 + r2));
            cd_e = 0.5 * (tt2 + (1.0 - r1) * (1.0 + r2));
            const double t2_e = (t < nc) ? t2 : 0.0;
            it2_e = (t < nc) ? fast_rcp(t2) : 0.0;
            const double du_e = in_u ? stg.d[item_u * stg.vec_stride + (in_u ? t : 0)] : 1.0;
            const double dui_e = fast_rcp(du_e);
            // Y^T = a - C'^T b, Nn^T = c - C'^T d (column factors and diagonals); then the factors of the result
            E0[t] = in_e ? -cb_e : 0.0; E1[t] = in_e ? ca : 1.0;
            E2[t] = in_e ? -cd_e : 0.0; E3[t] = in_e ? cc : 0.0;
            E6[t] = in_e ? cb_e * c_e : 0.0;                                // b c'
            E7[t] = -it2_e * dui_e; E8[t] = t2_e * du_e; E9[t] = extra_e;
        }
        block_sync();
        Strip Nn;   // Nn^T
        zero(Nn);
        if (w < nt) {
            const int col = 16 * w + Ln.c;
            const double f0 = E0[col], f2 = E2[col];
#pragma unroll
            for (int ti = 0; ti < NTT; ++ti)
                if (ti < nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * ti + 4 * r + Ln.g;
                        const bool dg = (ti == w && 4 * r + Ln.g == Ln.c);
                        const double cv = C.v[ti][r];
                        C.v[ti][r] = cv * f0 + (dg ? E1[row] : 0.0);
                        Nn.v[ti][r] = cv * f2 + (dg ? E3[row] : 0.0);
                    }
                }
        }
        block_sync();                                                       // (At in the region is read)
        SMRT_ST(STP_INV3);
        strip_invert(C, nt, w, Lw, SMRT_SI_PTR);                                         // Y^-T
        block_sync();
        SMRT_ST(STP_Z);
        if (w < nt) put_strip(C, nt, Lw);
        block_sync();
        // ---- Z^T = Y^-T Nn^T;  C_u = -t2^-1 Z t2 on the common streams, (1 - R) / (1 + R) on the diagonal of the upper layer's
        //      extra streams;  c_u = (d c' - Z b c') / t2;  in the hats of the layer above, C^ = D^-1 C D -- transposed
        zero(X2);
        if (w < nt) {
            strip_gemm<false>(X2, Nn, nt, Lw);
            strip_matvec_t(X2, E6, E10, nt, w, Ln);                         // Z b c' -> E10
        }
        if (w < ntm) {
            const int col = 16 * w + Ln.c;
            const double cf = E7[col];                                      // (-1 / (t2 d_u)) of the column = row of C_u
#pragma unroll
            for (int ti = 0; ti < NTT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + 4 * r + Ln.g;
                    double x = 0.0;
                    if (ti < ntm) {
                        x = (row < nc && col < nc) ? X2.v[ti][r] * E8[row] * cf : 0.0;   // (X2 is zero outside nt x nt)
                        if (row == col) x += E9[row];
                    }
                    C.v[ti][r] = x;
                }
        } else zero(C);
        block_sync();
        if (t < kVecLen) c_e = (t < nc) ? (cd_e * c_e - E10[t]) * it2_e : 0.0;
        block_sync();
    }

    if (t < kVecLen) E0[t] = tb_e;
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
#ifdef SMRT_STRIP_TIMING
    SMRT_ST(STP_SETUP);
    if (t == 0 && b.stage_out) for (int k = 0; k < 16; ++k) b.stage_out[p * 16 + k] = (k < STP_COUNT) ? st_acc[k] : 0.0;
#endif
}

};   // StripFinish

// LDS layout of the strip finish kernels: the matrix region, the exchange vectors, then the tables pair_setup fills (stream
// tables 3 x n_max_stream, layer tables 15 x Lmax, 8 doubles of flags)
SMRT_HD int finish_strip_lds_doubles(int n_max_stream, int Lmax, int ntt = 8) {
    return ntt * ntt * 16 * 17 + 12 * 16 * ntt + 3 * n_max_stream + 15 * Lmax + 8;
}
SMRT_DEV void dort_pair_passive_strip(const DevBatch& b, long long p, double* lds_base, const DevStage& stg) {
    StripFinish<8>::run<true>(b, p, lds_base, stg);
}
// MAY_DIRECT: the batch may hold layers the Rayleigh kernel diagonalised in closed form (DevBatch::rayleigh_direct)
template <bool MAY_DIRECT>
SMRT_DEV void dort_pair_passive_strip4(const DevBatch& b, long long p, double* lds_base, const DevStage& stg) {
    StripFinish<4>::run<MAY_DIRECT>(b, p, lds_base, stg);
}

}  // namespace smrt
