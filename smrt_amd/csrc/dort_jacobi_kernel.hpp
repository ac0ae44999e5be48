// The Jacobi kernel of the three-kernel pipelines: one (pair, layer[, azimuth mode]) item per workgroup.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_passive.hpp"

namespace smrt {

#ifndef SMRT_JACOBI_GS
#define SMRT_JACOBI_GS 8     // lanes per column pair in the Jacobi kernel
#endif
#ifndef SMRT_JACOBI_NT
#define SMRT_JACOBI_NT 256   // threads per workgroup of the Jacobi kernel
#endif

// Rotation that annihilates the inner product g of two columns with squared norms a, b (dd = b - a):
// t = sign(dd) 2g / (|dd| + sqrt(dd^2 + 4 g^2)), c = 1 / sqrt(1 + t^2), s = c t.  From the third sweep on almost every
// rotation is a small one, and the parameters sit on the sequential chain of the kernel: for g^2 < 1e-8 dd^2 the series
// t = (g / dd)(1 - (g / dd)^2), c = 1 - t^2 / 2, s = c t (relative errors 2e-16, 4e-17, 4e-17) replaces two of the
// three reciprocal (square root)s and their Newton steps.
#ifndef SMRT_JACOBI_SMALL_ANGLE
#define SMRT_JACOBI_SMALL_ANGLE 1e-8
#endif
SMRT_DEV void jacobi_rotation(double gg, double g2, double dd, double& tt, double& c, double& sn) {
    if (g2 < SMRT_JACOBI_SMALL_ANGLE * (dd * dd)) {
        const double t0 = gg * fast_rcp1(dd);
        tt = t0 - t0 * (t0 * t0);
        c = 1.0 - 0.5 * (tt * tt);
        sn = c * tt;
    } else {
        const double hh = dd * dd + 4.0 * g2;
        const double h = hh * fast_rsqrt1(hh);
        tt = (dd >= 0.0 ? 2.0 : -2.0) * gg * fast_rcp1(fabs(dd) + h);
        c = fast_rsqrt(1.0 + tt * tt);
        sn = c * tt;
    }
}

// ---- zero-padded Jacobi step for the split pipeline ----------------------------------------------------------
// The matrix is padded with zero rows/columns up to CP = NB*m columns and RPL*GS rows, so no lane ever needs a
// validity test or a masked load (a zero column never rotates: g = 0).  Column norms are tracked in LDS (a rotation
// changes them by -/+ t g exactly) and refreshed once per sweep, so a step needs ONE dot product and ONE group sum.
template <int GS, int RPL>
SMRT_DEV void rotate_pair_padded(double* Bm, int LD, int p, int q, int sub, double* nrm, int* flag, double skip2,
                                 double exit2) {
    double x[RPL], y[RPL];
    double gg = 0.0, gg2 = 0.0;
    double* cp = Bm + p * LD;
    double* cq = Bm + q * LD;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        const int r = sub + i * GS;
        x[i] = cp[r];
        y[i] = cq[r];
        if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
    }
    const double a = nrm[p], bb = nrm[q];
    gg = group_sum<GS>(gg + gg2);
    const double g2 = gg * gg, ab = a * bb;
    if (g2 > skip2 * ab) {
        const double dd = bb - a;
        double tt, c, sn;
        jacobi_rotation(gg, g2, dd, tt, c, sn);
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int r = sub + i * GS;
            cp[r] = c * x[i] - sn * y[i];
            cq[r] = sn * x[i] + c * y[i];
        }
        if (sub == 0) {
            nrm[p] = a - tt * gg;
            nrm[q] = bb + tt * gg;
            if (g2 > exit2 * ab) lds_or(flag, 1);
        }
    }
}

// One pass with the X columns resident in registers: lane group `slot` takes pass slot ps (ps < nslots; several rounds when
// there are more pass slots than lane groups), loads its column xcol(ps) once, rotates it against ycol(ps, j) for
// j = 0 .. steps - 1 -- only the Y column moves through LDS -- and stores it once.  The Y columns of one step must be
// distinct among the pass slots; an idle lane group works on the all-zero padding column CP (g = 0: never rotates).
template <int GS, int RPL, class FX, class FY>
SMRT_DEV void rotate_resident(double* Bm, int LD, int CP, int nslots, int steps, int slot, int sub, double* nrm, int* flag,
                              double skip2, double exit2, FX xcol, FY ycol) {
    constexpr int SLOTS = SMRT_LANES / GS;
    for (int ps0 = 0; ps0 < nslots; ps0 += SLOTS) {
        const int ps = ps0 + slot;
        const bool valid = ps < nslots;
        const int pc = valid ? xcol(ps) : CP;
        double* cp = Bm + pc * LD;
        double x[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) x[i] = cp[sub + i * GS];
        double a = nrm[pc];
        for (int j = 0; j < steps; ++j) {
            const int qc = valid ? ycol(ps, j) : CP;
            double* cq = Bm + qc * LD;
            double y[RPL];
            double gg = 0.0, gg2 = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                y[i] = cq[sub + i * GS];
                if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
            }
            const double bb = nrm[qc];
            gg = group_sum<GS>(gg + gg2);
            const double g2 = gg * gg, ab = a * bb;
            if (g2 > skip2 * ab) {
                const double dd = bb - a;
                double tt, c, sn;
                jacobi_rotation(gg, g2, dd, tt, c, sn);
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    const double xn = c * x[i] - sn * y[i];
                    cq[sub + i * GS] = sn * x[i] + c * y[i];
                    x[i] = xn;
                }
                if (sub == 0) {
                    nrm[qc] = bb + tt * gg;
                    if (g2 > exit2 * ab) lds_or(flag, 1);
                }
                a -= tt * gg;
            }
            wave_sync_lds();  // the Y columns just written are read by other lane groups in the next step
        }
#pragma unroll
        for (int i = 0; i < RPL; ++i) cp[sub + i * GS] = x[i];
        if (sub == 0) nrm[pc] = a;
        wave_sync_lds();
    }
}

// Wavefronts that rotate block pairs for an N-column item: 64 / GS = 8 column pairs per wavefront and step, so one
// wavefront per 16 columns keeps every lane group busy (NB = 2 JW column blocks of m = ceil(N / NB) <= 8 columns).  With
// the fixed four wavefronts of the first versions a 42 ... 48-column item (60 % of the headline batch) left a quarter
// of every wavefront rotating the padding column.
SMRT_HD int jacobi_waves(int N, int jw_max, int gs = SMRT_JACOBI_GS) {
#ifdef SMRT_JACOBI_ALL_WAVES   // ablation build (tools/build_variant.py): the fixed wavefront count of the first versions
    return jw_max;
#endif
    const int w = (N + 2 * (SMRT_LANES / gs) - 1) / (2 * (SMRT_LANES / gs));
    return w < 1 ? 1 : (w > jw_max ? jw_max : w);
}

// Two-level ordering as jacobi_onesided, on a zero-padded LDS matrix (rows < RPL*GS <= LD, columns < NB*m).
// skip2 / exit2: squared-cosine thresholds below which a rotation is skipped / does not count against convergence.
// Passive brightness temperatures (1e-6 K of ~250 K) tolerate 1e-26 / 1e-15; the backscatter is a small difference
// of intensities (coherent part subtracted, azimuth modes cancelling in cross-pol), so active mode uses 1e-30 / 1e-22
// (5e-10 -> 2e-11 relative error on the fixtures, about a third of a sweep more).
template <int NT, int GS, int RPL>
SMRT_DEV bool jacobi_padded(double* Bm, int N, int LD, double* sigma, double* nrm, int* flag, int JW,
                            double skip2 = SMRT_JACOBI_SKIP_COS2, double exit2 = SMRT_JACOBI_EXIT_COS2) {
    const int t = tid();
    const int lane = t & (SMRT_LANES - 1), wave = t / SMRT_LANES;
    const int NB = 2 * JW;
    constexpr int SLOTS = SMRT_LANES / GS;
    const int slot = lane / GS, sub = lane % GS;
    const int m = (N + NB - 1) / NB;
    const int CP = NB * m;
    const int me = m + (m & 1);
    (void)me;
    constexpr int NG = NT / GS;
    const int grp = t / GS;
    bool converged = false;
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        // refresh the tracked squared column norms (also the first computation)
        for (int c0 = 0; c0 < CP; c0 += NG) {  // uniform trip count (the group sum is a wavefront operation)
            const int c = c0 + grp;
            const int cc = c < CP ? c : CP;
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) { const double xx = Bm[cc * LD + sub + i * GS]; a += xx * xx; }
            a = group_sum<GS>(a);
            if (sub == 0 && c < CP) nrm[c] = a;
        }
        if (t == 0) { *flag = 0; nrm[CP] = 0.0; }
        block_sync();
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
#ifdef SMRT_JACOBI_INTRA_ROUND_ROBIN   // ablation build: the round-robin of the first versions (both columns through LDS)
                const int half = me / 2;
                for (int u = 0; u < me - 1; ++u) {
                    for (int ps0 = 0; ps0 < 2 * half; ps0 += SLOTS) {
                        const int ps = ps0 + slot;
                        const int base = (ps < half) ? i0 : j0;
                        const int k = (ps < half) ? ps : ps - half;
                        int a, b;
                        if (k == 0) { a = me - 1; b = u; }
                        else {
                            a = u + k; if (a >= me - 1) a -= me - 1;
                            b = u - k; if (b < 0) b += me - 1;
                        }
                        const bool valid = (ps < 2 * half) && (a < m) && (b < m);
                        // an idle slot rotates the (all-zero) last padding column with itself: a no-op
                        rotate_pair_padded<GS, RPL>(Bm, LD, valid ? base + a : CP, valid ? base + b : CP, sub, nrm, flag, skip2, exit2);
                    }
                    wave_sync_lds();
                }
#else
                // pairs INSIDE the two blocks of this wavefront, by recursive halving: the columns of a block split into
                // sub-blocks of 2h, whose halves are rotated against each other like two blocks (h inner steps, the column
                // of the first half resident in registers), for h = mp / 2, mp / 4, ... 1 -- every pair of the block once,
                // mp - 1 steps like the round-robin it replaces, but one column through LDS per rotation instead of two
                int mp = 1;
                while (mp < m) mp <<= 1;
                for (int h = mp >> 1; h >= 1; h >>= 1) {
                    rotate_resident<GS, RPL>(Bm, LD, CP, mp, h, slot, sub, nrm, flag, skip2, exit2,
                        [&](int ps) { const int q = ps & ((mp >> 1) - 1); const int x = (q / h) * 2 * h + (q % h);
                                      return x < m ? ((ps < (mp >> 1)) ? i0 : j0) + x : CP; },
                        [&](int ps, int j) { const int q = ps & ((mp >> 1) - 1); int a = (q % h) + j; if (a >= h) a -= h;
                                             const int y = (q / h) * 2 * h + h + a;
                                             return y < m ? ((ps < (mp >> 1)) ? i0 : j0) + y : CP; });
                }
#endif
            }
            // cross pairs (I_a, J_(a+j)): the lane group of slot a keeps column I_a (and its tracked norm) in
            // registers for all m inner steps -- loaded once, stored once -- only the J column moves through LDS
            rotate_resident<GS, RPL>(Bm, LD, CP, m, m, slot, sub, nrm, flag, skip2, exit2,
                [&](int ps) { return i0 + ps; },
                [&](int ps, int j) { int bq = ps + j; if (bq >= m) bq -= m; return j0 + bq; });
            block_sync();
        }
        converged = (*flag == 0);
        block_sync();  // everyone has read the flag before it is cleared again
    }
    for (int c0 = 0; c0 < N; c0 += NG) {
        const int c = c0 + grp;
        const int cc = c < N ? c : N - 1;
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) { const double xx = Bm[cc * LD + sub + i * GS]; a += xx * xx; }
        a = group_sum<GS>(a);
        if (sub == 0 && c < N) sigma[c] = a * fast_rsqrt(a);
    }
    block_sync();
    return converged;
}

// ---- Jacobi kernel of the split pipeline: one (pair, layer) item per workgroup, ONE matrix in LDS, so that four
// workgroups share a CU and hide each other's dependency latency (the rotation sequence of one matrix is strictly
// sequential: N-1 steps per sweep).
// LDJ: leading dimension of the LDS matrix inside the Jacobi kernel, == 8 (mod 32) eight-byte slots: the column
// pairs a 32-lane group rotates together are ADJACENT columns (8 lanes each), so consecutive columns must start 8
// bank slots apart to be conflict-free (with the odd LD of the other kernels they overlapped: 41 % of the LDS cycles
// of this kernel were bank conflicts, profiles/r1f_pmc_counters.txt).
struct JacobiPlan { int NMAX, LD, LDJ, NCOL, o_sigma, o_rsig, o_int, total; };
// Padded row count of an N-column item = RPL * GS of the instantiation dort_jacobi_item dispatches it to.
SMRT_HD int jacobi_padded_rows(int N) {
    return N > 112 ? 128 : N > 96 ? 112 : N > 80 ? 96 : N > 64 ? 80 : N > 56 ? 64 : N > 48 ? 56 : N > 40 ? 48
         : N > 32 ? 40 : N > 16 ? 32 : N > 8 ? 16 : 8;
}
// cap > 0: the LDS layout of a SIZE CLASS -- items of at most `cap` columns (the staged matrices in global memory keep
// the layout of the batch maximum) -- so that the small items of a batch do not pay the LDS of its largest one.
// gs = 16 (64 < N <= 128 only): the layout of the sixteen-wavefront kernel -- sixteen lanes per column pair, four pairs per
// wavefront, so that one 128-column matrix (64 disjoint pairs per step, the whole LDS of a CU) keeps 1024 threads busy
// instead of 512: the leading dimension is == 16 (mod 32) (the two column pairs of a 32-lane group are adjacent columns).
SMRT_HD JacobiPlan make_jacobi_plan(int n_max_stream, int P, int cap = 0, int gs = SMRT_JACOBI_GS) {
    JacobiPlan p;
    p.NMAX = n_max_stream * P;
    p.LD = (p.NMAX + 1) | 1;                            // layout of the staged matrices in global memory (make_plan)
    const int nmax = (cap > 0 && cap < p.NMAX) ? cap : p.NMAX;   // columns of the largest item held in this layout
    // leading dimension: the first value >= the padded rows that is == 8 or == 24 (mod 32) -- four adjacent columns
    // then start at bank slots {0, 8, 16, 24} in one order or the other
    int ldj = (gs == 16) ? ((nmax + 15) / 16) * 16 : jacobi_padded_rows(nmax);
    while ((ldj & 31) != gs && (ldj & 31) != 32 - gs) ++ldj;
    p.LDJ = ldj;
    // NB * ceil(N / NB) <= this - 1 for every N <= nmax and its NB = 2 * jacobi_waves(N) column blocks (the padded column
    // count grows with N inside a wavefront count and reaches 16 JW at its upper end, so nmax decides), for workgroups of
    // four and of eight wavefronts (k_jacobi.hip launches either on 64 < N <= 128); plus the idle-slot column
    int cpmax = 0;
    for (int jw_max = (gs == 16 ? 16 : 4); jw_max <= (gs == 16 ? 16 : 8); jw_max += 4) {
        const int nb = 2 * jacobi_waves(nmax, jw_max, gs);
        const int cp = nb * ((nmax + nb - 1) / nb);
        if (cp > cpmax) cpmax = cp;
    }
    p.NCOL = cpmax + 1;
    int o = p.NCOL * p.LDJ;
    p.o_sigma = o; o += nmax + 16;
    p.o_rsig = o; o += nmax + 16; // tracked column norms (padded columns included)
    p.o_int = o; o += 4;
    p.total = o;
    return p;
}

// Size classes of the N <= 64 pipelines (k_jacobi.hip launches one kernel per class over all items; an item outside the
// class leaves at once): a kernel that only holds the instantiations of ITS row counts needs 68-80 registers instead of
// the 114 of the one that holds them all, and the LDS of its own largest item -- 7 / 6 / 4 workgroups per CU for the
// items of <= 48 / <= 56 / <= 64 columns instead of 4 for all (62 % / 26 % / 12 % of the headline batch).
struct JacobiClass { int nt, lo, hi; };   // threads per workgroup, items with lo < N <= hi
// one wavefront per 2 * (64 / GS) columns of the largest item of the class (jacobi_waves)
constexpr int jacobi_class_nt(int hi) {
    return SMRT_LANES * ((hi + 2 * (SMRT_LANES / SMRT_JACOBI_GS) - 1) / (2 * (SMRT_LANES / SMRT_JACOBI_GS)));
}
SMRT_HD int jacobi_classes(int NMAX, JacobiClass* out) {
    const JacobiClass all[4] = {{jacobi_class_nt(32), 0, 32}, {jacobi_class_nt(48), 32, 48}, {jacobi_class_nt(56), 48, 56},
                                {jacobi_class_nt(64), 56, 64}};
    int n = 0;
    for (int i = 0; i < 4; ++i)
        if (NMAX > all[i].lo) out[n++] = all[i];
    return n;
}

template <int NT, int RPL, int GS = SMRT_JACOBI_GS>
SMRT_DEV void dort_jacobi_item_impl(const DevBatch& b, const DevStage& stg, long long item, double* lds, int cap = 0) {
    constexpr int JWMAX = (GS == 16) ? NT / SMRT_LANES : (NT / SMRT_LANES >= 8) ? 8 : NT / SMRT_LANES;   // wavefronts rotating block pairs: at most 4, or 8 (N > 64); 16 with sixteen lanes per pair
    const int t = tid();
    const int nmodes = (b.mode == 1) ? b.m_max + 1 : 1;   // active: items are (pair, azimuth mode, layer)
    const long long p = item / ((long long)b.Lmax * nmodes);
    const int l = (int)(item % b.Lmax);
    const long long gp = global_pair(b, p);
    const int si = (int)(gp % b.S);
    if (l >= b.n_layers[si]) return;          // uniform
    if (b.status[p] != ST_OK) return;         // the prep kernel flagged this pair (uniform)
    const JacobiPlan plan = make_jacobi_plan(b.n_max_stream, b.mode == 1 ? 3 : 2, cap, GS);
    const int LD = plan.LD, LDJ = plan.LDJ;
    const int N = stg.n[item];
    if (N <= 0) return;                       // the prep kernel flagged this layer (uniform)
    double* M = lds;
    double* sigma = lds + plan.o_sigma;
    double* nrm = lds + plan.o_rsig;
    int* ints = (int*)(lds + plan.o_int);
    double* gB = stg.B + item * stg.mat_stride;
    // load B and zero the padding: rows N..RPL*GS-1 of every used column, columns N..CP (CP = NB*m, plus the idle
    // slot column CP itself)
    const int JW = jacobi_waves(N, JWMAX, GS);
    const int NB = 2 * JW;
    const int m = (N + NB - 1) / NB;
    const int CP = NB * m;
    for_2d<NT>(RPL * GS, CP + 1, [&](int r, int c) { M[c * LDJ + r] = (r < N && c < N) ? gB[c * LD + r] : 0.0; });
    if (t == 0) ints[0] = 0;
    block_sync();
    const bool ok = jacobi_padded<NT, GS, RPL>(M, N, LDJ, sigma, nrm, &ints[0], JW, b.jacobi_skip2, b.jacobi_exit2);
    if (!ok) { if (t == 0) stg.n[item] = -ST_EIGEN; return; }   // per layer, like the prep kernel's failures
#ifdef SMRT_GJ_FAST_PANEL
    // eigenpairs out in ascending order of the singular value -- the order of the streams in the no-scattering limit,
    // where column c of the recursion matrices then belongs to row c: what lets the Gauss-Jordan solves of the finish
    // kernel take their pivots from the diagonal blocks (gj_panel16_fast).  Rank by counting; the dead norm buffer holds
    // the permutation.
    int* src = (int*)nrm;
    for (int r = t; r < N; r += NT) {
        const double sg = sigma[r];
        int rank = 0;
        for (int j = 0; j < N; ++j) { const double sj = sigma[j]; rank += (sj < sg || (sj == sg && j < r)) ? 1 : 0; }
        src[rank] = r;
    }
    block_sync();
    for_2d<NT>(N, N, [&](int r, int c) { gB[c * LD + r] = M[src[c] * LDJ + r]; });
    for (int r = t; r < N; r += NT) stg.sigma[item * stg.vec_stride + r] = sigma[src[r]];
#else
    for_2d<NT>(N, N, [&](int r, int c) { gB[c * LD + r] = M[c * LDJ + r]; });
    for (int r = t; r < N; r += NT) stg.sigma[item * stg.vec_stride + r] = sigma[r];
#endif
}

// Rows per lane follow the item's OWN size N = stg.n[item] (streams x polarisations of that layer: total reflection
// removes streams, so N varies from layer to layer), not the batch maximum: the padded rows RPL * GS are the first
// multiple of GS >= N (a few sizes are merged to bound the number of instantiations).  The LDS layout (LDJ) is the
// one of the batch maximum, so every variant fits.
template <int NT, int LO = 0, int HI = 128>
SMRT_DEV void dort_jacobi_item(const DevBatch& b, const DevStage& stg, long long item, double* lds) {
    constexpr int G = SMRT_JACOBI_GS;   // lanes per column pair
    const int rows = stg.n[item];       // <= 0: nothing to do
    if (rows <= LO || rows > HI) return;   // another size class's item (uniform); HI = 128, LO = 0: every item
    constexpr int cap = HI < 128 ? HI : 0;
    // (PREV, R]: the row counts of one instantiation; only those that meet (LO, HI] exist in this kernel
#define SMRT_JACOBI_ROWS(PREV, R) \
    if constexpr (LO < (R) && (PREV) < HI) if (rows <= (R)) { dort_jacobi_item_impl<NT, (R) / G>(b, stg, item, lds, cap); return; }
    SMRT_JACOBI_ROWS(0, 8) SMRT_JACOBI_ROWS(8, 16) SMRT_JACOBI_ROWS(16, 32) SMRT_JACOBI_ROWS(32, 40)
    SMRT_JACOBI_ROWS(40, 48) SMRT_JACOBI_ROWS(48, 56) SMRT_JACOBI_ROWS(56, 64) SMRT_JACOBI_ROWS(64, 80)
    SMRT_JACOBI_ROWS(80, 96) SMRT_JACOBI_ROWS(96, 112) SMRT_JACOBI_ROWS(112, 128)
#undef SMRT_JACOBI_ROWS
}

// The same with SIXTEEN lanes per column pair (64 < N <= 128, 1024 threads): rows per lane from the item's own size in steps of
// sixteen rows.
template <int NT>
SMRT_DEV void dort_jacobi_item16(const DevBatch& b, const DevStage& stg, long long item, double* lds) {
    const int rows = stg.n[item];       // <= 0: nothing to do
    if (rows <= 0 || rows > 128) return;   // (> 128: a layer the Rayleigh kernel diagonalises, dort_layout.hpp: kStageDirect)
    if (rows <= 16) { dort_jacobi_item_impl<NT, 1, 16>(b, stg, item, lds); return; }
    if (rows <= 32) { dort_jacobi_item_impl<NT, 2, 16>(b, stg, item, lds); return; }
    if (rows <= 48) { dort_jacobi_item_impl<NT, 3, 16>(b, stg, item, lds); return; }
    if (rows <= 64) { dort_jacobi_item_impl<NT, 4, 16>(b, stg, item, lds); return; }
    if (rows <= 80) { dort_jacobi_item_impl<NT, 5, 16>(b, stg, item, lds); return; }
    if (rows <= 96) { dort_jacobi_item_impl<NT, 6, 16>(b, stg, item, lds); return; }
    if (rows <= 112) { dort_jacobi_item_impl<NT, 7, 16>(b, stg, item, lds); return; }
    dort_jacobi_item_impl<NT, 8, 16>(b, stg, item, lds);
}

}  // namespace smrt
