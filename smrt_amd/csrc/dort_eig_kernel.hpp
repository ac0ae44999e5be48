// The symmetric eigensolver of the N <= 64 pipelines: an alternative to the one-sided Jacobi kernel between prep and finish.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
//
// What the finish kernels need from the diagonalisation stage (reference: smrt/rtsolver/dort.py:821-962) is B' = U Sigma and
// Sigma, where B = L+^T L- = U Sigma V^T.  The Jacobi kernel gets them by orthogonalising the columns of B (5.3 sweeps of
// N (N - 1) / 2 rotations with an N-long dot product each: a 25 N^3-class iteration).  Here they come from the symmetric
// eigenproblem S = B B^T = U Sigma^2 U^T:
//
//   gram      one WAVEFRONT per item: S = B B^T on the matrix core, in place
//   tridiag   one wavefront per item, lane r = row r of S with the whole row in registers: N - 2 Householder steps
//   chase     one LANE per item: implicit QL with Wilkinson shifts on the tridiagonal form only -- a strictly sequential
//             scalar chain, so 64 items share a wavefront -- that records every plane rotation (c, s) in a list
//   vectors   one wavefront per item, lane r = row r of Z with the whole row in registers: Z = Q (the reflectors applied to
//             the identity), then the rotation list -- streamed through an LDS ring well ahead of its use, sixteen records
//             at a time into registers, each (c, s) broadcast by one 64-bit DPP move per value -- then B' = Z Sigma
//
// ~0.85 N^2 rotations of 6 N flops + the reduction: an 8 N^3-class algorithm with U orthogonal to rounding by construction
// (what the pivot-free layer recursion relies on: A+^T A- = -Sigma).  What is given up is the high RELATIVE accuracy of the
// small singular values (S squares the condition number; the reference works on the squared problem too, dort.py:926-944):
// tests/studies/symmetric_eigen_route.py -- <= 2e-9 K against the oracle on the hard-input sweep, like the SVD itself.
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_layout.hpp"

namespace smrt {

// doubles of rotation-list space per item.  The list is a sequence of 16-byte slots: per QL iteration one header slot and
// one chunk of sixteen record slots per sixteen columns the iteration touches (the record (c, s) of column I sits at slot
// I mod 16 of its chunk, so that the consumer finds it at a STATIC lane of a register); ~0.85 N^2 rotations in ~1.6 N
// iterations on average.  An item that needs more fails like a Jacobi iteration that does not converge in 40 sweeps.
SMRT_HD long long eig_rot_doubles(int NMAX) { return 2LL * (3LL * NMAX * NMAX + 32LL * NMAX + 320); }
constexpr int kEigRingSlots = 512;   // LDS ring of the vectors kernel: 8 KB per wavefront

#if defined(SMRT_HOST_EMU)
SMRT_DEV double uload(const double* p) { return *p; }
SMRT_DEV int uload(const int* p) { return *p; }
#else
// wavefront-uniform load through the scalar data cache (s_load): the address must be the same in every lane, and the data
// must have been written by an EARLIER kernel (the scalar cache is not coherent with this kernel's own vector stores)
SMRT_DEV double uload(const double* p) {
    return *(const __attribute__((address_space(4))) double*)(unsigned long long)p;
}
SMRT_DEV int uload(const int* p) {
    return *(const __attribute__((address_space(4))) int*)(unsigned long long)p;
}
#endif

// a value the compiler cannot prove wavefront-uniform, pinned into scalar registers
#if defined(SMRT_HOST_EMU)
SMRT_DEV int uniform(int v) { return v; }
SMRT_DEV long long uniform(long long v) { return v; }
#else
SMRT_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
SMRT_DEV long long uniform(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffLL));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
#endif

// Rows of a staging item if this launch has staged it, else 0: the counts of layers beyond a snowpack's own (ragged
// batches) and of pairs the prep kernel refused are leftovers of earlier launches (like dort_jacobi_item_impl's guards).
SMRT_DEV int eig_item_rows(const DevBatch& b, const DevStage& stg, long long item) {
    const int nmodes = (b.mode == 1) ? b.m_max + 1 : 1;
    const long long p = item / ((long long)b.Lmax * nmodes);
    const int l = (int)(item % b.Lmax);
    const int si = (int)(global_pair(b, p) % b.S);
    if (l >= b.n_layers[si] || b.status[p] != ST_OK) return 0;
    const int n = stg.n[item];
    return (n > 0 && n <= stg.vec_stride) ? n : 0;
}

// sum over the 64 lanes of a wavefront, result in every lane: DPP inside the 16-lane rows, four readlanes across
SMRT_DEV double wave_sum64(double v) {
    v = group_sum<16>(v);
    return (wave_bcast(v, 0) + wave_bcast(v, 16)) + (wave_bcast(v, 32) + wave_bcast(v, 48));
}

// a[k] of a register array for a wavefront-uniform k: a tree of uniform branches, not a chain of selects
#define SMRT_PICK8(B) \
    case (B): if constexpr ((B) < NP) x = a[(B) < NP ? (B) : 0]; break; \
    case (B) + 1: if constexpr ((B) + 1 < NP) x = a[(B) + 1 < NP ? (B) + 1 : 0]; break; \
    case (B) + 2: if constexpr ((B) + 2 < NP) x = a[(B) + 2 < NP ? (B) + 2 : 0]; break; \
    case (B) + 3: if constexpr ((B) + 3 < NP) x = a[(B) + 3 < NP ? (B) + 3 : 0]; break; \
    case (B) + 4: if constexpr ((B) + 4 < NP) x = a[(B) + 4 < NP ? (B) + 4 : 0]; break; \
    case (B) + 5: if constexpr ((B) + 5 < NP) x = a[(B) + 5 < NP ? (B) + 5 : 0]; break; \
    case (B) + 6: if constexpr ((B) + 6 < NP) x = a[(B) + 6 < NP ? (B) + 6 : 0]; break; \
    case (B) + 7: if constexpr ((B) + 7 < NP) x = a[(B) + 7 < NP ? (B) + 7 : 0]; break;
template <int NP>
SMRT_DEV double eig_pick(const double (&a)[NP], int k) {
    double x = 0.0;
    switch (k) {
        SMRT_PICK8(0) SMRT_PICK8(8) SMRT_PICK8(16) SMRT_PICK8(24) SMRT_PICK8(32) SMRT_PICK8(40) SMRT_PICK8(48) SMRT_PICK8(56)
        default: break;
    }
    return x;
}
#undef SMRT_PICK8

// ---- gram: S = B B^T in place of B -------------------------------------------------------------------------------
// One wavefront per item, NP: padded row count of the instantiation (a multiple of 16, >= N).  The tiles of the lower
// triangle on the matrix core with the operands straight from global memory (lane (lg, lr) of the k-slab kk of tile row ti
// is B[16 ti + lr][4 kk + lg]: sixteen consecutive doubles of four columns -- no LDS, so the short kernel runs at the
// occupancy of its registers), both triangles written back.  (Its own kernel: the accumulation registers of this step would
// otherwise cap the occupancy of the tridiagonalisation.)
template <int NP>
SMRT_DEV void eig_gram_item(const DevStage& stg, long long item) {
    constexpr int T16 = (NP + 15) / 16;
    const int lane = tid() & (SMRT_LANES - 1);
    const int N = uniform(stg.n[item]);
    const int vs = stg.vec_stride;
    const int LD = (vs + 1) | 1;
    double* gB = stg.B + item * stg.mat_stride;
    const int kend = (N + 3) & ~3;
    tile4 acc[T16 * (T16 + 1) / 2];
#pragma unroll
    for (int i = 0; i < T16 * (T16 + 1) / 2; ++i) acc[i] = tile_zero();
    const int lr = lane & 15, lg = lane >> 4;
    for (int kk = 0; 4 * kk < kend; ++kk) {
        double a[T16];
        const int col = 4 * kk + lg;
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) {
            const int row = 16 * ti + lr;
            a[ti] = (row < N && col < N) ? gB[col * LD + row] : 0.0;
        }
        int idx = 0;
#pragma unroll
        for (int ti = 0; ti < T16; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) mfma_tile(a[ti], a[tj], acc[idx++]);
    }
    wave_sync();   // every lane has read B before S takes its place
    // tile (ti, tj): register reg of lane (lg, lr) is S[16 ti + lg + 4 reg][16 tj + lr]
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < T16; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = 16 * ti + lg + 4 * reg, col = 16 * tj + lr;
                const double v = acc[idx][reg];
                if (row < N && col < N) {
                    gB[col * LD + row] = v;
                    if (ti != tj) gB[row * LD + col] = v;
                }
            }
            ++idx;
        }
}

// ---- tridiag: S scaled by a power of two and reduced to tridiagonal form ------------------------------------------
// One wavefront per item, lane r = row r of S with the whole row in registers.  In: stg.B[item] = S (gram).  Out:
// stg.sigma[item] = diagonal d, stg.eig_e[item] = [ off-diagonal e (e[i] couples i, i + 1), then at index N - 1 the scale |
// tau ], and the Householder vectors over S: column k holds v (rows < k).  The elimination runs from the LAST row
// upwards (EISPACK tred2's direction), so that the small end of a graded matrix is where the QL iteration starts.
// Uniform operands (element j of the lane vectors v, w) come from registers that hold a 16-lane row of the vector in
// every row (rows_bcast: two lane-row swaps per half) by ONE 64-bit DPP move each (row_newbcast), not by two v_readlane.
template <int NP, int Q>
SMRT_DEV void eig_tri_matvec(const double (&a)[NP], const double (&vr)[(NP + 15) / 16], int k, double& p, double& p2) {
    if (16 * Q < k) {   // uniform; v_j = 0 from j = k on
#define SMRT_TRI_MV(J, ACC) if constexpr (16 * Q + (J) < NP) ACC = fma(a[16 * Q + (J) < NP ? 16 * Q + (J) : 0], row_bcast16<(J)>(vr[Q]), ACC);
        SMRT_TRI_MV(0, p) SMRT_TRI_MV(1, p2) SMRT_TRI_MV(2, p) SMRT_TRI_MV(3, p2) SMRT_TRI_MV(4, p) SMRT_TRI_MV(5, p2) SMRT_TRI_MV(6, p) SMRT_TRI_MV(7, p2)
        if (16 * Q + 8 < k) {
            SMRT_TRI_MV(8, p) SMRT_TRI_MV(9, p2) SMRT_TRI_MV(10, p) SMRT_TRI_MV(11, p2) SMRT_TRI_MV(12, p) SMRT_TRI_MV(13, p2) SMRT_TRI_MV(14, p) SMRT_TRI_MV(15, p2)
        }
#undef SMRT_TRI_MV
    }
    if constexpr (Q > 0) eig_tri_matvec<NP, Q - 1>(a, vr, k, p, p2);
}
template <int NP, int Q>
SMRT_DEV void eig_tri_update(double (&a)[NP], const double (&vr)[(NP + 15) / 16], const double (&wr)[(NP + 15) / 16], int k, double v, double w) {
    if (16 * Q < k) {
#define SMRT_TRI_UP(J) if constexpr (16 * Q + (J) < NP) a[16 * Q + (J) < NP ? 16 * Q + (J) : 0] = \
        fma(-v, row_bcast16<(J)>(wr[Q]), fma(-w, row_bcast16<(J)>(vr[Q]), a[16 * Q + (J) < NP ? 16 * Q + (J) : 0]));
        SMRT_TRI_UP(0) SMRT_TRI_UP(1) SMRT_TRI_UP(2) SMRT_TRI_UP(3) SMRT_TRI_UP(4) SMRT_TRI_UP(5) SMRT_TRI_UP(6) SMRT_TRI_UP(7)
        if (16 * Q + 8 < k) {
            SMRT_TRI_UP(8) SMRT_TRI_UP(9) SMRT_TRI_UP(10) SMRT_TRI_UP(11) SMRT_TRI_UP(12) SMRT_TRI_UP(13) SMRT_TRI_UP(14) SMRT_TRI_UP(15)
        }
#undef SMRT_TRI_UP
    }
    if constexpr (Q > 0) eig_tri_update<NP, Q - 1>(a, vr, wr, k, v, w);
}
// the 16-lane row Q of a lane vector in every row
template <int NQ, int Q>
SMRT_DEV void eig_rows(double x, double (&xr)[NQ]) {
    xr[Q] = rows_bcast<Q>(x);
    if constexpr (Q > 0) eig_rows<NQ, Q - 1>(x, xr);
}

template <int NP>
SMRT_DEV void eig_tridiag_item(const DevStage& stg, long long item) {
    constexpr int NQ = (NP + 15) / 16;
    const int lane = tid() & (SMRT_LANES - 1);
    const int N = uniform(stg.n[item]);
    const int vs = stg.vec_stride;
    const int LD = (vs + 1) | 1;
    double* gB = stg.B + item * stg.mat_stride;
    // scale by a power of two (exact) so that the largest diagonal element is in [1, 2): the QL chain never meets an
    // overflow or an underflow of its squares whatever the units of the extinction
    unsigned long long dbits = 0;
    {
        const double dg = (lane < N) ? gB[lane * LD + lane] : 0.0;
        if (dg > 0.0) memcpy(&dbits, &dg, 8);
    }
    dbits = wave_max_u64(dbits);
    const unsigned long long ex = (dbits >> 52) & 0x7ffULL;
    if (ex == 0 || ex >= 2046) {   // S has no positive diagonal (or not a finite one): not a product B B^T of a regular B
        if (lane == 0) stg.n[item] = -ST_EIGEN;
        return;
    }
    double scale;
    {
        const unsigned long long sb = (2046ULL - ex) << 52;
        memcpy(&scale, &sb, 8);
    }
    double a[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) a[c] = (lane < N && c < N) ? gB[c * LD + lane] * scale : 0.0;
    wave_sync();   // (the reflectors go where S was read from)
    // Householder steps k = N - 1 ... 2: row k is reduced to (.., 0, alpha, d_k) by H = I - tau v v^T on the leading k x k block
    double dd = 0.0, ee = 0.0;   // lane j: d_j, e_j
    double* ge = stg.eig_e + item * 2 * vs;
    for (int k = N - 1; k >= 2; --k) {
        const double xall = eig_pick<NP>(a, k);   // lane j: S[j][k]
        if (lane == k) dd = xall;
        const double x = lane < k ? xall : 0.0;
        const double xk1 = wave_bcast(x, k - 1);
        const double sig0 = wave_sum64(lane < k - 1 ? x * x : 0.0);
        double tau = 0.0, alpha = xk1;
        double v = x;
        if (uniform((int)(sig0 > 0.0))) {
            const double sig = sig0 + xk1 * xk1;
            const double nrm = sig * fast_rsqrt(sig);
            alpha = xk1 >= 0.0 ? -nrm : nrm;
            tau = fast_rcp(sig + fabs(xk1) * nrm);   // 2 / v^T v
            if (lane == k - 1) v = xk1 - alpha;
            double vr[NQ], wr[NQ];
            eig_rows<NQ, NQ - 1>(v, vr);
            // p = tau S v on the leading block (lanes >= k hold other rows: masked), w = p - (tau / 2)(p^T v) v
            double p = 0.0, p2 = 0.0;
            eig_tri_matvec<NP, NQ - 1>(a, vr, k, p, p2);
            p = lane < k ? tau * (p + p2) : 0.0;
            const double kk2 = 0.5 * tau * wave_sum64(p * v);
            const double w = p - kk2 * v;
            eig_rows<NQ, NQ - 1>(w, wr);
            // S <- S - v w^T - w v^T (rows / columns >= k have v = w = 0)
            eig_tri_update<NP, NQ - 1>(a, vr, wr, k, v, w);
        }
        if (lane == k - 1) ee = alpha;
        // the reflector for the back-transformation: rows < k of column k
        if (lane < k) gB[k * LD + lane] = v;
        if (lane == k) ge[vs + k] = tau;
    }
    if (lane == 0) { dd = a[0]; if (N >= 2) ee = a[1]; }
    if (lane == 1 && N >= 2) dd = a[1];
    if (lane < N) stg.sigma[item * vs + lane] = dd;
    if (lane < N - 1) ge[lane] = ee;
    if (lane == N - 1) ge[lane] = scale;
}

// ---- chase: implicit QL on the tridiagonal form, one lane per item ------------------------------------------------
// dl / el: this lane's d and e in LDS, element i at [kEigChaseLanes i] (conflict-free whatever index each lane is at).  (Measured
// with d and e left in global memory instead -- every wavefront of a launch resident at once, loads a step ahead: 11.2
// against 3.4 ms per 102 400 items; 64 lanes at 64 different indices are 64 cache lines per access.)
// In: d, e, scale (tridiag).  Out: stg.sigma[item] = singular values sqrt(lambda / scale), and the rotation list
// stg.eig_rot[item] in 16-byte slots: per QL iteration a header slot (top group gt, bottom group gb of four columns) and
// the chunks ct = gt / 4 ... cb = gb / 4 of sixteen record slots each, highest chunk first; the record (c, s) of column I is
// slot I mod 16 of chunk I / 16.  The columns of the groups gt ... gb outside the block hold identity records (the
// consumer works in whole groups); other slots of a chunk are never read.  A header with a negative gt ends the list.
// The algorithm is EISPACK tql2 / LAPACK dsteqr's QL branch; the negligibility tests of an iteration are made while its
// chase runs (every e[j] of the block is rewritten by it), so that no separate scan is needed afterwards.
// (Measured with sixteen items per wavefront instead of 64, so that its 16 KB of LDS fit beside three workgroups of the
// prep or finish kernel of ANOTHER pipeline pass on a CU: 3.62 against 3.38 ms alone and 165.3 k against 169.0 k solves/s
// with three concurrent passes -- the overlap is not limited by the LDS the kernel holds.  48 / 40 / 32 items per wavefront,
// i.e. 3 / 4 / 5 wavefronts per CU instead of 2 -- a wavefront issues 68 % of its cycles and two of a CU's four SIMDs are
// idle with 2: the diagonalisation alone 14.6 -> 14.4 ms at 40, but the step with three passes 29.3 -> 29.6 ms.)
#ifndef SMRT_EIG_CHASE_LANES
#define SMRT_EIG_CHASE_LANES 64
#endif
constexpr int kEigChaseLanes = SMRT_EIG_CHASE_LANES;

SMRT_DEV void eig_chase_lane(const DevStage& stg, long long item, double* dl, double* el) {
    const int n = stg.n[item];
    if (n <= 0) return;
    const int vs = stg.vec_stride;
    double* gd = stg.sigma + item * vs;
    const double* ge = stg.eig_e + item * 2 * vs;
    for (int i = 0; i < n; ++i) { dl[kEigChaseLanes * i] = gd[i]; el[kEigChaseLanes * i] = (i < n - 1) ? ge[i] : 0.0; }
    const double scale = ge[n - 1];
    double* rp = stg.eig_rot + item * stg.rot_stride;
    double* const rend = rp + stg.rot_stride - 2;   // room for the terminating header
    const double eps = 2.220446049250313e-16;
    bool fail = false;
    int l = 0, mk = -1;      // mk: known end of the block that starts at l (e[mk] negligible or mk = n - 1), or < l: unknown
    int iters = 0;
    while (true) {
        // -- the block [l, m] of this iteration
        bool done = false;
        while (true) {
            if (l >= n) { done = true; break; }
            if (mk < l) {
                int mm = l;
                while (mm < n - 1) {
                    const double em = fabs(el[kEigChaseLanes * (mm)]);
                    if (em <= eps * (fabs(dl[kEigChaseLanes * (mm)]) + fabs(dl[kEigChaseLanes * ((mm + 1))]))) break;
                    ++mm;
                }
                mk = mm;
            }
            if (mk == l) { ++l; continue; }   // eigenvalue l stands alone
            break;
        }
        if (done) break;
        const int m = mk;
        if (++iters > 30 * n) { fail = true; break; }
        const int top = m - 1;
        const int ct = top >> 4;
        if (rp + 2 * (1 + 16 * (ct - (l >> 4) + 1)) > rend) { fail = true; break; }
        double* const blk = rp + 2;   // first record slot of the block; column I at slot 16 (ct - I / 16) + I mod 16
        // -- Wilkinson shift from the leading 2 x 2 of the block
        const double dl0 = dl[kEigChaseLanes * (l)], el0 = el[kEigChaseLanes * (l)];
        double g = (dl[kEigChaseLanes * ((l + 1))] - dl0) / (2.0 * el0);
        double r = sqrt(g * g + 1.0);
        g = dl[kEigChaseLanes * (m)] - dl0 + el0 / (g + (g >= 0.0 ? r : -r));
        double s = 1.0, c = 1.0, p = 0.0;
        double dip1 = dl[kEigChaseLanes * (m)];    // d[i + 1] as it was before this iteration
        double dfin = 0.0;           // d[i + 2] as this iteration leaves it
        int j1 = -1, j2 = -1;        // smallest / second smallest j in [l, m) whose e[j] is negligible after this iteration
        int bottom = l;
        bool underflow = false;
        double di = 0.0;
        double en = el[kEigChaseLanes * (m - 1)], dn = dl[kEigChaseLanes * (m - 1)];   // requested a step ahead of their use
        for (int i = m - 1; i >= l; --i) {
            const double ei = en;
            di = dn;
            if (i > l) { en = el[kEigChaseLanes * (i - 1)]; dn = dl[kEigChaseLanes * (i - 1)]; }
            const double f = s * ei, b = c * ei;
            const double r2 = f * f + g * g;
            if (r2 == 0.0) {   // (tql2: recover from underflow)
                dl[kEigChaseLanes * ((i + 1))] = dip1 - p;
                el[kEigChaseLanes * (m)] = 0.0;
                bottom = i + 1;
                underflow = true;
                break;
            }
            const double rinv = fast_rsqrt(r2);
            r = r2 * rinv;
            el[kEigChaseLanes * ((i + 1))] = r;
            s = f * rinv;
            c = g * rinv;
            g = dip1 - p;
            const double rr2 = (di - g) * s + 2.0 * c * b;
            p = s * rr2;
            const double dnew = g + p;
            dl[kEigChaseLanes * ((i + 1))] = dnew;
            g = c * rr2 - b;
            {
                double* rec = blk + 2 * (16 * (ct - (i >> 4)) + (i & 15));
                rec[0] = c; rec[1] = s;
            }
            if (i + 1 < m && r <= eps * (fabs(dnew) + fabs(dfin))) { j2 = j1; j1 = i + 1; }   // e[i + 1] is final now
            dfin = dnew;
            dip1 = di;
        }
        // header, and identity records for the columns of the first / last group that lie outside the block
        const int gt = top >> 2, gb = bottom >> 2;
        ((int*)rp)[0] = gt; ((int*)rp)[1] = gb;
        for (int i = 4 * gt + 3; i > top; --i) { double* rec = blk + 2 * (16 * (ct - (i >> 4)) + (i & 15)); rec[0] = 1.0; rec[1] = 0.0; }
        for (int i = bottom - 1; i >= 4 * gb; --i) { double* rec = blk + 2 * (16 * (ct - (i >> 4)) + (i & 15)); rec[0] = 1.0; rec[1] = 0.0; }
        rp = blk + 2 * 16 * (ct - (gb >> 2) + 1);
        if (underflow) { mk = -1; continue; }   // (a fresh scan decides what the next block is)
        const double dlnew = di - p;
        dl[kEigChaseLanes * (l)] = dlnew;
        el[kEigChaseLanes * (l)] = g;
        el[kEigChaseLanes * (m)] = 0.0;
        if (fabs(g) <= eps * (fabs(dlnew) + fabs(dfin))) { j2 = j1; j1 = l; }
        if (j1 == l) { ++l; mk = (j2 >= 0) ? j2 : m; }
        else if (j1 >= 0) mk = j1;
        // else: the block stands as it is
    }
    if (!fail) {
        // singular values of B: sqrt(lambda / scale)
        const double rs = 1.0 / scale;
        for (int i = 0; i < n; ++i) {
            const double lam = dl[kEigChaseLanes * (i)] * rs;
            if (!(lam > 0.0)) fail = true;
            gd[i] = sqrt(lam);
        }
    }
    if (fail) { stg.n[item] = -ST_EIGEN; return; }
    ((int*)rp)[0] = -1; ((int*)rp)[1] = 0;
}

// ---- vectors: Z = Q, the rotation list, B' = Z Sigma --------------------------------------------------------------
// Uniform operands without scalar loads (a scalar load from HBM is ~2000 cycles, the scalar-memory counter cannot be waited
// on partially, and 64 cycles of work per load leave the wavefront waiting 90 % of the time -- measured: profiles/
// r6_eig_first.txt): a register holds sixteen consecutive values, one per lane of every 16-lane row, and element K reaches
// all the lanes by ONE 64-bit DPP move (row_newbcast, K a compile-time constant).
struct EigChunk { double c, s; };   // this lane's record of a chunk of sixteen

// the group G of four columns: (c, s) from the chunk's records in registers by DPP moves (two vector instructions per
// rotation on top of its four).  Measured and switched off (SMRT_EIG_LDS_BROADCAST=1 builds it): every other group reading
// its records from the ring with ONE uniform-address ds_read_b128 each (no vector instruction, eight cycles of the CU's LDS
// pipe) -- on paper the two pipes share the broadcasts, in the kernel 6.71 against 6.29 ms per 102 400 items.
#ifndef SMRT_EIG_LDS_BROADCAST
#define SMRT_EIG_LDS_BROADCAST 0
#endif
template <int NP, int G>
SMRT_DEV void eig_group(double (&z)[NP], const EigChunk& ch, const double* ring, int chunk_slot) {
#define SMRT_EIG_ROT(I) \
    if constexpr ((I) + 1 < NP) { \
        double cc, ss; \
        if constexpr (SMRT_EIG_LDS_BROADCAST && (G & 1) == 0) { \
            const int slot = (chunk_slot + ((I) & 15)) & (kEigRingSlots - 1); \
            cc = ring[2 * slot]; ss = ring[2 * slot + 1]; \
        } else { \
            cc = row_bcast16<(I) & 15>(ch.c); ss = row_bcast16<(I) & 15>(ch.s); \
        } \
        const double t1 = ss * z[(I) + 1 < NP ? (I) : 0], t2 = ss * z[(I) + 1 < NP ? (I) + 1 : 0]; \
        z[(I) + 1 < NP ? (I) + 1 : 0] = fma(cc, z[(I) + 1 < NP ? (I) + 1 : 0], t1); \
        z[(I) + 1 < NP ? (I) : 0] = fma(cc, z[(I) + 1 < NP ? (I) : 0], -t2); \
    }
    SMRT_EIG_ROT(4 * G + 3) SMRT_EIG_ROT(4 * G + 2) SMRT_EIG_ROT(4 * G + 1) SMRT_EIG_ROT(4 * G)
#undef SMRT_EIG_ROT
}
// chunk Q (columns 16 Q ... 16 Q + 15) of an iteration whose groups are gt ... gb: its four groups, each guarded (uniform)
template <int NP, int Q>
SMRT_DEV void eig_chunk(double (&z)[NP], const double* ring, int base_slot, int ct, int gt, int gb, int l16) {
    if (Q <= ct && 4 * Q + 3 >= gb) {   // uniform: the iteration touches this chunk
        const int chunk_slot = base_slot + 16 * (ct - Q);
        const int slot = (chunk_slot + l16) & (kEigRingSlots - 1);
        EigChunk ch;
        ch.c = ring[2 * slot]; ch.s = ring[2 * slot + 1];
        if (4 * Q + 3 <= gt && 4 * Q + 3 >= gb) eig_group<NP, 4 * Q + 3>(z, ch, ring, chunk_slot);
        if (4 * Q + 2 <= gt && 4 * Q + 2 >= gb) eig_group<NP, 4 * Q + 2>(z, ch, ring, chunk_slot);
        if (4 * Q + 1 <= gt && 4 * Q + 1 >= gb) eig_group<NP, 4 * Q + 1>(z, ch, ring, chunk_slot);
        if (4 * Q <= gt && 4 * Q >= gb) eig_group<NP, 4 * Q>(z, ch, ring, chunk_slot);
    }
    if constexpr (Q > 0) eig_chunk<NP, Q - 1>(z, ring, base_slot, ct, gt, gb, l16);
}

// one reflector: t = sum_j z[j] v[j], z[j] += (-tau t) v[j], v in registers sixteen values at a time
template <int NP, int Q>
SMRT_DEV void eig_reflect_dot(const double (&z)[NP], const double (&vq)[(NP + 15) / 16], int k, double& t, double& t2) {
    if (16 * Q < k) {   // uniform
#define SMRT_EIG_DOT(J, ACC) if constexpr (16 * Q + (J) < NP) ACC = fma(z[16 * Q + (J) < NP ? 16 * Q + (J) : 0], row_bcast16<(J)>(vq[Q]), ACC);
        SMRT_EIG_DOT(0, t) SMRT_EIG_DOT(1, t2) SMRT_EIG_DOT(2, t) SMRT_EIG_DOT(3, t2) SMRT_EIG_DOT(4, t) SMRT_EIG_DOT(5, t2)
        SMRT_EIG_DOT(6, t) SMRT_EIG_DOT(7, t2)
        if (16 * Q + 8 < k) {
            SMRT_EIG_DOT(8, t) SMRT_EIG_DOT(9, t2) SMRT_EIG_DOT(10, t) SMRT_EIG_DOT(11, t2) SMRT_EIG_DOT(12, t) SMRT_EIG_DOT(13, t2)
            SMRT_EIG_DOT(14, t) SMRT_EIG_DOT(15, t2)
        }
#undef SMRT_EIG_DOT
    }
    if constexpr (Q > 0) eig_reflect_dot<NP, Q - 1>(z, vq, k, t, t2);
}
template <int NP, int Q>
SMRT_DEV void eig_reflect_axpy(double (&z)[NP], const double (&vq)[(NP + 15) / 16], int k, double t) {
    if (16 * Q < k) {
#define SMRT_EIG_AXPY(J) if constexpr (16 * Q + (J) < NP) z[16 * Q + (J) < NP ? 16 * Q + (J) : 0] = fma(t, row_bcast16<(J)>(vq[Q]), z[16 * Q + (J) < NP ? 16 * Q + (J) : 0]);
        SMRT_EIG_AXPY(0) SMRT_EIG_AXPY(1) SMRT_EIG_AXPY(2) SMRT_EIG_AXPY(3) SMRT_EIG_AXPY(4) SMRT_EIG_AXPY(5) SMRT_EIG_AXPY(6) SMRT_EIG_AXPY(7)
        if (16 * Q + 8 < k) {
            SMRT_EIG_AXPY(8) SMRT_EIG_AXPY(9) SMRT_EIG_AXPY(10) SMRT_EIG_AXPY(11) SMRT_EIG_AXPY(12) SMRT_EIG_AXPY(13) SMRT_EIG_AXPY(14) SMRT_EIG_AXPY(15)
        }
#undef SMRT_EIG_AXPY
    }
    if constexpr (Q > 0) eig_reflect_axpy<NP, Q - 1>(z, vq, k, t);
}

// lds: the ring, 2 * kEigRingSlots doubles per wavefront
template <int NP>
SMRT_DEV void eig_vectors_item(const DevStage& stg, long long item, double* ring) {
    constexpr int NQ = (NP + 15) / 16;
    const int lane = tid() & (SMRT_LANES - 1);
    const int l16 = lane & 15;
    const int N = uniform(stg.n[item]);
    if (N <= 0) return;
    const int vs = stg.vec_stride;
    const int LD = (vs + 1) | 1;
    double* gB = stg.B + item * stg.mat_stride;
    const double* gtau = stg.eig_e + item * 2 * vs + vs;
    // the list streams through the ring in pieces of 256 slots, requested a piece ahead of the one being consumed
    const double* src = stg.eig_rot + item * stg.rot_stride;
    const int cap = (int)(stg.rot_stride / 2);
    int prod = 0, cons = 0;   // slots committed to the ring / consumed
    double f[4][2];
    auto request = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int slot = prod + 64 * u + lane;
            const bool in = slot < cap;
            f[u][0] = in ? src[2 * slot] : 0.0;
            f[u][1] = in ? src[2 * slot + 1] : 0.0;
        }
    };
    auto commit = [&]() {
        wave_sync();   // every lane has read what the piece overwrites
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int slot = (prod + 64 * u + lane) & (kEigRingSlots - 1);
            ring[2 * slot] = f[u][0]; ring[2 * slot + 1] = f[u][1];
        }
        prod += 256;
        wave_sync();
    };
    request();
    const double tauv = (lane >= 2 && lane < N) ? gtau[lane] : 0.0;   // lane k: tau of reflector k
    double z[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) z[c] = (c == lane) ? 1.0 : 0.0;
    // Q = H_(N-1) ... H_2, every reflector applied from the right to the rows; v of the NEXT reflector is requested
    // while this one is applied
    double vq[NQ], vn[NQ];
    auto load_v = [&](int k, double (&v)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = 16 * q + l16;
            v[q] = (k >= 2 && j < k) ? gB[k * LD + j] : 0.0;
        }
    };
    load_v(N - 1, vq);
    for (int k = N - 1; k >= 2; --k) {
        load_v(k - 1, vn);
        const double tau = wave_bcast(tauv, k);
        if (uniform((int)(tau != 0.0))) {
            double t = 0.0, t2 = 0.0;
            eig_reflect_dot<NP, NQ - 1>(z, vq, k, t, t2);
            t = -tau * (t + t2);
            eig_reflect_axpy<NP, NQ - 1>(z, vq, k, t);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) vq[q] = vn[q];
    }
    // the rotations of the QL iterations
    commit();
    if (prod < cap) request();
    bool pending = prod < cap;
    while (true) {
        if (pending && prod - cons <= 256) {
            commit();
            pending = prod < cap;
            if (pending) request();
        }
        union { double d; int i[2]; } hd;
        hd.d = ring[2 * (cons & (kEigRingSlots - 1))];
        const int gt = uniform(hd.i[0]), gb = uniform(hd.i[1]);
        if (gt < 0) break;
        const int ct = gt >> 2;
        eig_chunk<NP, NQ - 1>(z, ring, cons + 1, ct, gt, gb, l16);
        cons += 1 + 16 * (ct - (gb >> 2) + 1);
    }
    // B' = Z Sigma over the reflectors (every lane has consumed them: program order on the GPU, a rendezvous of the fibers
    // in the emulator)
    wave_sync();
    const double sgv = (lane < N) ? stg.sigma[item * vs + lane] : 0.0;   // lane c: sigma_c
#pragma unroll
    for (int c = 0; c < NP; ++c)
        if (c < N) {   // uniform
            const double sg = wave_bcast(sgv, c);   // (static lane)
            if (lane < N) gB[c * LD + lane] = z[c] * sg;
        }
}

}  // namespace smrt
