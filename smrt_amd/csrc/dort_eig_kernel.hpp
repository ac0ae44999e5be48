// The symmetric eigensolver of the N <= 64 pipelines: an alternative to the one-sided Jacobi kernel between prep and finish.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
//
// What the finish kernels need from the diagonalisation stage (reference: smrt/rtsolver/dort.py:821-962) is B' = U Sigma and
// Sigma, where B = L+^T L- = U Sigma V^T.  The Jacobi kernel gets them by orthogonalising the columns of B (5.3 sweeps of
// N (N - 1) / 2 rotations with an N-long dot product each: a 25 N^3-class iteration).  Here they come from the symmetric
// eigenproblem S = B B^T = U Sigma^2 U^T:
//
//   tridiag   one WAVEFRONT per item, lane r = row r of S with the whole row in registers: S = B B^T on the matrix core,
//             then N - 2 Householder steps (the broadcasts of a step are v_readlane's of static lanes / registers)
//   chase     one LANE per item: implicit QL with Wilkinson shifts on the tridiagonal form only -- a strictly sequential
//             scalar chain, so 64 items share a wavefront -- that records every plane rotation (c, s) in a list
//   vectors   one wavefront per item, lane r = row r of Z with the whole row in registers: Z = Q (the reflectors applied to
//             the identity, operands through scalar loads), then the rotation list applied as 4 FMAs per rotation with
//             (c, s) in scalar registers, then B' = Z Sigma
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

// doubles of rotation-list space per item: records of two doubles -- (c, s) of a rotation, or the (top, bottom) header of
// a QL iteration -- for up to 2 N^2 rotations (0.85 N^2 on average; an item that needs more fails like a Jacobi iteration
// that does not converge in 40 sweeps)
SMRT_HD long long eig_rot_doubles(int NMAX) { return 2LL * (2LL * NMAX * NMAX + 16LL * NMAX + 16); }

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

// ---- tridiag: S = B B^T, scaled by a power of two, reduced to tridiagonal form -----------------------------------
// NP: padded row count of the instantiation (a multiple of 8, >= N).  One wavefront; lds: 16 ceil(NP / 16) x (that + 1) doubles.
// In: stg.B[item] = B.  Out: stg.sigma[item] = diagonal d, stg.eig_e[item] = [ off-diagonal e (e[i] couples i, i + 1),
// then at index N - 1 the scale | tau ], and the Householder vectors over B: column k holds v (rows < k, zero up to
// the next multiple of 8).  The elimination runs from the LAST row upwards (EISPACK tred2's direction), so that the small
// end of a graded matrix is where the QL iteration starts.
template <int NP>
constexpr int eig_tridiag_lds_doubles() { return (16 * ((NP + 15) / 16)) * (16 * ((NP + 15) / 16) + 1); }

template <int NP>
SMRT_DEV void eig_tridiag_item(const DevStage& stg, long long item, double* lds) {
    constexpr int T16 = (NP + 15) / 16;
    constexpr int R16 = 16 * T16;
    constexpr int LDL = R16 + 1;
    const int lane = tid() & (SMRT_LANES - 1);
    const int N = uniform(stg.n[item]);
    const int vs = stg.vec_stride;
    const int LD = (vs + 1) | 1;
    double* gB = stg.B + item * stg.mat_stride;
    // 1. B -> LDS, zero-padded to R16 x (multiple of 4) columns
    const int kend = (N + 3) & ~3;
    for (int c = 0; c < kend; ++c)
        for (int r = lane; r < R16; r += SMRT_LANES) lds[c * LDL + r] = (r < N && c < N) ? gB[c * LD + r] : 0.0;
    wave_sync();
    // 2. S = B B^T: the tiles of the lower triangle on the matrix core
    tile4 acc[T16 * (T16 + 1) / 2];
#pragma unroll
    for (int i = 0; i < T16 * (T16 + 1) / 2; ++i) acc[i] = tile_zero();
    const int lr = lane & 15, lg = lane >> 4;
    for (int kk = 0; 4 * kk < kend; ++kk) {
        double a[T16];
#pragma unroll
        for (int ti = 0; ti < T16; ++ti) a[ti] = lds[(4 * kk + lg) * LDL + 16 * ti + lr];
        int idx = 0;
#pragma unroll
        for (int ti = 0; ti < T16; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) mfma_tile(a[ti], a[tj], acc[idx++]);
    }
    wave_sync();   // every read of B is done before S takes its place
    {
        int idx = 0;
#pragma unroll
        for (int ti = 0; ti < T16; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * ti + lg + 4 * reg, col = 16 * tj + lr;
                    const double v = acc[idx][reg];
                    lds[col * LDL + row] = v;
                    if (ti != tj) lds[row * LDL + col] = v;
                }
                ++idx;
            }
    }
    wave_sync();
    // 3. scale by a power of two (exact) so that the largest diagonal element is in [1, 2): the QL chain never meets an
    // overflow or an underflow of its squares whatever the units of the extinction
    const int rr = lane < R16 ? lane : R16 - 1;
    unsigned long long dbits = 0;
    {
        const double dg = (lane < N) ? lds[rr * LDL + rr] : 0.0;
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
    for (int c = 0; c < NP; ++c) a[c] = (lane < N && c < N) ? lds[c * LDL + rr] * scale : 0.0;
    // 4. Householder steps k = N - 1 ... 2: row k is reduced to (.., 0, alpha, d_k) by H = I - tau v v^T on the leading k x k block
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
            // p = tau S v on the leading block (lanes >= k hold other rows: masked), w = p - (tau / 2)(p^T v) v
            double p = 0.0, p2 = 0.0;
#pragma unroll
            for (int j0 = 0; j0 < NP; j0 += 8) {
                if (j0 < k) {   // uniform; v_j = 0 from j = k on
#pragma unroll
                    for (int jj = 0; jj < 8; jj += 2) {
                        p = fma(a[j0 + jj], wave_bcast(v, j0 + jj), p);
                        p2 = fma(a[j0 + jj + 1], wave_bcast(v, j0 + jj + 1), p2);
                    }
                }
            }
            p = lane < k ? tau * (p + p2) : 0.0;
            const double kk2 = 0.5 * tau * wave_sum64(p * v);
            const double w = p - kk2 * v;
            // S <- S - v w^T - w v^T (rows / columns >= k have v = w = 0)
#pragma unroll
            for (int j0 = 0; j0 < NP; j0 += 8) {
                if (j0 < k) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = j0 + jj;
                        a[j] = fma(-v, wave_bcast(w, j), fma(-w, wave_bcast(v, j), a[j]));
                    }
                }
            }
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
// dl / el: this lane's d and e in LDS, element i at [64 i] (conflict-free whatever index each lane is at).
// In: d, e, scale (tridiag).  Out: stg.sigma[item] = singular values sqrt(lambda / scale), and the rotation list
// stg.eig_rot[item]: per QL iteration a header (top group, bottom group) followed by the records (c, s) of columns
// 4 gt + 3 ... 4 gb in descending order -- the columns of the block, padded at both ends with identity records to whole
// groups of four (the consumer works in groups of four columns); a header with a negative top group ends the list.
// The algorithm is EISPACK tql2 / LAPACK dsteqr's QL branch; the negligibility tests of an iteration are made while its
// chase runs (every e[j] of the block is rewritten by it), so that no separate scan is needed afterwards.
SMRT_DEV void eig_chase_lane(const DevStage& stg, long long item, double* dl, double* el) {
    const int n = stg.n[item];
    if (n <= 0) return;
    const int vs = stg.vec_stride;
    double* gd = stg.sigma + item * vs;
    const double* ge = stg.eig_e + item * 2 * vs;
    for (int i = 0; i < n; ++i) { dl[64 * i] = gd[i]; el[64 * i] = (i < n - 1) ? ge[i] : 0.0; }
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
                    const double em = fabs(el[64 * mm]);
                    if (em <= eps * (fabs(dl[64 * mm]) + fabs(dl[64 * (mm + 1)]))) break;
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
        if (rp + 2 * ((top | 3) - (l & ~3) + 2) > rend) { fail = true; break; }
        int* hdr = (int*)rp;
        rp += 2;
        for (int i = top | 3; i > top; --i) { rp[0] = 1.0; rp[1] = 0.0; rp += 2; }   // identity records above the block
        // -- Wilkinson shift from the leading 2 x 2 of the block
        const double dl0 = dl[64 * l], el0 = el[64 * l];
        double g = (dl[64 * (l + 1)] - dl0) / (2.0 * el0);
        double r = sqrt(g * g + 1.0);
        g = dl[64 * m] - dl0 + el0 / (g + (g >= 0.0 ? r : -r));
        double s = 1.0, c = 1.0, p = 0.0;
        double dip1 = dl[64 * m];    // d[i + 1] as it was before this iteration
        double dfin = 0.0;           // d[i + 2] as this iteration leaves it
        int j1 = -1, j2 = -1;        // smallest / second smallest j in [l, m) whose e[j] is negligible after this iteration
        int bottom = l;
        bool underflow = false;
        double di = 0.0;
        for (int i = m - 1; i >= l; --i) {
            const double ei = el[64 * i];
            di = dl[64 * i];
            const double f = s * ei, b = c * ei;
            const double r2 = f * f + g * g;
            if (r2 == 0.0) {   // (tql2: recover from underflow)
                dl[64 * (i + 1)] = dip1 - p;
                el[64 * m] = 0.0;
                bottom = i + 1;
                underflow = true;
                break;
            }
            const double rinv = fast_rsqrt(r2);
            r = r2 * rinv;
            el[64 * (i + 1)] = r;
            s = f * rinv;
            c = g * rinv;
            g = dip1 - p;
            const double rr2 = (di - g) * s + 2.0 * c * b;
            p = s * rr2;
            const double dnew = g + p;
            dl[64 * (i + 1)] = dnew;
            g = c * rr2 - b;
            rp[0] = c; rp[1] = s; rp += 2;
            if (i + 1 < m && r <= eps * (fabs(dnew) + fabs(dfin))) { j2 = j1; j1 = i + 1; }   // e[i + 1] is final now
            dfin = dnew;
            dip1 = di;
        }
        hdr[0] = top >> 2; hdr[1] = bottom >> 2;
        for (int i = bottom - 1; i >= (bottom & ~3); --i) { rp[0] = 1.0; rp[1] = 0.0; rp += 2; }   // identity records down to the group's end
        if (underflow) { mk = -1; continue; }   // (a fresh scan decides what the next block is)
        const double dlnew = di - p;
        dl[64 * l] = dlnew;
        el[64 * l] = g;
        el[64 * m] = 0.0;
        if (fabs(g) <= eps * (fabs(dlnew) + fabs(dfin))) { j2 = j1; j1 = l; }
        if (j1 == l) { ++l; mk = (j2 >= 0) ? j2 : m; }
        else if (j1 >= 0) mk = j1;
        // else: the block stands as it is
    }
    if (!fail) {
        // singular values of B: sqrt(lambda / scale)
        const double rs = 1.0 / scale;
        for (int i = 0; i < n; ++i) {
            const double lam = dl[64 * i] * rs;
            if (!(lam > 0.0)) fail = true;
            gd[i] = sqrt(lam);
        }
    }
    if (fail) { stg.n[item] = -ST_EIGEN; return; }
    ((int*)rp)[0] = -1; ((int*)rp)[1] = 0;
}

// ---- vectors: Z = Q, the rotation list, B' = Z Sigma --------------------------------------------------------------
// One plane rotation of columns I, I + 1 with record K of the current group of four.  A QL iteration covers whole groups of
// four columns (the list pads both ends with identity records), so the consumer is a fixed sequence of guarded groups --
// structured control flow with static register indices -- and the four rotations of a group are straight-line code
// whose scalar loads the scheduler batches.
template <int NP, int I>
SMRT_DEV void eig_rot(double (&z)[NP], const double* rec) {
    if constexpr (I + 1 < NP) {
        const double cc = uload(rec), ss = uload(rec + 1);
        const double t1 = ss * z[I], t2 = ss * z[I + 1];
        z[I + 1] = fma(cc, z[I + 1], t1);
        z[I] = fma(cc, z[I], -t2);
    }
}
template <int NP, int G>
SMRT_DEV void eig_groups(double (&z)[NP], const double*& rq, int gt, int gb) {
    if (G <= gt && G >= gb) {   // uniform
        eig_rot<NP, 4 * G + 3>(z, rq);
        eig_rot<NP, 4 * G + 2>(z, rq + 2);
        eig_rot<NP, 4 * G + 1>(z, rq + 4);
        eig_rot<NP, 4 * G>(z, rq + 6);
        rq += 8;
    }
    if constexpr (G > 0) eig_groups<NP, G - 1>(z, rq, gt, gb);
}

template <int NP>
SMRT_DEV void eig_vectors_item(const DevStage& stg, long long item) {
    const int lane = tid() & (SMRT_LANES - 1);
    const int N = uniform(stg.n[item]);
    if (N <= 0) return;
    const int vs = stg.vec_stride;
    const int LD = (vs + 1) | 1;
    double* gB = stg.B + item * stg.mat_stride;
    const double* gtau = stg.eig_e + item * 2 * vs + vs;
    double z[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) z[c] = (c == lane) ? 1.0 : 0.0;
    // Q = H_(N-1) ... H_2, every reflector applied from the right to the rows
    for (int k = N - 1; k >= 2; --k) {
        const double tau = uload(gtau + k);
        if (tau == 0.0) continue;   // uniform
        const double* vk = gB + k * LD;
        double t = 0.0, t2 = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < NP; j0 += 8) {
            if (j0 + 8 <= k) {   // uniform
#pragma unroll
                for (int jj = 0; jj < 8; jj += 2) {
                    t = fma(z[j0 + jj], uload(vk + j0 + jj), t);
                    t2 = fma(z[j0 + jj + 1], uload(vk + j0 + jj + 1), t2);
                }
            } else if (j0 < k) {   // the chunk that holds row k - 1: the column ends there
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    if (j0 + jj < k) t = fma(z[j0 + jj], uload(vk + j0 + jj), t);
            }
        }
        t = -tau * (t + t2);
#pragma unroll
        for (int j0 = 0; j0 < NP; j0 += 8) {
            if (j0 + 8 <= k) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) z[j0 + jj] = fma(t, uload(vk + j0 + jj), z[j0 + jj]);
            } else if (j0 < k) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    if (j0 + jj < k) z[j0 + jj] = fma(t, uload(vk + j0 + jj), z[j0 + jj]);
            }
        }
    }
    // the rotations of the QL iterations: per iteration the groups gt ... gb of four columns, top column first
    const double* list = stg.eig_rot + item * stg.rot_stride;
    while (true) {
        const int gt = uload((const int*)list), gb = uload((const int*)list + 1);
        if (gt < 0) break;
        const double* rq = list + 2;
        eig_groups<NP, NP / 4 - 1>(z, rq, gt, gb);
        list += 2 + 8 * (gt - gb + 1);
    }
    // B' = Z Sigma over the reflectors: every lane has consumed them (program order on the GPU; a rendezvous of the fibers
    // in the emulator)
    wave_sync();
    const double* gs = stg.sigma + item * vs;
#pragma unroll
    for (int c = 0; c < NP; ++c)
        if (c < N) {   // uniform
            const double sg = uload(gs + c);
            if (lane < N) gB[c * LD + lane] = z[c] * sg;
        }
}

}  // namespace smrt
