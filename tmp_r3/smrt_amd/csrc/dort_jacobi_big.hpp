// The Jacobi kernel for matrices that do not fit in LDS (streams x polarisations N in 129 .. 384: BASELINE configs[3]
// with 128 streams has N = 256 for azimuth mode 0 and 384 for modes 1, 2): one (pair, [azimuth mode,] layer) item per
// workgroup, the matrix B = L+^T L- stays in the staging area (HBM / L2) and moves through LDS two column blocks at
// a time.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
//
// One-sided (Hestenes) Jacobi, block row-cyclic: the columns are cut into NB blocks of W; block I is resident in LDS
// slot 0 while the blocks J > I pass through slot 1; a visit (I, J) rotates all W x W cross pairs -- W rounds of W
// disjoint pairs, lane group a keeps column I_a in registers for the whole visit, the J columns go round through LDS
// (one workgroup barrier per round) -- plus, once per sweep and block, the pairs inside the block.  Per sweep the matrix
// crosses the LDS boundary (NB + 1) / 2 times instead of once per column pair.  Rotation formulas, tracked column norms
// and the skip / exit thresholds are those of the LDS-resident kernel (dort_jacobi_kernel.hpp).
#pragma once
#include "dort_jacobi_kernel.hpp"

namespace smrt {

#ifndef SMRT_JACOBI_BIG_NT
#define SMRT_JACOBI_BIG_NT 768      // threads per workgroup = W column pairs x GS lanes
#endif
constexpr int kJacobiBigGS = 32;    // lanes per column pair
constexpr int kJacobiBigW = SMRT_JACOBI_BIG_NT / kJacobiBigGS;   // columns per block (24)

struct JacobiBigPlan { int NMAX, LD, LDJ, W, NBMAX, o_nrm, o_int, total; };
SMRT_HD JacobiBigPlan make_jacobi_big_plan(int n_max_stream, int P) {
    JacobiBigPlan p;
    p.NMAX = n_max_stream * P;
    p.LD = (p.NMAX + 1) | 1;                           // layout of the staged matrices in global memory (make_plan)
    p.LDJ = ((p.NMAX + 63) / 64) * 64 + 8;             // padded rows (an even number of 32-row slices: RPL is even) + 8
    p.W = kJacobiBigW;
    p.NBMAX = (p.NMAX + p.W - 1) / p.W;
    int o = 2 * p.W * p.LDJ;                           // slot 0 (block I), slot 1 (block J)
    p.o_nrm = o; o += p.NBMAX * p.W + 8;               // tracked squared column norms of the WHOLE matrix
    p.o_int = o; o += 4;
    p.total = o;
    return p;
}

// rotation of one column pair given g = x.y and the squared norms a, b; returns tan (0 if skipped)
SMRT_DEV double jacobi_tangent(double gg, double a, double bb, double skip2, double* c, double* sn) {
    const double g2 = gg * gg;
    if (!(g2 > skip2 * a * bb)) return 0.0;
    const double dd = bb - a;
    const double hh = dd * dd + 4.0 * g2;
    const double h = hh * fast_rsqrt1(hh);
    const double tt = (dd >= 0.0 ? 2.0 : -2.0) * gg * fast_rcp1(fabs(dd) + h);
    *c = fast_rsqrt(1.0 + tt * tt);
    *sn = *c * tt;
    return tt;
}

template <int NT, int RPL>
SMRT_DEV void dort_jacobi_big_impl(const DevBatch& b, const DevStage& stg, long long item, int N, double* lds) {
    constexpr int GS = kJacobiBigGS, W = NT / GS;
    static_assert(W % 2 == 0, "the in-block tournament needs an even number of columns");
    const int t = tid();
    const int grp = t / GS, sub = t % GS;
    const JacobiBigPlan plan = make_jacobi_big_plan(b.n_max_stream, b.mode == 1 ? 3 : 2);
    const int LD = plan.LD, LDJ = plan.LDJ;
    double* S0 = lds;
    double* S1 = lds + W * LDJ;
    double* nrm = lds + plan.o_nrm;
    int* flag = (int*)(lds + plan.o_int);
    double* gB = stg.B + item * stg.mat_stride;
    const int NB = (N + W - 1) / W;
    const double skip2 = b.jacobi_skip2, exit2 = b.jacobi_exit2;

    // column block blk of the global matrix <-> LDS slot (zero padding beyond N rows / columns)
    auto load_block = [&](double* slot, int blk) {
        for (int c = grp; c < W; c += NT / GS) {   // one lane group per column, RPL rows per lane
            const int gc = blk * W + c;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = sub + i * GS;
                const double v = (gc < N && r < N) ? gB[(long long)gc * LD + r] : 0.0;
                slot[c * LDJ + r] = v;
            }
        }
    };
    auto store_block = [&](const double* slot, int blk) {
        for (int c = grp; c < W; c += NT / GS) {
            const int gc = blk * W + c;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int r = sub + i * GS;
                if (gc < N && r < N) gB[(long long)gc * LD + r] = slot[c * LDJ + r];
            }
        }
    };
    // squared norms of the columns of the block in `slot` -> nrm[blk * W + c]
    auto block_norms = [&](const double* slot, int blk) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) { const double xx = slot[grp * LDJ + sub + i * GS]; a += xx * xx; }
        a = group_sum<GS>(a);
        if (sub == 0) nrm[blk * W + grp] = a;
    };
    // pairs inside the block in `slot`: W - 1 rounds of W / 2 pairs (lane groups 0 .. W/2 - 1)
    auto inblock_pairs = [&](double* slot, int blk) {
        for (int u = 0; u < W - 1; ++u) {
            if (grp < W / 2) {
                int pa, pb;
                if (grp == 0) { pa = W - 1; pb = u; }
                else {
                    pa = u + grp; if (pa >= W - 1) pa -= W - 1;
                    pb = u - grp; if (pb < 0) pb += W - 1;
                }
                double x[RPL], y[RPL];
                double gg = 0.0, gg2 = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    x[i] = slot[pa * LDJ + sub + i * GS];
                    y[i] = slot[pb * LDJ + sub + i * GS];
                    if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
                }
                // (the norms are read before the reduction: lane 0 of the group rewrites them below, and only the
                // lockstep of a wavefront would otherwise keep the other lanes from seeing the new values)
                const double a = nrm[blk * W + pa], bb = nrm[blk * W + pb];
                gg = group_sum<GS>(gg + gg2);
                double c, sn;
                const double tt = jacobi_tangent(gg, a, bb, skip2, &c, &sn);
                if (tt != 0.0) {
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        slot[pa * LDJ + sub + i * GS] = c * x[i] - sn * y[i];
                        slot[pb * LDJ + sub + i * GS] = sn * x[i] + c * y[i];
                    }
                    if (sub == 0) {
                        nrm[blk * W + pa] = a - tt * gg;
                        nrm[blk * W + pb] = bb + tt * gg;
                        if (gg * gg > exit2 * a * bb) lds_or(flag, 1);
                    }
                }
            }
            block_sync();
        }
    };

    bool converged = false;
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        if (t == 0) *flag = 0;
        for (int I = 0; I < NB - 1 || (NB == 1 && I == 0); ++I) {
            load_block(S0, I);
            block_sync();
            block_norms(S0, I);              // fresh norms of the resident block once per sweep
            block_sync();
            inblock_pairs(S0, I);
            for (int J = I + 1; J < NB; ++J) {
                load_block(S1, J);
                block_sync();
                if (I == 0) { block_norms(S1, J); block_sync(); }     // first time this sweep: fresh norms of block J
                if (I == NB - 2) inblock_pairs(S1, J);                 // the last block is never resident in slot 0
                // cross pairs (I_a, J_(a + r)): column I_a and its norm stay in registers for the W rounds
                double x[RPL];
#pragma unroll
                for (int i = 0; i < RPL; ++i) x[i] = S0[grp * LDJ + sub + i * GS];
                double a = nrm[I * W + grp];
                for (int r = 0; r < W; ++r) {
                    int bq = grp + r; if (bq >= W) bq -= W;
                    double* cq = S1 + bq * LDJ;
                    double y[RPL];
                    double gg = 0.0, gg2 = 0.0;
#pragma unroll
                    for (int i = 0; i < RPL; ++i) {
                        y[i] = cq[sub + i * GS];
                        if (i & 1) gg2 += x[i] * y[i]; else gg += x[i] * y[i];
                    }
                    const double bb = nrm[J * W + bq];
                    gg = group_sum<GS>(gg + gg2);
                    double c, sn;
                    const double tt = jacobi_tangent(gg, a, bb, skip2, &c, &sn);
                    if (tt != 0.0) {
#pragma unroll
                        for (int i = 0; i < RPL; ++i) {
                            const double xn = c * x[i] - sn * y[i];
                            cq[sub + i * GS] = sn * x[i] + c * y[i];
                            x[i] = xn;
                        }
                        if (sub == 0) {
                            nrm[J * W + bq] = bb + tt * gg;
                            if (gg * gg > exit2 * a * bb) lds_or(flag, 1);
                        }
                        a -= tt * gg;
                    }
                    block_sync();   // the J columns just written are read by other lane groups in the next round
                }
#pragma unroll
                for (int i = 0; i < RPL; ++i) S0[grp * LDJ + sub + i * GS] = x[i];
                if (sub == 0) nrm[I * W + grp] = a;
                store_block(S1, J);
                block_sync();
            }
            store_block(S0, I);
            block_sync();
        }
        converged = (*flag == 0);
        block_sync();   // everyone has read the flag before it is cleared again
    }
    if (!converged) { if (t == 0) stg.n[item] = -ST_EIGEN; return; }
    // singular values = column norms, recomputed from the final matrix
    for (int blk = 0; blk < NB; ++blk) {
        load_block(S0, blk);
        block_sync();
        block_norms(S0, blk);
        block_sync();
        const int gc = blk * W + grp;
        if (sub == 0 && gc < N) { const double a = nrm[gc]; stg.sigma[item * stg.vec_stride + gc] = a * fast_rsqrt(a); }
        block_sync();
    }
}

template <int NT>
SMRT_DEV void dort_jacobi_big_item(const DevBatch& b, const DevStage& stg, long long item, double* lds) {
    const int nmodes = (b.mode == 1) ? b.m_max + 1 : 1;
    const long long p = item / ((long long)b.Lmax * nmodes);
    const int l = (int)(item % b.Lmax);
    const int si = (int)(global_pair(b, p) % b.S);
    if (l >= b.n_layers[si]) return;          // uniform
    if (b.status[p] != ST_OK) return;         // the prep kernel flagged this pair (uniform)
    const int N = stg.n[item];
    if (N <= 0) return;                       // the prep kernel flagged this layer (uniform)
    const int rpl = (N + kJacobiBigGS - 1) / kJacobiBigGS;
    if (rpl > 10) dort_jacobi_big_impl<NT, 12>(b, stg, item, N, lds);
    else if (rpl > 8) dort_jacobi_big_impl<NT, 10>(b, stg, item, N, lds);
    else if (rpl > 6) dort_jacobi_big_impl<NT, 8>(b, stg, item, N, lds);
    else if (rpl > 4) dort_jacobi_big_impl<NT, 6>(b, stg, item, N, lds);
    else dort_jacobi_big_impl<NT, 4>(b, stg, item, N, lds);
}

}  // namespace smrt
