// The layer eigenproblem of the Rayleigh-phase emmodels without an iteration: one workgroup per (pair, layer) item.
// Part of the DORT device code (see dort_device.hpp for the overview and the reference map).
//
// For azimuth mode 0 the Rayleigh phase matrix depends on mu^2 only (smrt/emmodel/rayleigh.py:70-76, shared by
// dmrt_qca_shortrange.py:65-112, dmrt_qcacp_shortrange.py, rayleigh.py, prescribed_kskaeps.py): P = pa (u1 u1^T / 2 + u2 u2^T)
// with u1 = [mu^2 (V); 1 (H)], u2 = [1 - mu^2 (V); 0 (H)], so P(mu, mu') = P(mu, -mu') and of the two symmetric factors of
// the reduced problem (reference: smrt/rtsolver/dort.py:891-962)
//     X- = diag(ke / mu) =: D^2                       (diagonal)
//     X+ = diag(ke / mu) - Y0 Y0^T,  Y0 = u o [sqrt(pa / 2) u1, sqrt(pa) u2]      (diagonal minus rank 2; u: the row scaling)
// What the pivot-free finish kernels need -- A+ = L+^-T B', A- = -L+ B' Sigma^-1 with A+^T A- = -Sigma -- is, with L- = D and
// B^T B = D X+ D = V Sigma^2 V^T:  A+ = D V,  A- = -D^-1 V Sigma = -D^-2 A+ Sigma: no Cholesky, no product, no Jacobi sweep.
// D X+ D = diag(a) - Y Y^T, a = (ke / mu)^2 (every pole twice: V and H of a stream), Y = D Y0, has the 2 x 2 secular problem
//     K(lam) = I - sum_r y_r y_r^T / (a_r - lam),   det K(lam) = 0,   v_r = y_r . c / (a_r - lam),  K(lam) c = 0.
// K decreases monotonically (Loewner order) between two poles, each of its eigenvalues kappa_0 <= kappa_1 from +inf to -inf:
// exactly two roots per interval (two in (0, a_0): the matrix is positive definite while the albedo is below one), each the
// zero of a MONOTONE function -- Newton inside a bracket, from the nearer pole so that a_r - lam carries no cancellation.
// The two roots of an interval may be nearly equal (weak scattering): their vectors are orthogonalised against each other.
// tests/studies/rayleigh_secular.py: V^T V - I <= 1e-12, <= 2e-10 K against the oracle on hard inputs (like the SVD route).
#pragma once
#include "spmd.hpp"
#include <math.h>
#include <string.h>
#include "dort_dense.hpp"

namespace smrt {

// doubles of LDS of the Rayleigh kernel: four pole vectors (n <= 64) + nine row / root vectors (N <= 128) + flags
SMRT_HD int rayleigh_lds_doubles() { return 4 * 64 + 9 * 128 + 8; }

// What the prep kernel leaves in the Linv slot of a direct item: [0] ke, [1] pa, [2 .. 2 + n) mu_j, [2 + n .. 2 + n + N) the
// row scaling u_r = sqrt(norm_r w_r / mu_r).  What this kernel leaves there: [0 .. N) 1 / D_r^2 = mu_r / ke.
template <int NT>
SMRT_DEV void dort_rayleigh_item(const DevBatch& b, const DevStage& stg, long long item, double* lds) {
    const int t = tid();
    const int nraw = stg.n[item];
    if (!stage_direct(nraw)) return;          // (uniform) not a layer of this kind, failed, or not staged in this launch
    {   // the guards of the other diagonalisation kernels: layers beyond a snowpack's own, pairs the prep kernel refused
        const long long p = item / b.Lmax;
        const int l = (int)(item % b.Lmax);
        const int si = (int)(global_pair(b, p) % b.S);
        if (l >= b.n_layers[si] || b.status[p] != ST_OK) return;
    }
    const int N = stage_rows(nraw), n = N >> 1;
    const int vs = stg.vec_stride;
    const int LD = (vs + 1) | 1;
    double* gI = stg.Linv + item * stg.linv_stride;
    double* gB = stg.B + item * stg.mat_stride;
    double* a = lds;                 // poles, ascending (mu descends)
    double* g11 = lds + 64, * g12 = lds + 128, * g22 = lds + 192;
    double* y1 = lds + 256, * y2 = y1 + 128, * dr = y1 + 256;
    double* org = y1 + 384;          // per root: a_origin (the pole the root is measured from)
    double* xx = y1 + 512;           // ... signed offset: lam = org - xx
    double* c0v = y1 + 640, * c1v = y1 + 768, * nrm = y1 + 896, * gsc = y1 + 1024;
    int* flags = (int*)(lds + 256 + 9 * 128);
    const double ke = gI[0], pa = gI[1];
    if (t == 0) flags[0] = 0;
    const double s1 = sqrt(0.5 * pa), s2 = sqrt(pa);
    for (int j = t; j < n; j += NT) {
        const double mu = gI[2 + j], m2 = mu * mu;
        const double d2 = ke / mu, d = sqrt(d2);
        const double uv = gI[2 + n + 2 * j], uh = gI[2 + n + 2 * j + 1];
        const double yv1 = d * s1 * uv * m2, yv2 = d * s2 * uv * (1.0 - m2), yh1 = d * s1 * uh;
        a[j] = d2 * d2;
        y1[2 * j] = yv1; y2[2 * j] = yv2; y1[2 * j + 1] = yh1; y2[2 * j + 1] = 0.0;
        dr[2 * j] = d; dr[2 * j + 1] = d;
        g11[j] = yv1 * yv1 + yh1 * yh1; g12[j] = yv1 * yv2; g22[j] = yv2 * yv2;
    }
    block_sync();
    // ---- the N roots, thread k: interval j = k / 2 (between the poles a[j - 1] and a[j]; below a[0] for j = 0), the zero of
    //      kappa_(k & 1)
    if (t < N) {
        const int k = t, j = k >> 1, which = k & 1;
        const double hi = a[j], lo = (j > 0) ? a[j - 1] : 0.0, width = hi - lo;
        double c0 = which ? 0.0 : 1.0, c1 = which ? 1.0 : 0.0, dk = 1.0, origin = hi, x = 0.0;
        if (pa > 0.0) {
            // kappa and d kappa / d lam-offset at lam = o - sg x  (sg = +1: measured down from the upper pole; -1: up from the lower)
            auto eval = [&](double o, double sg, double xv, double& kap, double& dkap, double& e0, double& e1) {
                double F11 = 0.0, F12 = 0.0, F22 = 0.0, Q11 = 0.0, Q12 = 0.0, Q22 = 0.0;
                for (int i = 0; i < n; ++i) {
                    const double den = (a[i] - o) + sg * xv;
                    const double inv = fast_rcp(den), inv2 = inv * inv;
                    F11 = fma(g11[i], inv, F11); F12 = fma(g12[i], inv, F12); F22 = fma(g22[i], inv, F22);
                    Q11 = fma(g11[i], inv2, Q11); Q12 = fma(g12[i], inv2, Q12); Q22 = fma(g22[i], inv2, Q22);
                }
                // eigenpair `which` of K = I - F (kappa_0 <= kappa_1)
                const double tr = 0.5 * (F11 + F22), df = 0.5 * (F11 - F22);
                const double rad = sqrt(df * df + F12 * F12);
                kap = which ? 1.0 - tr + rad : 1.0 - tr - rad;
                const double K11 = 1.0 - F11, K12 = -F12, K22 = 1.0 - F22;
                double a0 = K12, a1 = kap - K11, b0 = kap - K22, b1 = K12;
                if (b0 * b0 + b1 * b1 > a0 * a0 + a1 * a1) { a0 = b0; a1 = b1; }
                const double nn = a0 * a0 + a1 * a1;
                if (nn > 0.0) { const double r = fast_rsqrt(nn); e0 = a0 * r; e1 = a1 * r; }
                else { e0 = which ? 0.0 : 1.0; e1 = which ? 1.0 : 0.0; }
                dkap = e0 * e0 * Q11 + 2.0 * e0 * e1 * Q12 + e1 * e1 * Q22;    // c^T (sum y y^T / den^2) c = |v|^2 before scaling
            };
            double kap, dkap, e0, e1;
            // which half of the interval?  (kappa decreases with lam; the lowest interval has no pole at its lower end)
            double sg = 1.0;
            double xlo = 0.0, xhi = 0.5 * width;
            eval(hi, 1.0, 0.5 * width, kap, dkap, e0, e1);
            if (j == 0) {
                xhi = width;
                double k0, dk0, f0, f1;
                eval(hi, 1.0, width, k0, dk0, f0, f1);          // lam = 0: positive definite <=> both kappa > 0
                if (!(k0 > 0.0)) lds_max(&flags[0], ST_ALBEDO);
            } else if (kap < 0.0) {   // kappa(mid) < 0: the root lies below the middle, nearer the lower pole
                sg = -1.0; origin = lo;
            }
            // bracket in x: the root has kappa = 0; f(x) = kappa(o - sg x) increases with x for sg = +1, decreases for sg = -1
            x = 0.5 * xhi;
            for (int it = 0; it < 100; ++it) {
                eval(origin, sg, x, kap, dkap, e0, e1);
                const bool below = (sg > 0.0) ? (kap < 0.0) : (kap > 0.0);     // x is below the root
                if (below) xlo = x; else xhi = x;
                // Newton on h(x) = x kappa(x), d kappa / dx = sg dkap: kappa ~ 1 - c / x near the pole the distance is measured
                // from, so h is nearly linear there and a root that hugs its pole is met in a few steps (on kappa itself the
                // step from the middle of the interval overshoots the bracket and the bisection has to find the scale first)
                double step = x * (x * sg * dkap) / (kap + x * sg * dkap);
                if (kap == 0.0 || fabs(step - x) <= 4.0e-16 * x) break;         // converged (x itself is the better point)
                if (!(step > xlo && step < xhi))   // outside the bracket: bisect -- geometrically while the scale is unknown
                    step = (xlo <= 0.0) ? 0.125 * xhi : (xhi > 4.0 * xlo) ? sqrt(xlo) * sqrt(xhi) : 0.5 * (xlo + xhi);
                if (!(step > xlo && step < xhi)) break;                         // the bracket is down to neighbouring numbers
                x = step;
            }
            eval(origin, sg, x, kap, dkap, e0, e1);
            c0 = e0; c1 = e1; dk = dkap;
            x = sg * x;                                                         // lam = origin - x
        }
        org[k] = origin; xx[k] = x; c0v[k] = c0; c1v[k] = c1; nrm[k] = fast_rsqrt(dk);
        stg.sigma[item * vs + k] = sqrt(origin - x);
    }
    block_sync();
    if (flags[0] != 0) { if (t == 0) stg.n[item] = -flags[0]; return; }        // (uniform) per layer, like a failed Cholesky
    // ---- the second vector of every interval against the first: g = v0 . v1
    if (t < N) {
        double g = 0.0;
        if ((t & 1) && pa > 0.0) {
            const int k1 = t, k0 = t - 1;
            const double o0 = org[k0], x0 = xx[k0], o1 = org[k1], x1 = xx[k1];
            const double p11 = c0v[k0] * c0v[k1], p12 = c0v[k0] * c1v[k1] + c1v[k0] * c0v[k1], p22 = c1v[k0] * c1v[k1];
            for (int i = 0; i < n; ++i) {
                const double d0 = (a[i] - o0) + x0, d1 = (a[i] - o1) + x1;
                g += (p11 * g11[i] + p12 * g12[i] + p22 * g22[i]) * fast_rcp(d0 * d1);
            }
            g *= nrm[k0] * nrm[k1];
        }
        gsc[t] = g;
    }
    block_sync();
    // ---- A+ = D V into the matrix slot, 1 / D^2 into the vector slot the finish kernel reads
    for_2d<NT>(N, N, [&](int r, int k) {
        double v;
        if (pa > 0.0) {
            const double ai = a[r >> 1];
            v = (y1[r] * c0v[k] + y2[r] * c1v[k]) * fast_rcp((ai - org[k]) + xx[k]) * nrm[k];
            if (k & 1) {
                const double g = gsc[k];
                const double v0 = (y1[r] * c0v[k - 1] + y2[r] * c1v[k - 1]) * fast_rcp((ai - org[k - 1]) + xx[k - 1]) * nrm[k - 1];
                v = (v - g * v0) * fast_rsqrt(1.0 - g * g);
            }
        } else v = (r == k) ? 1.0 : 0.0;
        gB[k * LD + r] = dr[r] * v;
    });
    for (int r = t; r < N; r += NT) gI[r] = 1.0 / (dr[r] * dr[r]);
}

}  // namespace smrt
