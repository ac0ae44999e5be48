"""Numerical study (NOT a test): can the LAST sweeps of the one-sided Jacobi iteration on B = L+^T L- be replaced by matrix
updates on the matrix core?  Once every pair is at a small angle, the rotations of a whole sweep are, to first order, the
skew-symmetric K with K_pq = G_pq / (G_qq - G_pp) (G = B^T B), and B <- B (I + K + K^2 / 2) is an orthogonal update to
second order -- one Gram product and two more products instead of N (N - 1) / 2 rotations through LDS.  What decides whether
it works: nearly degenerate singular values (G_qq - G_pp small against G_pq), where the first-order angle is wrong.

    python tests/studies/jacobi_matrix_tail.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import dort_oracle as O  # noqa: E402
import bench  # noqa: E402


def layer_B(em, mu, w):
    n = len(mu); P = 2; N = n * P
    full = np.concatenate((mu, -mu))
    ft = em.ft_even_phase(mu, full, 0, 2)[:, :, 0]
    Pc = O.compress(ft)
    Pp, Pm = Pc[:, :N], Pc[:, N:]
    wv = np.repeat(w, P); mv = np.repeat(mu, P)
    ke = em.ks + em.ka
    rows = 0.5 * ((Pp + Pm) * wv[None, :]).sum(axis=1)
    norm = em.ks / rows if em.ks != 0 else np.ones(N)
    sc = np.sqrt(norm * wv)
    Xp = (ke * np.eye(N) - 0.5 * sc[:, None] * (Pp + Pm) * sc[None, :]) / np.sqrt(mv[:, None] * mv[None, :])
    Xm = (ke * np.eye(N) - 0.5 * sc[:, None] * (Pp - Pm) * sc[None, :]) / np.sqrt(mv[:, None] * mv[None, :])
    Lp = np.linalg.cholesky(0.5 * (Xp + Xp.T)); Lm = np.linalg.cholesky(0.5 * (Xm + Xm.T))
    return Lp.T @ Lm


def sweep(B, skip2=1e-26):
    """One cyclic sweep of one-sided Jacobi (columns sorted by norm first, like the kernel); returns max cos^2 met."""
    N = B.shape[1]
    worst = 0.0
    for p in range(N - 1):
        for q in range(p + 1, N):
            x, y = B[:, p], B[:, q]
            g = x @ y; a = x @ x; b = y @ y
            c2 = g * g / (a * b)
            worst = max(worst, c2)
            if c2 > skip2:
                dd = b - a
                t = np.sign(dd if dd != 0 else 1.0) * 2 * g / (abs(dd) + np.sqrt(dd * dd + 4 * g * g))
                c = 1 / np.sqrt(1 + t * t); s = c * t
                B[:, p], B[:, q] = c * x - s * y, s * x + c * y
    return worst


def max_cos2(B):
    G = B.T @ B
    d = np.diag(G)
    C = G * G / np.outer(d, d)
    np.fill_diagonal(C, 0)
    return C.max()


def matrix_update(B, theta=1e-2):
    """B (I + K + K^2 / 2) with the first-order angles; pairs whose angle estimate exceeds theta are left to rotations
    (returned as a count)."""
    G = B.T @ B
    d = np.diag(G)
    D = d[None, :] - d[:, None]                 # G_qq - G_pp at [p, q]
    with np.errstate(divide="ignore", invalid="ignore"):
        K = np.where(D != 0, G / D, np.inf)
    np.fill_diagonal(K, 0.0)
    unsafe = ~(np.abs(K) <= theta)
    np.fill_diagonal(unsafe, False)
    K = np.where(unsafe, 0.0, K)
    # rotation of columns p, q by the small angle t: x' = x - t y, y' = y + t x  <=>  B' = B (I + K), K[p, q] = +t, K[q, p] = -t
    # with t = g / (b - a): K[p, q] = G_pq / (G_qq - G_pp) -- antisymmetric by construction
    BK = B @ K
    return B + BK + 0.5 * (BK @ K), int(unsafe.sum() // 2)


def study(label, mats):
    print("==", label, "(%d matrices)" % len(mats))
    for pre in (2, 3, 4):
        res = []
        for B0 in mats:
            B = B0.copy()
            order = np.argsort(-(B * B).sum(axis=0)); B = B[:, order]
            sv = np.linalg.svd(B0, compute_uv=False)
            for _ in range(pre):
                sweep(B)
            c_pre = max_cos2(B)
            ups, unsafe_tot = 0, 0
            while max_cos2(B) > 1e-26 and ups < 4:
                B, unsafe = matrix_update(B)
                unsafe_tot += unsafe; ups += 1
            s_got = np.sort(np.sqrt((B * B).sum(axis=0)))[::-1]
            res.append((np.sqrt(c_pre), np.sqrt(max_cos2(B)), ups, unsafe_tot, np.abs(s_got / sv - 1).max()))
        r = np.array(res)
        print("  %d sweeps then matrix updates: cos before %.1e .. %.1e | after %.1e (worst) | updates %.1f mean, %d max | pairs left to rotations %d | "
              "singular values rel. err %.1e" % (pre, r[:, 0].min(), r[:, 0].max(), r[:, 1].max(), r[:, 2].mean(), int(r[:, 2].max()), int(r[:, 3].sum()), r[:, 4].max()))
    # plain Jacobi: sweeps to cos^2 <= 1e-26 everywhere
    n_sw = []
    for B0 in mats[:6]:
        B = B0.copy(); order = np.argsort(-(B * B).sum(axis=0)); B = B[:, order]
        k = 0
        while k < 12:
            k += 1
            if sweep(B) <= 1e-26:
                break
        n_sw.append(k)
    print("  plain cyclic Jacobi to cos <= 1e-13: sweeps", n_sw)


def matrices(freqs, n_sp, emmodel="iba", n_max_stream=32, seed=2):
    thick, dens, temp, lc = bench.synthetic_snowpacks(seed, S=n_sp)
    out = []
    for s in range(n_sp):
        sp = dict(thickness=thick[s], density=dens[s], temperature=temp[s], microstructure="exponential", corr_length=lc[s])
        for f in freqs:
            ems = O.make_layers(emmodel, f, sp)
            eps = np.array([e.eps_eff for e in ems])
            st = O.compute_streams(n_max_stream, eps)
            for l in (0, len(ems) // 2, len(ems) - 1):
                out.append(layer_B(ems[l], st.mu[l], st.weight[l]))
    return out


if __name__ == "__main__":
    study("headline laws, 18.7 / 36.5 / 89 GHz", matrices([18.7e9, 36.5e9, 89e9], 2))
    study("weak scattering, 1.4 GHz (nearly degenerate singular values)", matrices([1.4e9], 3))
