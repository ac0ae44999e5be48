"""Numerical study (NOT a test, not collected by pytest): the layer eigenproblem of the Rayleigh-phase emmodels (DMRT-QCA
short range = BASELINE configs[2], rayleigh, prescribed_kskaeps) WITHOUT an iteration.

For azimuth mode 0 the Rayleigh phase matrix depends on mu^2 only (smrt/emmodel/rayleigh.py:70-76):
P = 3 ks / 4 * (u1 u1^T / 2 + u2 u2^T) with u1 = [mu^2 (V); 1 (H)], u2 = [1 - mu^2 (V); 0 (H)], so P(mu, mu') = P(mu, -mu') and
of the two symmetric factors of the reduced problem

    X- = M^-1/2 T (ke I - c N (P+ - P-) W) T^-1 M^-1/2 = diag(ke / mu)              =: D^2   (diagonal)
    X+ = M^-1/2 T (ke I - c N (P+ + P-) W) T^-1 M^-1/2 = diag(ke / mu) - Y0 Y0^T            (diagonal minus rank 2)

The device needs A+ = L+^-T B' and A- = -L+ B' Sigma^-1 (B = L+^T L- = U Sigma V^T, B' = U Sigma) with A+^T A- = -Sigma.
With L- = D and B^T B = D X+ D = V Sigma^2 V^T they are A+ = D V and A- = -D^-1 V Sigma -- no Cholesky, no product, and
D X+ D = diag(a) - Y Y^T, a = (ke / mu)^2 (every pole twice: V and H of a stream), Y = D Y0 (N x 2), whose eigenpairs
follow from a 2 x 2 secular problem:

    K(lam) = I - sum_r y_r y_r^T / (a_r - lam),   det K(lam) = 0,   v_r = y_r . c / (a_r - lam),  K(lam) c = 0.

K decreases monotonically (Loewner order) between two poles, each of its two eigenvalues kappa_1 <= kappa_2 from +inf to
-inf: exactly two roots per interval (two below the lowest pole), each the zero of a monotone function -> bisection /
Newton without safeguards beyond the bracket.  The two roots of an interval may be nearly equal (weak scattering): their
vectors are orthogonalised against each other (they span the invariant subspace either way).

Prints orthogonality, residuals and the brightness temperatures against the oracle with the pivot-free recursion of
tests/studies/admittance_recursion.py around the new eigenpairs.

    python tests/studies/rayleigh_secular.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "studies"))
from oracle import dort_oracle as O  # noqa: E402
import admittance_recursion as AR  # noqa: E402

STATS = dict(orth=0.0, resid=0.0, evals=0, roots=0, xm_offdiag=0.0, rank2=0.0, layers=0)


def kappa(F11, F12, F22, which):
    """Eigenvalue `which` (0: smaller, 1: larger) of K = I - F for the symmetric 2 x 2 F."""
    tr = 0.5 * (F11 + F22)
    df = 0.5 * (F11 - F22)
    rad = np.hypot(df, F12)
    # eigenvalues of F: tr +- rad; of K: 1 - tr -+ rad
    return 1.0 - tr - rad if which == 0 else 1.0 - tr + rad


def secular_eig(a, Y):
    """Eigen-decomposition of diag(a) - Y Y^T (a ascending with every value twice: rows 2 j, 2 j + 1; Y: N x 2).
    Returns (lam ascending, V).  Roots are located as offsets from the upper pole of their interval (delta = a_j - lam > 0),
    so that a_r - lam is formed without cancellation for the nearest poles."""
    N = len(a)
    n = N // 2
    poles = a[0::2]
    G11 = Y[:, 0] * Y[:, 0]; G12 = Y[:, 0] * Y[:, 1]; G22 = Y[:, 1] * Y[:, 1]
    lam = np.empty(N); V = np.empty((N, N))
    for j in range(n):
        hi = poles[j]
        lo = poles[j - 1] if j > 0 else 0.0
        width = hi - lo

        def F(delta):
            # a_r - lam = (a_r - hi) + delta
            den = (a - hi) + delta
            inv = 1.0 / den
            STATS["evals"] += 1
            return (G11 * inv).sum(), (G12 * inv).sum(), (G22 * inv).sum()

        for which in (0, 1):
            # kappa_which(lam) decreases in lam, i.e. increases in delta = hi - lam on (0, width)
            d_lo, d_hi = 0.0, width
            for _ in range(200):
                mid = 0.5 * (d_lo + d_hi)
                if mid <= d_lo or mid >= d_hi:
                    break
                if kappa(*F(mid), which) < 0.0:
                    d_lo = mid
                else:
                    d_hi = mid
            delta = 0.5 * (d_lo + d_hi)
            F11, F12, F22 = F(delta)
            # null vector of K = I - F for the eigenvalue kappa_which
            K11, K12, K22 = 1.0 - F11, -F12, 1.0 - F22
            kap = kappa(F11, F12, F22, which)
            # (K - kap I) c = 0: c = (K12, kap - K11) or (kap - K22, K12), whichever is larger
            c1 = np.array([K12, kap - K11]); c2 = np.array([kap - K22, K12])
            c = c1 if c1 @ c1 >= c2 @ c2 else c2
            if c @ c == 0.0:
                c = np.array([1.0, 0.0]) if which == 0 else np.array([0.0, 1.0])
            v = (Y @ c) / ((a - hi) + delta)
            v /= np.linalg.norm(v)
            # kappa_0 <= kappa_1 and both decrease with lam: the zero of kappa_0 is the SMALLER lam of the interval
            k = 2 * j + which
            lam[k] = hi - delta
            V[:, k] = v
            STATS["roots"] += 1
        # the pair of an interval against each other (nearly equal roots: any orthonormal basis of the subspace will do)
        v0, v1 = V[:, 2 * j], V[:, 2 * j + 1]
        v1 = v1 - (v0 @ v1) * v0
        V[:, 2 * j + 1] = v1 / np.linalg.norm(v1)
    return lam, V


def rayleigh_layer_eigen(em, mu, w):
    """Same outputs as admittance_recursion.layer_eigen (S, A+, A-, d) from the secular route."""
    n = len(mu); P = 2; N = n * P
    full = np.concatenate((mu, -mu))
    ft = em.ft_even_phase(mu, full, 0, 2)[:, :, 0]
    Pc = O.compress(ft)
    Pp, Pm = Pc[:, :N], Pc[:, N:]
    wv = np.repeat(w, P); mv = np.repeat(mu, P)
    c = 0.5
    ke = em.ks + em.ka
    if em.ks != 0:
        rows = c * ((Pp + Pm) * wv[None, :]).sum(axis=1)
        norm = em.ks / rows
    else:
        norm = np.ones(N)
    sc = np.sqrt(norm * wv)
    Xp = (ke * np.eye(N) - c * sc[:, None] * (Pp + Pm) * sc[None, :]) / np.sqrt(mv[:, None] * mv[None, :])
    Xm = (ke * np.eye(N) - c * sc[:, None] * (Pp - Pm) * sc[None, :]) / np.sqrt(mv[:, None] * mv[None, :])
    STATS["xm_offdiag"] = max(STATS["xm_offdiag"], np.abs(Xm - np.diag(np.diag(Xm))).max() / np.abs(Xm).max())
    D2 = ke / mv                                   # X- (diagonal)
    R = np.diag(D2) - 0.5 * (Xp + Xp.T)            # = Y0 Y0^T, rank 2
    # its two columns from the closed form: R = (sc / sqrt(mu)) [2 c P] (sc / sqrt(mu)),  2 P = (3 ks / 4)(u1 u1^T + 2 u2 u2^T)
    # (read off numerically here: the study checks the rank and takes an exact factor)
    ev, evec = np.linalg.eigh(R)
    STATS["rank2"] = max(STATS["rank2"], np.abs(ev[:-2]).max() / max(ev[-1], 1e-300)) if em.ks != 0 else STATS["rank2"]
    Y0 = evec[:, -2:] * np.sqrt(np.maximum(ev[-2:], 0.0))[None, :]
    D = np.sqrt(D2)
    a = D2 * D2
    Y = D[:, None] * Y0
    order = np.argsort(a, kind="stable")           # poles ascending; (V, H) of a stream stay adjacent
    lam_s, V_s = secular_eig(a[order], Y[order])
    V = np.empty_like(V_s); V[order, :] = V_s
    M = np.diag(a) - Y @ Y.T
    STATS["orth"] = max(STATS["orth"], np.abs(V.T @ V - np.eye(N)).max())
    STATS["resid"] = max(STATS["resid"], np.abs(M @ V - V * lam_s[None, :]).max() / np.abs(lam_s).max())
    STATS["layers"] += 1
    S = np.sqrt(lam_s)
    d = np.sqrt(norm / wv) / np.sqrt(mv)
    Ap = D[:, None] * V
    Am = -(V * S[None, :]) / D[:, None]
    return S, Ap, Am, d


def compare(cases, label):
    worst_sec = worst_svd = 0.0
    n = 0
    svd_eigen = AR.layer_eigen
    for sp, f, theta, nstr, emmodel in cases:
        try:
            ref = O.solve(sp, f, theta, n_max_stream=nstr, emmodel=emmodel)
        except O.OracleError:
            continue
        n += 1
        AR.layer_eigen = rayleigh_layer_eigen
        got = AR.solve_pair(sp, f, theta, emmodel=emmodel, n_max_stream=nstr)
        worst_sec = max(worst_sec, float(np.abs(got - ref).max()))
        AR.layer_eigen = svd_eigen
        got = AR.solve_pair(sp, f, theta, emmodel=emmodel, n_max_stream=nstr)
        worst_svd = max(worst_svd, float(np.abs(got - ref).max()))
    print("== %s: %d pairs; max |dTb| against the oracle: secular %.2e K, SVD route %.2e K" % (label, n, worst_sec, worst_svd))
    print("   %d layers: max |V^T V - I| = %.1e, residual |M V - V L| / |L|max = %.1e, X- off-diagonal %.1e, third eigenvalue of "
          "diag - X+ %.1e of the first; %.1f function evaluations per root (plain bisection)"
          % (STATS["layers"], STATS["orth"], STATS["resid"], STATS["xm_offdiag"], STATS["rank2"], STATS["evals"] / max(STATS["roots"], 1)))
    sys.stdout.flush()


def cfg3_cases(n_sp=2, nstr=64, L=50):
    import bench
    thick, dens, temp, radius = bench.synthetic_snowpacks(3, S=n_sp, L=L, size_range=(5e-5, 1.5e-4))
    freqs = [6.925e9, 10.65e9, 18.7e9, 36.5e9, 89e9]
    for s in range(n_sp):
        sp = dict(thickness=thick[s], density=dens[s], temperature=temp[s], microstructure="sticky_hard_spheres", radius=radius[s],
                  stickiness=np.full(L, 0.2))
        for f in freqs:
            yield sp, f, [55.0], nstr, "dmrt_qca_shortrange"


def hard_cases(seed, n_cases):
    rng = np.random.default_rng(seed)
    for _ in range(n_cases):
        L = int(rng.integers(1, 7))
        nstr = int(rng.choice([8, 16, 32, 40, 64]))
        thick = 10.0 ** rng.uniform(-4, 0.5, L); thick[-1] = rng.choice([0.3, 100.0])
        sp = dict(thickness=thick, frac_volume=rng.uniform(0.05, 0.4, L), temperature=rng.uniform(200, 272.9, L),
                  microstructure="sticky_hard_spheres", radius=10.0 ** rng.uniform(-5, -3.3, L), stickiness=rng.choice([0.1, 0.2, 1000.0], L))
        for f in rng.choice([1.4e9, 6.9e9, 18.7e9, 36.5e9, 89e9], 2, replace=False):
            yield sp, float(f), [float(rng.uniform(0, 20)), float(rng.uniform(40, 75))], nstr, "dmrt_qca_shortrange"


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "small"
    if what == "small":
        compare(cfg3_cases(1, 16, 8), "DMRT-QCA-SR, 8 layers, 16 streams")
    elif what == "cfg3":
        compare(cfg3_cases(), "configs[2] shape: DMRT-QCA-SR, 50 layers, 64 streams")
    elif what == "hard":
        compare(hard_cases(1, 20), "hard DMRT inputs")
