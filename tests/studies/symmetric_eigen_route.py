"""Numerical study (NOT a test, not collected by pytest): can the one-sided Jacobi iteration on B = L+^T L- (25 N^3-class
work, 5.3 sweeps) be replaced by a SYMMETRIC eigensolver of S = B B^T = L+^T X- L+ ?

The finish kernels need B' = U Sigma (left singular vectors of B scaled by the singular values) and Sigma; with
S = U Sigma^2 U^T both come straight from the eigen-decomposition of S -- no product with V, and U is orthogonal to
rounding by construction (what the pivot-free recursion relies on: A+^T A- = -Sigma).  What is given up: S squares the
condition number, so small singular values carry an ABSOLUTE error eps * sigma_max^2 in sigma^2 (the reference itself
works on the squared problem: smrt/rtsolver/dort.py:926-944 half_rank_eig, :835-889 Schur of the product).

Three eigensolvers are compared end to end (brightness temperatures against the oracle, the NumPy statement of the
pivot-free recursion of tests/studies/admittance_recursion.py around them):
  svd     numpy's SVD of B (what the Jacobi kernel converges to)
  eigh    LAPACK's symmetric solver on S
  tql     Householder tridiagonalisation + implicit QL with Wilkinson shifts, written out here the way a kernel would
          run it (one sequential chase that only touches the tridiagonal + a rotation list applied to the rows of Z)
and the number of plane rotations of `tql` is counted: it is what prices the kernel.

    python tests/studies/symmetric_eigen_route.py [headline | hard | big]
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

STATS = dict(rot=0, n2=0, items=0, iters=0, orth=0.0, resid=0.0, nmax=0)


def tridiagonalise(S):
    """Householder reduction of the symmetric S to tridiagonal form, eliminating from the LAST row upwards (tred2's
    direction: the small end of a graded matrix stays at the top, where QL starts).  Returns d, e (e[i] couples i, i+1)
    and Q with Q^T S Q = T."""
    A = S.copy(); n = len(A)
    Q = np.eye(n)
    for k in range(n - 1, 1, -1):
        x = A[k, :k].copy()
        alpha = -np.copysign(np.linalg.norm(x), x[k - 1])
        if alpha == 0.0:
            continue
        v = x.copy(); v[k - 1] -= alpha
        vn = v @ v
        if vn == 0.0:
            continue
        tau = 2.0 / vn
        # H = I - tau v v^T on the leading k x k block
        p = tau * (A[:k, :k] @ v)
        w = p - (0.5 * tau * (p @ v)) * v
        A[:k, :k] -= np.outer(v, w) + np.outer(w, v)
        A[k, :k] = 0.0; A[:k, k] = 0.0
        A[k, k - 1] = A[k - 1, k] = alpha
        Q[:, :k] -= tau * np.outer(Q[:, :k] @ v, v)
    return np.diag(A).copy(), np.diag(A, 1).copy(), Q


def tql(d, e, Z, count=STATS):
    """Implicit QL with Wilkinson shifts (EISPACK tql2 / LAPACK dsteqr's QL branch) on the tridiagonal (d, e); the
    rotations are applied to the columns of Z.  e[i] couples i and i + 1."""
    n = len(d)
    d = d.copy(); e = np.concatenate((e, [0.0]))
    eps = np.finfo(float).eps
    for l in range(n):
        it = 0
        while True:
            m = l
            while m < n - 1:
                dd = abs(d[m]) + abs(d[m + 1])
                if abs(e[m]) <= eps * dd:
                    break
                m += 1
            if m == l:
                break
            it += 1
            if it > 60:
                raise RuntimeError("QL: no convergence")
            g = (d[l + 1] - d[l]) / (2.0 * e[l])
            r = np.hypot(g, 1.0)
            g = d[m] - d[l] + e[l] / (g + np.copysign(r, g))
            s = c = 1.0; p = 0.0
            underflow = False
            for i in range(m - 1, l - 1, -1):
                f = s * e[i]; b = c * e[i]
                r = np.hypot(f, g)
                e[i + 1] = r
                if r == 0.0:
                    d[i + 1] -= p; e[m] = 0.0
                    underflow = True
                    break
                s = f / r; c = g / r
                g = d[i + 1] - p
                r = (d[i] - g) * s + 2.0 * c * b
                p = s * r
                d[i + 1] = g + p
                g = c * r - b
                zi, zi1 = Z[:, i].copy(), Z[:, i + 1].copy()
                Z[:, i + 1] = s * zi + c * zi1
                Z[:, i] = c * zi - s * zi1
                count["rot"] += 1
            count["iters"] += 1
            if underflow:
                continue
            d[l] -= p; e[l] = g; e[m] = 0.0
    return d, Z


def eig_tql(S):
    d, e, Q = tridiagonalise(S)
    STATS["n2"] += len(S) ** 2; STATS["items"] += 1; STATS["nmax"] = max(STATS["nmax"], len(S))
    lam, Z = tql(d, e, Q)
    return lam, Z


def make_layer_eigen(kind):
    def layer_eigen(em, mu, w):
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
        Xp = 0.5 * (Xp + Xp.T); Xm = 0.5 * (Xm + Xm.T)
        Lp = np.linalg.cholesky(Xp)
        if kind == "svd":
            Lm = np.linalg.cholesky(Xm)
            U, Sg, _ = np.linalg.svd(Lp.T @ Lm)
        else:
            if kind.endswith("_bbt"):     # S formed from the staged B (no change to the prep kernel)
                Lm = np.linalg.cholesky(Xm)
                B = Lp.T @ Lm
                Ssym = B @ B.T
            else:                          # S = L+^T X- L+: no second Cholesky
                Ssym = Lp.T @ Xm @ Lp
            Ssym = 0.5 * (Ssym + Ssym.T)
            if kind.startswith("eigh"):
                lam, U = np.linalg.eigh(Ssym)
            else:
                lam, U = eig_tql(Ssym)
            STATS["orth"] = max(STATS["orth"], np.abs(U.T @ U - np.eye(N)).max())
            STATS["resid"] = max(STATS["resid"], np.abs(Ssym @ U - U * lam[None, :]).max() / np.abs(lam).max())
            Sg = np.sqrt(lam)
        Bp = U * Sg[None, :]
        d = np.sqrt(norm / wv) / np.sqrt(mv)
        Ap = np.linalg.solve(Lp.T, Bp)
        Am = -(Lp @ Bp) / Sg[None, :]
        return Sg, Ap, Am, d
    return layer_eigen


KINDS = ("svd", "eigh", "eigh_bbt", "tql")


def compare(cases, label):
    """cases: iterable of (sp, frequency, theta_deg list, n_max_stream, substrate, atmosphere)."""
    worst = {k: 0.0 for k in KINDS}
    n = 0
    for sp, f, theta, nstr, sub, atm in cases:
        try:
            ref = O.solve(sp, f, theta, n_max_stream=nstr, substrate=sub, atmosphere=atm)
        except O.OracleError:
            continue
        n += 1
        for k in KINDS:
            AR.layer_eigen = make_layer_eigen(k)
            got = AR.solve_pair(sp, f, theta, n_max_stream=nstr, substrate=sub, atmosphere=atm)
            worst[k] = max(worst[k], float(np.abs(got - ref).max()))
    print("== %s: %d pairs; max |dTb| against the oracle: %s" % (label, n, ", ".join("%s %.2e K" % kv for kv in worst.items())))
    if STATS["items"]:
        print("   tql: %d matrices (largest N = %d), %.3f N^2 rotations and %.2f QL iterations per eigenvalue on average; "
              "max |U^T U - I| = %.1e, max residual |S U - U L| / |L|max = %.1e"
              % (STATS["items"], STATS["nmax"], STATS["rot"] / STATS["n2"], STATS["iters"] / np.sqrt(STATS["n2"] * STATS["items"]),
                 STATS["orth"], STATS["resid"]))
    sys.stdout.flush()


def headline_cases(n_sp=3):
    import bench
    thick, dens, temp, lc = bench.synthetic_snowpacks(1, S=n_sp)
    for s in range(n_sp):
        sp = dict(thickness=thick[s], density=dens[s], temperature=temp[s], microstructure="exponential", corr_length=lc[s])
        for f in bench.FREQS:
            yield sp, f, [bench.THETA_DEG], 32, None, None


def hard_cases(seed, n_cases, streams=(4, 7, 12, 16, 24, 32), max_layers=8, S=2):
    """The generator of tools/stress_reg_extremes.py (0.1 mm ... 100 m layers, 1.4 ... 183 GHz, volume fractions up to 0.49,
    correlation lengths up to the renormalisation limit)."""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        L = int(rng.integers(1, max_layers + 1))
        n_str = int(rng.choice(list(streams)))
        thick = 10.0 ** rng.uniform(-4, 0.5, (S, L)); thick[:, -1] = rng.choice([0.3, 100.0], S)
        fv = rng.uniform(0.05, 0.49, (S, L)); temp = rng.uniform(200, 272.9, (S, L))
        lc = 10.0 ** rng.uniform(-5, -3.2, (S, L))
        freqs = np.sort(rng.choice([1.4e9, 6.9e9, 18.7e9, 36.5e9, 89e9, 150e9, 183e9], 3, replace=False))
        theta = [float(rng.uniform(0, 20)), float(rng.uniform(40, 75))]
        sub = atm = None
        if rng.random() < 0.5:
            sub = dict(kind="flat", eps=complex(rng.uniform(2, 30), rng.uniform(0.01, 5)), temperature=float(rng.uniform(240, 273)))
        if rng.random() < 0.4:
            atm = dict(tb_down=float(rng.uniform(3, 80)), tb_up=float(rng.uniform(2, 60)), transmittance=float(rng.uniform(0.4, 1.0)))
        for s in range(S):
            sp = dict(thickness=thick[s], frac_volume=fv[s], temperature=temp[s], microstructure="exponential", corr_length=lc[s])
            for f in freqs:
                yield sp, float(f), theta, n_str, sub, atm


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "headline"
    if what == "headline":
        compare(headline_cases(), "headline batch (20 layers, 32 streams, 5 channels)")
    elif what == "hard":
        for seed in (1, 2, 3):
            compare(hard_cases(seed, 10), "hard inputs, seed %d" % seed)
    elif what == "big":
        compare(hard_cases(5, 4, streams=(40, 64), max_layers=6, S=1), "hard inputs at 40 / 64 streams")
