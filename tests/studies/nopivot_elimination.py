"""Numerical study (NOT a test, not collected by pytest): how much pivoting do the two N x N solves of the layer
recursion need?  Uses the CPU oracle for the layer physics, rebuilds the device formulation (F, G from the symmetric
reduction) in NumPy and eliminates M1 = F - R~G and W without pivoting / with pivoting restricted to 16-row diagonal
blocks, for several orderings of the eigenpairs.  Result quoted in DESIGN.md section 7.1:

    python tests/studies/nopivot_elimination.py

  eigenpairs matched to rows (assignment)      : no pivoting, growth 2.0, errors 1e-15
  eigenpairs sorted by beta, 16-row block pivot : growth <= 4, errors 1e-15
  eigenpairs sorted by beta, no pivoting        : growth <= 220, errors <= 2e-13
  eigenpairs in LAPACK SVD order, no pivoting   : breaks down (growth 1e19)
"""
import numpy as np, sys
import numpy as np, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import dort_oracle as O
from conftest import load_golden, snowpack_dict, fixture_options

def layer_FG(em, mu, w):
    n = len(mu); P = 2; N = n*P
    full = np.concatenate((mu, -mu))
    ft = em.ft_even_phase(mu, full, 0, 2)[:, :, 0]           # (2,2,n,2n)
    Pc = O.compress(ft)                                        # (N, 2N)
    Pp, Pm = Pc[:, :N], Pc[:, N:]
    wv = np.repeat(w, P); mv = np.repeat(mu, P)
    c = 0.5
    rows = c * ((Pp + Pm) * wv[None, :]).sum(axis=1)
    norm = em.ks / rows
    assert np.all(np.abs(norm - 1) < 0.3)
    ke = em.ks + em.ka
    sc = np.sqrt(norm * wv)
    Xp = (ke*np.eye(N) - c * sc[:, None]*(Pp+Pm)*sc[None, :]) / np.sqrt(mv[:, None]*mv[None, :])
    Xm = (ke*np.eye(N) - c * sc[:, None]*(Pp-Pm)*sc[None, :]) / np.sqrt(mv[:, None]*mv[None, :])
    Xp = 0.5*(Xp+Xp.T); Xm = 0.5*(Xm+Xm.T)
    Lp = np.linalg.cholesky(Xp); Lm = np.linalg.cholesky(Xm)
    B = Lp.T @ Lm
    # one-sided Jacobi == SVD: B V = U S
    U, S, Vt = np.linalg.svd(B)
    Bp = U * S[None, :]      # B' = B V
    beta = S
    d = np.sqrt(norm / wv) / np.sqrt(mv)
    # roles: X_- = Lm Lm^T plays "L", X_+ = Lp Lp^T
    # Ep = d*(Lm V) = d * (Lp^-T B'),  Em = -d*(Lm^-T V) beta = -d * (Lp B') / beta
    Ep = d[:, None] * np.linalg.solve(Lp.T, Bp)
    Em = -d[:, None] * (Lp @ Bp) / beta[None, :]
    F = 0.5*(Ep - Em); G = 0.5*(Ep + Em)
    return beta, F, G

from scipy.optimize import linear_sum_assignment
def lu_nopiv_growth(A):
    A = A.copy(); n = len(A); amax = np.abs(A).max(); g = 1.0
    for k in range(n):
        if A[k, k] == 0: return np.inf
        A[k+1:, k] /= A[k, k]
        A[k+1:, k+1:] -= np.outer(A[k+1:, k], A[k, k+1:])
        g = max(g, np.abs(A[k+1:, k+1:]).max() / amax if k < n-1 else 0)
    return g, np.abs(np.tril(A, -1)).max()
def solve_nopiv(A, B):
    A = A.copy(); B = B.copy(); n = len(A)
    for k in range(n):
        piv = A[k, k]
        A[k, :] /= piv; B[k, :] /= piv
        for r in range(n):
            if r != k:
                l = A[r, k]; A[r, :] -= l*A[k, :]; B[r, :] -= l*B[k, :]
    return B
def run(name, order="match"):
    d = load_golden(name); sp = snowpack_dict(d)
    worst = [0, 0, 0, 0]
    for i, f in enumerate(d["frequency"]):
        ems = O.make_layers(str(d["emmodel"]), float(f), sp); eps = np.array([e.eps_eff for e in ems])
        st = O.compute_streams(fixture_options(d)["n_max_stream"], eps); itf = O.interface_diagonals(eps, st, 2)
        L = len(ems); thick = np.asarray(sp["thickness"], float)
        Rt = None
        for l in range(L-1, -1, -1):
            beta, F, G = layer_FG(ems[l], st.mu[l], st.weight[l])
            N = len(beta)
            if order == "match":
                r, c = linear_sum_assignment(-np.abs(F)); perm = c[np.argsort(r)]
                F, G, beta = F[:, perm], G[:, perm], beta[perm]
            t = np.exp(-beta*thick[l])
            Rtop = O._flatten_pol(itf["Rtop"][l], 0); Ttop = O._flatten_pol(itf["Ttop"][l], 0)
            if l == L-1: Rt = np.zeros((N, N))
            M1 = F - Rt @ G; rhs = Rt @ F - G
            Q = np.linalg.solve(M1, rhs)
            Qn = solve_nopiv(M1, rhs)
            g1 = lu_nopiv_growth(M1)
            tQt = t[:, None]*Q*t[None, :]
            Y = F @ tQt + G; W = (G - Rtop[:, None]*F) @ tQt + (F - Rtop[:, None]*G)
            K = np.linalg.solve(W.T, Y.T).T
            Kn = solve_nopiv(W.T, Y.T).T
            g2 = lu_nopiv_growth(W.T)
            worst[0] = max(worst[0], g1[0]); worst[1] = max(worst[1], g2[0])
            worst[2] = max(worst[2], np.abs(Qn-Q).max()/np.abs(Q).max()); worst[3] = max(worst[3], np.abs(Kn-K).max()/np.abs(K).max())
            if l > 0:
                Nu = st.n[l-1]*2; Rb = O._flatten_pol(itf["Rbot"][l-1], 0); Tb = O._flatten_pol(itf["Tbot"][l-1], 0)
                nc = min(N, Nu); Rn = np.diag(Rb).astype(float); Rn[:nc, :nc] += (Ttop[:nc, None]*K[:nc, :nc])*Tb[None, :nc]; Rt = Rn
    print(name, order, "growth M1 %.2f W %.2f  relerr Q %.2e K %.2e" % tuple(worst))
for nm in ["cfg2_iba_L20_n32_sp0", "cfg2_iba_L20_n32_sp1", "dmrt_L8_n16", "iba_L3_n16_shallow"]:
    run(nm, "match"); run(nm, "svd")

def solve_blockpiv(A, B, bs=16):
    """Gauss-Jordan with partial pivoting restricted to the rows of the running diagonal block."""
    A = A.copy(); B = B.copy(); n = len(A); gmax = np.abs(A).max(); g = 1.0
    for k in range(n):
        b0 = (k // bs) * bs; b1 = min(b0 + bs, n)
        p = k + np.argmax(np.abs(A[k:b1, k]))
        if p != k: A[[k, p]] = A[[p, k]]; B[[k, p]] = B[[p, k]]
        piv = A[k, k]
        A[k, :] /= piv; B[k, :] /= piv
        for r in range(n):
            if r != k:
                l = A[r, k]; A[r, :] -= l*A[k, :]; B[r, :] -= l*B[k, :]
        g = max(g, np.abs(A).max()/gmax, np.abs(B).max()/max(np.abs(B).max(),1e-300))
    return B, g
def jacobi_natural(B):
    """cyclic one-sided Jacobi without any column sorting (what the device does, different pair order)."""
    B = B.copy(); n = B.shape[1]
    for sweep in range(30):
        off = 0
        for p in range(n-1):
            for q in range(p+1, n):
                a = B[:, p] @ B[:, p]; b = B[:, q] @ B[:, q]; g = B[:, p] @ B[:, q]
                if g*g > 1e-30*a*b:
                    off = max(off, g*g/(a*b))
                    dd = b - a; tt = (2.0 if dd >= 0 else -2.0)*g/(abs(dd) + np.sqrt(dd*dd + 4*g*g))
                    c = 1/np.sqrt(1+tt*tt); s = c*tt
                    x = B[:, p].copy(); y = B[:, q].copy()
                    B[:, p] = c*x - s*y; B[:, q] = s*x + c*y
        if off < 1e-15: break
    return B
def run2(name, mode):
    d = load_golden(name); sp = snowpack_dict(d)
    worst = [0, 0, 0, 0]
    for i, f in enumerate(d["frequency"]):
        ems = O.make_layers(str(d["emmodel"]), float(f), sp); eps = np.array([e.eps_eff for e in ems])
        st = O.compute_streams(fixture_options(d)["n_max_stream"], eps); itf = O.interface_diagonals(eps, st, 2)
        L = len(ems); thick = np.asarray(sp["thickness"], float)
        Rt = None
        for l in range(L-1, -1, -1):
            em, mu, w = ems[l], st.mu[l], st.weight[l]
            n = len(mu); N = 2*n
            full = np.concatenate((mu, -mu)); Pc = O.compress(em.ft_even_phase(mu, full, 0, 2)[:, :, 0]); Pp, Pm = Pc[:, :N], Pc[:, N:]
            wv = np.repeat(w, 2); mv = np.repeat(mu, 2)
            norm = em.ks/(0.5*((Pp+Pm)*wv[None, :]).sum(axis=1)); ke = em.ks+em.ka; sc = np.sqrt(norm*wv)
            Xp = (ke*np.eye(N) - 0.5*sc[:, None]*(Pp+Pm)*sc[None, :])/np.sqrt(mv[:, None]*mv[None, :]); Xm = (ke*np.eye(N) - 0.5*sc[:, None]*(Pp-Pm)*sc[None, :])/np.sqrt(mv[:, None]*mv[None, :])
            Lp = np.linalg.cholesky(0.5*(Xp+Xp.T)); Lm = np.linalg.cholesky(0.5*(Xm+Xm.T)); B = Lp.T @ Lm
            if mode.startswith("jac"):
                Bp = jacobi_natural(B); beta = np.linalg.norm(Bp, axis=0)
            else:
                U, S, Vt = np.linalg.svd(B); Bp = U*S[None, :]; beta = S
            if mode.endswith("sorted"):
                o = np.argsort(beta, kind="stable"); Bp, beta = Bp[:, o], beta[o]
            dd = np.sqrt(norm/wv)/np.sqrt(mv)
            Ep = dd[:, None]*np.linalg.solve(Lp.T, Bp); Em = -dd[:, None]*(Lp @ Bp)/beta[None, :]
            F = 0.5*(Ep-Em); G = 0.5*(Ep+Em)
            t = np.exp(-beta*thick[l]); Rtop = O._flatten_pol(itf["Rtop"][l], 0); Ttop = O._flatten_pol(itf["Ttop"][l], 0)
            if l == L-1: Rt = np.zeros((N, N))
            M1 = F - Rt @ G; rhs = Rt @ F - G
            Q = np.linalg.solve(M1, rhs); Qn, g1 = solve_blockpiv(M1, rhs)
            tQt = t[:, None]*Q*t[None, :]; Y = F @ tQt + G; W = (G - Rtop[:, None]*F) @ tQt + (F - Rtop[:, None]*G)
            K = np.linalg.solve(W.T, Y.T).T; Kn, g2 = solve_blockpiv(W.T, Y.T); Kn = Kn.T
            worst[0] = max(worst[0], g1); worst[1] = max(worst[1], g2)
            worst[2] = max(worst[2], np.abs(Qn-Q).max()/np.abs(Q).max()); worst[3] = max(worst[3], np.abs(Kn-K).max()/np.abs(K).max())
            if l > 0:
                Nu = st.n[l-1]*2; Rb = O._flatten_pol(itf["Rbot"][l-1], 0); Tb = O._flatten_pol(itf["Tbot"][l-1], 0)
                nc = min(N, Nu); Rn = np.diag(Rb).astype(float); Rn[:nc, :nc] += (Ttop[:nc, None]*K[:nc, :nc])*Tb[None, :nc]; Rt = Rn
    print(name, mode, "blockpiv16: growth M1 %.2f W %.2f  relerr Q %.2e K %.2e" % tuple(worst))
print()
for nm in ["cfg2_iba_L20_n32_sp1", "dmrt_L8_n16", "iba_L3_n16_shallow"]:
    for mode in ["jac_natural", "jac_sorted", "svd_sorted"]: run2(nm, mode)

def pairfix(F, G, beta):
    F = F.copy(); G = G.copy(); beta = beta.copy()
    for i in range(0, len(beta) - 1, 2):
        if abs(F[i, i+1]*F[i+1, i]) > abs(F[i, i]*F[i+1, i+1]):
            F[:, [i, i+1]] = F[:, [i+1, i]]; G[:, [i, i+1]] = G[:, [i+1, i]]; beta[[i, i+1]] = beta[[i+1, i]]
    return F, G, beta
def run3(name, mode):
    d = load_golden(name); sp = snowpack_dict(d)
    worst = [0, 0, 0, 0]
    for i, f in enumerate(d["frequency"]):
        ems = O.make_layers(str(d["emmodel"]), float(f), sp); eps = np.array([e.eps_eff for e in ems])
        st = O.compute_streams(fixture_options(d)["n_max_stream"], eps); itf = O.interface_diagonals(eps, st, 2)
        L = len(ems); thick = np.asarray(sp["thickness"], float)
        Rt = None
        for l in range(L-1, -1, -1):
            em, mu, w = ems[l], st.mu[l], st.weight[l]
            n = len(mu); N = 2*n
            full = np.concatenate((mu, -mu)); Pc = O.compress(em.ft_even_phase(mu, full, 0, 2)[:, :, 0]); Pp, Pm = Pc[:, :N], Pc[:, N:]
            wv = np.repeat(w, 2); mv = np.repeat(mu, 2)
            norm = em.ks/(0.5*((Pp+Pm)*wv[None, :]).sum(axis=1)); ke = em.ks+em.ka; sc = np.sqrt(norm*wv)
            Xp = (ke*np.eye(N) - 0.5*sc[:, None]*(Pp+Pm)*sc[None, :])/np.sqrt(mv[:, None]*mv[None, :]); Xm = (ke*np.eye(N) - 0.5*sc[:, None]*(Pp-Pm)*sc[None, :])/np.sqrt(mv[:, None]*mv[None, :])
            Lp = np.linalg.cholesky(0.5*(Xp+Xp.T)); Lm = np.linalg.cholesky(0.5*(Xm+Xm.T)); B = Lp.T @ Lm
            if mode.startswith("jac"):
                Bp = jacobi_natural(B); beta = np.linalg.norm(Bp, axis=0)
            else:
                U, S, Vt = np.linalg.svd(B); Bp = U*S[None, :]; beta = S
            if "sorted" in mode:
                o = np.argsort(beta, kind="stable"); Bp, beta = Bp[:, o], beta[o]
            dd = np.sqrt(norm/wv)/np.sqrt(mv)
            Ep = dd[:, None]*np.linalg.solve(Lp.T, Bp); Em = -dd[:, None]*(Lp @ Bp)/beta[None, :]
            F = 0.5*(Ep-Em); G = 0.5*(Ep+Em)
            if "fix" in mode: F, G, beta = pairfix(F, G, beta)
            t = np.exp(-beta*thick[l]); Rtop = O._flatten_pol(itf["Rtop"][l], 0); Ttop = O._flatten_pol(itf["Ttop"][l], 0)
            if l == L-1: Rt = np.zeros((N, N))
            M1 = F - Rt @ G; rhs = Rt @ F - G
            Q = np.linalg.solve(M1, rhs); Qn = solve_nopiv(M1, rhs); g1 = lu_nopiv_growth(M1)[0]
            tQt = t[:, None]*Q*t[None, :]; Y = F @ tQt + G; W = (G - Rtop[:, None]*F) @ tQt + (F - Rtop[:, None]*G)
            K = np.linalg.solve(W.T, Y.T).T; Kn = solve_nopiv(W.T, Y.T).T; g2 = lu_nopiv_growth(W.T)[0]
            worst[0] = max(worst[0], g1); worst[1] = max(worst[1], g2)
            worst[2] = max(worst[2], np.abs(Qn-Q).max()/np.abs(Q).max()); worst[3] = max(worst[3], np.abs(Kn-K).max()/np.abs(K).max())
            if l > 0:
                Nu = st.n[l-1]*2; Rb = O._flatten_pol(itf["Rbot"][l-1], 0); Tb = O._flatten_pol(itf["Tbot"][l-1], 0)
                nc = min(N, Nu); Rn = np.diag(Rb).astype(float); Rn[:nc, :nc] += (Ttop[:nc, None]*K[:nc, :nc])*Tb[None, :nc]; Rt = Rn
    print(name, mode, "NO pivoting: growth M1 %.2f W %.2f  relerr Q %.2e K %.2e" % tuple(worst))
print()
for nm in ["cfg2_iba_L20_n32_sp1", "cfg2_iba_L20_n32_sp0", "dmrt_L8_n16", "iba_L3_n16_shallow", "iba_L6_n8_angles"]:
    for mode in ["jac_natural", "jac_sorted", "jac_sorted_fix", "svd_sorted_fix"]: run3(nm, mode)
