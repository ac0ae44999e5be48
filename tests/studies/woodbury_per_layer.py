"""Numerical study (NOT a test): the layer step of the pivot-free recursion with ONE inversion inside the layer (Woodbury
on M3, tests/studies/admittance_recursion.py: two_inversions) instead of two -- per layer, only where it is safe.

Theta = diag((1 + t^2) / (Sigma (1 - t^2))) - 4 G (H + Sigma (1 + t^2) / (1 - t^2))^-1 G,  G = t / (1 - t^2), t = exp(-sigma d),
is a difference of O(1 / (sigma d)) terms: digits are lost in proportion to 1 / (sigma d) for optically thin layers.  The
question: with the shortcut taken only when min_i sigma_i d >= tau, what does it cost in brightness temperature on the hard
inputs, as a function of tau?

    python tests/studies/woodbury_per_layer.py

What the study did NOT model, and the kernel found: the device's 16 x 16 elimination writes the diagonal of an inverse
through a unit-vector trick that costs |pivot|^2 ulps (dort_finish_reg.hpp: inv16_step).  With pivots of the size of
Sigma / (sigma d) and the 1 / (sigma d)^2 amplification of the form on top, the first kernel build was wrong by 2.5e-4 K on
the hard inputs where this study (inv_nopiv: plain IEEE Gauss-Jordan) says 1e-11.  The kernel now inverts I + s H s,
s = (Sigma (1 + t^2) / (1 - t^2))^-1/2 -- unit pivots -- and agrees with the study (dort_finish_strip.hpp).
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
import symmetric_eigen_route as SE  # noqa: E402

TAU = [1e9]
COUNT = [0, 0]


def solve_pair(sp, frequency, theta_deg, n_max_stream=32, substrate=None, atmosphere=None):
    """admittance_recursion.solve_pair with the choice made per layer."""
    ems = O.make_layers("iba", frequency, sp)
    eps = np.array([e.eps_eff for e in ems])
    thick = np.asarray(sp["thickness"], float)
    st = O.compute_streams(n_max_stream, eps)
    itf = O.interface_diagonals(eps, st, 2, substrate)
    L = len(ems)
    BT = [O.planck(frequency, float(t)) for t in sp["temperature"]]
    flat = lambda a: O._flatten_pol(a, 0)  # noqa: E731
    Rs = flat(itf["Rbot"][L - 1])
    src = np.zeros_like(Rs)
    if substrate is not None and substrate.get("temperature") is not None:
        src = flat(itf["Tbot"][L - 1]) * O.planck(frequency, float(substrate["temperature"]))
    Cdiag = (1.0 - Rs) / (1.0 + Rs)
    C = np.diag(Cdiag)
    c = (Cdiag + 1.0) * src
    inverse = AR.inv_nopiv
    for l in range(L - 1, -1, -1):
        S, Ap, Am, d = AR.layer_eigen(ems[l], st.mu[l], st.weight[l])
        N = len(S)
        t = np.exp(-S * thick[l])
        Bl = BT[l]
        Chat = C * (d[None, :] / d[:, None])
        chat = c / d
        one_hat = 1.0 / d
        H = Ap.T @ (Chat @ Ap)
        r = Ap.T @ (chat - 2.0 * Bl * (Chat @ one_hat))
        short = (S * thick[l]).min() >= TAU[0]
        COUNT[short] += 1
        if short:
            omt2 = -np.expm1(-2.0 * S * thick[l])
            G = t / omt2
            K2i = inverse(H + np.diag(S * (1.0 + t * t) / omt2))
            Theta = np.diag((1.0 + t * t) / (S * omt2)) - 4.0 * G[:, None] * K2i * G[None, :]
            Chat_top = Am @ Theta @ Am.T
            chat_top = 2.0 * Bl * (Chat_top @ one_hat) - 2.0 * (Am @ (G * (K2i @ r)))
        else:
            Pm_ = inverse(H + np.diag(S))
            q = Pm_ @ r
            st_ = S * t
            M3 = np.diag(S * (1.0 - t * t)) + 2.0 * st_[:, None] * Pm_ * st_[None, :]
            M3i = inverse(M3)
            Theta = 2.0 * M3i - np.diag(1.0 / S)
            Chat_top = Am @ Theta @ Am.T
            chat_top = 2.0 * Bl * (Chat_top @ one_hat) - 2.0 * (Am @ (M3i @ (st_ * q)))
        Ctop = Chat_top * (d[:, None] / d[None, :])
        ctop = chat_top * d
        if l == 0:
            break
        Nu = 2 * st.n[l - 1]
        nc = min(N, Nu)
        r2 = flat(itf["Rtop"][l]); t1 = flat(itf["Ttop"][l])
        r1 = np.ones(N); t2 = np.zeros(N)
        r1[:nc] = flat(itf["Rbot"][l - 1])[:nc]; t2[:nc] = flat(itf["Tbot"][l - 1])[:nc]
        t1 = t1.copy(); t1[nc:] = 0.0
        tt = t1 * t2
        a_ = 0.5 * (tt + (1 + r1) * (1 - r2)); b_ = 0.5 * (tt - (1 + r1) * (1 + r2))
        c_ = 0.5 * (tt - (1 - r1) * (1 - r2)); d_ = 0.5 * (tt + (1 - r1) * (1 + r2))
        Y = np.diag(a_) - b_[:, None] * Ctop
        Yi = inverse(Y)
        Nn = np.diag(c_) - d_[:, None] * Ctop
        Z = Nn @ Yi
        Cu = np.zeros((Nu, Nu)); cu = np.zeros(Nu)
        Cu[:nc, :nc] = -(Z[:nc, :nc] * t2[None, :nc]) / t2[:nc, None]
        cu[:nc] = ((d_ * ctop - Z @ (b_ * ctop))[:nc]) / t2[:nc]
        if Nu > nc:
            rb = flat(itf["Rbot"][l - 1])[nc:]
            Cu[np.arange(nc, Nu), np.arange(nc, Nu)] = (1.0 - rb) / (1.0 + rb)
        C, c = Cu, cu
    N0 = len(ctop)
    r2 = flat(itf["Rtop"][0]); t1 = flat(itf["Ttop"][0])
    Rair = flat(itf["Rbot_air"]); Tair = flat(itf["Tbot_air"])
    na = len(Rair)
    Isky = 0.0 if atmosphere is None else O.planck(frequency, float(atmosphere["tb_down"]))
    t2 = np.zeros(N0); t2[:na] = Tair
    Smat = np.diag(1.0 - r2) + Ctop * (1.0 + r2)[None, :]
    rhs = ctop + (np.eye(N0) - Ctop) @ (t2 * Isky)
    Iup = inverse(Smat) @ rhs
    I0 = Rair * Isky + (t1 * Iup)[:na]
    if atmosphere is not None:
        I0 = O.planck(frequency, float(atmosphere["tb_up"])) + atmosphere["transmittance"] * I0
    tb = O.inverse_planck(frequency, I0).reshape(st.n_air, 2).T
    return O.interpolate_passive(st.outmu, tb, np.cos(np.deg2rad(np.atleast_1d(theta_deg))))


if __name__ == "__main__":
    cases = list(SE.hard_cases(1, 12)) + list(SE.hard_cases(2, 12)) + list(SE.headline_cases(1))
    refs = []
    for sp, f, theta, nstr, sub, atm in cases:
        try:
            refs.append(O.solve(sp, f, theta, n_max_stream=nstr, substrate=sub, atmosphere=atm))
        except O.OracleError:
            refs.append(None)
    for tau in (1e9, 1.0, 0.3, 0.1, 0.03, 0.01, 1e-3, 1e-4, 0.0):
        TAU[0] = tau; COUNT[0] = COUNT[1] = 0
        worst = 0.0
        for (sp, f, theta, nstr, sub, atm), ref in zip(cases, refs):
            if ref is None:
                continue
            got = solve_pair(sp, f, theta, n_max_stream=nstr, substrate=sub, atmosphere=atm)
            worst = max(worst, float(np.abs(got - ref).max()))
        print("tau = %-8g  max |dTb| = %.2e K   layers on the shortcut: %d of %d" % (tau, worst, COUNT[1], COUNT[0] + COUNT[1]), flush=True)
