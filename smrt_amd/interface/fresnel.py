"""Fresnel coefficients between two lossy media on the host (NumPy), for the interface models that are evaluated in
Python (rough interfaces and substrates: iem_fung92.py, geometrical_optics.py).  The Flat interfaces of the DORT path
never come here: the device evaluates the same expressions per stream (csrc/dort_physics.hpp: fresnel_RvRh, fresnel_RT3).

Field coefficients after Maezawa & Miyauchi (2009), "Rigorous expressions for the Fresnel equations at interfaces between
absorbing media", JOSA A 26(2) eqs. 8, 59, 61 -- the formulation smrt/core/fresnel.py:99-146 uses; power 'diagonals' per
polarisation (V, H and the coherency term U of the third Stokes component, Tsang et al. 2000 vol. I eqs. 7.2.93 / 7.2.95)
as smrt/core/fresnel.py:417-474 lays them out: one row per polarisation, one column per cosine."""
import numpy as np


def field_reflection(eps_1, eps_2, mu):
    """(r_v, r_h, mu_2): field reflection coefficients seen from medium 1 at the cosines `mu` (in medium 1), and the
    cosine of the refracted direction in medium 2.  The tangential wavenumber is conserved (Snell); the normal
    components carry the losses."""
    eps_1, eps_2 = complex(eps_1), complex(eps_2)
    mu = np.asarray(mu, dtype=np.float64)
    n1 = np.sqrt(eps_1)
    tangential2 = n1.real ** 2 * (1.0 - mu * mu)
    k_in = -np.sqrt(eps_1 - tangential2 + 0j)
    k_out = -np.sqrt(eps_2 - tangential2 + 0j)
    r_h = (k_in - k_out) / (np.conj(k_in) + k_out)
    r_v = np.conj(n1) * (eps_2 * k_in - eps_1 * k_out) / (n1 * (eps_2 * np.conj(k_in) + np.conj(eps_1) * k_out))
    return r_v, r_h, -k_out.real / np.sqrt(eps_2).real


def _abs2(z):
    return z.real * z.real + z.imag * z.imag


def reflection_diagonal(eps_1, eps_2, mu, npol):
    """[npol, len(mu)] power reflection: |r_v|^2, |r_h|^2 and, for npol = 3, Re(r_v conj r_h)."""
    mu = np.atleast_1d(np.asarray(mu, float))
    r_v, r_h, _ = field_reflection(eps_1, eps_2, mu)
    out = np.ones((npol, len(mu)))
    out[0], out[1] = _abs2(r_v), _abs2(r_h)
    if npol >= 3:
        out[2] = (r_v * np.conj(r_h)).real
    return out


def transmission_diagonal(eps_1, eps_2, mu, npol):
    """[npol, len(mu)] power transmission: 1 - |r_v|^2, 1 - |r_h|^2 and, for npol = 3, (mu_2 / mu) Re((1 + r_v) conj(1 + r_h))."""
    mu = np.atleast_1d(np.asarray(mu, float))
    r_v, r_h, mu_2 = field_reflection(eps_1, eps_2, mu)
    out = np.zeros((npol, len(mu)))
    out[0], out[1] = 1.0 - _abs2(r_v), 1.0 - _abs2(r_h)
    if npol >= 3:
        out[2] = mu_2 / mu * ((1.0 + r_v) * np.conj(1.0 + r_h)).real
    return out
