"""CPU ORACLE for the DORT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy/SciPy restatement of the reference algorithm (smrt-model/smrt, path relative to /root/reference) for one
(snowpack, frequency) pair: IBA / DMRT-QCA-short-range layer electromagnetics, Gauss-Legendre streams with Snell
propagation, Flat (Fresnel) interfaces, per-layer eigen-decomposition of the discrete-ordinate matrix, the banded
boundary-condition system and its solution, mode summation, Planck inversion and interpolation to the sensor angles.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module, and only as the
checker / the timed CPU baseline.  The product (smrt_amd) never imports it and has no CPU fallback.

Parity pinning: tests/test_oracle_golden.py checks every function below against fixtures generated from the real
reference in the build container (tests/golden/make_golden.py), including the reference's own known answers
(smrt/test/test_integration_iba.py:48-49,67-69; examples/iba_onelayer_example.py).

Written from the equations/conventions the reference fixes; every function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.special import roots_legendre

# smrt/core/globalconstants.py:24-43
C_SPEED = 299792458.0
PLANCK_CONSTANT = 6.62607015e-34
BOLTZMANN_CONSTANT = 1.380649e-23
DENSITY_OF_ICE = 916.7
FREEZING_POINT = 273.15


class OracleError(Exception):
    """Numerical failure of one solve (the reference raises SMRTError, smrt/core/error.py:6-9)."""

    def __init__(self, msg, status=1):
        super().__init__(msg)
        self.status = status


# ----------------------------------------------------------------------------------------------------------------
# permittivity and microstructure
# ----------------------------------------------------------------------------------------------------------------
def ice_permittivity_maetzler06(frequency, temperature):
    """Pure-ice permittivity, smrt/permittivity/ice.py:52-73 (dry branch of wetice.py:12-41)."""
    f_ghz = frequency * 1e-9
    t_c = temperature - FREEZING_POINT
    e_real = 3.1884 + 9.1e-4 * t_c
    theta = 300.0 / temperature - 1.0
    alpha = (0.00504 + 0.0062 * theta) * np.exp(-22.1 * theta)
    eb = np.exp(335.0 / temperature)
    beta_m = (0.0207 / temperature) * (eb / (eb - 1.0) ** 2) + 1.16e-11 * f_ghz**2
    delta_beta = np.exp(-9.963 + 0.0372 * t_c)
    return e_real + 1j * (alpha / f_ghz + (beta_m + delta_beta) * f_ghz)


def water_permittivity_maetzler87(frequency, temperature):
    """Liquid water, double Debye model of Maetzler & Wegmuller 1987: smrt/permittivity/water.py:14-43."""
    if temperature < FREEZING_POINT:
        raise OracleError("The water temperature must be higher or equal to the freezing point", 5)
    f_ghz = frequency * 1e-9
    theta = 1.0 - 300.0 / temperature
    e0 = 77.66 - 103.3 * theta
    e1 = 0.0671 * e0
    f1 = 20.2 + 146.4 * theta + 316.0 * theta**2
    e2 = 3.52 + 7.52 * theta
    f2 = 39.8 * f1
    return e2 + (e1 - e2) / complex(1, -f_ghz / f2) + (e0 - e1) / complex(1, -f_ghz / f1)


def scatterer_permittivity(frequency, temperature, liquid_water=0.0):
    """Permittivity of the ice grains of a snow layer, the reference's default model wetice_permittivity_bohren83
    (smrt/permittivity/wetice.py:12-45): pure ice when dry; when wet, ice inclusions (1 - liquid_water) in a water host
    mixed by Maxwell Garnett (smrt/permittivity/generic_mixing_formula.py:352-380)."""
    eps_ice = ice_permittivity_maetzler06(frequency, temperature)
    if not liquid_water > 0.0:
        return eps_ice
    e0 = water_permittivity_maetzler87(frequency, temperature)
    c_plus = eps_ice + 2 * e0
    c_minus = (eps_ice - e0) * (1.0 - liquid_water)
    return (c_plus + 2 * c_minus) / (c_plus - c_minus) * e0


def polder_van_santen_spheres(frac_volume, e0, eps):
    """Positive root of 2x^2 + bx - eps*e0 = 0, smrt/permittivity/generic_mixing_formula.py:117-145."""
    b = eps - 2.0 * e0 - 3.0 * frac_volume * (eps - e0)
    return (-b + np.sqrt(b * b + 8.0 * eps * e0)) / 4.0


def ft_autocorr_exponential(k, frac_volume, corr_length):
    """smrt/microstructure_model/exponential.py:53-58."""
    x = (k * corr_length) ** 2
    return frac_volume * (1.0 - frac_volume) * 8.0 * np.pi * corr_length**3 / (1.0 + x) ** 2


def ft_autocorr_independent_sphere(k, frac_volume, radius):
    """smrt/microstructure_model/independent_sphere.py:54-72."""
    x = radius * np.asarray(k)                    # (k may be complex: the strong-contrast-expansion emmodels)
    bessel = np.ones_like(x)
    nz = ~np.isclose(x, 0)
    bessel[nz] = 9 * ((np.sin(x[nz]) - x[nz] * np.cos(x[nz])) / x[nz] ** 3) ** 2
    return frac_volume * (1.0 - frac_volume) * 4.0 / 3 * np.pi * radius**3 * bessel


def ft_autocorr_teubner_strey(k, frac_volume, corr_length, repeat_distance):
    """smrt/microstructure_model/teubner_strey.py:45-55."""
    x = (np.asarray(k) * corr_length) ** 2          # (k may be complex: the strong-contrast-expansion emmodels)
    y = (2 * np.pi * corr_length / repeat_distance) ** 2
    return frac_volume * (1.0 - frac_volume) * 8 * np.pi * corr_length**3 / ((1 + y) ** 2 + 2 * (1 - y) * x + x**2)


# ---- the models on the unified parameters (porod_length, polydispersity): smrt/microstructure_model/unified_*.py ---------
def ft_autocorr_unified_scaled_exponential(k, frac_volume, porod_length, polydispersity):
    """smrt/microstructure_model/unified_scaled_exponential.py:26-35 (corr_length = polydispersity * porod_length)."""
    corr_length = polydispersity * porod_length
    x = (np.asarray(k) * corr_length) ** 2          # (k may be complex: the strong-contrast-expansion emmodels)
    return frac_volume * (1.0 - frac_volume) * 8 * np.pi * corr_length**3 / (1.0 + x) ** 2


def ft_autocorr_unified_teubner_strey(k, frac_volume, porod_length, polydispersity):
    """smrt/microstructure_model/unified_teubner_strey.py:24-36 (the two lengths) and :63-78 (the transform), as written
    there: two Lorentzians from polydispersity 1 on, the factored Teubner-Strey denominator below."""
    k = np.asarray(k)                                # (may be complex: the strong-contrast-expansion emmodels)
    k32 = polydispersity ** (3 / 2)
    if polydispersity >= 1:
        b = porod_length * k32
        delta = np.sqrt(1 - 1 / k32)
        z1, z2 = b * (1 - delta), b * (1 + delta)
        ft = (4 * np.pi * z1 * z2 * (z1 + z2)) / ((1 + (z1 * k) ** 2) * (1 + (z2 * k) ** 2))
    else:
        z1 = porod_length
        z2 = porod_length * np.sqrt(1 / (1 / k32 - 1))
        x1, r12 = k * z1, z1 / z2
        ft = 8 * np.pi * z1**3 / ((1 + (x1 - r12) ** 2) * (1 + (x1 + r12) ** 2))
    return frac_volume * (1.0 - frac_volume) * ft


def ft_autocorr_unified_shs(k, frac_volume, porod_length, polydispersity):
    """smrt/microstructure_model/unified_sticky_hard_spheres.py:22-27 (radius and t from the unified parameters), :43-76."""
    f = frac_volume
    radius = 3 / 4 * porod_length / (1 - f)
    t = (1 + 2 * f - 3 / (8 * np.sqrt(2)) * polydispersity ** (-3 / 2)) / (f * (1.0 - f))
    return ft_autocorr_shs(k, f, radius, None, t=t)


def shs_t_parameter(frac_volume, stickiness):
    """Tsang vol II eq 8.4.22 root selection, smrt/microstructure_model/sticky_hard_spheres.py:132-167."""
    f = frac_volume
    if np.isinf(stickiness):
        return 0.0
    a = f / 12.0
    b = -(stickiness + f / (1.0 - f))
    c = (1.0 + f / 2.0) / (1.0 - f) ** 2
    disc = b * b - 4.0 * a * c
    if disc < 0:
        raise OracleError("negative discriminant for the SHS t parameter")
    t = (-b - np.sqrt(disc)) / (2.0 * a)
    if t * f * (1.0 - f) > 1.0 + 2.0 * f:
        t = (-b + np.sqrt(disc)) / (2.0 * a)
    if t * f * (1.0 - f) > 1.0 + 2.0 * f:
        raise OracleError("no solution for the SHS t parameter")
    return t


def ft_autocorr_shs(k, frac_volume, radius, stickiness, t=None):
    """Percus-Yevick sticky-hard-sphere structure factor form, sticky_hard_spheres.py:63-130; with `t` given the same
    expression as smrt/microstructure_model/unified_sticky_hard_spheres.py:43-76 evaluates it (its t comes from the
    polydispersity, :24-27, not from a stickiness)."""
    f, tau = frac_volume, stickiness
    x = np.atleast_1d(np.asarray(k)) * radius    # (k may be complex: the strong-contrast-expansion emmodels)
    if t is not None:
        pass
    elif np.isfinite(tau) and f > 0:
        t = (
            6 * tau * f - 6 * f - 6 * tau
            + (36 * tau**2 * f**2 - 72 * tau * f**2 - 72 * tau**2 * f + 30 * f**2 + 72 * tau * f + 36 * tau**2
               - 12 * f) ** 0.5
        ) / (f * (f - 1.0))
    else:
        t = 0.0
    vd = 4.0 / 3.0 * np.pi * radius**3
    small = np.isclose(x, 0, atol=1e-3)
    xs = np.where(small, 1.0, x)
    vint = np.where(small, 1.0, 3.0 * (np.sinc(xs / np.pi) - np.cos(xs)) / xs**2)
    psi = np.sinc(x / np.pi) / vint
    a_ = f / (1 - f) * ((1 - t * f + 3 * f / (1 - f)) + (3 - t * (1 - f)) * psi) + np.cos(x) / vint
    b_ = f / (1 - f) * x + np.sin(x) / vint
    c_tilde = f * vd / (a_**2 + b_**2)
    c0 = f * vd / (f / (1 - f) * ((1 - t * f + 3 * f / (1 - f)) + (3 - t * (1 - f))) + 1) ** 2
    return np.where(small, c0, c_tilde).reshape(np.shape(k))


UNIFIED_FT = {"unified_scaled_exponential": ft_autocorr_unified_scaled_exponential,
              "unified_teubner_strey": ft_autocorr_unified_teubner_strey,
              "unified_sticky_hard_spheres": ft_autocorr_unified_shs}


# ----------------------------------------------------------------------------------------------------------------
# layer electromagnetic models
# ----------------------------------------------------------------------------------------------------------------
class LayerEM:
    """What the rtsolver needs from an emmodel instance (contract at smrt/emmodel/iba.py:36-40)."""

    kind: str
    eps_eff: complex
    ks: float
    ka: float

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):  # -> (npol, npol, m_max+1, len(mu_s), len(mu_i))
        raise NotImplementedError


class IBALayer(LayerEM):
    """Improved Born approximation, smrt/emmodel/iba.py:85-265 (spherical depolarisation factors 1/3)."""

    kind = "iba"

    dense_snow_correction = None

    def __init__(self, frequency, frac_volume, temperature, microstructure, **mp):
        self.frequency = frequency
        self.k0 = 2.0 * np.pi * frequency / C_SPEED
        e0 = 1.0
        eps = scatterer_permittivity(frequency, temperature, mp.get("liquid_water", 0.0))
        self.eps_ice = eps
        if frac_volume > 0.5 and self.dense_snow_correction == "auto":
            # iba.py:95-96 -> core/layer.py:186-201, microstructure_model/autocorrelation.py:146-153: the inverted
            # medium -- same autocorrelation family with frac_volume -> 1 - frac_volume, permittivities swapped
            frac_volume, e0, eps = 1.0 - frac_volume, eps, e0
        self.f = frac_volume
        self.eps_eff = self.mixing(frac_volume, e0, eps)  # emmodel/common.py:269-289
        if microstructure == "exponential":
            lc = mp["corr_length"]
            self.ft_corr = lambda k: ft_autocorr_exponential(k, frac_volume, lc)
        elif microstructure == "sticky_hard_spheres":
            r, tau = mp["radius"], mp["stickiness"]
            self.ft_corr = lambda k: ft_autocorr_shs(k, frac_volume, r, tau)
        elif microstructure == "independent_sphere":
            r = mp["radius"]
            self.ft_corr = lambda k: ft_autocorr_independent_sphere(np.atleast_1d(k), frac_volume, r).reshape(np.shape(k))
        elif microstructure == "teubner_strey":
            lc, rd = mp["corr_length"], mp["repeat_distance"]
            self.ft_corr = lambda k: ft_autocorr_teubner_strey(k, frac_volume, lc, rd)
        elif microstructure in UNIFIED_FT:
            lp, pk, ft = mp["porod_length"], mp["polydispersity"], UNIFIED_FT[microstructure]
            self.ft_corr = lambda k: ft(k, frac_volume, lp, pk)
        else:
            raise ValueError(microstructure)
        # mean squared field ratio with depolarisation 1/3 on each axis (iba.py:152-162)
        depol = 1.0 / 3.0
        app = self.apparent_permittivity(e0, depol)
        y2 = abs(app / (app + (eps - e0) * depol)) ** 2
        self.iba_coeff = (1.0 / (4.0 * np.pi)) * abs(eps - e0) ** 2 * y2 * self.k0**4  # iba.py:148-150
        self.ka = self.absorption(frac_volume, eps, y2)
        # ks by Romberg on 65 samples of mu in [1,-1] (iba.py:176-226); note abs(sqrt(eps_eff)) here
        mu = np.linspace(1.0, -1.0, 65)
        kd = 2.0 * self.k0 * np.sqrt((1.0 - mu) / 2.0) * abs(np.sqrt(self.eps_eff))
        y = (self.iba_coeff * self.ft_corr(kd)).real * (mu**2 + 1.0)
        self.ks = romberg65(y, mu[0] - mu[1]) / 4.0

    # the three places where the other members of IBA's family differ (iba_original.py, iba_maxwell_garnett.py)
    mixing = staticmethod(polder_van_santen_spheres)

    def apparent_permittivity(self, e0, depol):
        return self.eps_eff * (1.0 - depol) + e0 * depol           # iba.py:152-162

    def absorption(self, frac_volume, eps, y2):
        return 2.0 * self.k0 * np.sqrt(self.eps_eff).imag          # iba.py:265

    def phase(self, mu_s, mu_i, dphi, npol):
        """iba.py:228-244: Rayleigh matrix times the microstructure term at the half scattering angle."""
        p, sin_half = rayleigh_matrix_and_half_angle(mu_s, mu_i, dphi, npol)
        kd = 2.0 * self.k0 * np.sqrt(self.eps_eff).real * sin_half
        return self.ft_corr(kd) * self.iba_coeff * p

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        nsamples = int(2 ** np.ceil(4 + np.log(m_max + 1) / np.log(2)))  # emmodel/common.py:401-414
        return ft_even_matrix(lambda dphi: self.phase(mu_s, mu_i, dphi, npol), m_max, nsamples, npol)


class IBADenseAutoLayer(IBALayer):
    """IBA with emmodel_options=dict(dense_snow_correction="auto") (iba.py:85-105)."""

    dense_snow_correction = "auto"


class IBAOriginalLayer(IBALayer):
    """smrt/emmodel/iba_original.py:29-44 (Maetzler 1998): IBA with the absorption k0 f Im(eps) |y2| instead of the one of
    the effective medium."""

    kind = "iba_original"

    def absorption(self, frac_volume, eps, y2):
        return self.k0 * frac_volume * complex(eps).imag * abs(y2)


def maxwell_garnett_spheres(frac_volume, e0, eps):
    """smrt/permittivity/generic_mixing_formula.py:361-380."""
    cplus = eps + 2 * e0
    cminus = (eps - e0) * frac_volume
    return (cplus + 2 * cminus) / (cplus - cminus) * e0


class IBAMaxwellGarnettLayer(IBALayer):
    """smrt/emmodel/iba_maxwell_garnett.py:36-52: Maxwell Garnett mixing and the apparent permittivity of the background."""

    kind = "iba_maxwell_garnett"
    mixing = staticmethod(maxwell_garnett_spheres)

    def apparent_permittivity(self, e0, depol):
        return e0


def romberg_pow2(y, dx):
    """scipy.integrate.romb on 2**k + 1 samples (romberg65 for any k)."""
    n = len(y) - 1
    k = int(round(np.log2(n)))
    assert 2**k == n
    r = np.array([(n >> i) * dx * (0.5 * (y[:: n >> i][0] + y[:: n >> i][-1]) + y[:: n >> i][1:-1].sum()) for i in range(k + 1)])
    for j in range(1, k + 1):
        r = (4.0**j * r[1:] - r[:-1]) / (4.0**j - 1.0)
    return r[0]


def sce_a2_nonlocal(q_big, ft_corr):
    """smrt/emmodel/sce_common.py:284-320 (compute_A2_nonlocal): the second-order coefficient of the strong-contrast
    expansion in its non-local form, from the Fourier transform of the autocorrelation function -- a cumulative trapezoid
    for Im F, a Romberg integral with the two singularities taken out and an asymptotic tail for Re F."""
    margin, k = 4, 12
    n = 2**k
    nq = n // margin
    maxq = margin * q_big
    q = np.linspace(0, maxq, n + 1)
    y = 2 * q * ft_corr(2 * q)
    x = 2 * q.real
    primitive = np.concatenate(([0.0], np.cumsum(0.5 * (y.real[1:] + y.real[:-1]) * np.diff(x))))   # cumulative_trapezoid, initial=0
    im_f = -1 / (2 * (2 * np.pi) ** 1.5) * q * primitive
    with np.errstate(invalid="ignore", divide="ignore"):
        y1 = im_f / ((q_big + q) * q)
        y1[0] = 0
        y2 = (im_f - im_f[nq]) / (q_big**2 - q**2)
        y2[nq] = (y2[nq - 1] + y2[nq + 1]) / 2
    yy = y1 + y2
    tail = (im_f[nq] - q_big / maxq * im_f[-1]) * np.log(np.abs((maxq + q_big) / (maxq - q_big)))
    re_f = -2 / np.pi * q_big * romberg_pow2(yy.real, maxq.real / n) - 1 / np.pi * tail
    return -(2 * np.pi) / (2**1.5 * 0.5 * np.sqrt(np.pi)) * (re_f + 1j * im_f[nq])


class SymSCELayer(LayerEM):
    """Symmetrised strong-contrast expansion, Torquato & Kim 2021 as smrt implements it (smrt/emmodel/symsce_torquato21.py:
    37-45 -> sce_common.py:24-62 with local=False, symmetrical=True, scaled=True): the effective permittivity of Polder-van
    Santen, ke / ks from the quadratic of compute_ke_ks_symmetrical (:95-114) with A2 of the medium and of its inverse at the
    wavenumber of the effective medium (:64-77), ka = 2 k0 Im sqrt(eps_eff) (:236-247), and IBA's phase function normalised to
    ks (:145-160) -- evaluated at the COMPLEX wavenumber 2 k0 sqrt(eps_eff) sin(Theta / 2) (:222), of which the Fourier
    decomposition keeps the real part (emmodel/common.py:107-117)."""

    kind = "symsce_torquato21"

    def __init__(self, frequency, frac_volume, temperature, microstructure, **mp):
        self.frequency = frequency
        self.k0 = 2.0 * np.pi * frequency / C_SPEED
        e0 = 1.0
        eps = scatterer_permittivity(frequency, temperature, mp.get("liquid_water", 0.0))
        f = self.f = frac_volume
        self.eps_eff = polder_van_santen_spheres(f, e0, eps)

        def ft_of(fv):
            if microstructure == "exponential":
                return lambda k: ft_autocorr_exponential(k, fv, mp["corr_length"])
            if microstructure == "unified_scaled_exponential":
                return lambda k: ft_autocorr_unified_scaled_exponential(k, fv, mp["porod_length"], mp["polydispersity"])
            if microstructure == "teubner_strey":
                return lambda k: ft_autocorr_teubner_strey(k, fv, mp["corr_length"], mp["repeat_distance"])
            if microstructure == "sticky_hard_spheres":
                return lambda k: ft_autocorr_shs(k, fv, mp["radius"], mp["stickiness"])
            if microstructure == "unified_sticky_hard_spheres":
                # (the inverted medium of the reference is a COPY with frac_volume -> 1 - f, autocorrelation.py:146-153: the
                # radius and the t parameter this model derives from f at construction stay those of the medium itself)
                radius = 3 / 4 * mp["porod_length"] / (1 - f)
                t = (1 + 2 * f - 3 / (8 * np.sqrt(2)) * mp["polydispersity"] ** (-3 / 2)) / (f * (1.0 - f))
                return lambda k: ft_autocorr_shs(k, fv, radius, None, t=t)
            if microstructure == "independent_sphere":
                return lambda k: ft_autocorr_independent_sphere(np.atleast_1d(k), fv, mp["radius"]).reshape(np.shape(k))
            if microstructure == "unified_teubner_strey":
                return lambda k: ft_autocorr_unified_teubner_strey(k, fv, mp["porod_length"], mp["polydispersity"])
            raise ValueError(microstructure)

        self.ft_corr = ft_of(f)
        k_sym = self.k0 * np.sqrt(self.eps_eff)
        a2, a2inv = sce_a2_nonlocal(k_sym, ft_of(f)), sce_a2_nonlocal(k_sym, ft_of(1.0 - f))   # (inverted_medium: 1 - f)
        big = 2 if f in (0, 1) else 2 + a2 / f + a2inv / (1 - f)
        se, pe = e0 + eps, e0 * eps
        wm = e0 * f + eps * (1 - f)
        eeff = se / 2 + 1 / (2 * big) * (-3 * wm + np.sqrt(4 * big * (3 - big) * pe + (se * big - 3 * wm) ** 2))
        eeff0 = se / 2 + 1 / 4 * (-3 * wm + np.sqrt(8 * pe + (se * 2 - 3 * wm) ** 2))
        ke = 2 * self.k0 * np.sqrt(eeff).imag
        self.ks = float(ke - 2 * self.k0 * np.sqrt(eeff0).imag)
        self.ka = float(2 * self.k0 * np.sqrt(self.eps_eff).imag)
        # phase norm: ks over the integral of the unnormalised phase function (|sqrt(eps_eff)| there, :189)
        mu = np.linspace(1.0, -1.0, 65)
        kd = 2.0 * self.k0 * np.sqrt((1.0 - mu) / 2.0) * abs(np.sqrt(self.eps_eff))
        integral = romberg65(self.ft_corr(kd).real * (mu**2 + 1.0), mu[0] - mu[1])
        self.iba_coeff = 0.0 if self.ks == 0 or integral == 0 else self.ks / (integral / 4.0)   # (_phase_norm)

    def phase(self, mu_s, mu_i, dphi, npol):
        p, sin_half = rayleigh_matrix_and_half_angle(mu_s, mu_i, dphi, npol)
        kd = 2.0 * self.k0 * np.sqrt(self.eps_eff) * sin_half           # complex
        return (self.iba_coeff * self.ft_corr(kd) * p).real               # what the real part of the FFT keeps for mode 0

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        if npol != 2 or m_max != 0:
            raise ValueError("the oracle restates the strong-contrast expansion for passive mode")
        nsamples = int(2 ** np.ceil(4 + np.log(m_max + 1) / np.log(2)))
        return ft_even_matrix(lambda dphi: self.phase(mu_s, mu_i, dphi, npol), m_max, nsamples, npol)


class DMRTQCAShortRangeLayer(LayerEM):
    """DMRT QCA short range, smrt/emmodel/dmrt_qca_shortrange.py:65-112; Rayleigh phase (rayleigh.py:52-127)."""

    kind = "dmrt_qca_shortrange"

    def __init__(self, frequency, frac_volume, temperature, microstructure, **mp):
        if microstructure != "sticky_hard_spheres":
            raise ValueError("DMRT_QCA_ShortRange needs sticky_hard_spheres")
        f = frac_volume
        radius = mp["radius"]
        e0 = 1.0
        es = scatterer_permittivity(frequency, temperature, mp.get("liquid_water", 0.0))
        if f > 0.5:  # dense_snow_correction="auto": inverted medium (core/layer.py inverted_medium)
            f, e0, es = 1.0 - f, es, e0
        self.f = f
        t = shs_t_parameter(f, mp["stickiness"])
        y = (es - e0) / (es + 2.0 * e0)
        fy = f * y
        k0 = (2.0 * np.pi * frequency / C_SPEED) * np.sqrt(complex(e0)).real
        den = 1.0 + 2.0 * f - t * f * (1.0 - f)
        e_eff = e0 + 3.0 * fy * e0 / (1.0 - fy) * (
            1.0 + 2j / 3.0 * (k0 * radius) ** 3 * y * (1.0 - f) ** 4 / ((1.0 - fy) * den**2)
        )
        ks = 2.0 / (9.0 * f) * k0 * (k0 * radius) ** 3 * (abs(e_eff / e0 - 1.0) ** 2 * (1.0 - f) ** 4 / den**2)
        beta = 2.0 * k0 * np.sqrt(complex(e_eff)).imag
        self.eps_eff = complex(e_eff)
        self.ks = float(ks)
        self.ka = float(beta - ks)

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        return rayleigh_ft_even_phase(self.ks, mu_s, mu_i, m_max, npol)


class DMRTQCACPShortRangeLayer(LayerEM):
    """DMRT QCA-CP short range as in DMRT-ML, smrt/emmodel/dmrt_qcacp_shortrange.py:63-125; Rayleigh phase."""

    kind = "dmrt_qcacp_shortrange"

    def __init__(self, frequency, frac_volume, temperature, microstructure, **mp):
        if microstructure != "sticky_hard_spheres":
            raise ValueError("DMRT_QCACP_ShortRange needs sticky_hard_spheres")
        f = frac_volume
        radius = mp["radius"]
        e0 = 1.0
        es = scatterer_permittivity(frequency, temperature, mp.get("liquid_water", 0.0))
        if f > 0.5:  # dense_snow_correction="auto": inverted medium
            f, e0, es = 1.0 - f, es, e0
        t = shs_t_parameter(f, mp["stickiness"])
        lmda = C_SPEED / frequency
        b = (es - e0) * (1.0 - 4.0 * f) / 3.0 - e0
        c = -e0 * (es - e0) * (1.0 - f) / 3.0
        disc = np.sqrt(complex(b * b - 4.0 * c))
        e_eff0 = 0.5 * (-b + disc)
        if e_eff0.real < 1:
            e_eff0 = 0.5 * (-b - disc)
        x3 = (2.0 * np.pi * radius / lmda) ** 3
        shape = (1.0 - f) ** 4 / (1.0 + 2.0 * f - t * f * (1.0 - f)) ** 2
        corr = (es - e0) / (1.0 + (es - e0) / (3.0 * e_eff0) * (1.0 - f))
        e_eff = e0 + (e_eff0 - e0) * (1.0 + 2j / 9.0 * x3 * np.sqrt(complex(e_eff0)) * corr * shape)
        sq_im = np.sqrt(complex(e_eff)).imag
        albedo = 2.0 / 9.0 * x3 * f / (2.0 * sq_im) * abs(corr) ** 2 * shape
        beta = 2.0 * np.pi / lmda * 2.0 * sq_im
        self.f = f
        self.eps_eff = complex(e_eff)
        self.ks = float(albedo * beta)
        self.ka = float(beta - self.ks)

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        return rayleigh_ft_even_phase(self.ks, mu_s, mu_i, m_max, npol)


class NonScatteringLayer(LayerEM):
    """smrt/emmodel/nonscattering.py: Polder-van Santen permittivity, absorption only, null phase matrix."""

    kind = "nonscattering"

    def __init__(self, frequency, frac_volume, temperature, microstructure, **mp):
        eps = scatterer_permittivity(frequency, temperature, mp.get("liquid_water", 0.0))
        self.f = frac_volume
        self.eps_eff = polder_van_santen_spheres(frac_volume, 1.0, eps)
        self.ka = float(2.0 * (2.0 * np.pi * frequency / C_SPEED) * np.sqrt(self.eps_eff).imag)
        self.ks = 0.0

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        return np.zeros((npol, npol, m_max + 1, len(mu_s), len(mu_i)))


class RayleighLayer(LayerEM):
    """smrt/emmodel/rayleigh.py:22-51: sparse medium of small independent spheres (background permittivity as the
    effective one), Rayleigh phase matrix."""

    kind = "rayleigh"

    def __init__(self, frequency, frac_volume, temperature, microstructure, radius, **mp):
        e0, eps = 1.0, scatterer_permittivity(frequency, temperature, mp.get("liquid_water", 0.0))
        k0 = 2.0 * np.pi * frequency / C_SPEED
        self.eps_eff = complex(e0)
        self.ks = float(frac_volume * 2 * abs((eps - e0) / (eps + 2 * e0)) ** 2 * radius**3 * abs(e0) ** 2 * k0**4)
        self.ka = float(frac_volume * k0 * eps.imag * abs(3 * e0 / (eps + 2 * e0)) ** 2
                        + (1 - frac_volume) * 2 * k0 * np.sqrt(complex(e0)).imag)

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        return rayleigh_ft_even_phase(self.ks, mu_s, mu_i, m_max, npol)


class PrescribedLayer(LayerEM):
    """smrt/emmodel/prescribed_kskaeps.py: ks, ka and the effective permittivity are layer attributes."""

    kind = "prescribed_kskaeps"

    def __init__(self, frequency, frac_volume, temperature, microstructure, ks, ka, eps_re, eps_im, **mp):
        self.ks, self.ka, self.eps_eff = float(ks), float(ka), complex(eps_re, eps_im)

    def ft_even_phase(self, mu_s, mu_i, m_max, npol):
        return rayleigh_ft_even_phase(self.ks, mu_s, mu_i, m_max, npol)


def romberg65(y, dx):
    """Romberg extrapolation of the trapezoid rule on 2**6+1 equally spaced samples (scipy.integrate.romb, called at
    smrt/emmodel/iba.py:179)."""
    n = len(y) - 1
    k = int(round(np.log2(n)))
    assert 2**k == n
    rows = []
    for i in range(k + 1):
        step = n >> i
        yy = y[::step]
        rows.append(step * dx * (0.5 * (yy[0] + yy[-1]) + yy[1:-1].sum()))
    r = np.array(rows)
    for j in range(1, k + 1):
        r = (4.0**j * r[1:] - r[:-1]) / (4.0**j - 1.0)
    return float(r[0])


def rayleigh_matrix_and_half_angle(mu_s, mu_i, dphi, npol):
    """Tsang (2000) scattering amplitudes -> phase matrix, smrt/emmodel/common.py:9-53, core/lib.py:623-652.
    Returns p[ps, pi, phi, mu_s, mu_i] and sin(Theta/2)."""
    mu_s = np.atleast_1d(mu_s)[None, :, None]
    mu_i = np.atleast_1d(mu_i)[None, None, :]
    dphi = np.atleast_1d(dphi)[:, None, None]
    sin_s = np.sqrt(1.0 - mu_s**2)
    sin_i = np.sqrt(1.0 - mu_i**2)
    cphi, sphi = np.cos(dphi), np.sin(dphi)
    fvv = cphi * mu_s * mu_i + sin_s * sin_i
    fhv = -sphi * mu_i
    fhh = cphi + 0.0 * mu_s * mu_i
    fvh = sphi * mu_s
    fvv, fvh, fhv, fhh = np.broadcast_arrays(fvv, fvh, fhv, fhh)
    if npol == 2:
        p = np.array([[fvv**2, fvh**2], [fhv**2, fhh**2]])
    else:
        p = np.array(
            [
                [fvv**2, fvh**2, fvh * fvv],
                [fhv**2, fhh**2, fhh * fhv],
                [2.0 * fvv * fhv, 2.0 * fvh * fhh, fvv * fhh + fvh * fhv],
            ]
        )
    cos_t = np.clip(mu_s * mu_i + sin_s * sin_i * cphi, -1.0, 1.0)
    return p, np.sqrt(0.5 * (1.0 - cos_t))


def ft_even_matrix(phase_fn, m_max, nsamples, npol):
    """Azimuthal Fourier modes by FFT of the mirrored samples, smrt/emmodel/common.py:56-131."""
    dphi = np.linspace(0.0, np.pi, nsamples // 2 + 1)
    p = phase_fn(dphi)
    mirror = p[:, :, -2:0:-1].copy()
    if npol >= 3:
        mirror[0:2, 2] *= -1.0
        mirror[2, 0:2] *= -1.0
    ft = np.fft.fft(np.concatenate((p, mirror), axis=2), axis=2)
    out = np.empty((npol, npol, m_max + 1) + p.shape[3:])
    out[:, :, 0] = ft[:, :, 0].real / nsamples
    d = 2.0 / nsamples
    if npol == 2:
        out[:, :, 1:] = ft[:, :, 1 : m_max + 1].real * d
    else:
        out[0:2, 0:2, 1:] = ft[0:2, 0:2, 1 : m_max + 1].real * d
        out[0:2, 2, 1:] = ft[0:2, 2, 1 : m_max + 1].imag * d
        out[2, 0:2, 1:] = -ft[2, 0:2, 1 : m_max + 1].imag * d
        out[2, 2, 1:] = ft[2, 2, 1 : m_max + 1].real * d
    return out


def rayleigh_ft_even_phase(ks, mu_s, mu_i, m_max, npol):
    """Closed-form Rayleigh modes m=0,1,2 (Ulaby et al.), smrt/emmodel/rayleigh.py:52-127."""
    mu_s, mu_i = np.asarray(mu_s, float), np.asarray(mu_i, float)
    s2, i2 = mu_s**2, mu_i**2
    one_s, one_i = np.ones_like(mu_s), np.ones_like(mu_i)
    P = np.zeros((npol, npol, m_max + 1, len(mu_s), len(mu_i)))
    o = np.outer
    P[0, 0, 0] = 0.5 * o(s2, i2) + o(1 - s2, 1 - i2)
    P[0, 1, 0] = 0.5 * o(s2, one_i)
    P[1, 0, 0] = 0.5 * o(one_s, i2)
    P[1, 1, 0] = 0.5
    if m_max >= 1:
        ss, si = np.sqrt(1 - s2), np.sqrt(1 - i2)
        cs, ci = mu_s * ss, mu_i * si
        P[0, 0, 1] = 2 * o(cs, ci)
        if npol >= 3:
            P[0, 2, 1] = o(cs, si)
            P[2, 0, 1] = -2 * o(ss, ci)
            P[2, 2, 1] = o(ss, si)
    if m_max >= 2:
        P[0, 0, 2] = 0.5 * o(s2, i2)
        P[0, 1, 2] = -0.5 * o(s2, one_i)
        P[1, 0, 2] = -0.5 * o(one_s, i2)
        P[1, 1, 2] = 0.5
        if npol >= 3:
            P[0, 2, 2] = 0.5 * o(s2, mu_i)
            P[1, 2, 2] = -0.5 * o(one_s, mu_i)
            P[2, 0, 2] = -o(mu_s, i2)
            P[2, 1, 2] = o(mu_s, one_i)
            P[2, 2, 2] = o(mu_s, mu_i)
    if npol == 3:
        P[0, 2] *= -1.0
        P[1, 2] *= -1.0
    return P * (1.5 * ks)


def make_layers(emmodel, frequency, sp):
    """One LayerEM per layer (smrt/core/model.py:529-582).  `sp` is a dict of arrays: thickness, density (or
    frac_volume), temperature, microstructure name and its parameters."""
    classes = {"iba": IBALayer, "iba_dense_auto": IBADenseAutoLayer, "iba_original": IBAOriginalLayer,
               "iba_maxwell_garnett": IBAMaxwellGarnettLayer, "symsce_torquato21": SymSCELayer, "dmrt_qca_shortrange": DMRTQCAShortRangeLayer,
               "dmrt_qcacp_shortrange": DMRTQCACPShortRangeLayer, "nonscattering": NonScatteringLayer,
               "rayleigh": RayleighLayer, "prescribed_kskaeps": PrescribedLayer}
    L = len(sp["thickness"])
    fv = sp["frac_volume"] if "frac_volume" in sp else np.asarray(sp["density"]) / DENSITY_OF_ICE
    # heterogeneous snowpacks (model.py:529-582: a list of emmodels, one per layer; per-layer microstructure models in
    # make_snowpack): `emmodel` and sp["microstructure"] may be sequences of L names
    ems = [str(e) for e in np.broadcast_to(np.atleast_1d(emmodel), (L,))]
    micros = [str(m) for m in np.broadcast_to(np.atleast_1d(sp["microstructure"]), (L,))]
    args = {"exponential": ("corr_length",), "sticky_hard_spheres": ("radius", "stickiness"),
            "independent_sphere": ("radius",), "homogeneous": (), "teubner_strey": ("corr_length", "repeat_distance"),
            **{n: ("porod_length", "polydispersity") for n in UNIFIED_FT}}
    extra = {"prescribed_kskaeps": ("ks", "ka", "eps_re", "eps_im")}   # layer attributes that emmodel reads
    wet = ("liquid_water",) if "liquid_water" in sp else ()              # water / (ice + water) volume per layer
    return [
        classes[ems[l]](frequency, float(fv[l]), float(sp["temperature"][l]), micros[l],
                        **{n: float(np.broadcast_to(sp[n], (L,))[l]) for n in args[micros[l]] + extra.get(ems[l], ()) + wet})
        for l in range(L)
    ]


# ----------------------------------------------------------------------------------------------------------------
# streams, interfaces
# ----------------------------------------------------------------------------------------------------------------
class Streams:
    pass


def compute_streams(n_max_stream, eps):
    """Gauss-Legendre nodes in the most refringent layer + Snell, smrt/rtsolver/streams.py:136-223,300-330."""
    eps = np.asarray(eps, complex)
    k_star = int(np.argmax(eps))  # complex argmax: lexicographic (real, imag)
    x, _ = roots_legendre(2 * n_max_stream)
    mu_star = x[-1 : n_max_stream - 1 : -1]  # positive nodes, descending
    s_star = np.sqrt(1.0 - mu_star**2)
    st = Streams()
    st.mu, st.weight, st.n = [], [], []
    for e in eps:
        relsin = np.sqrt(eps[k_star] / e).real * s_star
        mu = np.sqrt(1.0 - relsin[relsin < 1.0] ** 2)
        st.mu.append(mu)
        st.n.append(len(mu))
        st.weight.append(_fd_weights(mu, absolute=True))
    relsin = np.sqrt(eps[k_star]).real * s_star
    st.outmu = np.sqrt(1.0 - relsin[relsin < 1.0] ** 2)
    st.n_air = len(st.outmu)
    st.outweight = _fd_weights(st.outmu, absolute=False)
    st.n = np.array(st.n)
    return st


def _fd_weights(mu, absolute):
    w = np.empty_like(mu)
    w[0] = 1.0 - 0.5 * (mu[0] + mu[1])
    w[-1] = 0.5 * (mu[-2] + mu[-1])
    w[1:-1] = 0.5 * (mu[:-2] - mu[2:])
    return np.abs(w) if absolute else w


def fresnel_rigorous(eps1, eps2, mu1):
    """Maezawa & Miyauchi (2009) field coefficients for lossy media, smrt/core/fresnel.py:99-146."""
    eps1, eps2 = complex(eps1), complex(eps2)
    n1 = np.sqrt(eps1)
    kz2 = n1.real**2 * (1.0 - mu1**2)
    kyi = -np.sqrt(eps1 - kz2 + 0j)
    kyt = -np.sqrt(eps2 - kz2 + 0j)
    rh = (kyi - kyt) / (kyi.conjugate() + kyt)
    rv = n1.conjugate() * (eps2 * kyi - eps1 * kyt) / (n1 * (eps2 * kyi.conjugate() + eps1.conjugate() * kyt))
    mu2 = -kyt.real / np.sqrt(eps2).real
    return rv, rh, mu2


def flat_reflection(eps1, eps2, mu1, npol):
    """(npol, n) power reflection, smrt/core/fresnel.py:417-443 via interface/flat.py:20-36."""
    rv, rh, _ = fresnel_rigorous(eps1, eps2, mu1)
    out = [abs(rv) ** 2, abs(rh) ** 2]
    if npol >= 3:
        out.append((rv * np.conj(rh)).real)
    return np.array(out)


def flat_transmission(eps1, eps2, mu1, npol):
    """(npol, n) power transmission, smrt/core/fresnel.py:446-474."""
    rv, rh, mu2 = fresnel_rigorous(eps1, eps2, mu1)
    out = [1.0 - abs(rv) ** 2, 1.0 - abs(rh) ** 2]
    if npol >= 3:
        out.append(mu2 / mu1 * ((1.0 + rv) * np.conj(1.0 + rh)).real)
    return np.array(out)


def coherent_slab(frequency, eps1, eps2, mu1, slab_eps, slab_thickness):
    """A thin layer and its two flat interfaces collapsed into one (smrt/interface/coherent_flat.py:163-186):
    field reflection / transmission coefficients of the slab seen from medium 1, and the cosine in medium 2."""
    r01v, r01h, mu_1 = fresnel_rigorous(eps1, slab_eps, mu1)  # (core/fresnel.py:342: the rigorous field coefficients)
    r1tv, r1th, mu_t = fresnel_rigorous(slab_eps, eps2, np.maximum(mu_1, 1e-4))
    k1 = 2 * np.pi / C_SPEED * frequency * np.sqrt(complex(slab_eps))
    phase = k1 * mu_1 * slab_thickness  # (coherent_flat.py:179-181: the "incoherent" reset has no effect)
    e1, e2 = np.exp(1j * phase), np.exp(2j * phase)
    Rv = (r01v + r1tv * e2) / (1 + r01v * r1tv * e2)
    Rh = (r01h + r1th * e2) / (1 + r01h * r1th * e2)
    Tv = (1 + r01v) * (1 + r1tv) * e1 / (1 + r01v * r1tv * e2)
    Th = (1 + r01h) * (1 + r1th) * e1 / (1 + r01h * r1th * e2)
    return Rv, Rh, Tv, Th, mu_t


def coherent_reflection(frequency, eps1, eps2, mu1, npol, slab):
    """coherent_flat.py:76-105."""
    Rv, Rh, _, _, _ = coherent_slab(frequency, eps1, eps2, mu1, *slab)
    out = [abs(Rv) ** 2, abs(Rh) ** 2]
    if npol >= 3:
        out.append((Rv * np.conj(Rh)).real)
    return np.array(out)


def coherent_transmission(frequency, eps1, eps2, mu1, npol, slab):
    """coherent_flat.py:110-147."""
    Rv, Rh, Tv, Th, mu_t = coherent_slab(frequency, eps1, eps2, mu1, *slab)
    nt = np.sqrt(complex(eps2) / complex(eps1)).real
    out = [abs(Tv) ** 2 * mu_t / mu1 / nt, abs(Th) ** 2 * mu_t / mu1 * nt]
    if npol >= 3:
        out.append(mu_t / mu1 * ((1 + Rv) * np.conj(1 + Rh)).real)
    return np.array(out)


def process_coherent_layers(frequency, eps, thickness):
    """smrt/interface/coherent_flat.py:16-57: the layers thinner than 3/8 of a wavelength (k Re(n) d < 3 pi / 4) are
    removed; each becomes the slab of the interface ON TOP of the layer that followed it.  Returns the indices of the
    kept layers and, per kept layer, None or (slab permittivity, slab thickness)."""
    k0 = 2 * np.pi * frequency / C_SPEED
    coherent = np.array([k0 * np.sqrt(complex(e)).real * d < 3 * np.pi / 4 for e, d in zip(eps, thickness)])
    if coherent[-1]:
        raise OracleError("The last layer is coherent, this is not supported", status=6)
    keep, slabs, pending = [], [], None
    for l in range(len(eps)):
        if coherent[l]:
            if coherent[l - 1]:  # (index -1 = the last layer: never coherent here)
                raise OracleError("Two successive layers are coherent, this is not yet supported", status=6)
            pending = (complex(eps[l]), float(thickness[l]))
        else:
            keep.append(l)
            slabs.append(pending)
            pending = None
    return keep, slabs


def _flatten_pol(d, mode):
    """(npol, n) -> stream-major, polarisation-fastest vector; mode 0 keeps V,H only (core/lib.py:355-363)."""
    if mode == 0:
        d = d[0:2]
    return d.T.reshape(-1)


def interface_diagonals(eps, st, npol, substrate=None, slabs=None, frequency=None):
    """Flat interfaces: smrt/rtsolver/rtsolver_utils.py:473-644 (coherent terms only).  substrate: None, or a dict
    {"kind": "flat", "eps": complex} (smrt/substrate/flat.py via core/interface.py:169-240: Fresnel reflection /
    transmission against the substrate permittivity) or {"kind": "reflector", "R": (R_V, R_H)}
    (smrt/substrate/reflector.py: prescribed specular reflection, emissivity 1 - R; two polarisations only)."""
    L = len(eps)
    itf = dict(Rtop=[], Ttop=[], Rbot=[], Tbot=[])
    slabs = slabs or [None] * L  # slabs[l]: the coherent layer collapsed into the interface on top of layer l

    def refl(e1, e2, mu, slab):
        return flat_reflection(e1, e2, mu, npol) if slab is None else coherent_reflection(frequency, e1, e2, mu, npol, slab)

    def trans(e1, e2, mu, slab):
        return flat_transmission(e1, e2, mu, npol) if slab is None else coherent_transmission(frequency, e1, e2, mu, npol, slab)

    for l in range(L):
        e_up = eps[l - 1] if l > 0 else 1.0
        itf["Rtop"].append(refl(eps[l], e_up, st.mu[l], slabs[l]))
        itf["Ttop"].append(trans(eps[l], e_up, st.mu[l], slabs[l]))
        if l < L - 1:
            itf["Rbot"].append(refl(eps[l], eps[l + 1], st.mu[l], slabs[l + 1]))
            itf["Tbot"].append(trans(eps[l], eps[l + 1], st.mu[l], slabs[l + 1]))
        elif substrate is not None and substrate["kind"] == "flat":  # rtsolver_utils.py:544-547,579-584
            itf["Rbot"].append(flat_reflection(eps[l], substrate["eps"], st.mu[l], npol))
            itf["Tbot"].append(flat_transmission(eps[l], substrate["eps"], st.mu[l], npol))
        elif substrate is not None and substrate["kind"] == "host":
            # dense reflection matrices per azimuth mode handed over by the caller (rtsolver_utils.py:567-597,690-707:
            # specular diagonal + 2 pi | pi x weighted diffuse modes), picked up in dort_mode; no emission terms (active)
            itf["Rbot"].append(np.zeros((npol, st.n[l])))
            # passive: the emissivity diagonal (substrate.emissivity_matrix, rtsolver_utils.py:533-536), [npol, n]
            itf["Tbot"].append(np.asarray(substrate["emissivity"], float) if "emissivity" in substrate else np.zeros((npol, st.n[l])))
            itf["Rbot_dense"] = substrate["R"]          # list over modes of (n P x n P) arrays
            itf["Rbot_coh"] = substrate["Rcoh"]         # list over modes of the specular diagonals
        elif substrate is not None and substrate["kind"] == "reflector":
            if npol > 2:
                raise NotImplementedError("reflector substrate in active mode (reflector.py: not implemented)")
            R = np.repeat(np.asarray(substrate["R"], float)[:, None], st.n[l], axis=1)
            itf["Rbot"].append(R)
            itf["Tbot"].append(1.0 - R)
        else:  # nothing below (rtsolver_utils.py:548-551,601-603)
            itf["Rbot"].append(np.zeros((npol, st.n[l])))
            itf["Tbot"].append(np.zeros((npol, st.n[l])))
    itf["Rbot_air"] = refl(1.0, eps[0], st.outmu, slabs[0])
    itf["Tbot_air"] = trans(1.0, eps[0], st.outmu, slabs[0])
    return itf


# ----------------------------------------------------------------------------------------------------------------
# per-layer eigenproblem
# ----------------------------------------------------------------------------------------------------------------
def compress(P4):
    """(ps, pi, mu_s, mu_i) -> (mu_s*npol + ps, mu_i*npol + pi), smrt/core/lib.py:336-347."""
    a, b, ns, ni = P4.shape
    return np.transpose(P4, (2, 0, 3, 1)).reshape(ns * a, ni * b)


class LayerEigen:
    """Eq 12-13 for one layer: smrt/rtsolver/dort.py:617-962."""

    def __init__(self, em, mu, weight, m_max, npol, method="half_rank_eig", normalization=True):
        self.em, self.mu, self.w, self.m_max, self.npol = em, np.asarray(mu), np.asarray(weight), m_max, npol
        self.method, self.normalization = method, normalization
        self.norm0 = None
        self._ft = None

    def build_A(self, m):
        """dort.py:714-749.  Returns the (K x K) matrix A (K = 2 n P), or None when the phase matrix is null."""
        npol = 2 if m == 0 else 3
        if self._ft is None:
            full = np.concatenate((self.mu, -self.mu))
            self._ft = self.em.ft_even_phase(full, full, self.m_max, self.npol)
        ft = self._ft
        P4 = ft[0:2, 0:2, m] if (m == 0) else ft[:, :, m]
        A = compress(P4).copy()
        if not np.any(A):
            return None
        coef = 0.5 if m == 0 else 0.25
        A *= np.tile(np.repeat(-coef * self.w, npol), 2)[None, :]
        ks = np.full(A.shape[0], self.em.ks)
        if self.normalization and np.all(ks != 0):
            if m == 0:
                self.norm0 = -ks / A.sum(axis=1)
                if self.normalization != "forced" and np.any(np.abs(self.norm0 - 1.0) > 0.3):
                    raise OracleError("phase renormalisation exceeds 30 % (dort.py:792-801)", status=2)
                norm = self.norm0
            else:
                if self.norm0 is None:
                    raise RuntimeError("mode 0 must be solved first (dort.py:803-807)")
                norm = np.empty(len(self.norm0) // 2 * npol)
                norm[0::npol] = self.norm0[0::2]
                norm[1::npol] = self.norm0[1::2]
                norm[2::npol] = np.sqrt(self.norm0[0::2] * self.norm0[1::2])
            A *= norm[:, None]
        ke = self.em.ks + self.em.ka  # isotropic, all polarisations (emmodel/common.py:134-152,326-345)
        A[np.diag_indices(A.shape[0])] += ke
        inv_mu = np.repeat(1.0 / self.mu, npol)
        return np.concatenate((inv_mu, -inv_mu))[:, None] * A

    def no_scattering(self, m):
        """dort.py:765-780."""
        npol = 2 if m == 0 else 3
        n = npol * len(self.mu)
        inv_mu = np.repeat(1.0 / self.mu, npol)
        beta = np.concatenate((inv_mu, -inv_mu)) * (self.em.ks + self.em.ka)
        E = np.eye(2 * n)
        return beta, E[:n], E[n:]

    def solve(self, m, coherent_only=False):
        if coherent_only:
            return self.no_scattering(m)
        A = self.build_A(m)
        if A is None:
            return self.no_scattering(m)
        n = A.shape[0] // 2
        if self.method == "eig":  # dort.py:821-833
            beta, E = scipy.linalg.eig(A)
            return _validated(beta, E[:n], E[n:])
        if self.method == "schur_forcedtriu":  # dort.py:835-889 (reference default)
            T, Z = scipy.linalg.schur(A)
            T[np.tril_indices(T.shape[0], k=-1)] = 0
            beta, E = scipy.linalg.eig(T)
            E = Z @ E
            return _validated(beta, E[:n], E[n:])
        # half-rank reduction (Stamnes et al. 1988 eq 8), dort.py:891-962
        alpha, bmat = -A[:n, :n], -A[:n, n:].copy()
        if m > 0:
            bmat[:, 2::3] *= -1.0
        lam, Ep = scipy.linalg.eig((alpha - bmat) @ (alpha + bmat))
        if np.any(lam.real <= 0):
            raise OracleError("non-positive eigenvalue of the half-rank matrix (albedo >= 1)", status=3)
        beta = np.sqrt(lam.real)
        Em = (alpha + bmat) @ (Ep / beta[None, :])
        Eu = np.hstack((0.5 * (Ep - Em), 0.5 * (Ep + Em)))
        Ed = np.hstack((Eu[:, n:], Eu[:, :n]))
        if m > 0:
            Ed[2::3, :] *= -1.0
        return _validated(np.concatenate((beta, -beta)), Eu, Ed)


def _validated(beta, Eu, Ed):
    """dort.py:1068-1103."""
    bad = (
        (not np.allclose(beta.imag, 0, atol=np.max(beta.real) * 1e-7))
        or (not np.allclose(Eu.imag, 0, atol=1e-6))
        or (not np.allclose(Ed.imag, 0, atol=1e-6))
    )
    if bad:
        raise OracleError("complex eigen-pairs (dort.py:1068-1085)", status=1)
    return beta.real, Eu.real, Ed.real


# ----------------------------------------------------------------------------------------------------------------
# boundary-condition system for one azimuthal mode
# ----------------------------------------------------------------------------------------------------------------
def planck(frequency, T):
    """smrt/core/lib.py:594-607."""
    return (2.0 * PLANCK_CONSTANT / C_SPEED**2) * frequency**3 / np.expm1(
        (PLANCK_CONSTANT / BOLTZMANN_CONSTANT) * frequency / T
    ) if T > 1e-10 else 0.0


def inverse_planck(frequency, radiance):
    """smrt/core/lib.py:610-620."""
    radiance = np.asarray(radiance, float)
    out = np.zeros_like(radiance)
    ok = radiance > 1e-40
    x = (2.0 * PLANCK_CONSTANT / C_SPEED**2) * frequency**3 / radiance[ok]
    out[ok] = (PLANCK_CONSTANT / BOLTZMANN_CONSTANT) * frequency / np.log1p(x)
    return out


def _put_block(ab, u, i0, j0, blk):
    """Scatter a dense block into LAPACK band storage ab[u + i - j, j] (role of _todiag, dort.py:556-587)."""
    n, m = blk.shape
    ii = np.arange(n)[:, None] + i0
    jj = np.arange(m)[None, :] + j0
    ab[u + ii - jj, jj + 0 * ii] = blk


def dort_mode(m, layers_eig, st, itf, thickness, planck_T, intensity_down, coherent_only=False,
              return_x0=False, planck_substrate=None, prune_deep_snowpack=None, pruned_at=None):
    """Assemble and solve the block-tridiagonal boundary system for mode m: smrt/rtsolver/dort.py:263-488.

    planck_T: per-layer black-body radiance B(T_l) (None in active mode).  intensity_down: (n_air*P, R).
    Returns the upwelling intensity above the surface, (n_air*P, R).

    prune_deep_snowpack: optical depth (sum over the layers of min|beta| * thickness) beyond which the deeper layers
    are dropped from the system (dort.py:443-452): the boundary rows and the unknowns below the bottom of the layer in
    which the threshold is passed are cut away, i.e. that layer keeps its bottom reflection and sees nothing coming
    up from below.  pruned_at (a list) receives the number of layers kept.
    """
    P = 2 if m == 0 else 3
    L = len(layers_eig)
    N = st.n * P  # half block size per layer
    col0 = 2 * (np.cumsum(N) - N)
    row_top = col0
    row_bot = col0 + N
    ntot = int(2 * N.sum())
    if L >= 2:
        nband = int(max(np.max(2 * N[1:] + N[:-1]), np.max(N[1:] + 2 * N[:-1])))
    else:
        nband = int(3 * N.max())
    ab = np.zeros((2 * nband + 1, ntot))
    R = intensity_down.shape[1]
    b = np.zeros((ntot, R))
    optical_depth = 0.0
    dense_itf = itf.get("dense", {})   # rough interfaces: {i: matrices of the interface on top of layer i (0: the surface)}

    def imat(i, kind, diag):
        """Matrix `kind` (Rtop / Ttop of layer i, Rbot / Tbot of the layer above it, or of the air side for i = 0) of
        interface i: the caller's dense matrix of this mode for a rough interface (its specular diagonal in the coherent
        pass), else the Flat diagonal `diag` (rtsolver_utils.py:473-642,690-707)."""
        if i in dense_itf:
            return np.asarray(dense_itf[i][kind + ("_coh" if coherent_only else "")][m], float)
        return np.diag(diag)

    def rowsum(M):   # _muleye (dort.py:514-530): action on the isotropic black-body field
        return np.asarray(M).sum(axis=1)

    for l in range(L):
        beta, Eu, Ed = layers_eig[l].solve(m, coherent_only)
        tt = np.exp(-np.maximum(beta, 0.0) * thickness[l])  # reference at the bottom (dort.py:339)
        tb = np.exp(np.minimum(beta, 0.0) * thickness[l])  # reference at the top (dort.py:341)
        Rtop = _flatten_pol(itf["Rtop"][l], m)
        Ttop = _flatten_pol(itf["Ttop"][l], m)
        Rbot = _flatten_pol(itf["Rbot"][l], m)
        Tbot = _flatten_pol(itf["Tbot"][l], m)
        j = col0[l]
        if l == 0:
            Eu0, tt0 = Eu, tt
        # top of layer l (eq 17 & 19, dort.py:364-395)
        RtopM = imat(l, "Rtop", Rtop)
        _put_block(ab, nband, row_top[l], j, (Ed - RtopM @ Eu) * tt[None, :])
        if l < L - 1:
            TbotM = imat(l + 1, "Tbot", Tbot)      # (rows: streams of layer l + 1 for a dense matrix, of layer l for a diagonal)
            nc = min(TbotM.shape[0], N[l + 1])
            _put_block(ab, nband, row_top[l + 1], j, -((TbotM @ Ed) * tb[None, :])[:nc])
        if m == 0 and planck_T is not None:
            b[row_top[l] : row_top[l] + N[l]] -= ((1.0 - rowsum(RtopM)) * planck_T[l])[:, None]
            if l < L - 1:
                b[row_top[l + 1] : row_top[l + 1] + nc] += (rowsum(TbotM) * planck_T[l])[:nc, None]
        if l == 0:
            TairM = imat(0, "Tbot", _flatten_pol(itf["Tbot_air"], m))
            nc0 = min(TairM.shape[0], N[0])
            b[row_top[0] : row_top[0] + nc0] += (TairM @ intensity_down)[:nc0]
        # bottom of layer l (eq 18 & 22, dort.py:400-427)
        if l == L - 1 and "Rbot_dense" in itf:   # rough substrate: dense reflection matrix of this mode (diagonal if coherent only)
            Rmat = np.diag(itf["Rbot_coh"][m]) if coherent_only else np.asarray(itf["Rbot_dense"][m])
            _put_block(ab, nband, row_bot[l], j, (Eu - Rmat @ Ed) * tb[None, :])
        else:
            RbotM = imat(l + 1, "Rbot", Rbot) if l < L - 1 else np.diag(Rbot)
            _put_block(ab, nband, row_bot[l], j, (Eu - RbotM @ Ed) * tb[None, :])
        if l > 0:
            TtopM = imat(l, "Ttop", Ttop)
            nc = min(TtopM.shape[0], N[l - 1])
            _put_block(ab, nband, row_bot[l - 1], j, -((TtopM @ Eu) * tt[None, :])[:nc])
        if m == 0 and planck_T is not None:
            rb = Rbot   # dense reflection matrix: its row sums (_muleye, dort.py:514-530)
            if l == L - 1 and "Rbot_dense" in itf:
                rb = np.asarray(itf["Rbot_dense"][m]).sum(axis=1)
            elif l < L - 1:
                rb = rowsum(imat(l + 1, "Rbot", Rbot))
            b[row_bot[l] : row_bot[l] + N[l]] -= ((1.0 - rb) * planck_T[l])[:, None]
            if l > 0:
                b[row_bot[l - 1] : row_bot[l - 1] + nc] += (rowsum(TtopM) * planck_T[l])[:nc, None]
            if l == L - 1 and planck_substrate is not None:  # emission of the substrate, dort.py:429-441
                b[row_bot[l] : row_bot[l] + N[l]] += (Tbot * planck_substrate)[:, None]
        optical_depth += np.min(np.abs(beta)) * thickness[l]  # dort.py:444
        if prune_deep_snowpack is not None and optical_depth > prune_deep_snowpack:  # dort.py:446-452
            nkeep = int(2 * N[: l + 1].sum())
            ab = ab[:, :nkeep]
            b = b[:nkeep]
            if pruned_at is not None:
                pruned_at.append(l + 1)
            break
    x = scipy.linalg.solve_banded((nband, nband), ab, b)  # dort.py:469
    x0 = x[: 2 * N[0]]
    I1 = Eu0 @ (tt0[:, None] * x0)  # dort.py:476
    if m == 0 and planck_T is not None:
        I1 = I1 + planck_T[0]
    RairM = imat(0, "Rbot", _flatten_pol(itf["Rbot_air"], m))
    Ttop0M = imat(0, "Ttop", _flatten_pol(itf["Ttop"][0], m))
    I0 = RairM @ intensity_down + (Ttop0M @ I1)[: st.n_air * P]  # dort.py:484
    return (I0, x0) if return_x0 else I0


# ----------------------------------------------------------------------------------------------------------------
# full solve for one (snowpack, frequency)
# ----------------------------------------------------------------------------------------------------------------
def solve(sp, frequency, theta_deg, emmodel="iba", mode="P", theta_inc_deg=None, phi=np.pi, n_max_stream=32,
          m_max=2, method="half_rank_eig", phase_normalization=True, rayleigh_jeans=False, details=None,
          substrate=None, atmosphere=None, prune_deep_snowpack=None, process_coherent_layers_=False, interfaces=None):
    """DORT.solve (smrt/rtsolver/dort.py:189-261) for Flat interfaces -- or rough ones handed over as matrices:
    interfaces = {i: {"Rtop": [per azimuth mode], "Ttop": [...], "Rbot": [...], "Tbot": [...], and the same keys + "_coh"
    (specular parts, for the coherent pass of active mode)}} for the interface on top of layer i (0: the surface), every
    matrix as compute_interface_properties combines it (rtsolver_utils.py:473-642,690-707: specular diagonal + 2 pi | pi x
    the normalised diffuse mode); Rtop / Ttop belong to layer i looking up, Rbot / Tbot to the medium above looking down.

    process_coherent_layers_: DORT option process_coherent_layers (dort.py:110,156,203; rtsolver_utils.py:349-365).

    substrate: None or a dict, see interface_diagonals, plus "temperature" (None: no emission).
    atmosphere: None or a dict {"tb_down", "tb_up", "transmittance"} (K, K, -) of a SimpleIsotropicAtmosphere at this
    frequency (smrt/atmosphere/simple_isotropic_atmosphere.py, core/atmosphere.py:131-160): its downwelling radiation
    illuminates the snowpack and the result is tb_up + transmittance * (...) (rtsolver_utils.py:251-260,302-305);
    ignored in active mode like in the reference.
    prune_deep_snowpack: None, True (= 6, dort.py:176-177) or the optical depth beyond which layers are dropped.

    Passive: returns Tb[(V,H), theta].  Active: returns intensity[(pol V,H,U), (pol_inc V,H,U), theta_inc]
    (the layout of the reference's Result.data; sigma = 4 pi cos(theta) I, smrt/core/result.py:484-486).
    """
    if prune_deep_snowpack is True:
        prune_deep_snowpack = 6.0
    elif prune_deep_snowpack is False:
        prune_deep_snowpack = None
    prune = dict(prune_deep_snowpack=prune_deep_snowpack)
    if details is not None:
        details["pruned_at"] = prune["pruned_at"] = []
    ems = make_layers(emmodel, frequency, sp)
    eps = np.array([e.eps_eff for e in ems])
    thickness = np.asarray(sp["thickness"], float)
    slabs = None
    if process_coherent_layers_:
        keep, slabs = process_coherent_layers(frequency, eps, thickness)
        ems, eps, thickness = [ems[l] for l in keep], eps[keep], thickness[keep]
        sp = dict(sp, temperature=np.asarray(sp["temperature"], float)[keep])
        if details is not None:
            details["kept_layers"] = keep
    st = compute_streams(n_max_stream, eps)
    active = mode == "A"
    npol = 3 if active else 2
    mm = m_max if active else 0
    itf = interface_diagonals(eps, st, npol, substrate, slabs, frequency)
    if interfaces:
        itf["dense"] = dict(interfaces)
    leig = [LayerEigen(ems[l], st.mu[l], st.weight[l], mm, npol, method, phase_normalization)
            for l in range(len(ems))]
    if details is not None:
        details.update(streams=st, itf=itf, ems=ems, eig=leig)
    if not active:
        if rayleigh_jeans:
            BT = [float(t) for t in sp["temperature"]]
        else:
            BT = [planck(frequency, float(t)) for t in sp["temperature"]]
        to_I = (lambda t: float(t)) if rayleigh_jeans else (lambda t: planck(frequency, float(t)))
        Bsub = None
        if substrate is not None and substrate.get("temperature") is not None:
            Bsub = to_I(substrate["temperature"])
        I_down = np.zeros((2 * st.n_air, 1))
        if atmosphere is not None:
            I_down[:] = to_I(atmosphere["tb_down"])
        I0 = dort_mode(0, leig, st, itf, thickness, BT, I_down, planck_substrate=Bsub, **prune)[:, 0]
        if atmosphere is not None:
            I0 = to_I(atmosphere["tb_up"]) + atmosphere["transmittance"] * I0
        tb = I0 if rayleigh_jeans else inverse_planck(frequency, I0)
        tb = tb.reshape(st.n_air, 2).T  # (pol, stream), dort.py:503-505
        if details is not None:
            details["tb_streams"] = tb
        return interpolate_passive(st.outmu, tb, np.cos(np.deg2rad(np.atleast_1d(theta_deg))))

    # ---- active: rtsolver_utils.py:91-135,241-320
    mu_inc = np.cos(np.deg2rad(np.atleast_1d(theta_inc_deg)))
    inc = set()
    for mi in mu_inc:
        i0 = int(np.searchsorted(-st.outmu, -mi))
        if i0 == 0:
            inc.add(0)
        elif i0 == st.n_air:
            inc.add(i0 - 1)
        else:
            inc.update((i0, i0 - 1))
    inc = sorted(inc)
    I_0 = np.zeros((2 * st.n_air, 2 * len(inc)))
    I_h = np.zeros((3 * st.n_air, 3 * len(inc)))
    for j, i in enumerate(inc):
        power = 1.0 / (2.0 * np.pi * st.outweight[i])
        for p in range(2):
            I_0[2 * i + p, 2 * j + p] = power
        for p in range(3):
            I_h[3 * i + p, 3 * j + p] = 2.0 * power
    total = np.zeros((3, st.n_air, 3, len(inc)))

    def reshape(I, P):  # dort.py:506-508
        return I.reshape(I.shape[0] // P, P, I.shape[1] // P, P).transpose(1, 0, 3, 2)

    coh = reshape(dort_mode(0, leig, st, itf, thickness, None, I_0, coherent_only=True, **prune), 2)
    for m in range(mm + 1):
        P = 2 if m == 0 else 3
        Im = reshape(dort_mode(m, leig, st, itf, thickness, None, I_0 if m == 0 else I_h, **prune), P)
        Im[0:2, :, 0:2, :] -= coh * (1.0 + float(m > 0))
        if m == 0:
            total[0:2, :, 0:2] += Im[0:2, :, 0:2]
        else:
            total[0:2] += Im[0:2] * np.cos(m * phi)
            total[2:] += Im[2:] * np.sin(m * phi)
    back = np.empty((3, 3, len(inc)))
    for j, i in enumerate(inc):
        back[:, :, j] = total[:, i, :, j]
    outmu = st.outmu[inc]
    if details is not None:
        details["backscatter_streams"] = back
        details["incident_streams"] = inc
    user_mu = np.cos(np.deg2rad(np.atleast_1d(theta_deg)))
    return interpolate_active(outmu, back, user_mu)


def _lin_interp_extrap(x, y, xq):
    """scipy.interpolate.interp1d(kind='linear', fill_value='extrapolate') on the last axis
    (smrt/rtsolver/rtsolver_utils.py:234-237); x need not be sorted."""
    order = np.argsort(x)
    xs, ys = x[order], y[..., order]
    idx = np.clip(np.searchsorted(xs, xq) - 1, 0, len(xs) - 2)
    x0, x1 = xs[idx], xs[idx + 1]
    t = (xq - x0) / (x1 - x0)
    return ys[..., idx] + (ys[..., idx + 1] - ys[..., idx]) * t


def interpolate_passive(outmu, tb, user_mu):
    """rtsolver_utils.py:179-239, passive branch."""
    if np.max(user_mu) > np.max(outmu):
        imax = int(np.argmax(outmu))
        tb = np.insert(tb, 0, np.mean(tb[:, imax]), axis=1)
        outmu = np.insert(outmu, 0, 1.0)
    return _lin_interp_extrap(outmu, tb, user_mu)


def interpolate_active(outmu, I, user_mu):
    """rtsolver_utils.py:199-239, active branch (pol, pol_inc, incidence)."""
    if np.max(user_mu) > np.max(outmu):
        imax = int(np.argmax(outmu))
        co = 0.5 * (I[0, 0, imax] + I[1, 1, imax])
        cx = 0.5 * (I[1, 0, imax] + I[0, 1, imax])
        new = np.array([[co, cx, I[0, 2, imax]], [cx, co, I[1, 2, imax]], list(I[2, :, imax])])
        I = np.insert(I, 0, new, axis=2)
        outmu = np.insert(outmu, 0, 1.0)
    if len(outmu) == 1:
        return np.repeat(I, len(user_mu), axis=2)
    return _lin_interp_extrap(outmu, I, user_mu)


def sigma_dB(intensity, theta_deg):
    """smrt/core/result.py:484-486 and smrt/utils/__init__.py:13-23."""
    x = 4.0 * np.pi * np.cos(np.deg2rad(theta_deg)) * intensity
    return 10.0 * np.log10(np.maximum(x, 1e-20))
