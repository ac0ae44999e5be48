"""IEM rough interface after Fung, Li & Chen (1992), "Backscattering from a randomly rough dielectric surface", IEEE TGRS
30(2) -- smrt_amd's own evaluator of what smrt/interface/iem_fung92.py + interface_utils.py:17-68 provide: the
specular reflection and coherent transmission of a moderately rough surface (Fresnel x the Kirchhoff attenuation, Tsang
et al. 2001 vol. I eq. 2.1.94) and the BACKSCATTER diffuse reflection (single-scattering IEM series), spread over the
azimuth modes.  Host NumPy: the DORT solver evaluates the object on the streams of the two media it separates and
hands the device dense matrices (rtsolver/dort.py:interface_matrices); the same class serves as a substrate through
smrt_amd/substrate/iem_fung92.py.

Validity (checked like the reference: warning, or NaN with warning_handling="nan"): k s < 3 and k s . k l < sqrt(eps_r)."""
import numpy as np

from ..core.error import SMRTError, smrt_warn
from ..core.globalconstants import C_SPEED
from .fresnel import field_reflection, reflection_diagonal, transmission_diagonal


class KirchhoffCoherentPart:
    """Coherent (specular) part of a surface with rms height `roughness_rms`: the flat-surface coefficients times
    exp(-(Delta k_z s)^2), Delta k_z being the change of the normal wavenumber in reflection (2 k_z) or in transmission."""

    roughness_rms = 0.0

    def specular_reflection_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        mu1 = np.atleast_1d(np.asarray(mu1, float))
        # (the reference scales the free-space wavenumber by |eps_1|^2 here, smrt/interface/interface_utils.py:37 -- kept)
        k2 = (2 * np.pi * frequency / C_SPEED) ** 2 * abs(complex(eps_1)) ** 2
        return reflection_diagonal(eps_1, eps_2, mu1, npol) * np.exp(-4 * k2 * self.roughness_rms ** 2 * mu1 ** 2)

    def coherent_transmission_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        mu1 = np.atleast_1d(np.asarray(mu1, float))
        k0 = 2 * np.pi * frequency / C_SPEED
        kz_in = k0 * np.sqrt(complex(eps_1)).real * mu1
        kz_out = k0 * np.sqrt(complex(eps_2) - (1 - mu1 ** 2) * complex(eps_1)).real
        return transmission_diagonal(eps_1, eps_2, mu1, npol) * np.exp(-((kz_out - kz_in) * self.roughness_rms) ** 2)


class IEM_Fung92(KirchhoffCoherentPart):
    """roughness_rms, corr_length [m]; autocorrelation_function "exponential" (default) or "gaussian"; series_truncation:
    number of terms of the roughness-spectrum series; warning_handling "print" | "nan" | anything else: silent."""

    args = ["roughness_rms", "corr_length"]
    optional_args = {"autocorrelation_function": "exponential", "warning_handling": "print", "series_truncation": 10}

    def __init__(self, roughness_rms=None, corr_length=None, autocorrelation_function="exponential",
                 warning_handling="print", series_truncation=10):
        if roughness_rms is None or corr_length is None:
            raise SMRTError("Parameter roughness_rms and corr_length must be specified")
        if autocorrelation_function not in ("exponential", "gaussian"):
            raise SMRTError("The autocorrelation function must be exponential or gaussian")
        self.roughness_rms, self.corr_length = float(roughness_rms), float(corr_length)
        self.autocorrelation_function = autocorrelation_function
        self.warning_handling, self.series_truncation = warning_handling, int(series_truncation)

    # -- the roughness spectrum of order n at the horizontal wavenumber q ------------------------------------------------
    def spectrum(self, n, q):
        lc = self.corr_length
        if self.autocorrelation_function == "gaussian":
            return lc ** 2 / (2 * n) * np.exp(-((q * lc) ** 2) / (4 * n))
        return (lc / n) ** 2 * (1 + (q * lc / n) ** 2) ** -1.5

    def _outside_validity(self, ks, kl, eps_r):
        if ks > 3:
            return f"iem_fung92: the surface is too rough at this wavelength for the single-scattering series (k * rms height = {ks:g}, the model holds below 3)"
        if ks * kl > np.sqrt(eps_r):
            return (f"iem_fung92: (k * rms height)(k * correlation length) = {ks * kl:g} exceeds sqrt(eps_r) = "
                    f"{np.sqrt(eps_r):g}, the bound of the model's validity for this pair of media")
        return None

    def backscatter(self, frequency, eps_1, eps_2, mu):
        """[2, len(mu)] sigma0_pp / (4 pi mu) in the backscattering direction, p = V, H: the single-scattering series of
        Fung et al. 1992 (Kirchhoff term eqs. 44-45 in eq. 82, complementary term eq. 95), or None -> NaN when the
        surface is outside the validity range and warning_handling == "nan"."""
        mu = np.atleast_1d(np.asarray(mu, float))
        eps_1, eps_2 = complex(eps_1), complex(eps_2)
        k = 2 * np.pi * frequency / C_SPEED * np.sqrt(eps_1).real
        eps_r = eps_2 / eps_1
        problem = self._outside_validity(abs(k * self.roughness_rms), abs(k * self.corr_length), eps_r)
        if problem and self.warning_handling == "print":
            smrt_warn(problem)
        elif problem and self.warning_handling == "nan":
            return np.full((2, len(mu)), np.nan)
        r_v, r_h, _ = field_reflection(eps_1, eps_2, mu)
        sin2 = 1 - mu ** 2
        kz, kx = k * mu, k * np.sqrt(sin2)
        s2 = self.roughness_rms ** 2
        n = np.arange(1, self.series_truncation + 1, dtype=np.float64)[:, None]
        kirchhoff = (2 * kz) ** n * np.exp(-s2 * kz ** 2)                        # x f_pp: eq. 82
        i_vv = kirchhoff * (2 * r_v / mu) + kz ** n * (sin2 / mu * (1 + r_v) ** 2 * (1 - 1 / eps_r) * (1 + sin2 / mu ** 2 / eps_r))
        i_hh = kirchhoff * (-2 * r_h / mu) - kz ** n * (sin2 / mu * (1 + r_h) ** 2 * (eps_r - 1) / mu ** 2)
        weight = np.cumprod(s2 / n)[:, None] * self.spectrum(n, -2 * kx)            # s^2n / n! . W^(n)(-2 k_x)
        front = k ** 2 / 2 * np.exp(-2 * s2 * kz ** 2)
        out = np.empty((2, len(mu)))
        for row, series in ((0, i_vv), (1, i_hh)):
            out[row] = front * np.sum(weight * (series.real ** 2 + series.imag ** 2), axis=0) / (4 * np.pi * mu)
        return out

    def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        """[npol, m_max + 1, len(mu_i)]: diagonal in the streams (backscatter only, mu_s must equal mu_i); the energy of the
        backscatter lobe is spread over the azimuth modes: coefficient 1, -2, +2, -2 ... over 1 + 2 m_max
        (smrt/interface/iem_fung92.py:192-214)."""
        if not np.allclose(mu_s, mu_i):
            raise NotImplementedError("Only the backscattering coefficient is implemented at this stage in iem_fung92.")
        gamma = self.backscatter(frequency, eps_1, eps_2, mu_i)
        out = np.zeros((npol, m_max + 1, gamma.shape[1]))
        for m in range(m_max + 1):
            out[:2, m] = (1.0 if m == 0 else (-2.0 if m % 2 else 2.0)) / (1 + 2 * m_max) * gamma
        return out
