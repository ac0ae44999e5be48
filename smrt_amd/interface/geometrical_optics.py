"""Geometrical-optics (very rough, k s >> 1) interface after Tsang & Kong, "Scattering of Electromagnetic Waves", vol. III
(2001) section 2.1 -- smrt_amd's own evaluator of what smrt/interface/geometrical_optics.py and
geometrical_optics_backscatter.py provide: purely diffuse, bistatic reflection (eqs. 2.1.122-124) and transmission
(eqs. 2.1.128-132) of a Gaussian-slope surface with Smith's shadowing function (eq. 2.1.154), as azimuth Fourier modes
on (mu_s, mu_i) grids.  Host NumPy: the DORT solver samples the object on the streams of the two media it separates
(rtsolver/dort.py:interface_matrices); smrt_amd/substrate/geometrical_optics*.py reuse it under the last layer.

Own layout: every directional quantity is an array [azimuth, scattered cosine, incident cosine]; the facet geometry is
worked out once per (incident, outgoing) direction pair by `_facets`, for reflection and refraction alike."""
import numpy as np
from scipy.special import erfc

from ..core.error import SMRTError, smrt_warn
from ..core.globalconstants import C_SPEED
from .fresnel import field_reflection

AZIMUTH_SAMPLES = 256          # samples of the full circle behind the Fourier modes (the reference's choice)
MU_FLOOR = 0.1                 # grazing directions are clipped to this cosine (smrt/interface/geometrical_optics.py:521-523)


def shadowing(mean_square_slope, cotangent):
    """Smith's shadowing term Lambda(cot theta) of a Gaussian-slope surface (Tsang vol. III eq. 2.1.154)."""
    v = cotangent / np.sqrt(2.0 * mean_square_slope)
    return 0.5 * (np.exp(-v * v) / (np.sqrt(np.pi) * v) - erfc(v))


def _grid(mu_out, mu_in, dphi):
    mu_in = np.clip(np.atleast_1d(np.asarray(mu_in, float)), MU_FLOOR, 1.0)[None, None, :]
    mu_out = np.clip(np.atleast_1d(np.asarray(mu_out, float)), MU_FLOOR, 1.0)[None, :, None]
    phi = np.atleast_1d(np.asarray(dphi, float))[:, None, None]
    return mu_out, mu_in, phi


def _polarisation_couplings(k_in, k_out, h_out, v_out, v_in):
    """(h_out.k_in, v_out.k_in, h_in.k_out, v_in.k_out) / |k_in x k_out| with h_in = y: the geometric factors of the
    tangent-plane amplitudes; for colinear directions their limits (-1, 0, 1, 0)."""
    cross = np.sqrt((k_in[1] * k_out[2] - k_in[2] * k_out[1]) ** 2 + (k_in[2] * k_out[0] - k_in[0] * k_out[2]) ** 2
                    + (k_in[0] * k_out[1] - k_in[1] * k_out[0]) ** 2)
    colinear = cross < 1e-4
    cross = np.where(colinear, 1.0, cross)
    dot = lambda a, b: a[0] * b[0] + a[1] * b[1] + a[2] * b[2]          # noqa: E731
    zero = np.zeros_like(cross)
    h_in = (zero, zero + 1.0, zero)
    terms = (dot(h_out, k_in), dot(v_out, k_in), dot(h_in, k_out), dot(v_in, k_out))
    limits = (-1.0, 0.0, 1.0, 0.0)
    return [np.where(colinear, lim, t / cross) for t, lim in zip(terms, limits)]


class GeometricalOptics:
    """mean_square_slope, or roughness_rms and corr_length (mean_square_slope = 2 | 1 | 3 x (rms / length)^2 for a
    gaussian | exponential | power1.5 autocorrelation); shadow_correction (default True)."""

    args = []
    optional_args = {"mean_square_slope": None, "roughness_rms": None, "corr_length": None, "shadow_correction": True,
                     "autocorrelation_function": "gaussian", "warning_handling": "print"}

    def __init__(self, mean_square_slope=None, roughness_rms=None, corr_length=None, shadow_correction=True,
                 autocorrelation_function="gaussian", warning_handling="print"):
        both = roughness_rms is not None and corr_length is not None
        if (mean_square_slope is None) != both:
            raise SMRTError("Either mean_square_slope or both roughness_rms and corr_length must be set.")
        if mean_square_slope is None:
            factor = {"gaussian": 2, "exponential": 1, "power1.5": 3}[autocorrelation_function]
            mean_square_slope = factor * (roughness_rms / corr_length) ** 2
        self.mean_square_slope = float(mean_square_slope)
        self.roughness_rms, self.corr_length = roughness_rms, corr_length
        self.shadow_correction = bool(shadow_correction)
        self.autocorrelation_function, self.warning_handling = autocorrelation_function, warning_handling

    # -- no coherent part: all the power is scattered --------------------------------------------------------------------
    def specular_reflection_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        return 0.0

    def coherent_transmission_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        return 0.0

    def _check_validity(self, frequency, eps_1):
        if self.roughness_rms is None or self.corr_length is None:
            return False
        k = 2 * np.pi * frequency / C_SPEED * np.sqrt(complex(eps_1)).real
        for value, what in ((k * self.roughness_rms, "roughness_rms"), (k * self.corr_length, "corr_length")):
            if value < 3:
                message = (f"Warning, {what} is too small for the given wavelength. Limit is set to "
                           f"k{'s' if what == 'roughness_rms' else 'l'} > 3. Here it is {value:g}")
                if self.warning_handling == "print":
                    smrt_warn(message)
                return self.warning_handling == "nan"
        return False

    def _slope_density(self, q, mu_in):
        """The common factor of eqs. 2.1.124 / 2.1.130 without its numerator: 1 / (4 pi) x Gaussian slope density of the
        facet that turns k_in into k_out (q = their difference) / (2 s^2 mu_in q_z^4)."""
        s2 = self.mean_square_slope
        return np.exp(-(q[0] ** 2 + q[1] ** 2) / (2 * q[2] ** 2 * s2)) / (4 * np.pi * 2 * s2 * mu_in * q[2] ** 4)

    def diffuse_reflection_matrix(self, frequency, eps_1, eps_2, mu_s, mu_i, dphi, npol):
        """[npol, npol, len(dphi), len(mu_s), len(mu_i)] bistatic reflection coefficient / (4 pi) (V, H block; the third
        Stokes component is not modelled: zeros)."""
        if self._check_validity(frequency, eps_1):
            return np.full((npol, len(np.atleast_1d(mu_i))), np.nan)
        mu_s, mu_i, phi = _grid(mu_s, mu_i, dphi)
        sin_i, sin_s = np.sqrt(1 - mu_i ** 2), np.sqrt(1 - mu_s ** 2)
        cos_p, sin_p = np.cos(phi), np.sin(phi)
        one = np.ones(np.broadcast_shapes(mu_s.shape, mu_i.shape, phi.shape))
        k_in = (sin_i * one, 0.0 * one, -mu_i * one)
        k_out = (sin_s * cos_p * one, sin_s * sin_p * one, mu_s * one)
        q = tuple(a - b for a, b in zip(k_in, k_out))
        q2 = q[0] ** 2 + q[1] ** 2 + q[2] ** 2
        # the facet normal is q / (sign(q_z) |q|); the local incidence cosine on it: -n . k_in
        mu_local = -(q[0] * k_in[0] + q[1] * k_in[1] + q[2] * k_in[2]) / (np.sign(q[2]) * np.sqrt(q2))
        r_v, r_h, _ = field_reflection(eps_1, eps_2, np.clip(mu_local, MU_FLOOR, 1.0))
        h_out = (-sin_p * one, cos_p * one, 0.0 * one)
        v_out = (mu_s * cos_p * one, mu_s * sin_p * one, -sin_s * one)
        v_in = (-mu_i * one, 0.0 * one, -sin_i * one)
        hs_ki, vs_ki, hi_ks, vi_ks = _polarisation_couplings(k_in, k_out, h_out, v_out, v_in)
        amp = lambda z: z.real ** 2 + z.imag ** 2                                 # noqa: E731
        out = np.zeros((npol, npol) + one.shape)
        out[0, 0] = amp(hs_ki * hi_ks * r_h + vs_ki * vi_ks * r_v)               # eqs. 2.1.122
        out[1, 1] = amp(vs_ki * vi_ks * r_h + hs_ki * hi_ks * r_v)
        out[1, 0] = amp(vs_ki * hi_ks * r_h - hs_ki * vi_ks * r_v)
        out[0, 1] = amp(hs_ki * vi_ks * r_h - vs_ki * hi_ks * r_v)
        factor = self._slope_density(q, mu_i) * q2 ** 2                           # eq. 2.1.124
        if self.shadow_correction:
            # in the backscattering half-plane the shadowing of the lower of the two directions is already counted
            backward = phi == np.pi
            lower_in = backward & (mu_s <= mu_i)
            lower_out = backward & ~(mu_s <= mu_i)
            cot_i, cot_s = mu_i / np.maximum(sin_i, 1e-3), mu_s / np.maximum(sin_s, 1e-3)
            factor = factor / (1 + (~lower_in) * shadowing(self.mean_square_slope, cot_i)
                               + (~lower_out) * shadowing(self.mean_square_slope, cot_s))
        return out * factor

    def diffuse_transmission_matrix(self, frequency, eps_1, eps_2, mu_t, mu_i, dphi, npol):
        """[npol, npol, len(dphi), len(mu_t), len(mu_i)] bistatic transmission coefficient / (4 pi) from medium 1 into
        medium 2 (real refractive indices in the facet geometry)."""
        n1, n2 = np.sqrt(complex(eps_1)), np.sqrt(complex(eps_2))
        impedance_ratio = n1 / n2
        if abs(impedance_ratio - 1) < 1e-6:
            raise NotImplementedError(f"the case of successive layers with identical index ({n2:g}) is not implemented")
        mu_t, mu_i, phi = _grid(mu_t, mu_i, dphi)
        sin_i, sin_t = np.sqrt(1 - mu_i ** 2), np.sqrt(1 - mu_t ** 2)
        cos_p, sin_p = np.cos(phi), np.sin(phi)
        one = np.ones(np.broadcast_shapes(mu_t.shape, mu_i.shape, phi.shape))
        k_in = (sin_i * one, 0.0 * one, -mu_i * one)
        k_out = (sin_t * cos_p * one, sin_t * sin_p * one, -mu_t * one)
        q = tuple(a * n1.real - b * n2.real for a, b in zip(k_in, k_out))        # eq. 2.1.87
        q2 = q[0] ** 2 + q[1] ** 2 + q[2] ** 2
        norm = np.sign(q[2]) * np.sqrt(q2)
        n_kt = -(q[0] * k_out[0] + q[1] * k_out[1] + q[2] * k_out[2]) / norm     # cosines on the facet (eq. 2.1.128)
        n_ki = -(q[0] * k_in[0] + q[1] * k_in[1] + q[2] * k_in[2]) / norm
        r_h = (n1.real * n_ki - n2.real * n_kt) / (n1.real * n_ki + n2.real * n_kt)   # eqs. 2.1.132
        r_v = (n2.real * n_ki - n1.real * n_kt) / (n2.real * n_ki + n1.real * n_kt)
        impossible = (n_kt < 0) | (n_ki < 0)          # no facet orientation refracts k_in into k_out
        r_h, r_v = np.where(impossible, -1.0, r_h), np.where(impossible, -1.0, r_v)
        h_out = (-sin_p * one, cos_p * one, 0.0 * one)
        v_out = (-mu_t * cos_p * one, -mu_t * sin_p * one, -sin_t * one)
        v_in = (-mu_i * one, 0.0 * one, -sin_i * one)
        ht_ki, vt_ki, hi_kt, vi_kt = _polarisation_couplings(k_in, k_out, h_out, v_out, v_in)
        amp = lambda z: z.real ** 2 + z.imag ** 2                                 # noqa: E731
        th, tv = 1 + r_h, (1 + r_v) * impedance_ratio
        out = np.zeros((npol, npol) + one.shape)
        out[0, 0] = amp(ht_ki * hi_kt * th + vt_ki * vi_kt * tv)                   # eqs. 2.1.130
        out[1, 1] = amp(vt_ki * vi_kt * th + ht_ki * hi_kt * tv)
        out[1, 0] = amp(-vt_ki * hi_kt * th + ht_ki * vi_kt * tv)
        out[0, 1] = amp(ht_ki * vi_kt * th - vt_ki * hi_kt * tv)
        s2 = self.mean_square_slope
        factor = (2 * complex(eps_2) * q2 * n_kt ** 2 / (4 * np.pi * impedance_ratio * s2 * mu_i * q[2] ** 4)
                  * np.exp(-(q[0] ** 2 + q[1] ** 2) / (2 * q[2] ** 2 * s2)))
        if self.shadow_correction:
            cot_i, cot_t = mu_i / np.maximum(sin_i, 1e-3), mu_t / np.maximum(sin_t, 1e-3)
            factor = factor / (1 + shadowing(s2, cot_i) + shadowing(s2, cot_t))
        return out * factor.real

    # -- azimuth Fourier modes ---------------------------------------------------------------------------------------------
    @staticmethod
    def _even_modes(sampled, m_max):
        """Cosine modes 0 .. m_max of a function even in the azimuth, sampled on [0, pi] (AZIMUTH_SAMPLES / 2 + 1 points,
        both ends included): mode 0 is the mean over the circle, mode m >= 1 twice the mean of f cos(m phi).  Input
        [p, p, phi, s, i] -> output [p, p, m, s, i].  (Only the V, H block is non-zero, so the sine modes of the third
        Stokes component -- smrt/core/lib.py:583-590 -- vanish.)"""
        n_half = sampled.shape[2] - 1
        phi = np.linspace(0.0, np.pi, n_half + 1)
        weight = np.full(n_half + 1, 2.0)
        weight[0] = weight[-1] = 1.0           # the interior samples stand for their mirror images too
        out = np.empty(sampled.shape[:2] + (m_max + 1,) + sampled.shape[3:])
        for m in range(m_max + 1):
            w = weight * np.cos(m * phi) * ((1.0 if m == 0 else 2.0) / (2 * n_half))
            out[:, :, m] = np.einsum("k,abkij->abij", w, sampled)
        return out

    def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        phi = np.linspace(0.0, np.pi, AZIMUTH_SAMPLES // 2 + 1)
        return self._even_modes(self.diffuse_reflection_matrix(frequency, eps_1, eps_2, mu_s, mu_i, phi, npol), m_max)

    def ft_even_diffuse_transmission_matrix(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        phi = np.linspace(0.0, np.pi, AZIMUTH_SAMPLES // 2 + 1)
        return self._even_modes(self.diffuse_transmission_matrix(frequency, eps_1, eps_2, mu_s, mu_i, phi, npol), m_max)

    def hemispherical_reflectivity(self, frequency, eps_1, eps_2, mu_i, n_mu=128, n_phi=128):
        """[2, len(mu_i)]: the diffuse reflection integrated over the upper hemisphere and summed over the scattered
        polarisations (Gauss-Legendre in the cosine, rectangle rule in the azimuth) -- smrt/interface/interface_utils.py:
        86-99."""
        from scipy.special import roots_legendre

        x, w = roots_legendre(n_mu)                 # the n_mu-point rule of [-1, 1] mapped onto [0, 1]
        mu, w = 0.5 * (x + 1.0), 0.5 * w
        phi = np.linspace(0.0, 2 * np.pi, n_phi, endpoint=False)
        r = self.diffuse_reflection_matrix(frequency, eps_1, eps_2, mu, mu_i, phi, 2)      # [p_s, p_i, phi, s, i]
        return 2 * np.pi / n_phi * np.einsum("s,pksi->pi", w, r.sum(axis=0))
