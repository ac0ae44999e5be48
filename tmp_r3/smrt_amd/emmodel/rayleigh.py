"""Rayleigh scattering by small independent spheres (smrt/emmodel/rayleigh.py:18-127), evaluated on the HOST.

This emmodel has no device implementation: it is the worked example of the route any emmodel with the reference's
protocol takes through smrt_amd's DORT -- `effective_permittivity()`, `ks`, `ka` and `ft_even_phase()` are called in
Python for every (layer, frequency), and the device solves the transfer problem with the numbers it is handed
(include/smrt_dort.h, SMRT_EM_HOST)."""
import numpy as np

from ..core.error import SMRTError
from ..core.globalconstants import C_SPEED


class _HostEMModel:
    """Scalar half of the emmodel protocol for an isotropic medium (emmodel/common.py:134-152,309-345): the subclass
    sets `_ks`, `ka` and `_effective_permittivity`."""

    def effective_permittivity(self):
        return self._effective_permittivity

    def ks(self, mu, npol=2):
        return np.full((npol, np.size(mu)), self._ks)

    def ke(self, mu, npol=2):
        return np.full((npol, np.size(mu)), self._ks + self.ka)


class _RayleighPhase(_HostEMModel):
    """Azimuth Fourier modes of the Rayleigh phase matrix.

    With a = sin(theta_s) sin(theta_i) and b = mu_s mu_i the scattering amplitudes are f_vv = a + b cos(phi),
    f_hh = cos(phi), f_vh = mu_s sin(phi), f_hv = -mu_i sin(phi); every element of the Stokes phase matrix (intensity
    terms f^2, and the U row / column products of emmodel/common.py:40-50) is then a three-term cosine or sine series in
    phi whose coefficients are written down below -- there is nothing to integrate.  Same conventions as the
    reference's table (rayleigh.py:52-127): mode 0 is the azimuth mean, mode m >= 1 the coefficient of cos(m phi) for
    the (V|H, V|H) and (U, U) elements and of sin(m phi) for the (V|H, U) and (U, V|H) ones, all scaled by 3 ks / 2;
    the modes above 2 vanish."""

    def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):
        if npol is None:
            npol = 2 if m_max == 0 else 3
        ms = np.asarray(mu_s, float)[:, None]
        mi = np.asarray(mu_i, float)[None, :]
        shape = np.broadcast_shapes(ms.shape, mi.shape)
        a = np.sqrt(1.0 - ms * ms) * np.sqrt(1.0 - mi * mi)
        b = ms * mi
        zero = np.zeros(shape)
        full = lambda x: np.broadcast_to(x, shape)  # noqa: E731
        # (scattered pol, incident pol) -> harmonics 0, 1, 2
        series = {
            (0, 0): (a * a + 0.5 * b * b, 2.0 * a * b, 0.5 * b * b),
            (0, 1): (full(0.5 * ms * ms), zero, full(-0.5 * ms * ms)),
            (1, 0): (full(0.5 * mi * mi), zero, full(-0.5 * mi * mi)),
            (1, 1): (full(0.5), zero, full(0.5)),
        }
        if npol >= 3:
            series.update({
                (2, 2): (zero, a, b),
                (0, 2): (zero, -ms * a, -0.5 * ms * b),
                (1, 2): (zero, zero, full(0.5 * mi)),
                (2, 0): (zero, -2.0 * a * mi, -b * mi),
                (2, 1): (zero, zero, full(ms)),
            })
        P = np.zeros((npol, npol, m_max + 1) + shape)
        for (ps, pi), harmonics in series.items():
            for m, h in enumerate(harmonics[: m_max + 1]):
                P[ps, pi, m] = h
        return P * (1.5 * self._ks)


class Rayleigh(_RayleighPhase):
    """Sparse medium of small spheres: needs a microstructure with a `radius` (rayleigh.py:22-51)."""

    def __init__(self, sensor, layer):
        radius = getattr(layer.microstructure, "radius", None)
        if radius is None:
            raise SMRTError("Only microstructure_model which defined a `radius` can be used with Rayleigh scattering")
        self.sensor, self.layer = sensor, layer
        fv = layer.frac_volume
        e_bg = layer.permittivity(0, sensor.frequency)
        e_sc = layer.permittivity(1, sensor.frequency)
        k0 = 2.0 * np.pi * sensor.frequency / C_SPEED
        polarisability = (e_sc - e_bg) / (e_sc + 2.0 * e_bg)         # Clausius-Mossotti factor of one sphere
        field_ratio = 3.0 * e_bg / (e_sc + 2.0 * e_bg)               # inner / outer field of the sphere
        self._effective_permittivity = e_bg                          # sparse medium
        self._ks = 2.0 * fv * abs(polarisability) ** 2 * radius ** 3 * abs(e_bg) ** 2 * k0 ** 4
        self.ka = fv * k0 * np.imag(e_sc) * abs(field_ratio) ** 2 + (1.0 - fv) * 2.0 * k0 * np.sqrt(e_bg + 0j).imag
