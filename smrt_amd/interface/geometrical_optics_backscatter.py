"""Geometrical optics reduced to its backscatter lobe (smrt/interface/geometrical_optics_backscatter.py): the diffuse
reflection is the closed-form backscattering coefficient of a Gaussian-slope surface, spread over the azimuth modes like
iem_fung92's; what is not reflected -- one minus the hemispherical reflectivity of the FULL geometrical-optics model -- is
transmitted in the specular direction ("first order" geometrical optics).  Usable in passive mode, where the plain model
has no coherent transmission to emit through."""
import numpy as np

from .fresnel import field_reflection
from .geometrical_optics import GeometricalOptics, shadowing


class GeometricalOpticsBackscatter(GeometricalOptics):
    def backscatter(self, eps_1, eps_2, mu_i):
        """[len(mu_i)] sigma0 / (4 pi mu): |R(0)|^2 / (2 s^2 mu^5) exp(-tan^2 / 2 s^2) / (4 pi), shadowed once."""
        mu_i = np.atleast_1d(np.asarray(mu_i, float))
        r0, _, _ = field_reflection(eps_1, eps_2, np.ones(1))
        tan2 = 1 / mu_i ** 2 - 1
        s2 = self.mean_square_slope
        gamma = abs(r0) ** 2 / (4 * np.pi * 2 * s2 * mu_i ** 5) * np.exp(-tan2 / (2 * s2))
        if self.shadow_correction:
            with np.errstate(divide="ignore", invalid="ignore"):
                gamma = gamma / (1 + shadowing(s2, 1 / np.sqrt(tan2)))
        return gamma

    def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        """[npol, m_max + 1, len(mu_i)], diagonal in the streams; modes weighted 1, -2, +2, ... over 1 + 2 m_max."""
        if not np.allclose(mu_s, mu_i):
            raise NotImplementedError("Only the backscattering coefficient is implemented at this stage.")
        gamma = self.backscatter(eps_1, eps_2, mu_i)
        out = np.zeros((npol, m_max + 1, len(gamma)))
        for m in range(m_max + 1):
            out[:2, m] = (1.0 if m == 0 else (-2.0 if m % 2 else 2.0)) / (1 + 2 * m_max) * gamma
        return out

    def ft_even_diffuse_transmission_matrix(self, frequency, eps_1, eps_2, mu_s, mu_i, m_max, npol):
        return 0.0

    def coherent_transmission_matrix(self, frequency, eps_1, eps_2, mu1, npol):
        mu1 = np.atleast_1d(np.asarray(mu1, float))
        # (the reference integrates the full model with its default shadow correction whatever this object's setting:
        # geometrical_optics_backscatter.py:139-142 passes the flag under a name the constructor does not know)
        full = GeometricalOptics(mean_square_slope=self.mean_square_slope, shadow_correction=True)
        reflected = full.hemispherical_reflectivity(frequency, eps_1, eps_2, mu1)
        out = np.zeros((npol, len(mu1)))
        out[:2] = 1.0 - reflected
        return out
