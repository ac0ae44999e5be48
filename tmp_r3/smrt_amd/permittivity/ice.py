"""Permittivity of pure ice on the host (for emmodels evaluated in Python; the device emmodels have their own copy in
dort_physics.hpp): Maetzler 2006 as used by smrt/permittivity/ice.py:52-73."""
import numpy as np

from ..core.error import SMRTError
from ..core.globalconstants import FREEZING_POINT


def ice_permittivity_maetzler06(frequency, temperature):
    if temperature > FREEZING_POINT:
        raise SMRTError("the ice temperature must be lower or equal to the freezing point")
    f_ghz = frequency * 1e-9
    t_c = temperature - FREEZING_POINT
    real = 3.1884 + 9.1e-4 * t_c
    theta = 300.0 / temperature - 1.0
    alpha = (0.00504 + 0.0062 * theta) * np.exp(-22.1 * theta)
    b1, b2, bb = 0.0207, 1.16e-11, 335.0
    delta_beta = np.exp(-9.963 + 0.0372 * t_c)
    e_b = np.exp(bb / temperature)
    beta = (b1 / temperature) * e_b / (e_b - 1.0) ** 2 + b2 * f_ghz ** 2 + delta_beta
    return real + 1j * (alpha / f_ghz + beta * f_ghz)
