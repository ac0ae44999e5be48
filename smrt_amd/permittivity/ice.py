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


def water_permittivity_maetzler87(frequency, temperature):
    """Liquid water (double Debye model of Maetzler & Wegmuller 1987, smrt/permittivity/water.py:14-43)."""
    if temperature < FREEZING_POINT:
        raise SMRTError(f"The water temperature must be higher or equal to {FREEZING_POINT}K")
    f_ghz = frequency * 1e-9
    theta = 1.0 - 300.0 / temperature
    e0 = 77.66 - 103.3 * theta
    e1 = 0.0671 * e0
    f1 = 20.2 + 146.4 * theta + 316.0 * theta ** 2
    e2 = 3.52 + 7.52 * theta
    f2 = 39.8 * f1
    return e2 + (e1 - e2) / complex(1, -f_ghz / f2) + (e0 - e1) / complex(1, -f_ghz / f1)


def wetice_permittivity_bohren83(frequency, temperature, liquid_water):
    """Ice grains coated in water (smrt/permittivity/wetice.py:12-45): pure ice when dry, else ice inclusions of volume
    fraction 1 - liquid_water in a water host by Maxwell Garnett (generic_mixing_formula.py:352-380) -- the default
    scatterer permittivity of a snow layer, on the host for emmodels evaluated in Python (the device: dort_physics.hpp)."""
    eps_ice = ice_permittivity_maetzler06(frequency, temperature)
    if not liquid_water > 0:
        return eps_ice
    e0 = water_permittivity_maetzler87(frequency, temperature)
    c_plus, c_minus = eps_ice + 2 * e0, (eps_ice - e0) * (1.0 - liquid_water)
    return (c_plus + 2 * c_minus) / (c_plus - c_minus) * e0
