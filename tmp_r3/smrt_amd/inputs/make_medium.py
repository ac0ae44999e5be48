"""make_snowpack / make_snow_layer with the reference's signature (smrt/inputs/make_medium.py:158-314) for dry snow,
Flat interfaces, optional Flat / Reflector substrate and SimpleIsotropicAtmosphere."""
import collections.abc

import numpy as np

from ..core.error import SMRTError
from ..core.globalconstants import FREEZING_POINT
from ..core.layer import Layer
from ..core.snowpack import Snowpack
from ..interface.flat import Flat


def _get(x, i):
    if isinstance(x, str) or x is None or not isinstance(x, (collections.abc.Sequence, np.ndarray)):
        return x
    return x[i]


def _check_size(x, n, name):
    if isinstance(x, (collections.abc.Sequence, np.ndarray)) and not isinstance(x, str) and len(x) != n:
        raise SMRTError(f"The length of '{name}' must be the same as the number of layers ({n}).")


def make_snow_layer(layer_thickness, microstructure_model, density, temperature=FREEZING_POINT, **kwargs):
    return Layer(layer_thickness, microstructure_model, density, temperature=temperature, **kwargs)


def make_snowpack(thickness, microstructure_model, density, interface=None, surface=None, substrate=None,
                  atmosphere=None, **kwargs):
    """Build a multi-layered snowpack; every parameter can be an array, a list or a constant."""
    if not isinstance(thickness, collections.abc.Iterable):
        raise SMRTError("The thickness argument must be iterable, that is, a list of numbers, numpy array or pandas "
                        "Series or DataFrame.")
    thickness = list(thickness)
    n = len(thickness)
    _check_size(density, n, "density")
    for k, v in kwargs.items():
        _check_size(v, n, k)
    def as_interface(itf):   # None / "flat" / the Flat class or an instance / an object with the reference's interface protocol
        if itf is None or itf == "flat" or itf is Flat:
            return Flat()
        if isinstance(itf, str):
            raise SMRTError(f"interface '{itf}' has no implementation in smrt_amd: pass Flat or an interface OBJECT with the "
                            "reference's protocol (e.g. smrt's iem_fung92 / geometrical_optics instance); it is evaluated on "
                            "the host")
        return itf() if isinstance(itf, type) else itf

    if isinstance(interface, (list, tuple)):
        _check_size(interface, n, "interface")
    sp = Snowpack(substrate=substrate, atmosphere=atmosphere)
    for i, dz in enumerate(thickness):
        if dz <= 0:
            continue
        layer = make_snow_layer(dz, _get(microstructure_model, i), density=_get(density, i),
                                **{k: _get(v, i) for k, v in kwargs.items()})
        # (one interface per layer, on top of it; `surface` replaces the first one, smrt/inputs/make_medium.py:127-133)
        itf = surface if (surface is not None and sp.nlayer == 0) else (_get(interface, i) if isinstance(interface, (list, tuple)) else interface)
        sp.append(layer, interface=as_interface(itf))
    if sp.nlayer == 0:
        raise SMRTError("a snowpack needs at least one layer with a positive thickness")
    return sp
