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


def make_interface(inst_class_or_modulename=None, broadcast=True, **kwargs):
    """An interface object from a name ("flat", "iem_fung92", "geometrical_optics", "geometrical_optics_backscatter" -- or
    any module of a registered plugin package), a class or a ready instance; with sequences among the keyword arguments and
    `broadcast`, one instance per element (smrt/core/interface.py:17-52).  Flat is evaluated on the device (Fresnel per
    stream); every other model is evaluated on the host through the interface protocol."""
    import inspect

    from ..core.plugin import import_class

    if inst_class_or_modulename is None or inst_class_or_modulename == "flat":
        cls = Flat
    elif isinstance(inst_class_or_modulename, str):
        cls = import_class("interface", inst_class_or_modulename)
    elif inspect.isclass(inst_class_or_modulename):
        cls = inst_class_or_modulename
    elif callable(getattr(inst_class_or_modulename, "specular_reflection_matrix", None)) or isinstance(inst_class_or_modulename, Flat):
        return inst_class_or_modulename
    else:
        raise SMRTError("The interface must be either the name of a module in the interface directory of a plugin package, "
                        "a class that implements the interface behavior, or an instance of such a class.")
    if broadcast and kwargs:
        lengths = [len(v) for v in kwargs.values() if isinstance(v, (collections.abc.Sequence, np.ndarray)) and not isinstance(v, str)]
        if lengths:
            return [cls(**{k: _get(v, i) for k, v in kwargs.items()}) for i in range(max(lengths))]
    return cls(**kwargs)


def make_soil(substrate_model, permittivity_model, temperature, **kwargs):
    """A substrate by name ("flat", "reflector", "iem_fung92", "geometrical_optics", "geometrical_optics_backscatter") or
    class, with a numeric permittivity or a callable of (frequency[, temperature]) -- the part of
    smrt/inputs/make_soil.py:41-139 that needs no soil dielectric model (those are outside the DORT path: give the
    permittivity as a number or your own function)."""
    from ..core.plugin import import_class

    if isinstance(permittivity_model, str):
        raise SMRTError(f"the soil permittivity model '{permittivity_model}' is outside the scope of smrt_amd: give the "
                        "permittivity as a number or as a function of (frequency, temperature)")
    cls = import_class("substrate", substrate_model) if isinstance(substrate_model, str) else substrate_model
    return cls(temperature=temperature, permittivity_model=permittivity_model, **kwargs)


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
    def as_interface(itf):   # None / a name / a class / an instance (smrt/core/interface.py:17-52)
        return make_interface(itf)

    if isinstance(interface, (list, tuple)):
        _check_size(interface, n, "interface")
        if surface is not None:   # smrt/inputs/make_medium.py:207-210
            raise SMRTError("Setting both 'surface' and 'interface' arguments is ambiguous when interface is a list or any "
                            "sequence: its first element already is the surface.")
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
    sp.all_interfaces_flat()
    sp.layer_facts()   # the per-layer columns the batching solver stacks, packed once here: a FIRST Model.run on a fresh
    return sp          # ensemble costs what a repeated one does (the caches stand until one of these layers is written to)
