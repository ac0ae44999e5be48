"""The reference's own objects at the plugin boundary.

`register_package("smrt_amd")` inside smrt-model/smrt makes `make_model("iba", "dort")` resolve to smrt_amd's classes, but
the Model, the Snowpack / Layer / interface / substrate / atmosphere objects, the Sensor and the Result stay the
reference's (smrt/core/model.py:395-398,596-617 hand them to the runner and to `rtsolver.solve`).  This module is the ONE
place that reads them, through their public attributes only:

* layers (smrt/core/layer.py:35-117, smrt/inputs/make_medium.py:316-434): thickness, temperature, frac_volume,
  `microstructure` (Exponential.corr_length; StickyHardSpheres.radius / .stickiness), permittivity_model,
  inclusion_shape, medium, emmodel, emmodel_options;
* snowpacks (smrt/core/snowpack.py:34-46): layers, interfaces (classes or instances, smrt/core/interface.py:17-52),
  substrate (smrt/core/interface.py:86-166: temperature, permittivity(frequency); Reflector.specular_reflection),
  atmosphere (smrt/atmosphere/simple_isotropic_atmosphere.py:52-56: constant_tbdown / constant_tbup / constant_trans, or
  any object with the `run(frequency, costheta, npol)` protocol as long as its answer is isotropic);
* models (smrt/core/model.py:254-283): emmodel (class, list, dict by medium), emmodel_options;
* results: the caller's own `make_result` (smrt/core/result.py:79-90, smrt/rtsolver/rtsolver_utils.py:322-344) so that
  the reference's `concat_results` can nest what comes back.

`adopt_snowpack` turns a foreign snowpack into smrt_amd's `Snowpack` of `AdoptedLayer`s: from there on the packing code
of rtsolver/dort.py sees one kind of object, and the device batch is bitwise the one smrt_amd's own objects give.
A layer the device emmodels cannot reproduce (saline snow, a user permittivity model, non-spherical inclusions,
a microstructure model without device code) is never computed as if it were dry snow: with one of the reference's own
emmodel classes it is evaluated on the host through the emmodel protocol, with a device emmodel it raises."""
import importlib

import numpy as np

from .error import SMRTError

NATIVE = "smrt_amd"
REFERENCE = "smrt"


def package_of(x):
    cls = x if isinstance(x, type) else type(x)
    return (getattr(cls, "__module__", None) or "").split(".")[0]


def is_native(x):
    return package_of(x) == NATIVE


def _is_exactly(cls, scope, module, name):
    """Is `cls` the class `name` defined in <package>.<scope>.<module> (not a subclass, not a specialised copy)?"""
    parts = (getattr(cls, "__module__", "") or "").split(".")
    # (the root package must be the reference itself: a third-party plugin that names its class and module alike keeps its
    # own code and takes the host route)
    return getattr(cls, "__name__", None) == name and parts[-2:] == [scope, module] and parts[0] == REFERENCE


# ---- layers ----------------------------------------------------------------------------------------------------------
DEVICE_MICROSTRUCTURE_CLASSES = {            # (module, class) -> (device name, parameters, defaults)
    ("exponential", "Exponential"): ("exponential", ("corr_length",)),
    ("sticky_hard_spheres", "StickyHardSpheres"): ("sticky_hard_spheres", ("radius", "stickiness")),
    ("independent_sphere", "IndependentSphere"): ("independent_sphere", ("radius",)),
    ("teubner_strey", "TeubnerStrey"): ("teubner_strey", ("corr_length", "repeat_distance")),
    ("unified_scaled_exponential", "UnifiedScaledExponential"): ("unified_scaled_exponential", ("porod_length", "polydispersity")),
    ("unified_teubner_strey", "UnifiedTeubnerStrey"): ("unified_teubner_strey", ("porod_length", "polydispersity")),
    ("unified_sticky_hard_spheres", "UnifiedStickyHardSpheres"): ("unified_sticky_hard_spheres", ("porod_length", "polydispersity")),
}
DRY_ICE_PERMITTIVITIES = ("wetice_permittivity_bohren83", "ice_permittivity_maetzler06")   # equal for dry, fresh ice


def _device_microstructure(ms):
    """(device name, p1, p2) of a reference microstructure object, or (None, 0, 0) when the device has no code for it."""
    for (module, name), (dev, params) in DEVICE_MICROSTRUCTURE_CLASSES.items():
        if _is_exactly(type(ms), "microstructure_model", module, name):
            from .layer import device_microstructure_params
            args = {p: float(getattr(ms, p)) for p in params}
            if dev == "unified_sticky_hard_spheres":
                # the object's OWN derived radius and t: smrt's inverted_medium() is a copy with frac_volume flipped that
                # keeps the radius and t of the medium it was copied from
                args["radius"], args["t"] = float(ms.radius), float(ms.t)
                if not args["t"] > 0:
                    return None, 0.0, 0.0   # no device encoding: the host route evaluates the layer
            return (dev,) + device_microstructure_params(dev, float(ms.frac_volume), **args)
    return None, 0.0, 0.0


def _why_not_on_device(layer, device_ms):
    """None when the device emmodels reproduce the electromagnetics of this reference layer (dry or wet snow: air
    background, Maetzler 2006 ice -- coated in water by wetice_permittivity_bohren83 when wet --, spherical inclusions:
    smrt/inputs/make_medium.py:234-312), else the reason as a string."""
    if device_ms is None:
        return f"the microstructure model {type(layer.microstructure).__name__} has no device implementation"
    pm = getattr(layer, "permittivity_model", None)
    if not isinstance(pm, (tuple, list)) or len(pm) != 2:
        return "the layer has no (background, scatterer) permittivity model pair"
    if callable(pm[0]) or complex(pm[0]) != 1.0:
        return "the background is not air (permittivity 1)"
    if not callable(pm[1]) or getattr(pm[1], "__name__", None) not in DRY_ICE_PERMITTIVITIES \
            or (getattr(pm[1], "__module__", "") or "").split(".")[-2:-1] != ["permittivity"]:
        return "the scatterer permittivity is not the default ice permittivity (Maetzler 2006 / wet: Bohren 1983)"
    if (getattr(layer, "liquid_water", None) or 0) > 0 and pm[1].__name__ != "wetice_permittivity_bohren83":
        return "the layer holds liquid water but its scatterer permittivity ignores it"
    if (getattr(layer, "salinity", None) or 0) > 0 and "salinity" in getattr(pm[1], "required_arguments", ()):
        return "the scatterer permittivity depends on the salinity"
    if getattr(layer, "inclusion_shape", None) not in (None, "spheres"):
        return f"inclusion_shape={layer.inclusion_shape!r}"
    if getattr(layer, "depolarization_factors", None) is not None or getattr(layer, "length_ratio", None) is not None:
        return "the layer prescribes depolarization factors"
    return None


class _AdoptedMicrostructure:
    def __init__(self, source, name, frac_volume, p1, p2):
        self.source, self.name, self.frac_volume, self.device_params = source, name, frac_volume, (p1, p2)

    def __getattr__(self, key):          # everything else is the reference object's business
        source = self.__dict__.get("source")
        if source is None:
            raise AttributeError(key)
        return getattr(source, key)


class AdoptedLayer:
    """What rtsolver/dort.py reads of a layer (smrt_amd/core/layer.py), taken from a reference layer; `source` is that
    layer, which emmodels evaluated on the host are instantiated with."""

    def __init__(self, source):
        self.source = source
        self.thickness = float(source.thickness)
        self.temperature = float(source.temperature)
        self.liquid_water = float(getattr(source, "liquid_water", None) or 0)
        self.medium = getattr(source, "medium", None)
        self.emmodel = getattr(source, "emmodel", None)
        self.emmodel_options = getattr(source, "emmodel_options", None)
        dev, p1, p2 = _device_microstructure(source.microstructure)
        self.microstructure_model = dev or type(source.microstructure).__name__
        self.microstructure = _AdoptedMicrostructure(source.microstructure, self.microstructure_model,
                                                     float(source.frac_volume), p1, p2)
        self.device_refusal = _why_not_on_device(source, dev)

    @property
    def frac_volume(self):
        return self.microstructure.frac_volume

    def __getattr__(self, key):          # density, permittivity(i, frequency), ks / ka of prescribed_kskaeps, ...
        source = self.__dict__.get("source")
        if source is None:
            raise AttributeError(key)
        return getattr(source, key)


# ---- interfaces, substrates, atmospheres -----------------------------------------------------------------------------
def adopt_interface(interface):
    """Flat -> the device's Fresnel interface; any other class / instance stays the caller's object and is evaluated on
    the host through the interface protocol (rtsolver/dort.py:interface_matrices)."""
    from ..interface.flat import Flat

    if interface is None or isinstance(interface, Flat):
        return Flat()
    cls = interface if isinstance(interface, type) else type(interface)
    if _is_exactly(cls, "interface", "flat", "Flat"):
        return Flat()
    return interface() if isinstance(interface, type) else interface


def adopt_substrate(substrate):
    """The reference's Flat and (angle-independent) Reflector substrates -> the device kinds; any other substrate stays
    the caller's object (evaluated on the host through the substrate protocol)."""
    if substrate is None or is_native(substrate):
        return substrate
    cls = type(substrate)
    if _is_exactly(cls, "substrate", "flat", "Flat"):
        from ..substrate.flat import Flat

        return Flat(temperature=substrate.temperature, permittivity_model=lambda frequency, _t=None: substrate.permittivity(frequency))
    if _is_exactly(cls, "substrate", "reflector", "Reflector"):
        refl = 1 if substrate.specular_reflection is None else substrate.specular_reflection
        values = list(refl.values()) if isinstance(refl, dict) else [refl]
        if not any(callable(v) for v in values) and getattr(substrate, "backscattering_coefficient", None) is None:
            from ..substrate.reflector import Reflector

            return Reflector(temperature=substrate.temperature, specular_reflection=refl)
    return substrate


def adopt_atmosphere(atmosphere):
    if atmosphere is None or is_native(atmosphere):
        return atmosphere
    from ..atmosphere.simple_isotropic_atmosphere import SimpleIsotropicAtmosphere

    if all(hasattr(atmosphere, a) for a in ("constant_tbdown", "constant_tbup", "constant_trans")):
        return SimpleIsotropicAtmosphere(tb_down=atmosphere.constant_tbdown, tb_up=atmosphere.constant_tbup,
                                         transmittance=atmosphere.constant_trans)
    if callable(getattr(atmosphere, "run", None)):
        return _IsotropicProbe(atmosphere)
    raise SMRTError("the atmosphere must be a SimpleIsotropicAtmosphere or speak the run(frequency, costheta, npol) protocol")


class _IsotropicProbe:
    """Any atmosphere with the reference's protocol (smrt/core/atmosphere.py:16-19,97-125) whose emission and
    transmittance do not depend on the angle: evaluated once per frequency at two cosines."""

    def __init__(self, source):
        self.source = source

    def device_params(self, frequency):
        res = self.source.run(frequency, np.array([1.0, 0.4]), 2)
        out = []
        for what in ("tb_down", "tb_up", "transmittance"):
            a = np.asarray(getattr(res, what), float)
            if not np.allclose(a, a.flat[0], rtol=1e-12, atol=0.0):
                raise SMRTError(f"the atmosphere's {what} depends on the angle or the polarisation: smrt_amd's DORT takes an "
                                "isotropic atmosphere (one number per frequency)")
            out.append(float(a.flat[0]))
        return tuple(out)


# ---- snowpacks -------------------------------------------------------------------------------------------------------
def adopt_snowpack(snowpack, memo=None):
    """smrt_amd's Snowpack for a reference snowpack (itself when it is native).  `memo`: id(foreign object) -> adopted
    object, shared over one run so that snowpacks sharing an atmosphere / a substrate keep sharing it."""
    if is_native(snowpack):
        return snowpack
    from .snowpack import Snowpack

    memo = {} if memo is None else memo
    key = id(snowpack)
    if key in memo:
        return memo[key]

    def shared(obj, adopt):
        if obj is None:
            return None
        if id(obj) not in memo:
            memo[id(obj)] = adopt(obj)
        return memo[id(obj)]

    layers = [AdoptedLayer(lay) for lay in snowpack.layers]
    interfaces = list(getattr(snowpack, "interfaces", None) or [])
    if len(interfaces) != len(layers):
        raise SMRTError("a snowpack needs one interface per layer (the interface lies on top of its layer)")
    adopted = Snowpack(layers=layers, interfaces=[adopt_interface(i) for i in interfaces],
                       substrate=shared(getattr(snowpack, "substrate", None), adopt_substrate),
                       atmosphere=shared(getattr(snowpack, "atmosphere", None), adopt_atmosphere))
    adopted.source = snowpack
    memo[key] = adopted
    return adopted


# ---- emmodels --------------------------------------------------------------------------------------------------------
REFERENCE_DEVICE_EMMODELS = {                # (module, class) of the reference -> device emmodel
    ("iba", "IBA"): "iba",
    ("dmrt_qca_shortrange", "DMRT_QCA_ShortRange"): "dmrt_qca_shortrange",
    ("dmrt_qcacp_shortrange", "DMRT_QCACP_ShortRange"): "dmrt_qcacp_shortrange",
    ("nonscattering", "NonScattering"): "nonscattering",
}


def _reference_device_name(cls):
    for (module, name), dev in REFERENCE_DEVICE_EMMODELS.items():
        if _is_exactly(cls, "emmodel", module, name):
            return dev
    return None


def device_entry(emmodel_class, options, layer):
    """How one layer's emmodel reaches the device: a device emmodel name (str) when `emmodel_class` is smrt_amd's
    descriptor or one of the reference's classes the device reproduces (smrt/emmodel/iba.py:85-105,
    dmrt_qca_shortrange.py:65-112, dmrt_qcacp_shortrange.py:63-125, nonscattering.py:20-32) AND the layer is one the
    device computes; otherwise the (class, options) pair, evaluated on the host through the emmodel protocol."""
    options = dict(options or {})
    refusal = getattr(layer, "device_refusal", None)
    if getattr(emmodel_class, "device_name", None):                 # smrt_amd's own descriptor
        if refusal:
            raise SMRTError(f"the device emmodel {emmodel_class.__name__} cannot compute this layer: {refusal}.  Give the "
                            "model the reference's emmodel class instead: it is then evaluated on the host")
        return emmodel_class.device_name_for(layer, options)
    dev = None if refusal else _reference_device_name(emmodel_class)
    if dev == "iba":
        dsc = options.pop("dense_snow_correction", None)
        if options or dsc not in (None, "auto"):
            return emmodel_class, dict(options, dense_snow_correction=dsc)
        return "iba_inverted" if dsc == "auto" and layer.frac_volume > 0.5 else "iba"
    if dev in ("dmrt_qca_shortrange", "dmrt_qcacp_shortrange"):
        dsc = options.pop("dense_snow_correction", "auto")
        if options or layer.microstructure_model != "sticky_hard_spheres" or (dsc != "auto" and layer.frac_volume > 0.5):
            return emmodel_class, dict(options, dense_snow_correction=dsc)
        return dev
    if dev == "nonscattering" and not options:
        return dev
    return emmodel_class, options


def entry_of_instance(instance, layer):
    """The same decision for a ready emmodel INSTANCE (the rtsolver protocol hands `solve` one per layer,
    smrt/core/model.py:571-582): the device name when the instance is smrt_amd's descriptor or a reference object whose
    numbers the device reproduces, else the instance itself (evaluated on the host)."""
    own = getattr(instance, "_device_name", None) or getattr(type(instance), "device_name", None)
    if own:
        refusal = getattr(layer, "device_refusal", None)
        if refusal:
            raise SMRTError(f"the device emmodel {type(instance).__name__} cannot compute this layer: {refusal}")
        return own
    if getattr(layer, "device_refusal", None):
        return instance
    dev = _reference_device_name(type(instance))
    f = float(layer.frac_volume)
    if dev == "iba":
        # the instance keeps the volume fraction it worked with: that of the inverted medium under "auto" (iba.py:96-99)
        used = float(getattr(instance, "frac_volume", f))
        if used == f:
            return "iba"
        if f > 0.5 and abs(used - (1.0 - f)) <= 1e-12:
            return "iba_inverted"
        return instance
    if dev in ("dmrt_qca_shortrange", "dmrt_qcacp_shortrange"):
        # the option is not kept by the instance: above half ice "auto" (the device's rule) and None differ
        return dev if f <= 0.5 and layer.microstructure_model == "sticky_hard_spheres" else instance
    return dev or instance


def model_make_emmodel(model):
    """The `make_emmodel` of the package the model belongs to (names are resolved by ITS plugin loader)."""
    root = package_of(model)
    try:
        return importlib.import_module(root + ".core.model").make_emmodel
    except (ImportError, AttributeError):
        from .model import make_emmodel

        return make_emmodel


# ---- results ---------------------------------------------------------------------------------------------------------
def result_factory(sensor):
    """(make_result, labelled-array constructor) of the package `sensor` belongs to: smrt_amd's xarray-free pair, or the
    reference's `make_result` with xarray.DataArray (smrt/core/result.py:79-121) so that its `concat_results` and
    accessors work on what the rtsolver returns."""
    from .result import LabeledArray, make_result

    root = package_of(sensor)
    if root == NATIVE:
        return make_result, LabeledArray
    try:
        foreign = importlib.import_module(root + ".core.result").make_result
        import xarray as xr
    except (ImportError, AttributeError):
        return make_result, LabeledArray

    def labelled(values, coords, name=None):
        return xr.DataArray(values, coords=[(dim, np.asarray(v)) for dim, v in coords], name=name)

    return foreign, labelled
