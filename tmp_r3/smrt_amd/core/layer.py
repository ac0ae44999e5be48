"""Snow layer container: the attributes of smrt/core/layer.py:35-156 and of SnowLayer
(smrt/inputs/make_medium.py:320-434) that the DORT path reads."""
import numpy as np

from .error import SMRTError
from .globalconstants import DENSITY_OF_ICE, DENSITY_OF_WATER, FREEZING_POINT


READ_ONLY_AFTER_INIT = ("density", "liquid_water", "volumetric_liquid_water")   # smrt/inputs/make_medium.py:355-359
# count of writes to ANY Layer / Microstructure object of the process: while it stands still, no snowpack cache can have
# gone stale through an attribute write (Snowpack._fresh then skips its per-layer comparison)
WRITES = [0]


class Microstructure:
    """Parameters of one of the two supported microstructure models (exponential: corr_length;
    sticky_hard_spheres: radius, stickiness)."""

    def __setattr__(self, key, value):
        # every write bumps the object's own version: Snowpack's per-run caches (packed columns, microstructure set,
        # per-layer emmodel flag) compare the versions of THEIR layers, nobody else's
        object.__setattr__(self, "_version", self.__dict__.get("_version", 0) + 1); WRITES[0] += 1
        object.__setattr__(self, key, value)

    def __init__(self, name, frac_volume, **params):
        self.name = name
        self.frac_volume = frac_volume
        for k, v in params.items():
            setattr(self, k, v)

    @property
    def device_params(self):
        if self.name == "exponential":
            return float(self.corr_length), 0.0
        if self.name == "homogeneous":
            return 0.0, 0.0
        return float(self.radius), float(getattr(self, "stickiness", np.inf))


MICROSTRUCTURE_ARGS = {"exponential": ("corr_length",), "sticky_hard_spheres": ("radius", "stickiness"),
                       # these two have no device emmodel: they serve the emmodels evaluated on the host
                       "independent_sphere": ("radius",), "homogeneous": ()}
DEVICE_MICROSTRUCTURES = ("exponential", "sticky_hard_spheres")


class Layer:
    def __setattr__(self, key, value):
        """Any change invalidates the caches of the snowpacks holding this layer; a microstructure parameter set on the
        layer (layer.corr_length = ...) goes to the microstructure object too, which is what the solver reads.  density,
        liquid_water and volumetric_liquid_water are read-only once the layer exists, like in the reference
        (smrt/inputs/make_medium.py:355-359, smrt/core/layer.py:203-208): the ice volume fraction derives from them --
        use update(density=...)."""
        if key in READ_ONLY_AFTER_INIT and self.__dict__.get("_constructed"):
            raise SMRTError(f"The attribute '{key}' is read-only, setting it would make the layer inconsistent "
                            "(frac_volume derives from it). Use the update method instead: layer.update(density=...).")
        object.__setattr__(self, "_version", self.__dict__.get("_version", 0) + 1); WRITES[0] += 1
        object.__setattr__(self, key, value)
        ms = self.__dict__.get("microstructure")
        if ms is not None and key in MICROSTRUCTURE_ARGS.get(self.__dict__.get("microstructure_model"), ()):
            setattr(ms, key, float(value))

    def __init__(self, thickness, microstructure_model, density, temperature=FREEZING_POINT, medium="snow",
                 liquid_water=None, volumetric_liquid_water=None, salinity=0, emmodel=None, emmodel_options=None,
                 **params):
        if isinstance(microstructure_model, str):
            name = microstructure_model
        else:
            name = getattr(microstructure_model, "__name__", str(microstructure_model)).lower()
        if name not in MICROSTRUCTURE_ARGS:
            raise SMRTError(f"microstructure model '{name}' is outside the scope of smrt_amd "
                            f"(available: {', '.join(MICROSTRUCTURE_ARGS)})")
        if (liquid_water or 0) > 0 or (volumetric_liquid_water or 0) > 0 or (salinity or 0) > 0:
            raise SMRTError("wet or saline snow is outside the scope of smrt_amd (dry snow only)")
        missing = [a for a in MICROSTRUCTURE_ARGS[name] if a not in params and a != "stickiness"]
        if missing:
            raise SMRTError(f"missing microstructure parameter(s) {missing} for '{name}'")
        self.thickness = float(thickness)
        self.density = float(density)
        self.temperature = float(temperature)
        self.medium = medium
        self.emmodel = emmodel
        self.emmodel_options = emmodel_options
        # SnowLayer.compute_frac_volumes with no liquid water (make_medium.py:390-434)
        frac_volume = self.density / DENSITY_OF_ICE
        if not (0 <= frac_volume <= 1.01):
            raise SMRTError(f"the frac_volume of ice in snow is {frac_volume} but must be between 0 and 1.")
        frac_volume = min(frac_volume, 1.0)
        self.microstructure_model = name
        mparams = {k: float(params[k]) for k in MICROSTRUCTURE_ARGS[name] if k in params}
        if name == "sticky_hard_spheres":
            mparams.setdefault("stickiness", np.inf)
        self.microstructure = Microstructure(name, frac_volume, **mparams)
        for k, v in mparams.items():
            setattr(self, k, v)
        for k, v in params.items():   # anything else rides along as a layer attribute (e.g. ks / ka / effective_permittivity
            if k not in mparams:      # for the prescribed_kskaeps emmodel)
                setattr(self, k, v)
        object.__setattr__(self, "_constructed", True)

    @property
    def frac_volume(self):
        return self.microstructure.frac_volume

    def update(self, **kwargs):
        """Change attributes consistently (SnowLayer.update, smrt/inputs/make_medium.py:361-388): density recomputes the
        ice volume fraction; liquid water stays outside the scope (dry snow only)."""
        if (kwargs.get("liquid_water") or 0) > 0 or (kwargs.get("volumetric_liquid_water") or 0) > 0:
            raise SMRTError("wet or saline snow is outside the scope of smrt_amd (dry snow only)")
        if "density" in kwargs:
            density = float(kwargs.pop("density"))
            frac_volume = density / DENSITY_OF_ICE
            if not (0 <= frac_volume <= 1.01):
                raise SMRTError(f"the frac_volume of ice in snow is {frac_volume} but must be between 0 and 1.")
            object.__setattr__(self, "density", density)
            object.__setattr__(self, "_version", self.__dict__.get("_version", 0) + 1); WRITES[0] += 1
            self.microstructure.frac_volume = min(frac_volume, 1.0)
        for k in ("liquid_water", "volumetric_liquid_water"):
            kwargs.pop(k, None)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def permittivity(self, i, frequency):
        """Permittivity of the background (i = 0: air) or of the scatterers (i = 1: pure ice), smrt/core/layer.py:120-156
        for dry snow.  The device emmodels compute the same on the GPU; this is for emmodels evaluated on the host."""
        if i == 0:
            return 1.0
        if i == 1:
            from ..permittivity.ice import ice_permittivity_maetzler06

            return ice_permittivity_maetzler06(frequency, self.temperature)
        raise SMRTError("a snow layer has two constituents (0: air, 1: ice)")
