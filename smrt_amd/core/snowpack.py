"""Snowpack container (counterpart of smrt/core/snowpack.py:34-260 for Flat interfaces; optional Flat / Reflector
substrate and SimpleIsotropicAtmosphere)."""
import numpy as np

from ..interface.flat import Flat
from .error import SMRTError
from .layer import WRITES as LAYER_WRITES


def substrate_kind(substrate):
    """"flat" / "reflector" for the substrates the device evaluates itself, "host" for any other object that speaks the
    reference's substrate protocol (smrt/core/interface.py:169-240: evaluated in Python, handed to the device as dense
    reflection matrices, active mode), None for anything else."""
    kind = getattr(substrate, "device_kind", None)
    if kind is None and callable(getattr(substrate, "specular_reflection_matrix", None)):
        kind = "host"
    return kind


class Snowpack:
    def __init__(self, layers=None, interfaces=None, substrate=None, atmosphere=None):
        if substrate is not None and substrate_kind(substrate) is None:
            raise SMRTError("smrt_amd implements the Flat and Reflector substrates (smrt_amd.substrate); any other substrate "
                            "must speak the reference's protocol (specular_reflection_matrix, and "
                            "ft_even_diffuse_reflection_matrix if it is rough): it is then evaluated on the host")
        if atmosphere is not None and not hasattr(atmosphere, "device_params"):
            raise SMRTError("smrt_amd implements the SimpleIsotropicAtmosphere (smrt_amd.atmosphere)")
        self.layers = list(layers) if layers is not None else []
        self.interfaces = list(interfaces) if interfaces is not None else [Flat() for _ in self.layers]
        if len(self.interfaces) != len(self.layers):
            raise SMRTError("a snowpack needs one interface per layer (the interface lies on top of its layer)")
        for itf in self.interfaces:
            self._check_interface(itf)
        self._packed = None
        self._micro = self._overrides = None
        self._wet = self._flat = None
        self._facts = None
        self.substrate = substrate
        self.atmosphere = atmosphere

    def __add__(self, other):
        """snowpack + substrate (smrt/core/snowpack.py:__add__)."""
        if substrate_kind(other) is not None:
            return Snowpack(layers=self.layers, interfaces=self.interfaces, substrate=other, atmosphere=self.atmosphere)
        raise SMRTError("only a substrate can be added to a snowpack in smrt_amd")

    @property
    def nlayer(self):
        return len(self.layers)

    @property
    def layer_thicknesses(self):
        return np.array([lay.thickness for lay in self.layers])

    @property
    def layer_densities(self):
        return np.array([lay.density for lay in self.layers])

    def profile(self, property_name):
        return np.array([getattr(lay, property_name) for lay in self.layers])

    @staticmethod
    def _check_interface(interface):
        """Flat (Fresnel on the device), or any object that speaks the reference's interface protocol
        (smrt/core/interface.py; specular_reflection_matrix + coherent_transmission_matrix, and the ft_even_diffuse_*
        matrices if it is rough): it is evaluated in Python and handed to the device as dense matrices."""
        if interface is None or isinstance(interface, Flat):
            return
        if not (callable(getattr(interface, "specular_reflection_matrix", None)) and
                callable(getattr(interface, "coherent_transmission_matrix", None))):
            raise SMRTError("an interface must be Flat or speak the reference's interface protocol (specular_reflection_matrix, "
                            "coherent_transmission_matrix, and ft_even_diffuse_reflection_matrix / "
                            "ft_even_diffuse_transmission_matrix if it is rough): it is then evaluated on the host")

    def append(self, layer, interface=None):
        self._check_interface(interface)
        self.layers.append(layer)
        self.interfaces.append(interface or Flat())
        self._packed = self._micro = self._overrides = self._wet = self._flat = self._facts = None

    def packed(self):
        """The per-layer columns of the device batch for this snowpack -- thickness, ice volume fraction, temperature,
        the two microstructure parameters -- as one (5, n_layers) array, built once (the batching runner stacks these
        rows of many snowpacks instead of walking their layer objects again for every run)."""
        fresh = self._fresh("_packed_key")
        if self._packed is None or not fresh:
            cols = [(lay.thickness, lay.frac_volume, lay.temperature) + lay.microstructure.device_params
                    for lay in self.layers]
            self._packed = np.array(cols, dtype=np.float64).T.reshape(5, len(self.layers))
        return self._packed

    def layer_facts(self):
        """What the batching solver asks of every snowpack of every run, behind ONE freshness check: (packed columns,
        microstructure model names, any per-layer emmodel setting?, liquid-water column or None).  The solver reads it once
        per run and snowpack (rtsolver/dort.py keeps it for the duration of a solve)."""
        fresh = self._fresh("_facts_key")
        f = self.__dict__.get("_facts")
        if f is None or not fresh:
            f = self._facts = (self.packed(), frozenset(self.microstructure_models), self.has_layer_emmodels(), self.liquid_water())
        return f

    def liquid_water(self):
        """Per-layer liquid water (water / (ice + water) volume) as one array, or None for a dry snowpack -- the optional
        sixth column of the device batch (include/smrt_dort.h: smrt_batch.liquid_water)."""
        fresh = self._fresh("_wet_key")
        if self._wet is None or not fresh:   # (looked up once per snowpack, like packed(): the runner asks on every run)
            lw = [float(getattr(lay, "liquid_water", 0) or 0) for lay in self.layers]
            self._wet = (np.array(lw) if any(lw) else None,)
        return self._wet[0]

    def all_interfaces_flat(self):
        """No interface of this snowpack needs the host (Flat everywhere): cached per interface list."""
        key = self._flat
        if key is None or key[0] != self.interfaces:   # (list equality: identity first, element by element, in C)
            key = self._flat = (list(self.interfaces), all(isinstance(itf, Flat) for itf in self.interfaces))
        return key[1]

    def _fresh(self, slot):
        """Is the cache `slot` still valid?  Only if the layer list holds the same objects and none of them (nor its
        microstructure object) has been written to since the cache was filled: every Layer / Microstructure counts its
        own writes, so building or changing OTHER snowpacks' layers does not invalidate this one.  Called on every use
        (it also records the state the cache is being filled for)."""
        writes = LAYER_WRITES[0]
        seen = getattr(self, slot, None)
        # fast path: no Layer / Microstructure of the process has been written to since this cache was checked, and the
        # list holds the same objects in the same order (list equality: identity first, element by element, in C)
        if seen is not None and seen[0] == writes and seen[1] == self.layers:
            return True
        state = tuple((id(lay), lay.__dict__.get("_version", 0), lay.microstructure.__dict__.get("_version", 0))
                      for lay in self.layers)
        ok = seen is not None and seen[2] == state
        setattr(self, slot, (writes, list(self.layers), state))
        return ok

    @property
    def microstructure_models(self):
        fresh = self._fresh("_micro_key")
        if self._micro is None or not fresh:
            self._micro = {lay.microstructure_model for lay in self.layers}
        return self._micro

    def has_layer_emmodels(self):
        """Does any layer carry its own emmodel or emmodel options (smrt/core/model.py:529-582)?  Looked up once per
        snowpack and layer count: the batching runner asks for every snowpack of every run."""
        fresh = self._fresh("_overrides_key")
        if self._overrides is None or not fresh:
            self._overrides = any(getattr(l, "emmodel", None) or getattr(l, "emmodel_options", None) for l in self.layers)
        return self._overrides
