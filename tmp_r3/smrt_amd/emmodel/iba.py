"""IBA electromagnetic model (smrt/emmodel/iba.py:53-265) as a host-side descriptor: the per-layer quantities
(effective permittivity, ks, ka) and the phase-matrix assembly are evaluated inside the HIP kernel; this class records
what the kernel needs and exposes the scalar accessors of the emmodel protocol (iba.py:36-40) by asking the device."""
import numpy as np

from ..core.error import SMRTError


# (emmodel, microstructure, frac_volume, temperature, p1, p2, frequency) -> eps / ks / ka as the device computes them: one
# tiny launch per distinct layer state and frequency, not one per emmodel INSTANCE (the sequential runner makes an
# instance per layer and simulation)
_PROPS_CACHE = {}


class _DeviceEMModel:
    device_name = None

    def __init__(self, sensor, layer, **options):
        if np.ndim(sensor.frequency) != 0:
            raise SMRTError("an emmodel instance needs a single-frequency sensor")
        self.sensor = sensor
        self.layer = layer
        self.frequency = float(sensor.frequency)
        self.npol = 2 if sensor.mode == "P" else 3
        self.options = options
        self._props = None
        self._device_name = self.device_name_for(layer, options)
        # the volume fraction of the INCLUSIONS: the air of an inverted medium (smrt/core/layer.py:186-201)
        self._frac_volume = float(1.0 - layer.frac_volume if self._device_name == "iba_inverted" else layer.frac_volume)

    @classmethod
    def device_name_for(cls, layer, options):
        """The device emmodel of one layer under these options (the class attribute, unless an option changes the medium)."""
        return cls.device_name

    def _device_properties(self):
        if self._props is None:
            from .._native import PackedBatch
            from ..rtsolver.dort import get_context

            p1, p2 = self.layer.microstructure.device_params
            key = (self._device_name, self.layer.microstructure_model, self._frac_volume,
                   float(self.layer.temperature), p1, p2, self.frequency)
            if key in _PROPS_CACHE:
                self._props = _PROPS_CACHE[key]
                return self._props
            batch = PackedBatch([1], [100.0], [self._frac_volume], [self.layer.temperature], [p1], [p2],
                                [self.frequency], [0.0], emmodel=self._device_name,
                                microstructure=self.layer.microstructure_model, n_max_stream=4,
                                phase_normalization="forced")
            out = get_context().run(batch)   # the shared, cached context of this process's default GPU (serialised by its lock)
            lay = out.layers[0, 0]
            self._props = dict(eps=complex(lay[0], lay[1]), ks=float(lay[2]), ka=float(lay[3]))
            if len(_PROPS_CACHE) > 65536:
                _PROPS_CACHE.clear()
            _PROPS_CACHE[key] = self._props
        return self._props

    def effective_permittivity(self):
        return self._device_properties()["eps"]

    @property
    def ka(self):
        return self._device_properties()["ka"]

    @property
    def _ks(self):
        return self._device_properties()["ks"]

    def ks(self, mu, npol=2):
        """(npol, len(mu)) isotropic scattering coefficient (emmodel/common.py:309-324,134-152)."""
        return np.full((npol, np.size(mu)), self._ks)

    def ke(self, mu, npol=2):
        return np.full((npol, np.size(mu)), self._ks + self.ka)

    def ft_even_phase(self, mu_s, mu_i, m_max, npol=None):
        """Azimuthal modes 0..m_max of the phase matrix on mu_s x mu_i: array [npol, npol, m_max + 1, len(mu_s),
        len(mu_i)] with the reference's conventions (smrt/emmodel/common.py:349-399, rayleigh.py:52-127), evaluated on
        the device (smrt_dort_ft_even_phase).  smrt_amd's own DORT never asks for it (its kernels assemble the modes in
        place); it is what a foreign rtsolver consumes (smrt/rtsolver/dort.py:231-247)."""
        from ..rtsolver.dort import get_context

        npol = self.npol if npol is None else npol
        if np.any(np.asarray(mu_i) == 1) and npol > 2:
            raise SMRTError("Phase matrix signs for sine elements of mode m = 2 incorrect")
        p1, p2 = self.layer.microstructure.device_params
        return get_context().ft_even_phase(self._device_name, self.layer.microstructure_model, self.frequency,
                                            self._frac_volume, self.layer.temperature, p1,
                                            p2, mu_s, mu_i, m_max, npol)


class IBA(_DeviceEMModel):
    device_name = "iba"

    def __init__(self, sensor, layer, dense_snow_correction=None):
        # dense_snow_correction="auto" (smrt/emmodel/iba.py:85-105) inverts the medium -- air inclusions in an ice
        # background -- for layers whose ice volume fraction exceeds 0.5 and leaves every other layer alone; the device
        # has both media (include/smrt_dort.h: SMRT_EM_IBA_INVERTED)
        if dense_snow_correction not in (None, False, "auto"):
            raise SMRTError(f"unknown dense_snow_correction '{dense_snow_correction}' (None or 'auto')")
        super().__init__(sensor, layer, dense_snow_correction=dense_snow_correction)
        self.frac_volume = self._frac_volume   # what the reference's instance exposes (iba.py:98-99)

    @classmethod
    def device_name_for(cls, layer, options):
        inverted = options.get("dense_snow_correction") == "auto" and layer.frac_volume > 0.5
        return "iba_inverted" if inverted else cls.device_name
