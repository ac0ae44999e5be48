"""DORT radiative-transfer solver, MI355X-native (drop-in for smrt/rtsolver/dort.py:84-261).

Same constructor options and `solve()` protocol as the reference class; the numerical work (layer electromagnetics,
streams, interfaces, eigen-decomposition, boundary conditions) runs in the HIP kernel through the C ABI.  In addition
to the reference's one-simulation `solve`, `solve_batch` takes a whole list of (sensor, snowpack) simulations and
launches them at once -- that is what the batching runner calls.
"""
import threading

import numpy as np

from .._native import STATUS_MESSAGES, DortContext, PackedBatch
from ..core.error import SMRTError, smrt_warn
from ..core.result import LabeledArray, make_result
from ..core.snowpack import Snowpack
from ..emmodel.dmrt_qca_shortrange import DMRT_QCA_ShortRange
from ..emmodel.iba import IBA

_DIAG_METHODS = ("eig", "schur", "schur_forcedtriu", "half_rank_eig", "stamnes88")


class DORT(object):
    """Discrete-ordinate and eigenvalue solver (Picard et al. 2018), device implementation.

    Args follow smrt/rtsolver/dort.py:148-161.  `diagonalization_method` is accepted for compatibility: the device
    always uses its own symmetric reduction (all reference methods agree with it to ~1e-11 K)."""

    _broadcast_capability = {"theta_inc", "polarization_inc", "theta", "phi", "polarization"}

    def __init__(self, n_max_stream=32, m_max=2, stream_mode="most_refringent", phase_normalization="auto",
                 phase_symmetrization=False, error_handling="exception", process_coherent_layers=False,
                 prune_deep_snowpack=None, diagonalization_method="schur_forcedtriu", diagonalization_cache=False,
                 rayleigh_jeans_approximation=False, devices=None, block_threads=0):
        if stream_mode != "most_refringent":
            raise SMRTError("smrt_amd's DORT implements stream_mode='most_refringent' only")
        if phase_symmetrization or process_coherent_layers:
            raise SMRTError("phase_symmetrization and process_coherent_layers are outside the scope of smrt_amd's DORT")
        # diagonalization_cache only saves the reference repeated eigen-decompositions of identical layers: accepted,
        # no effect here.  prune_deep_snowpack: True means an optical depth of 6 (smrt/rtsolver/dort.py:176-178)
        if prune_deep_snowpack is True:
            prune_deep_snowpack = 6
        if prune_deep_snowpack is not None and prune_deep_snowpack is not False and not float(prune_deep_snowpack) > 0:
            raise SMRTError("prune_deep_snowpack must be None, True or a positive optical depth")
        self.prune_deep_snowpack = float(prune_deep_snowpack) if prune_deep_snowpack else None
        if diagonalization_method not in _DIAG_METHODS:
            raise SMRTError(f"Unknown method '{diagonalization_method}' to diagonalize the matrix")
        if error_handling not in ("exception", "nan"):
            raise SMRTError("error_handling must be 'exception' or 'nan'")
        if phase_normalization not in (True, False, "auto", "forced"):
            raise SMRTError("phase_normalization must be True, False, 'auto' or 'forced'")
        self.n_max_stream = int(n_max_stream)
        self.m_max = int(m_max)
        self.stream_mode = stream_mode
        self.phase_normalization = phase_normalization
        self.error_handling = error_handling
        self.diagonalization_method = diagonalization_method
        self.rayleigh_jeans_approximation = bool(rayleigh_jeans_approximation)
        self.devices = devices
        self.block_threads = int(block_threads)

    # ---- the reference's protocol --------------------------------------------------------------------------
    def solve(self, snowpack, emmodels, sensor, atmosphere=None, parallel_computation=None):
        """Solve one (snowpack, sensor-configuration); `emmodels` are the per-layer instances made by
        Model.prepare_emmodels (only their class and layer are used: the device recomputes their numbers)."""
        if atmosphere is not None and snowpack.atmosphere is None:  # the deprecated route of Model.run (model.py:612)
            snowpack = Snowpack(layers=snowpack.layers, interfaces=snowpack.interfaces, substrate=snowpack.substrate,
                                atmosphere=atmosphere)
        emmodel_cls = type(emmodels[0]) if emmodels else IBA
        if any(type(e) is not emmodel_cls for e in emmodels):
            raise SMRTError("smrt_amd's DORT needs the same emmodel in all the layers")
        return self.solve_batch([(sensor, snowpack)], emmodel_cls)[0]

    # ---- batched entry point -------------------------------------------------------------------------------
    def solve_batch(self, simulations, emmodel_cls=IBA):
        """simulations: sequence of (sensor, snowpack) with single-frequency sensors.  Returns one Result each."""
        simulations = list(simulations)
        if not simulations:
            return []
        device_name = getattr(emmodel_cls, "device_name", None)
        if device_name is None:
            raise SMRTError(f"emmodel {emmodel_cls} has no device implementation in smrt_amd (iba, "
                            "dmrt_qca_shortrange)")
        results = [None] * len(simulations)
        # group by everything that must be uniform inside one device batch
        groups = {}
        for i, (sensor, sp) in enumerate(simulations):
            self._check_sensor(sensor)
            micro = {lay.microstructure_model for lay in sp.layers}
            if len(micro) != 1:
                raise SMRTError("smrt_amd's DORT needs the same microstructure model in all the layers")
            angles = sensor.theta_inc_deg if sensor.mode == "A" else sensor.theta_deg
            key = (sensor.mode, micro.pop(), tuple(np.round(angles, 12)), float(np.ravel(sensor.phi)[0]),
                   getattr(sp.substrate, "device_kind", None), id(sp.atmosphere) if sp.atmosphere is not None else None)
            groups.setdefault(key, []).append(i)
        for (mode, micro, _, phi, _sub, _atm), idx in groups.items():
            self._run_group(simulations, idx, device_name, mode, micro, phi, results)
        return results

    def _check_sensor(self, sensor):
        if np.ndim(sensor.frequency) != 0:
            raise SMRTError("DORT does not broadcast the frequency: split the sensor first (Model.run does)")
        if np.size(sensor.phi) > 1:
            raise SMRTError("phi as an array must be implemented")
        if sensor.mode == "A" and not np.array_equal(sensor.theta_deg, sensor.theta_inc_deg):
            raise SMRTError("smrt_amd's DORT computes the backscatter (theta == theta_inc) in active mode")

    def _run_group(self, simulations, idx, device_name, mode, micro, phi, results):
        # distinct snowpacks / frequencies; the device batch is the Cartesian product S x F
        sps, sp_index = [], {}
        freqs, f_index = [], {}
        for i in idx:
            sensor, sp = simulations[i]
            if id(sp) not in sp_index:
                sp_index[id(sp)] = len(sps)
                sps.append(sp)
            f = float(sensor.frequency)
            if f not in f_index:
                f_index[f] = len(freqs)
                freqs.append(f)
        S, F = len(sps), len(freqs)
        Lmax = max(sp.nlayer for sp in sps)
        shape = (S, Lmax)
        thick, fv, temp = np.ones(shape), np.full(shape, 0.3), np.full(shape, 260.0)
        p1, p2 = np.full(shape, 1e-4), np.full(shape, 0.2)
        nl = np.empty(S, np.int32)
        for s, sp in enumerate(sps):
            n = sp.nlayer
            nl[s] = n
            thick[s, :n] = [lay.thickness for lay in sp.layers]
            fv[s, :n] = [lay.frac_volume for lay in sp.layers]
            temp[s, :n] = [lay.temperature for lay in sp.layers]
            pp = [lay.microstructure.device_params for lay in sp.layers]
            p1[s, :n] = [a for a, _ in pp]
            p2[s, :n] = [b for _, b in pp]
        sensor0 = simulations[idx[0]][0]
        angles = sensor0.theta_inc if mode == "A" else sensor0.theta
        substrate = atmosphere = None
        sub0 = sps[0].substrate
        if sub0 is not None:  # one kind per group; permittivity / reflection per (frequency, snowpack)
            q = np.array([[sp.substrate.device_params(f) for sp in sps] for f in freqs])  # (F, S, 2)
            ts = [sp.substrate.temperature if sp.substrate.temperature is not None else 0.0 for sp in sps]
            substrate = (sub0.device_kind, q[:, :, 0], q[:, :, 1], ts)
        atm0 = sps[0].atmosphere
        if atm0 is not None and mode == "P":  # one atmosphere object per group; ignored in active mode (reference)
            a = np.array([atm0.device_params(f) for f in freqs])  # (F, 3)
            atmosphere = (a[:, 0], a[:, 1], a[:, 2])
        batch = PackedBatch(nl, thick, fv, temp, p1, p2, freqs, angles, emmodel=device_name, microstructure=micro,
                            mode=mode, n_max_stream=self.n_max_stream, m_max=self.m_max,
                            phase_normalization=self.phase_normalization,
                            rayleigh_jeans=self.rayleigh_jeans_approximation, phi=phi, substrate=substrate,
                            atmosphere=atmosphere, prune_deep_snowpack=self.prune_deep_snowpack)
        wanted = np.array([f_index[float(simulations[i][0].frequency)] * S + sp_index[id(simulations[i][1])]
                           for i in idx])
        out = run_on_devices(batch, self.devices, self.block_threads, needed=np.unique(wanted))
        for i, pidx in zip(idx, wanted):
            sensor, sp = simulations[i]
            st = int(out.status[pidx])
            if st != 0 and self.error_handling == "exception":
                raise SMRTError(STATUS_MESSAGES.get(st, f"DORT failed with status {st}"))
            results[i] = self._make_result(sensor, sp, out, pidx)

    def _make_result(self, sensor, sp, out, p):
        """Labels and diagnostics of DiscreteOrdinatesMixin.make_result (rtsolver_utils.py:322-344,373-398)."""
        L = sp.nlayer
        if sensor.mode == "P":
            coords = [("polarization", ["V", "H"]), ("theta", sensor.theta_deg)]
        else:
            pola = ["V", "H", "U"]
            coords = [("polarization_inc", pola), ("polarization", pola), ("theta_inc", sensor.theta_inc_deg)]
        n_air = int(out.streams[p, 0])
        outmu = out.streams[p, 1:1 + n_air]
        if sensor.mode == "A":  # only the incident streams are reported (rtsolver_utils.py:307-316, dort.py:210-226)
            keep = set()
            for mu_inc in np.cos(np.atleast_1d(sensor.theta_inc)):
                i0 = int(np.searchsorted(-outmu, -mu_inc))
                keep.update((0,) if i0 == 0 else ((n_air - 1,) if i0 == n_air else (i0, i0 - 1)))
            outmu = outmu[sorted(keep)]
        layer_idx = ("layer", np.arange(L))
        lay = out.layers[p, :L]
        other = {
            "stream_angles": LabeledArray(np.rad2deg(np.arccos(outmu)), [("dim_0", np.arange(len(outmu)))]),
            "effective_permittivity": LabeledArray(lay[:, 0] + 1j * lay[:, 1], [layer_idx]),
            "ks": LabeledArray(lay[:, 2].copy(), [layer_idx], name="ks"),
            "ke": LabeledArray(lay[:, 2] + lay[:, 3], [layer_idx], name="ke"),
            "ka": LabeledArray(lay[:, 3].copy(), [layer_idx], name="ka"),
            "thickness": LabeledArray(sp.layer_thicknesses, [layer_idx], name="thickness"),
        }
        return make_result(sensor, out.values[p], coords, other_data=other)


_ctx_cache = {}
_ctx_lock = threading.Lock()


def get_context(device):
    with _ctx_lock:
        if device not in _ctx_cache:
            _ctx_cache[device] = DortContext(device)
        return _ctx_cache[device]


def visible_devices():
    from .._native import device_count

    return list(range(device_count()))


def run_on_devices(batch, devices=None, block_threads=0, needed=None):
    """Run a packed batch, sharding the flattened pair list over the given GPUs (contiguous slices, one host thread
    and one context per GPU, no collective: the results land in disjoint rows of the same host arrays)."""
    from .._native import BatchOutput

    n = batch.n_pairs
    if devices is None:
        devices = visible_devices() if n >= 4096 else [0]
    devices = list(devices) or [0]
    if len(devices) == 1:
        ctx = get_context(devices[0])
        ctx.set_block_threads(block_threads)
        return ctx.run(batch)
    out = BatchOutput(batch, n)
    bounds = np.linspace(0, n, len(devices) + 1).astype(np.int64)
    errors = []

    def work(dev, lo, hi):
        try:
            ctx = get_context(dev)
            ctx.set_block_threads(block_threads)
            part = ctx.run(batch, int(lo), int(hi - lo))
            out.values[lo:hi], out.status[lo:hi] = part.values, part.status
            out.layers[lo:hi], out.streams[lo:hi] = part.layers, part.streams
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(d, bounds[k], bounds[k + 1]))
               for k, d in enumerate(devices) if bounds[k + 1] > bounds[k]]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out
