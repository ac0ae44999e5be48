"""DORT radiative-transfer solver, MI355X-native (drop-in for smrt/rtsolver/dort.py:84-261).

Same constructor options and `solve()` protocol as the reference class; the numerical work (layer electromagnetics,
streams, interfaces, eigen-decomposition, boundary conditions) runs in the HIP kernels through the C ABI
(include/smrt_dort.h).  Three entry points, from the reference's to the batched one:

* `solve(snowpack, emmodels, sensor, atmosphere)` -- the rtsolver protocol, one simulation (what
  `Model.run_single_simulation` calls, smrt/core/model.py:596-617);
* `solve_batch(simulations, emmodel)` -- a list of (single-frequency sensor, snowpack) pairs in one launch per group,
  one Result each;
* `solve_plan(model, plan)` -- a whole `SimulationPlan` (smrt_amd/core/model.py): index vectors instead of pair
  objects, distinct snowpacks packed once, the stacked Result built from the output array without per-pair objects.
"""
import operator
import threading
from collections.abc import Mapping

import numpy as np

from .._native import STATUS_MESSAGES, BatchOutput, DortContext, PackedBatch, device_count
from ..core.error import SMRTError, smrt_warn
from ..core.globalconstants import C_SPEED
from ..core.foreign import result_factory
from ..core.result import LabeledArray, make_result
from ..core.snowpack import Snowpack, substrate_kind
from ..interface.flat import Flat

# the row blocks and the stacked columns of the last group that was packed (DORT._pack): a repeated Model.run on the same,
# unchanged snowpacks copies the columns instead of stacking thousands of small arrays again

_DIAG_METHODS = ("eig", "schur", "schur_forcedtriu", "half_rank_eig", "stamnes88")


class DORT(object):
    """Discrete-ordinate and eigenvalue solver (Picard et al. 2018), device implementation.

    Arguments as smrt/rtsolver/dort.py:148-161.  `diagonalization_method` and `diagonalization_cache` are accepted for
    compatibility: the device always uses its own symmetric reduction -- Cholesky + a symmetric eigensolver of B B^T
    (passive mode up to 32 streams) or a one-sided Jacobi iteration on B (everything else); every reference method
    agrees with it to ~1e-9 K -- and has nothing to cache.  `devices` (list of GPU indices, default: all visible ones for large batches)
    and `block_threads` are smrt_amd's own knobs."""

    _broadcast_capability = {"theta_inc", "polarization_inc", "theta", "phi", "polarization"}

    def __init__(self, n_max_stream=32, m_max=2, stream_mode="most_refringent", phase_normalization="auto",
                 phase_symmetrization=False, error_handling="exception", process_coherent_layers=False,
                 prune_deep_snowpack=None, diagonalization_method="schur_forcedtriu", diagonalization_cache=False,
                 rayleigh_jeans_approximation=False, devices=None, block_threads=0):
        if stream_mode != "most_refringent":
            # the reference's two other modes do not run in the reference either: "uniform_air" always fails the
            # assertion at smrt/rtsolver/streams.py:288 (np.size of a scalar stream count is 1, never > 2), "air" is
            # announced as untested there (streams.py:164-175)
            raise SMRTError("smrt_amd's DORT implements stream_mode='most_refringent' only (the reference's "
                            "'uniform_air' raises an AssertionError for every snowpack, its 'air' is untested code)")
        if prune_deep_snowpack is True:  # True means an optical depth of 6 (smrt/rtsolver/dort.py:176-178)
            prune_deep_snowpack = 6
        if prune_deep_snowpack not in (None, False) and not float(prune_deep_snowpack) > 0:
            raise SMRTError("prune_deep_snowpack must be None, True or a positive optical depth")
        if diagonalization_method not in _DIAG_METHODS:
            raise SMRTError(f"Unknown method '{diagonalization_method}' to diagonalize the matrix")
        if diagonalization_method != "schur_forcedtriu":
            # not silently: 'stamnes88' in the reference differs from its other methods by up to ~1 K
            # (smrt/test/test_integration_iba.py atol table); the device has ONE route, equal to the default to 1e-8 K
            smrt_warn(f"diagonalization_method='{diagonalization_method}' is ignored: smrt_amd's DORT always diagonalises with "
                      "its own symmetric reduction (Cholesky + a symmetric eigensolver or a one-sided Jacobi iteration), which reproduces the reference's "
                      "default 'schur_forcedtriu'" + (" -- NOT the 'stamnes88' variant" if diagonalization_method == "stamnes88" else ""))
        if error_handling not in ("exception", "nan"):
            raise SMRTError("error_handling must be 'exception' or 'nan'")
        if phase_normalization not in (True, False, "auto", "forced"):
            raise SMRTError("phase_normalization must be True, False, 'auto' or 'forced'")
        self.n_max_stream = int(n_max_stream)
        self.m_max = int(m_max)
        self.stream_mode = stream_mode
        self.phase_normalization = phase_normalization
        self.error_handling = error_handling
        self.process_coherent_layers = bool(process_coherent_layers)
        # phase_symmetrization (dort.py:104,173; rtsolver_utils.py:743-765) averages the (up, up) / (down, down) and the
        # (up, down) / (down, up) blocks of the phase matrix.  The device formulation is built on exactly that mirror
        # symmetry (it only ever forms P(mu, +mu') and P(mu, -mu')), which the device emmodels have to the last bit -- the
        # reference's own result does not change by one ulp with the option on IBA -- so there it is a no-op; for emmodels
        # evaluated on the host the averaging is applied when their matrices are packed.
        self.phase_symmetrization = bool(phase_symmetrization)
        self.prune_deep_snowpack = float(prune_deep_snowpack) if prune_deep_snowpack else None
        self.diagonalization_method = diagonalization_method
        self.rayleigh_jeans_approximation = bool(rayleigh_jeans_approximation)
        self.devices = devices
        self.block_threads = int(block_threads)

    # ---- the reference's protocol --------------------------------------------------------------------------------
    def solve(self, snowpack, emmodels, sensor, atmosphere=None, parallel_computation=None):
        """One (snowpack, sensor configuration).  `emmodels`: the per-layer instances made by Model.prepare_emmodels;
        they must be smrt_amd's device-backed classes, all of one kind (the device recomputes their numbers from the
        layer properties)."""
        from ..core.foreign import adopt_atmosphere, adopt_snowpack, entry_of_instance

        snowpack = adopt_snowpack(snowpack)     # the reference's own Snowpack (smrt/core/model.py:609-615) or smrt_amd's
        if atmosphere is not None and snowpack.atmosphere is None:  # the deprecated route of Model.run (model.py:612)
            snowpack = Snowpack(layers=snowpack.layers, interfaces=snowpack.interfaces, substrate=snowpack.substrate,
                                atmosphere=adopt_atmosphere(atmosphere))
        if len(emmodels) != snowpack.nlayer:
            raise SMRTError("one emmodel per layer is needed")
        # device-backed instances (and the reference's instances the device reproduces) by the device emmodel of their
        # layer; any other instance is evaluated on the host
        entries = [entry_of_instance(e, layer) for e, layer in zip(emmodels, snowpack.layers)]
        return self.solve_batch([(sensor, snowpack)], [entries])[0]

    # ---- batched entry points ------------------------------------------------------------------------------------
    @staticmethod
    def _device_name(emmodel_cls):
        name = getattr(emmodel_cls, "device_name", None)
        if name is None:
            raise SMRTError(f"emmodel {emmodel_cls} has no device implementation in smrt_amd (iba, dmrt_qca_shortrange, "
                            "dmrt_qcacp_shortrange, nonscattering)")
        return name

    def solve_batch(self, simulations, emmodel):
        """simulations: sequence of (single-frequency sensor, snowpack).  One Result per simulation, in order.
        emmodel: one emmodel class (or device emmodel name) for every layer, or -- aligned with the distinct snowpacks in
        order of first appearance -- a list of per-layer lists of entries (device name, (class, options) pair or ready
        instance) for snowpacks that mix emmodels.  The snowpacks may be the reference's own objects (core/foreign.py)."""
        from ..core.foreign import adopt_snowpack

        sensors, packs, si, pi = [], [], [], []
        seen_s, seen_p, memo = {}, {}, {}
        for sensor, sp in simulations:
            sp = adopt_snowpack(sp, memo)
            si.append(seen_s.setdefault(id(sensor), len(sensors)))
            if si[-1] == len(sensors):
                sensors.append(sensor)
            pi.append(seen_p.setdefault(id(sp), len(packs)))
            if pi[-1] == len(packs):
                packs.append(sp)
        if not si:
            return []
        names = emmodel if isinstance(emmodel, (list, str)) else self._device_name(emmodel)
        sol = self._solve_indexed(sensors, packs, np.asarray(si), np.asarray(pi), names)
        return [sol.result(i) for i in range(len(si))]

    def solve_plan(self, model, plan):
        """The whole plan of a Model.run: returns the nested Result directly."""
        from ..core.model import nest_results

        # one freshness check per snowpack and run: the snapshot lives on THIS solver for the duration of the solve (never on
        # the snowpack: a layer changed after the run must not meet a stale tuple in a later solve_batch / solve)
        self._plan_facts = {id(sp): sp.layer_facts() for sp in plan.snowpacks}
        self._plan_model = model
        try:
            names = self.emmodel_names(model, plan, self._plan_facts)
            sol = self._solve_indexed(plan.sensors, plan.snowpacks, plan.sensor_index, plan.snowpack_index, names)
            stacked = sol.stacked_result(plan)
        finally:
            self._plan_facts = self._plan_model = None
        if stacked is not None:
            return stacked
        return nest_results([sol.result(i) for i in range(len(plan))], plan.dimensions)

    _plan_facts = _plan_model = None   # (set by solve_plan for the duration of one solve)

    @classmethod
    def emmodel_names(cls, model, plan, facts=None):
        """The emmodel of every layer as the device knows it, after the same checks the per-simulation route applies
        through Model.prepare_emmodels (per-layer overrides, lists / dicts of emmodels and emmodel options are honoured
        or refused, never dropped): one device name when all the layers of all the snowpacks share it, otherwise a list
        (per snowpack) of lists (per layer).  One instance is made per distinct (class, options) pair, which validates
        the options against the class.  `model` is smrt_amd's Model or the reference's (only its public attributes
        `emmodel` and `emmodel_options` are read, smrt/core/model.py:254-283); the snowpacks of the plan are smrt_amd's
        or adopted ones (core/foreign.py)."""
        from ..core.foreign import device_entry, is_native, model_make_emmodel
        from ..core.model import is_sequence, select_emmodel, select_emmodel_options

        emmodel, all_options = model.emmodel, model.emmodel_options
        make = None if is_native(model) else model_make_emmodel(model)
        simple = isinstance(emmodel, type)
        checked = set()
        per_pack, distinct = [], set()
        simple_name = getattr(emmodel, "device_name", None) if simple else None
        simple_options = None
        # (dense_snow_correction="auto" is checked layer by layer)
        plain_model = simple and not is_sequence(all_options) and all_options.get("dense_snow_correction") != "auto"
        for sp in plan.snowpacks:
            f = (facts.get(id(sp)) if facts else None) or sp.layer_facts()
            n = f[0].shape[1]
            plain = plain_model and not f[2]
            if plain and simple_name is not None and simple_options is not None and not hasattr(sp, "source"):
                # the common case -- one device emmodel, no per-layer settings, options already validated, smrt_amd's own
                # layers (nothing the device could not compute): no per-layer work
                distinct.add(simple_name)
                per_pack.append([simple_name] * n)
                continue
            if plain:
                kinds = [emmodel] * n
                options = [all_options] * n
                todo = [0]
                simple_options = True
            else:
                kinds = [select_emmodel(emmodel, k, layer, n, make) for k, layer in enumerate(sp.layers)]
                options = [select_emmodel_options(emmodel, all_options, layer, k, n) for k, layer in enumerate(sp.layers)]
                todo = range(n)
            for k in todo:
                kind, layer, opts = kinds[k], sp.layers[k], options[k]
                key = (kind, tuple(sorted(opts.items())), opts.get("dense_snow_correction") == "auto" and layer.frac_volume > 0.5)
                if key not in checked:
                    checked.add(key)
                    # validates the options against the class (a class of the reference sees the reference's layer)
                    kind(plan.sensors[0], layer if is_native(kind) else getattr(layer, "source", layer), **opts)
            # a class without a device implementation is evaluated on the host, layer by layer (_evaluate_on_host)
            names = [device_entry(kd, opts, layer) for kd, opts, layer in zip(kinds, options, sp.layers)]
            if any(not isinstance(e, str) for e in names):
                # the snowpack goes to the host route as a whole: a layer whose class of the REFERENCE package was mapped
                # to a device emmodel is evaluated by that class as well (one implementation per snowpack, and no device
                # round trip per layer through smrt_amd's descriptor)
                names = [(kd, dict(opts)) if isinstance(e, str) and not getattr(kd, "device_name", None) else e
                         for e, kd, opts in zip(names, kinds, options)]
            distinct.update(n if isinstance(n, str) else "host" for n in names)
            per_pack.append(names)
        return distinct.pop() if len(distinct) == 1 and "host" not in distinct else per_pack

    # ---- grouping, packing, launching ----------------------------------------------------------------------------
    def _check_sensor(self, sensor):
        if np.ndim(sensor.frequency) != 0:
            raise SMRTError("DORT does not broadcast the frequency: split the sensor first (Model.run does)")
        if np.size(sensor.phi) > 1:
            raise SMRTError("phi as an array must be implemented")
        if sensor.mode == "A" and not np.array_equal(sensor.theta_deg, sensor.theta_inc_deg):
            raise SMRTError("smrt_amd's DORT computes the backscatter (theta == theta_inc) in active mode")

    def _solve_indexed(self, sensors, packs, sens_idx, pack_idx, emmodel_names):
        # everything that must be uniform inside one device batch, as small integer codes per sensor / per snowpack
        sensor_keys, pack_keys = {}, {}
        s_code = np.empty(len(sensors), np.int64)
        for k, sensor in enumerate(sensors):
            self._check_sensor(sensor)
            angles = sensor.theta_inc_deg if sensor.mode == "A" else sensor.theta_deg
            key = (sensor.mode, tuple(np.round(angles, 12)), float(np.ravel(sensor.phi)[0]))
            s_code[k] = sensor_keys.setdefault(key, len(sensor_keys))
        p_code = np.empty(len(packs), np.int64)
        for k, sp in enumerate(packs):
            on_host = not isinstance(emmodel_names, str) and any(not isinstance(e, str) for e in emmodel_names[k])
            key = (substrate_kind(sp.substrate), id(sp.atmosphere) if sp.atmosphere is not None else None, on_host)
            p_code[k] = pack_keys.setdefault(key, len(pack_keys))
        freq = np.array([float(s.frequency) for s in sensors])
        code = s_code[sens_idx] * len(pack_keys) + p_code[pack_idx]
        sol = _Solution(self, sensors, packs, sens_idx, pack_idx)
        for g in np.unique(code):
            sel = np.nonzero(code == g)[0]
            u_packs, inv_p = np.unique(pack_idx[sel], return_inverse=True)
            u_freq, inv_f = np.unique(freq[sens_idx[sel]], return_inverse=True)
            sensor0, sp0 = sensors[sens_idx[sel[0]]], packs[u_packs[0]]
            names = emmodel_names if isinstance(emmodel_names, str) else [emmodel_names[k] for k in u_packs]
            sensor_of = {float(sensors[k].frequency): sensors[k] for k in sens_idx[sel]}
            batch = self._pack(sensor0, [packs[k] for k in u_packs], u_freq, names, sensor_of)
            pairs = inv_f * len(u_packs) + inv_p
            full = len(pairs) == batch.n_pairs and np.array_equal(pairs, np.arange(batch.n_pairs))
            out = run_on_devices(batch, self.devices, self.block_threads, pairs=None if full else pairs)
            bad = np.nonzero(out.status != 0)[0]
            if len(bad) and self.error_handling == "exception":
                st = int(out.status[bad[0]])
                raise SMRTError(STATUS_MESSAGES.get(st, f"DORT failed with status {st}"))
            sol.add_group(sel, out, sp0, (u_packs, np.array(batch.n_layers, np.int64), np.array(batch.thickness, float)))
        return sol

    def _pack(self, sensor0, sps, freqs, emmodel_names, sensor_of=None):
        """The device batch of one group: S distinct snowpacks x F distinct frequencies."""
        from .._native import EM_CODES, MS_CODES

        S = len(sps)
        known = self._plan_facts or {}
        facts = [known.get(id(sp)) or sp.layer_facts() for sp in sps]   # (packed, microstructures, overrides, liquid water)
        nl = np.fromiter((f[0].shape[1] for f in facts), np.int32, S)
        Lmax = int(nl.max())
        # emmodel + 16 * microstructure per layer; handed to the device only when the batch really mixes them
        micro = [f[1] for f in facts]
        uniform_micro = len(set().union(*micro)) == 1
        layer_kind = host = None
        if not isinstance(emmodel_names, str):
            for s, sp in enumerate(sps):
                if len(emmodel_names[s]) != nl[s]:
                    raise SMRTError("one emmodel per layer is needed")
        scalars = None
        if not isinstance(emmodel_names, str) and any(not isinstance(e, str) for row in emmodel_names for e in row):
            # at least one emmodel without a device implementation.  Emmodels of IBA's family on a microstructure model the
            # device has hand over their scalars only -- the phase matrices are assembled on the device
            # (SMRT_EM_IBA_HOST); otherwise the whole group is evaluated through the emmodel protocol on the host (the
            # device classes speak it too) and handed to the device as numbers, phase matrices included (SMRT_EM_HOST)
            scalars = self._iba_scalars_on_host(sensor0, sps, freqs, emmodel_names, nl, Lmax, sensor_of or {})
            if scalars is not None:
                layer_kind = scalars[2]
            else:
                host = self._evaluate_on_host(sensor0, sps, freqs, emmodel_names, nl, Lmax, sensor_of or {})
                layer_kind = np.full((S, Lmax), EM_CODES["host"], np.int32)
        elif not (isinstance(emmodel_names, str) and uniform_micro):
            layer_kind = np.zeros((S, Lmax), np.int32)
            for s, sp in enumerate(sps):
                em = [emmodel_names] * nl[s] if isinstance(emmodel_names, str) else emmodel_names[s]
                layer_kind[s, :nl[s]] = [EM_CODES[e] + 16 * self._ms_code(lay) for e, lay in zip(em, sp.layers)]
        if host is None and scalars is None:
            from ..core.layer import DEVICE_MICROSTRUCTURES

            foreign = set().union(*micro) - set(DEVICE_MICROSTRUCTURES)
            if foreign:
                raise SMRTError(f"the microstructure model(s) {sorted(foreign)} have no device implementation: they can only "
                                "be used with an emmodel evaluated on the host (e.g. rayleigh, prescribed_kskaeps)")
        device_name = "host" if host is not None else "iba" if scalars is not None else \
            (emmodel_names if isinstance(emmodel_names, str) else emmodel_names[0][0])
        if int(nl.min()) == Lmax:
            # (5, S, L); one concatenate + reshape (np.stack reshapes every one of the S small arrays in Python first), and
            # not even that when the row blocks are the very objects of the previous run (unchanged snowpacks keep theirs)
            # (kept on the MODEL of the run, behind its lock -- never process-wide: two models on two threads do not share a
            # slot, and the arrays go when the model goes)
            rows = [f[0] for f in facts]
            keeper = getattr(self._plan_model, "_kept_columns", None) if S >= 256 else None
            cols = None
            if keeper is not None:
                with keeper[0]:
                    kept = keeper[1]
                    if kept is not None and len(kept[0]) == S and all(map(operator.is_, rows, kept[0])):
                        cols = kept[1].copy()
            if cols is None:
                cols = np.concatenate(rows, axis=1).reshape(5, S, Lmax)
                if keeper is not None:
                    with keeper[0]:
                        keeper[1] = (rows, cols.copy())
        else:
            cols = np.empty((5, S, Lmax))
            cols[0], cols[1], cols[2], cols[3], cols[4] = 1.0, 0.3, 260.0, 1e-4, 0.2   # harmless padding
            for s, sp in enumerate(sps):
                cols[:, s, :nl[s]] = facts[s][0]
        if scalars is not None:
            # the medium every emmodel object works on (its own frac_volume and microstructure: inverted above half ice
            # under dense_snow_correction="auto"), read while the scalars were taken
            cols[1], cols[3], cols[4] = scalars[3], scalars[4], scalars[5]
        elif layer_kind is not None and host is None:
            # IBA on the inverted medium (dense_snow_correction="auto" above half ice): the device takes the volume
            # fraction of the inclusions, the air (include/smrt_dort.h: SMRT_EM_IBA_INVERTED)
            inverted = (layer_kind & 15) == EM_CODES["iba_inverted"]
            cols[1][inverted] = 1.0 - cols[1][inverted]
        elif device_name == "iba_inverted":   # every layer of the batch
            cols[1] = 1.0 - cols[1]
        # wet snow: the optional liquid-water column (water / (ice + water) volume; the frac_volume column is ice + water)
        wet = [f[3] for f in facts]
        liquid_water = None
        if any(w is not None for w in wet):
            liquid_water = np.zeros((S, Lmax))
            for s, w in enumerate(wet):
                if w is not None:
                    liquid_water[s, :nl[s]] = w
        mode = sensor0.mode
        substrate = atmosphere = None
        sub0 = sps[0].substrate
        if sub0 is not None and substrate_kind(sub0) == "host":
            substrate = self._substrates_on_host(sensor0, sps, freqs, cols, nl, emmodel_names, layer_kind, host,
                                                     scalars=scalars, liquid_water=liquid_water)
        elif sub0 is not None:  # one kind per group; permittivity / reflection per (frequency, snowpack)
            q = np.array([[sp.substrate.device_params(f) for sp in sps] for f in freqs])  # (F, S, 2)
            ts = [sp.substrate.temperature if sp.substrate.temperature is not None else 0.0 for sp in sps]
            substrate = (sub0.device_kind, q[:, :, 0], q[:, :, 1], ts)
        host_interfaces = None
        if not all(sp.all_interfaces_flat() for sp in sps):
            host_interfaces = self._interfaces_on_host(sensor0, sps, freqs, cols, nl, emmodel_names, layer_kind, host,
                                                            scalars=scalars, liquid_water=liquid_water)
        atm0 = sps[0].atmosphere
        if atm0 is not None and mode == "P":  # one atmosphere object per group; ignored in active mode (reference)
            a = np.array([atm0.device_params(f) for f in freqs])  # (F, 3)
            atmosphere = (a[:, 0], a[:, 1], a[:, 2])
        return PackedBatch(nl, cols[0], cols[1], cols[2], cols[3], cols[4], freqs,
                           sensor0.theta_inc if mode == "A" else sensor0.theta, emmodel=device_name,
                           microstructure=sps[0].layers[0].microstructure_model if host is None and scalars is None else "exponential",
                           mode=mode,
                           n_max_stream=self.n_max_stream, m_max=self.m_max,
                           phase_normalization=self.phase_normalization,
                           rayleigh_jeans=self.rayleigh_jeans_approximation, phi=float(np.ravel(sensor0.phi)[0]),
                           substrate=substrate, atmosphere=atmosphere, prune_deep_snowpack=self.prune_deep_snowpack,
                           layer_kind=layer_kind, host_emmodel=host,
                           host_scalars=None if scalars is None else scalars[:2],
                           process_coherent_layers=self.process_coherent_layers, host_interfaces=host_interfaces,
                           liquid_water=liquid_water)

    def _layer_permittivities(self, sps, freqs, cols, nl, emmodel_names, layer_kind, host, scalars=None, liquid_water=None):
        """Effective permittivity of every layer, (F, S, Lmax): from the host-evaluated emmodels if the group has them,
        otherwise from a cheap pre-pass of the device emmodels (four streams, layer diagnostics only) -- what the streams
        of matrices evaluated on the host (rough substrates / interfaces) are placed with."""
        from .._native import PackedBatch

        F, S, Lmax = len(freqs), len(sps), cols.shape[2]
        if host is not None:
            return host[0][..., 2] + 1j * host[0][..., 3]
        scal = scalars   # emmodels of IBA's family: their scalars travel with the probe
        name = "iba" if scal is not None else emmodel_names if isinstance(emmodel_names, str) else emmodel_names[0][0]
        probe = PackedBatch(nl, cols[0], cols[1], cols[2], cols[3], cols[4], freqs, [0.0], emmodel=name,
                            microstructure=sps[0].layers[0].microstructure_model if scal is None else "exponential",
                            n_max_stream=4, phase_normalization="forced", layer_kind=layer_kind,
                            host_scalars=None if scal is None else scal[:2],
                            liquid_water=liquid_water)
        # on the first of the solver's own devices (the device of this rank), not on GPU 0 whatever the caller chose
        res = get_context((self.devices or [default_device()])[0]).run(probe)
        bad = np.flatnonzero(res.status == 5)   # 5 = invalid layer input: the permittivities below would be meaningless
        if len(bad):
            raise SMRTError("the layer electromagnetics of pair %d are not computable (status 5): cannot place the "
                            "streams of the matrices evaluated on the host" % int(bad[0]))
        lay = res.layers.reshape(F, S, Lmax, 5)
        return lay[..., 0] + 1j * lay[..., 1]

    def _staying_layers(self, frequency, eps_layers, thickness):
        """Indices of the layers process_coherent_layers leaves in the snowpack at this frequency (all of them without the
        option): a layer with k0 Re(n) d < 3 pi / 4 is collapsed into the interface below it
        (smrt/interface/coherent_flat.py:16-30) -- the device applies the same criterion to the same permittivities."""
        if not self.process_coherent_layers:
            return list(range(len(eps_layers)))
        k0 = 2.0 * np.pi * float(frequency) / C_SPEED
        return [l for l in range(len(eps_layers))
                if not k0 * np.sqrt(complex(eps_layers[l])).real * thickness[l] < 0.75 * np.pi]

    # ---- rough interfaces evaluated on the host (include/smrt_dort.h: SMRT_INTERFACE_HOST) --------------------------
    @staticmethod
    def _streams_of(eps_layers, n_max_stream):
        """mu and weights of every layer and of the air for one (snowpack, frequency): streams.py:136-223,300-330."""
        from .._native import gauss_legendre_positive

        gmu, _ = gauss_legendre_positive(n_max_stream)
        gsin = np.sqrt(1.0 - gmu * gmu)
        e = np.asarray(eps_layers, complex)
        star = max(range(len(e)), key=lambda l: (e[l].real, e[l].imag, -l))

        def weights(mu, absolute):
            w = np.empty_like(mu)
            w[0], w[-1] = 1.0 - 0.5 * (mu[0] + mu[1]), 0.5 * (mu[-2] + mu[-1])
            w[1:-1] = 0.5 * (mu[:-2] - mu[2:])
            return np.abs(w) if absolute else w
        mus, ws = [], []
        for el in list(e) + [1.0 + 0j]:
            rs = np.sqrt(e[star] / el).real * gsin
            mu = np.sqrt(1.0 - rs[rs < 1.0] ** 2)
            mus.append(mu)
            ws.append(weights(mu, absolute=len(mus) <= len(e)))
        return mus[:-1], ws[:-1], mus[-1], ws[-1]

    @staticmethod
    def interface_matrices(interface, frequency, eps_low, eps_up, mu_low, mu_up, mu_t_up, w_low, w_up, m_max, npol):
        """The four matrices of a rough interface between a layer (eps_low, streams mu_low / weights w_low) and the medium
        above it (eps_up, mu_up / w_up) as compute_interface_properties combines them (smrt/rtsolver/rtsolver_utils.py:
        473-642,690-707,728-740): per azimuth mode m, in the compressed order (stream * P + polarisation, P = 2 for mode 0,
        3 above),  specular / coherent part on the diagonal + (2 pi | pi) x the diffuse mode with the column scaled by
        mu_i w_i and the row by 1 / mu_s (a diffuse part given as [P, m, n] is diagonal in the streams).  The diffuse
        transmissions carry the ratio of the real permittivities (incident / transmitted side).  mu_t_up: the cosines the
        reference evaluates the upward diffuse transmission on (streams.mu[layer - 1] for layer > 1, the air streams
        otherwise, rtsolver_utils.py:510).  Returns (list over modes of {"Rtop", "Ttop", "Rbot", "Tbot"},
        {"Rtop", ...: specular diagonal of mode 0})."""
        def raw(method, *args):
            f = getattr(interface, method, None)
            if not callable(f):
                return None
            v = f(frequency, *args)
            v = np.asarray(getattr(v, "values", v), float)
            return None if v.ndim == 0 else v

        def diag_of(spec, n, P):
            return np.zeros(n * P) if spec is None else spec.reshape(npol, n)[:P].T.reshape(n * P)

        def combined(spec, diff, m, mu_st, mu_i, w_i, scale, same):
            P = 2 if m == 0 else 3
            coef = 2 * np.pi if m == 0 else np.pi
            M = np.diag(diag_of(spec, len(mu_i), P))
            if diff is None:
                return M
            if diff.ndim == 5:      # [ps, pi, m, mu_st, mu_i]
                D = diff[:P, :P, m] * scale * (mu_i * w_i)[None, None, None, :] / mu_st[None, None, :, None]
                D = np.transpose(D, (2, 0, 3, 1)).reshape(len(mu_st) * P, len(mu_i) * P)
                if M.shape != D.shape:   # rectangular transmission: the specular part sits on the common streams
                    Mr = np.zeros_like(D)
                    k = min(D.shape[0], M.shape[0])
                    Mr[:k, :k] = M[:k, :k]
                    M = Mr
                return M + coef * D
            if diff.ndim == 3:      # [p, m, mu]: diagonal in the streams and in the polarisation
                fac = w_i if same else mu_i * w_i / mu_st
                return M + coef * np.diag((diff[:P, m] * scale * fac[None, :]).T.reshape(len(mu_i) * P))
            raise SMRTError("unsupported layout of a diffuse interface matrix (expected [p, p, m, mu_s, mu_i] or [p, m, mu])")

        spec_up = raw("specular_reflection_matrix", eps_low, eps_up, mu_low, npol)
        spec_dn = raw("specular_reflection_matrix", eps_up, eps_low, mu_up, npol)
        ctr_up = raw("coherent_transmission_matrix", eps_low, eps_up, mu_low, npol)
        ctr_dn = raw("coherent_transmission_matrix", eps_up, eps_low, mu_up, npol)
        drf_up = raw("ft_even_diffuse_reflection_matrix", eps_low, eps_up, mu_low, mu_low, m_max, npol)
        drf_dn = raw("ft_even_diffuse_reflection_matrix", eps_up, eps_low, mu_up, mu_up, m_max, npol)
        dtr_up = raw("ft_even_diffuse_transmission_matrix", eps_low, eps_up, mu_t_up, mu_low, m_max, npol)
        dtr_dn = raw("ft_even_diffuse_transmission_matrix", eps_up, eps_low, mu_low, mu_up, m_max, npol)
        r_up, r_dn = complex(eps_low).real / complex(eps_up).real, complex(eps_up).real / complex(eps_low).real
        modes = []
        for m in range(m_max + 1):
            modes.append(dict(Rtop=combined(spec_up, drf_up, m, mu_low, mu_low, w_low, 1.0, True),
                              Ttop=combined(ctr_up, dtr_up, m, mu_t_up, mu_low, w_low, r_up, False),
                              Rbot=combined(spec_dn, drf_dn, m, mu_up, mu_up, w_up, 1.0, True),
                              Tbot=combined(ctr_dn, dtr_dn, m, mu_low, mu_up, w_up, r_dn, False)))
        coh = dict(Rtop=diag_of(spec_up, len(mu_low), 2), Ttop=diag_of(ctr_up, len(mu_low), 2),
                   Rbot=diag_of(spec_dn, len(mu_up), 2), Tbot=diag_of(ctr_dn, len(mu_up), 2))
        return modes, coh

    def _interfaces_on_host(self, sensor0, sps, freqs, cols, nl, emmodel_names, layer_kind, host, scalars=None, liquid_water=None):
        """(slot, matrices, specular diagonals) of PackedBatch(host_interfaces=...) for a group with rough interfaces: every
        interface object that is not Flat is evaluated through the reference's interface protocol on the streams of the
        two media it separates."""
        act = sensor0.mode == "A"
        F, S, Lmax = len(freqs), len(sps), cols.shape[2]
        eps = self._layer_permittivities(sps, freqs, cols, nl, emmodel_names, layer_kind, host, scalars, liquid_water)
        nm, ne, npol = (self.m_max + 1 if act else 1), 3 * self.n_max_stream, (3 if act else 2)
        rough = [[i for i, itf in enumerate(sp.interfaces) if not isinstance(itf, Flat)] for sp in sps]
        nslots = max(1, max(len(r) for r in rough))
        slot = -np.ones((F, S, Lmax), np.int32)
        M = np.zeros((F, S, nslots, nm, 4, ne, ne))
        C = np.zeros((F, S, nslots, 4, ne))
        for fi, f in enumerate(freqs):
            for s, sp in enumerate(sps):
                if not rough[s]:
                    continue
                e = eps[fi, s, :nl[s]]
                # under process_coherent_layers the matrices are sampled on the streams of the REDUCED snowpack; a rough
                # interface on top of or right below a collapsed layer goes into the reference's CoherentFlat, which takes
                # it for a flat one (smrt/interface/coherent_flat.py:37-45): no slot, the device's Fabry-Perot slab stands
                stay = self._staying_layers(f, e, cols[0][s])
                if not stay:
                    continue
                at = {l: r for r, l in enumerate(stay)}
                mus, ws, outmu, outw = self._streams_of(e[stay], self.n_max_stream)
                for k, i in enumerate(rough[s]):
                    if i not in at or (i > 0 and (i - 1) not in at):
                        continue
                    r = at[i]
                    mu_up, w_up, e_up = (mus[r - 1], ws[r - 1], complex(e[i - 1])) if r > 0 else (outmu, outw, 1.0)
                    mu_t = mus[r - 1] if r > 1 else outmu      # (the reference's choice, rtsolver_utils.py:510)
                    modes, coh = self.interface_matrices(sp.interfaces[i], float(f), complex(e[i]), e_up, mus[r], mu_up, mu_t,
                                                         ws[r], w_up, self.m_max if act else 0, npol)
                    slot[fi, s, i] = k
                    for m in range(nm):
                        P = 2 if m == 0 else 3
                        n_low, n_up = len(mus[r]) * P, len(mu_up) * P
                        for q, (kind, rows) in enumerate((("Rtop", n_low), ("Ttop", n_up), ("Rbot", n_up), ("Tbot", n_low))):
                            A = modes[m][kind]
                            rr = min(A.shape[0], rows)     # cut to the common streams like dort.py:372-376,409-414
                            M[fi, s, k, m, q, :rr, :A.shape[1]] = A[:rr]
                    for q, (kind, rows) in enumerate((("Rtop", None), ("Ttop", len(mu_up) * 2), ("Rbot", None), ("Tbot", len(mus[r]) * 2))):
                        c = coh[kind] if rows is None else coh[kind][:rows]
                        C[fi, s, k, q, :len(c)] = c
        return slot, M, C

    # ---- substrates evaluated on the host (include/smrt_dort.h: SMRT_SUBSTRATE_HOST) -------------------------------
    @staticmethod
    def substrate_matrices(substrate, frequency, eps_last, mu, weight, m_max, npol=3):
        """Reflection matrices of the bottom boundary as compute_interface_properties builds them for a substrate object
        (smrt/rtsolver/rtsolver_utils.py:567-597,690-707,728-740): per azimuth mode m, in the compressed order (stream *
        P + polarisation, P = 2 for mode 0 and 3 above), diag(specular) + (2 pi | pi) x the diffuse mode with the column
        scaled by mu_i w_i and the row by 1 / mu_s (a diffuse part given as [P, m, n] is diagonal in the streams: scaled by
        w only).  Returns (list of dense matrices, list of specular diagonals)."""
        n = len(mu)
        spec = substrate.specular_reflection_matrix(frequency, eps_last, mu, npol)
        spec = np.asarray(getattr(spec, "values", spec), float)       # (an smrt_matrix keeps its array in .values)
        spec = np.zeros((npol, n)) if spec.ndim == 0 else spec.reshape(npol, n)
        diff = None
        if callable(getattr(substrate, "ft_even_diffuse_reflection_matrix", None)):
            diff = substrate.ft_even_diffuse_reflection_matrix(frequency, eps_last, mu, mu, m_max, npol)
            diff = np.asarray(getattr(diff, "values", diff), float)
            diff = None if diff.ndim == 0 else diff
        dense, coh = [], []
        for m in range(m_max + 1):
            P = 2 if m == 0 else 3
            c = spec[:P].T.reshape(n * P)
            R = np.diag(c)
            coef = 2 * np.pi if m == 0 else np.pi
            if diff is not None and diff.ndim == 5:      # [ps, pi, m, mu_s, mu_i]
                D = diff[:P, :P, m] * (mu * weight)[None, None, None, :] / mu[None, None, :, None]
                R = R + coef * np.transpose(D, (2, 0, 3, 1)).reshape(n * P, n * P)
            elif diff is not None and diff.ndim == 3:    # [p, m, mu]: diagonal in the streams and in the polarisation
                R = R + coef * np.diag((diff[:P, m] * weight[None, :]).T.reshape(n * P))
            elif diff is not None:
                raise SMRTError(f"ft_even_diffuse_reflection_matrix returned an array of {diff.ndim} dimensions")
            dense.append(R)
            coh.append(c)
        return dense, coh

    def _substrates_on_host(self, sensor0, sps, freqs, cols, nl, emmodel_names, layer_kind, host, scalars=None, liquid_water=None):
        """("host", R, Rcoh) of PackedBatch for a group whose snowpacks lie on substrates without a device implementation
        (rough ones: geometrical optics, IEM, ...).  Needs the streams of every last layer, hence the effective
        permittivity of every layer: from the host-evaluated emmodels if the group has them, otherwise from a cheap
        pre-pass of the device emmodels (four streams, layer diagnostics only)."""
        from .._native import PackedBatch, gauss_legendre_positive

        act = sensor0.mode == "A"
        F, S, Lmax = len(freqs), len(sps), cols.shape[2]
        eps = self._layer_permittivities(sps, freqs, cols, nl, emmodel_names, layer_kind, host, scalars, liquid_water)
        nm, ne = (self.m_max + 1 if act else 1), 3 * self.n_max_stream
        R = np.zeros((F, S, nm, ne, ne))
        Rc = np.zeros((F, S, nm, ne))
        gmu, _ = gauss_legendre_positive(self.n_max_stream)
        gsin = np.sqrt(1.0 - gmu * gmu)
        for fi, f in enumerate(freqs):
            for s, sp in enumerate(sps):
                e = eps[fi, s, :nl[s]]
                # (under process_coherent_layers: the streams of the last layer in the REDUCED snowpack -- the most
                # refringent layer is looked for among the layers that stay)
                stay = self._staying_layers(f, e, cols[0][s]) or list(range(len(e)))
                star = max(stay, key=lambda l: (e[l].real, e[l].imag, -l))
                rs = np.sqrt(e[star] / e[-1]).real * gsin
                mu = np.sqrt(1.0 - rs[rs < 1.0] ** 2)
                w = np.empty_like(mu)                       # streams.py:324-330
                w[0], w[-1] = 1.0 - 0.5 * (mu[0] + mu[1]), 0.5 * (mu[-2] + mu[-1])
                w[1:-1] = 0.5 * (mu[:-2] - mu[2:])
                dense, coh = self.substrate_matrices(sp.substrate, float(f), complex(e[-1]), mu, np.abs(w),
                                                     self.m_max if act else 0, 3 if act else 2)
                for m in range(nm):
                    k = dense[m].shape[0]
                    R[fi, s, m, :k, :k] = dense[m]
                    Rc[fi, s, m, :k] = coh[m]
                if not act:   # the emissivity diagonal takes the place of the specular one (rtsolver_utils.py:533-536)
                    if not callable(getattr(sp.substrate, "emissivity_matrix", None)):
                        raise SMRTError("a substrate evaluated on the host needs an emissivity_matrix method in passive mode")
                    em = sp.substrate.emissivity_matrix(float(f), complex(e[-1]), mu, 2)
                    em = np.asarray(getattr(em, "values", em), float)
                    Rc[fi, s, 0, :2 * len(mu)] = 0.0 if em.ndim == 0 else em.reshape(2, len(mu)).T.reshape(-1)
        if act:
            return ("host", R, Rc)
        return ("host", R, Rc, [sp.substrate.temperature if getattr(sp.substrate, "temperature", None) is not None else 0.0
                                for sp in sps])

    @staticmethod
    def _ms_code(layer):
        from .._native import MS_CODES

        code = MS_CODES.get(layer.microstructure_model)
        if code is None:
            raise SMRTError(f"the microstructure model '{layer.microstructure_model}' has no device implementation: it "
                            "can only be used with an emmodel evaluated on the host (e.g. rayleigh, prescribed_kskaeps)")
        return code

    # ---- emmodels evaluated on the host (include/smrt_dort.h: SMRT_EM_HOST) ----------------------------------------
    HOST_PHASE_BYTES_MAX = 8e9

    @staticmethod
    def _emmodel_instance(entry, sensor, layer):
        """The emmodel object of a layer from its entry: a device emmodel name, an (emmodel class, options) pair or a ready
        instance (rtsolver protocol)."""
        from ..core.foreign import is_native
        from ..core.plugin import import_class

        if isinstance(entry, str):
            # a device emmodel inside a group that is evaluated on the host: its descriptor class speaks the protocol
            # too ("iba_inverted" is IBA under dense_snow_correction="auto", the only option the device names carry)
            if entry == "iba_inverted":
                return import_class("emmodel", "iba")(sensor, layer, dense_snow_correction="auto")
            return import_class("emmodel", entry)(sensor, layer)
        if isinstance(entry, tuple):
            # a class of the reference package is given the reference's own layer object (core/foreign.py)
            target = layer if is_native(entry[0]) else getattr(layer, "source", layer)
            return entry[0](sensor, target, **entry[1])
        return entry

    @staticmethod
    def _isotropic(value, what):
        a = np.asarray(getattr(value, "values", value), float).ravel()   # (an smrt_matrix keeps its array in .values)
        if a.size == 0 or not np.allclose(a, a[0], rtol=1e-12, atol=0.0):
            raise SMRTError(f"smrt_amd's DORT needs an isotropic {what} (one number per layer)")
        return float(a[0])

    @staticmethod
    def _iba_phase_coefficient(em):
        """The coefficient of the phase matrix when `em` is an emmodel of IBA's family -- its phase matrix is IBA's own
        `phase` / `ft_even_phase` (smrt/emmodel/iba.py:228-244: coefficient x Fourier transform of the autocorrelation
        function x Rayleigh geometry), whatever it does to the scalars: IBA on any permittivity model / inclusion shape /
        background, IBA_original, IBA_MaxwellGarnett, derived_IBA(...); or a class that declares `iba_phase_family = True`
        and carries `iba_coeff`, `frac_volume`, `microstructure` next to the scalar accessors of the protocol -- else None."""
        cls = type(em)
        coeff = getattr(em, "iba_coeff", None)
        if coeff is None or not hasattr(em, "microstructure") or not hasattr(em, "frac_volume"):
            return None
        if not getattr(cls, "iba_phase_family", False):   # (a class of the caller's own may declare the family itself)
            base = next((c for c in cls.__mro__ if c.__name__ == "IBA" and (c.__module__ or "").endswith("emmodel.iba")), None)
            if base is None:
                return None
            for method in ("phase", "ft_even_phase"):
                if getattr(cls, method, None) is not getattr(base, method, None):
                    return None
        coeff = complex(coeff)
        return coeff.real if coeff.imag == 0.0 and coeff.real >= 0.0 else None

    @staticmethod
    def _sce_phase_coefficient(em):
        """The norm of the phase function when `em` is one of smrt's strong-contrast-expansion emmodels (sce_common.py:
        SCEBase -- symsce_torquato21, sce_torquato21, ... -- with `phase` / `ft_even_phase` unchanged): IBA's phase function
        normalised to ks and evaluated at the COMPLEX wavenumber 2 k0 sqrt(eps_eff) sin(Theta / 2) (:222), which the device
        does for the exponential model (SMRT_MS_EXPONENTIAL_COMPLEX_K); else None."""
        cls = type(em)
        base = next((c for c in cls.__mro__ if c.__name__ == "SCEBase" and (c.__module__ or "").endswith("emmodel.sce_common")), None)
        if base is None or not hasattr(em, "microstructure") or not hasattr(em, "frac_volume"):
            return None
        for method in ("phase", "ft_even_phase"):
            if getattr(cls, method, None) is not getattr(base, method, None):
                return None
        norm = getattr(em, "_phase_norm", None)
        if norm is None:
            norm = em.compute_phase_norm()
        norm = complex(norm)
        return norm.real if norm.imag == 0.0 and norm.real >= 0.0 else None

    @staticmethod
    def _has_rayleigh_phase(em):
        """Is the phase matrix of `em` the Rayleigh one, 3 ks / 2 x the geometry of smrt/emmodel/rayleigh.py:52-127?  True for a
        class that inherits `ft_even_phase` from smrt's `Rayleigh` unchanged (rayleigh, sft_rayleigh, prescribed_kskaeps, the
        dmrt short-range emmodels), for smrt_amd's own host emmodels of that shape (emmodel/rayleigh.py), and for a class
        that declares `rayleigh_phase_family = True`."""
        cls = type(em)
        if getattr(cls, "rayleigh_phase_family", False):
            return True
        for c in cls.__mro__:
            module = c.__module__ or ""
            if c.__name__ == "Rayleigh" and module.endswith("emmodel.rayleigh") and not module.startswith("smrt_amd."):
                return getattr(cls, "ft_even_phase", None) is getattr(c, "ft_even_phase", None)
            if c.__name__ == "_RayleighPhase" and module == "smrt_amd.emmodel.rayleigh":
                return getattr(cls, "ft_even_phase", None) is getattr(c, "ft_even_phase", None)
        return False

    def _iba_scalars_on_host(self, sensor0, sps, freqs, entries, nl, Lmax, sensor_of):
        """The cheap half of the host route for emmodels whose phase matrix the device can assemble itself -- IBA's family
        (include/smrt_dort.h: SMRT_EM_IBA_HOST) and the Rayleigh family (SMRT_EM_RAYLEIGH_HOST): effective permittivity, ks,
        ka (and, for IBA's family, the coefficient of the phase matrix) of every (frequency, snowpack, layer) from the
        emmodel objects; the phase matrices themselves are assembled on the device.  Returns
        (host_layer [F, S, Lmax, 4], coefficient [F, S, Lmax], per-layer kinds, frac_volume, micro_p1, micro_p2 [S, Lmax])
        -- the medium the emmodel works on (inverted above half ice under dense_snow_correction="auto") --, or None when
        a layer's emmodel is of another family or its microstructure model has no device code: the full host route
        (_evaluate_on_host) then applies."""
        import copy

        from .._native import EM_CODES, MS_CODES
        from ..core.foreign import _device_microstructure, is_native

        P = 2 if sensor0.mode == "P" else 3
        F, S = len(freqs), len(sps)
        hl = np.zeros((F, S, Lmax, 4)); hl[..., 2] = 1.0
        coeff = np.zeros((F, S, Lmax))
        kinds = np.zeros((S, Lmax), np.int32)
        fv = np.full((S, Lmax), 0.3); p1 = np.full((S, Lmax), 1e-4); p2 = np.full((S, Lmax), 0.2)   # harmless padding
        one = np.array([1.0, 0.55, 0.1])   # (three directions: an anisotropic ks / ka shows and is refused)
        for fi, f in enumerate(freqs):
            sensor = sensor_of.get(float(f))
            if sensor is None:
                sensor = copy.copy(sensor0)
                sensor.frequency = float(f)
            for s, sp in enumerate(sps):
                for l, layer in enumerate(sp.layers):
                    if isinstance(entries[s][l], str):     # a device emmodel: nothing to evaluate here
                        if fi == 0:
                            name = entries[s][l]
                            kinds[s, l] = EM_CODES[name] + 16 * self._ms_code(layer)
                            q = layer.microstructure.device_params
                            fv[s, l] = 1.0 - layer.frac_volume if name == "iba_inverted" else layer.frac_volume
                            p1[s, l], p2[s, l] = q
                        continue
                    em = self._emmodel_instance(entries[s][l], sensor, layer)
                    if self._has_rayleigh_phase(em):
                        # the Rayleigh phase matrix needs ks only (SMRT_EM_RAYLEIGH_HOST); no microstructure is read
                        ks = em.ks(one, P) if callable(getattr(em, "ks", None)) else em.ks
                        ka = em.ka(one, P) if callable(getattr(em, "ka", None)) else em.ka
                        eps = complex(em.effective_permittivity())
                        hl[fi, s, l] = self._isotropic(ks, "ks"), self._isotropic(ka, "ka"), eps.real, eps.imag
                        if fi == 0:
                            kinds[s, l] = EM_CODES["rayleigh_host"]
                            fv[s, l] = min(max(float(layer.frac_volume), 0.0), 1.0)
                        continue
                    c = self._iba_phase_coefficient(em)
                    # (a class of the caller's own may declare iba_phase_family = "complex_k": the strong-contrast form)
                    complex_k = c is not None and getattr(type(em), "iba_phase_family", None) == "complex_k"
                    if complex_k and sensor0.mode != "P":
                        return None
                    if c is None and sensor0.mode == "P":   # (the complex phase function in active mode: dense route)
                        c = self._sce_phase_coefficient(em)
                        complex_k = c is not None
                    if c is None:
                        return None
                    ms = em.microstructure
                    if is_native(ms) or hasattr(ms, "device_params"):
                        name, (q1, q2) = getattr(ms, "name", None), ms.device_params
                    else:
                        name, q1, q2 = _device_microstructure(ms)
                    if name not in MS_CODES:
                        return None
                    if complex_k:   # the model's transform at the complex wavenumber (4 + its own code): the two rational forms
                        name = {0: "exponential_complex_k", 3: "teubner_strey_complex_k"}.get(MS_CODES[name])
                        if name is None:   # the sphere models: the dense route
                            return None
                    ks = em.ks(one, P) if callable(getattr(em, "ks", None)) else em.ks
                    ka = em.ka(one, P) if callable(getattr(em, "ka", None)) else em.ka
                    eps = complex(em.effective_permittivity())
                    hl[fi, s, l] = self._isotropic(ks, "ks"), self._isotropic(ka, "ka"), eps.real, eps.imag
                    coeff[fi, s, l] = c
                    if fi == 0:
                        kinds[s, l] = EM_CODES["iba_host"] + 16 * MS_CODES[name]
                        fv[s, l], p1[s, l], p2[s, l] = float(em.frac_volume), q1, q2
        return hl, coeff, kinds, fv, p1, p2

    def _evaluate_on_host(self, sensor0, sps, freqs, entries, nl, Lmax, sensor_of):
        """What smrt/rtsolver/dort.py:189,231-247,714-762 asks of the emmodels -- effective permittivity, ks, ka and the
        azimuth modes of the phase matrix on the layer's own streams -- for every (frequency, snowpack, layer) of the
        group, as the three arrays of PackedBatch(host_emmodel=...).  An entry is a device emmodel name, an (emmodel
        class, options) pair or a ready instance (rtsolver protocol).  The streams are the ones the device will find
        (Gauss-Legendre nodes in the most refringent layer + Snell, streams.py:136-223); it checks the counts."""
        import copy

        from .._native import gauss_legendre_positive
        from ..core.foreign import is_native
        from ..core.plugin import import_class

        mode = sensor0.mode
        P = 2 if mode == "P" else 3
        modes = 1 if mode == "P" else self.m_max + 1
        m_arg = modes - 1
        F, S, NE = len(freqs), len(sps), self.n_max_stream * P
        nbytes = 8.0 * F * S * Lmax * modes * 2 * NE * NE
        if nbytes > self.HOST_PHASE_BYTES_MAX:
            raise SMRTError(f"the phase matrices of this batch would take {nbytes / 1e9:.1f} GB on the host: run emmodels "
                            "without a device implementation in smaller groups of snowpacks")
        hl = np.zeros((F, S, Lmax, 4))
        hl[..., 2] = 1.0
        hs = np.zeros((F, S, Lmax), np.int32)
        hp = np.zeros((F, S, Lmax, modes, 2, NE, NE))
        gmu, _ = gauss_legendre_positive(self.n_max_stream)
        gsin = np.sqrt(1.0 - gmu * gmu)

        instance, scalar = self._emmodel_instance, self._isotropic

        for fi, f in enumerate(freqs):
            sensor = sensor_of.get(float(f))
            if sensor is None:
                sensor = copy.copy(sensor0)
                sensor.frequency = float(f)
            for s, sp in enumerate(sps):
                ems = [instance(entries[s][l], sensor, layer) for l, layer in enumerate(sp.layers)]
                eps = np.array([complex(em.effective_permittivity()) for em in ems])
                stay = list(range(len(eps)))
                if self.process_coherent_layers:
                    # the layers the device will take out at this frequency (interface/coherent_flat.py:16-57, k0 Re(n) d
                    # < 3 pi / 4): the streams of the others are those of the REDUCED snowpack.  (Where the reference
                    # refuses -- the last layer or two in a row -- the device answers status 6 whatever is put here.)
                    k0 = 2.0 * np.pi * float(f) / C_SPEED
                    stay = [l for l, lay in enumerate(sp.layers) if not k0 * np.sqrt(eps[l]).real * lay.thickness < 0.75 * np.pi]
                    for l in set(range(len(eps))) - set(stay):
                        hl[fi, s, l, 2:] = eps[l].real, eps[l].imag
                    if not stay:
                        continue
                star = max(stay, key=lambda l: (eps[l].real, eps[l].imag, -l))   # np.argmax on complex
                for l in stay:
                    em = ems[l]
                    rs = np.sqrt(eps[star] / eps[l]).real * gsin
                    mu = np.sqrt(1.0 - rs[rs < 1.0] ** 2)
                    n = len(mu)
                    ks = em.ks(mu, P) if callable(getattr(em, "ks", None)) else em.ks
                    ka = em.ka(mu, P) if callable(getattr(em, "ka", None)) else em.ka
                    hl[fi, s, l] = scalar(ks, "ks"), scalar(ka, "ka"), eps[l].real, eps[l].imag
                    hs[fi, s, l] = n
                    if hl[fi, s, l, 0] == 0.0 or n == 0:
                        continue
                    full = np.concatenate((mu, -mu))
                    ft = em.ft_even_phase(full, full, m_arg, npol=P)
                    ft = np.asarray(getattr(ft, "values", ft), float)
                    if ft.shape != (P, P, modes, 2 * n, 2 * n):
                        raise SMRTError(f"ft_even_phase returned the shape {ft.shape}, expected {(P, P, modes, 2 * n, 2 * n)}")
                    # (ps, pi, m, mu_s, mu_i) -> m, (mu_s, ps), (mu_i, pi): the compressed order of core/lib.py:336-347
                    C = np.transpose(ft, (2, 3, 0, 4, 1)).reshape(modes, 2 * n * P, 2 * n * P)
                    nP = n * P
                    if self.phase_symmetrization:   # rtsolver_utils.py:743-765 (sign -1 between U and V | H for m >= 1)
                        sgn = np.where(np.arange(nP) % P < 2, 1.0, -1.0)
                        d = np.ones((modes, 1, 1)) * (sgn[:, None] * sgn[None, :])
                        d[0] = 1.0 if P == 2 else d[0]
                        C = C.copy()
                        C[:, :nP, :nP] = 0.5 * (C[:, :nP, :nP] + d * C[:, nP:, nP:])
                        C[:, :nP, nP:] = 0.5 * (C[:, :nP, nP:] + d * C[:, nP:, :nP])
                    vh = np.arange(nP)[np.arange(nP) % P < 2]
                    for blk in (C[0][:nP, :nP], C[0][:nP, nP:]):   # the device reads the lower triangles only
                        sub = blk[np.ix_(vh, vh)]
                        if not np.allclose(sub, sub.T, rtol=1e-9, atol=1e-12 * np.abs(sub).max()):
                            raise SMRTError("the phase matrix of this emmodel does not obey reciprocity: smrt_amd's DORT "
                                            "(symmetric eigenproblem) cannot solve it")
                    hp[fi, s, l, :, 0, :nP, :nP] = C[:, :nP, :nP]
                    hp[fi, s, l, :, 1, :nP, :nP] = C[:, :nP, nP:]
        return hl, hs, hp


class _Solution:
    """Outputs of the device batches of one call, addressable per simulation and stackable as one Result."""

    def __init__(self, solver, sensors, packs, sens_idx, pack_idx):
        self.solver, self.sensors, self.packs = solver, sensors, packs
        self.sens_idx, self.pack_idx = np.asarray(sens_idx), np.asarray(pack_idx)
        n = len(self.sens_idx)
        self.group_of = np.full(n, -1, np.int64)
        self.row_of = np.zeros(n, np.int64)
        self.outputs = []
        self.columns = []

    def add_group(self, sel, out, sp0, columns=None):
        self.group_of[sel] = len(self.outputs)
        self.row_of[sel] = np.arange(len(sel))
        self.outputs.append(out)
        # (snowpack indices of the group, their layer counts and thickness rows AS SOLVED: what the stacked diagnostics are
        # built from when somebody asks later -- never the live objects, which may have changed since)
        self.columns.append(columns)

    # -- labels ----------------------------------------------------------------------------------------------------
    @staticmethod
    def _coords(sensor):
        if sensor.mode == "P":
            return [("polarization", ["V", "H"]), ("theta", sensor.theta_deg)]
        pola = ["V", "H", "U"]
        return [("polarization_inc", pola), ("polarization", pola), ("theta_inc", sensor.theta_inc_deg)]

    @staticmethod
    def _reported_streams(sensor, streams_row):
        """Cosines of the air streams of Result.other_data['stream_angles']: all of them in passive mode, only the
        incident ones in active mode (rtsolver_utils.py:307-316, dort.py:210-226)."""
        n_air = int(streams_row[0])
        outmu = streams_row[1:1 + n_air]
        if sensor.mode == "A":
            keep = set()
            for mu_inc in np.cos(np.atleast_1d(sensor.theta_inc)):
                i0 = int(np.searchsorted(-outmu, -mu_inc))
                keep.update((0,) if i0 == 0 else ((n_air - 1,) if i0 == n_air else (i0, i0 - 1)))
            outmu = outmu[sorted(keep)]
        return outmu

    def result(self, i):
        """The Result of simulation i with the labels and diagnostics of DiscreteOrdinatesMixin.make_result
        (rtsolver_utils.py:322-344,373-398)."""
        sensor, sp = self.sensors[self.sens_idx[i]], self.packs[self.pack_idx[i]]
        out, row = self.outputs[self.group_of[i]], self.row_of[i]
        L = sp.nlayer
        thickness = sp.layer_thicknesses
        layers = out.layers[row]
        if self.solver.process_coherent_layers:   # the layers that were solved: column 4 = streams + 1024 x input index
            code = layers[:L, 4]
            kept = int(np.count_nonzero(code > 0))
            thickness = thickness[(code[:kept] // 1024).astype(int)]
            layers = layers.copy()
            layers[:, 4] = layers[:, 4] % 1024
            L = kept
        outmu = self._reported_streams(sensor, out.streams[row])
        layer_idx = ("layer", np.arange(L))
        lay = layers[:L]
        # smrt_amd's Result, or the caller's own when the sensor belongs to the reference package (core/foreign.py)
        make, labelled = result_factory(sensor)
        other = {
            "stream_angles": labelled(np.rad2deg(np.arccos(outmu)), [("dim_0", np.arange(len(outmu)))]),
            "effective_permittivity": labelled(lay[:, 0] + 1j * lay[:, 1], [layer_idx]),
            "ks": labelled(lay[:, 2].copy(), [layer_idx], name="ks"),
            "ke": labelled(lay[:, 2] + lay[:, 3], [layer_idx], name="ke"),
            "ka": labelled(lay[:, 3].copy(), [layer_idx], name="ka"),
            "thickness": labelled(np.asarray(thickness, float), [layer_idx], name="thickness"),
        }
        return make(sensor, out.values[row], self._coords(sensor), other_data=other)

    def stacked_result(self, plan):
        """All simulations as ONE Result whose leading dimensions are the plan's -- built from the output arrays by
        reshaping (no per-simulation objects).  None when the simulations are not one homogeneous grid (several device
        groups, different channel maps): the caller then nests per-simulation results."""
        if len(self.outputs) != 1 or not plan.dimensions:
            return None
        sensor0 = self.sensors[0]
        if any(s.channel_map != sensor0.channel_map or s.mode != sensor0.mode for s in self.sensors[1:]):
            return None
        out = self.outputs[0]
        order = self.row_of                       # simulation i -> row of the group output
        lead = [(name, np.asarray(list(values))) for name, values in plan.dimensions]
        shape = tuple(len(v) for _, v in lead)
        if int(np.prod(shape)) != len(order):
            return None
        data = LabeledArray(out.values[order].reshape(shape + out.values.shape[1:]), lead + self._coords(sensor0))
        # the layer / stream diagnostics of Result.other_data are built when somebody asks for them: on a 1024 x 5 batch
        # they are a third of the host time of Model.run, and most callers only read the brightness temperatures
        return make_result(sensor0, data, other_data=_LazyOther(lambda: self._stacked_other(out, order, lead, shape, sensor0)))

    def _stacked_other(self, out, order, lead, shape, sensor0):
        u_packs, nl_solved, thick_solved = self.columns[0]
        slot = np.full(len(self.packs), -1, np.int64)
        slot[u_packs] = np.arange(len(u_packs))            # snowpack -> its row in the group's batch
        nl = nl_solved[slot[self.pack_idx]]
        Lmax = int(nl.max())
        lay = out.layers[order][:, :Lmax].copy()
        lay[np.arange(Lmax)[None, :] >= nl[:, None]] = np.nan      # ragged packs: NaN below the last layer
        layer_dim = [("layer", np.arange(Lmax))]

        def stack(values, name=None):
            return LabeledArray(values.reshape(shape + (Lmax,)), lead + layer_dim, name=name)

        thick_rows = thick_solved.reshape(len(u_packs), -1)[slot[self.pack_idx], :Lmax].copy()
        thick_rows[np.arange(Lmax)[None, :] >= nl[:, None]] = np.nan
        if self.solver.process_coherent_layers:   # per simulation: the layers that were solved, top first, NaN after them
            code = np.nan_to_num(lay[:, :, 4])    # streams + 1024 x index in the input (include/smrt_dort.h)
            kept = code > 0
            thick_rows = np.where(kept, np.take_along_axis(thick_rows, (code // 1024).astype(np.int64), axis=1), np.nan)
            lay[~kept] = np.nan
        streams = [self._reported_streams(self.sensors[s], out.streams[r]) for s, r in zip(self.sens_idx, order)] \
            if sensor0.mode == "A" else None
        if streams is None:
            n_air = out.streams[order, 0].astype(np.int64)
            width = int(n_air.max())
            mu = out.streams[order, 1:1 + width].copy()
            mu[np.arange(width)[None, :] >= n_air[:, None]] = np.nan
        else:
            width = max(len(s) for s in streams)
            mu = np.full((len(order), width), np.nan)
            for k, s in enumerate(streams):
                mu[k, :len(s)] = s
        other = {
            "stream_angles": LabeledArray(np.rad2deg(np.arccos(mu)).reshape(shape + (width,)),
                                          lead + [("dim_0", np.arange(width))]),
            "effective_permittivity": stack(lay[:, :, 0] + 1j * lay[:, :, 1]),
            "ks": stack(lay[:, :, 2], "ks"),
            "ke": stack(lay[:, :, 2] + lay[:, :, 3], "ke"),
            "ka": stack(lay[:, :, 3], "ka"),
            "thickness": stack(thick_rows, "thickness"),
        }
        return other


class _LazyOther(Mapping):
    """Result.other_data of a stacked result, built on first access (a read-only mapping with the six keys of the reference:
    smrt/rtsolver/rtsolver_utils.py:338-342,373-398)."""
    KEYS = ("stream_angles", "effective_permittivity", "ks", "ke", "ka", "thickness")

    def __init__(self, build):
        self._build, self._data = build, None

    def _get(self):
        if self._data is None:
            self._data, self._build = self._build(), None
        return self._data

    def __getitem__(self, key):
        return self._get()[key]

    def __iter__(self):
        return iter(self.KEYS)

    def __len__(self):
        return len(self.KEYS)

    def __reduce__(self):   # (pickling / deep copies: as the plain dictionary it stands for)
        return (dict, (dict(self._get()),))


# ---- contexts and the multi-GPU fan-out ----------------------------------------------------------------------------
_ctx_cache = {}
_ctx_lock = threading.Lock()


def default_device():
    """The GPU the helpers outside a solver run use (emmodel accessors, ft_even_phase): SMRT_DORT_DEVICE, else this
    process's LOCAL_RANK in a one-process-per-GPU launch, else 0 -- never blindly GPU 0."""
    import os

    for key in ("SMRT_DORT_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(key)
        if v is not None and v.strip().isdigit():
            from .._native import device_count

            n = device_count()
            return int(v) % n if n > 0 else int(v)
    return 0


def get_context(device=None):
    """The cached context of a GPU (created on first use; its calls are serialised by DortContext.lock); None: the
    default device of this process (default_device)."""
    if device is None:
        device = default_device()
    with _ctx_lock:
        if device not in _ctx_cache:
            _ctx_cache[device] = DortContext(device)
        return _ctx_cache[device]


def visible_devices():
    return list(range(device_count()))


def shard_by_cost(cost, n_shards):
    """Contiguous slices [b[k], b[k+1]) of a cost vector with (nearly) equal cost each: the boundaries are where the
    running cost crosses k / n_shards of the total (shard by sum_l N_l^3, not by count)."""
    cost = np.asarray(cost, dtype=np.float64)
    running = np.concatenate([[0.0], np.cumsum(cost)])
    targets = running[-1] * np.arange(1, n_shards) / n_shards
    inner = np.searchsorted(running, targets, side="left")
    return np.concatenate([[0], np.clip(inner, 0, len(cost)), [len(cost)]]).astype(np.int64)


def run_on_devices(batch, devices=None, block_threads=0, pairs=None, cost=None):
    """Run a packed batch -- all of it, or the listed pair indices -- sharded over the given GPUs: contiguous slices of
    the work list (equal cost when `cost` per work item is given, equal counts otherwise), one host thread and one
    context per GPU, no collective: every GPU writes disjoint rows of the same host arrays."""
    n = batch.n_pairs if pairs is None else len(pairs)
    if devices is None:
        devices = visible_devices() if n >= 4096 else [0]
    devices = list(devices) or [0]

    def one(dev, lo, hi):
        ctx = get_context(dev)
        with ctx.lock:
            ctx.set_block_threads(block_threads)
            if pairs is None:
                return ctx.run(batch, int(lo), int(hi - lo))
            return ctx.run(batch, pairs=pairs[lo:hi])

    if len(devices) == 1:
        return one(devices[0], 0, n)
    if cost is None:   # the work of a pair varies with its stream counts: estimate it on the first device (cheap kernel)
        ctx = get_context(devices[0])
        with ctx.lock:
            if pairs is None:
                ctx.upload(batch)
            else:
                ctx.upload(batch, pairs=pairs)
            cost = ctx.pair_cost()
    bounds = shard_by_cost(cost, len(devices))
    out = BatchOutput(batch, n)
    errors = []

    def work(dev, lo, hi):
        try:
            part = one(dev, lo, hi)
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
