"""make_model / Model: the entry point of the reference's plugin surface (smrt/core/model.py:120-624) for the DORT path.

Kept from the reference: the signatures and meaning of `make_model`, `make_rtsolver`, `make_emmodel`,
`make_emmodel_instance`, `Model.run`, `Model.prepare_simulations`, `Model.prepare_emmodels`,
`Model.run_single_simulation`, the runner protocol `runner(function, argument_list)` and the order in which the
(sensor configuration, snowpack) pairs are flattened (every sensor axis the rtsolver does not broadcast, slowest first,
then the snowpacks -- frequency-major).

Own structure (GPU-first): the flattening is not a recursive generator but a `SimulationPlan` -- the list of
single-configuration sensors, the list of snowpacks and two index vectors built with NumPy.  A batching runner takes
the plan as a whole (`runner.run_plan`): it packs the distinct snowpacks once, launches once per device and returns the
stacked result directly; any other runner is fed the reference's `(function, argument_list)` protocol and the per-pair
results are nested afterwards by `nest_results`."""
import inspect
import threading
from collections.abc import Mapping, Sequence
from dataclasses import dataclass, field

import numpy as np
import pandas as pd

from .error import SMRTError, smrt_warn
from .plugin import import_class
from .result import concat_results
from .sensor import SensorBase, SensorList


def is_sequence(x):
    return isinstance(x, (Sequence, np.ndarray)) and not isinstance(x, str)


# ---- class factories ---------------------------------------------------------------------------------------------
def _with_options(scope, cls, options):
    """The class `cls` (or the plugin named `cls` in `scope`) with constructor keyword defaults bound."""
    if isinstance(cls, str):
        cls = import_class(scope, cls)
    if not options:
        return cls

    def __init__(self, *args, **kwargs):
        cls.__init__(self, *args, **{**options, **kwargs})

    return type(cls.__name__, (cls,), {"__init__": __init__, "__module__": cls.__module__})


def make_rtsolver(rtsolver_class, **options):
    """`make_model("iba", make_rtsolver("dort", n_max_stream=64))`: an rtsolver class with its options bound."""
    return _with_options("rtsolver", rtsolver_class, options)


def make_emmodel(emmodel_class, **options):
    """`make_model(make_emmodel("iba", ...), "dort")`: an emmodel class with its options bound."""
    return _with_options("emmodel", emmodel_class, options)


def make_emmodel_instance(emmodel, sensor, layer, **emmodel_options):
    """One emmodel object for one layer seen by one (single-frequency) sensor."""
    if not isinstance(sensor, SensorBase):
        raise SMRTError("the first argument of 'run' must be a sensor")
    return make_emmodel(emmodel)(sensor, layer, **emmodel_options)


def make_model(emmodel=None, rtsolver=None, emmodel_options=None, rtsolver_options=None, emmodel_kwargs=None,
               rtsolver_kwargs=None):
    """Create a model from an electromagnetic model and a radiative-transfer solver, each given by name (resolved by
    the plugin loader, smrt_amd.core.plugin) or as a class.  `emmodel` may also be a list (one per layer) or a dict
    (one per layer medium)."""
    if emmodel_kwargs is not None:
        raise DeprecationWarning("Use emmodel_options instead of emmodel_kwargs")
    if rtsolver_kwargs is not None:
        raise DeprecationWarning("Use rtsolver_options instead of rtsolver_kwargs")
    return Model(emmodel, rtsolver, emmodel_options=emmodel_options, rtsolver_options=rtsolver_options)


# ---- which emmodel, with which options, for which layer ------------------------------------------------------------
# Free functions of (model.emmodel, model.emmodel_options): they serve smrt_amd's Model and -- through
# rtsolver/dort.py:DORT.emmodel_names -- a Model of the reference package handed to HipBatchRunner by its own `run`.
def select_emmodel(emmodel, index, layer, n_layers, make=None):
    """The emmodel class for one layer: from the per-layer list, the per-medium dict, the layer's own attribute or the
    model-wide class, in the reference's order of precedence (smrt/core/model.py:529-554).  `make`: the make_emmodel
    that resolves a layer's own emmodel given by name (default: smrt_amd's)."""
    make = make or make_emmodel
    own = getattr(layer, "emmodel", None)
    if is_sequence(emmodel):
        if len(emmodel) != n_layers:
            raise SMRTError("the list of emmodels must have the same length as the number of layers")
        chosen = emmodel[index]
    elif isinstance(emmodel, Mapping):
        if layer.medium not in emmodel:
            raise SMRTError(f"no emmodel is given for the medium '{layer.medium}'")
        chosen = emmodel[layer.medium]
    else:
        if own is not None:
            return make(own)
        chosen = emmodel
    if own is not None:
        smrt_warn("a layer defines its own emmodel but the model was given a list / dict of emmodels: the layer's "
                  "emmodel is ignored")
    if chosen is None:
        raise SMRTError("no emmodel: give one to make_model or to every layer")
    return chosen


def select_emmodel_options(emmodel, options, layer, index=None, n_layers=None):
    """The options of one layer's emmodel, in the reference's order (smrt/core/model.py:556-569): the entry of a
    per-layer sequence, the per-medium dict of dicts that goes with a dict of emmodels, else the layer's own options or
    the model-wide dict."""
    if is_sequence(options):
        if index is None or n_layers is None or len(options) != n_layers:
            raise SMRTError("the list of emmodel_options must have the same length as the number of layers")
        return options[index]
    if isinstance(emmodel, Mapping) and options and all(isinstance(o, Mapping) for o in options.values()):
        if layer.medium not in options:
            raise SMRTError(f"no emmodel_options are given for the medium '{layer.medium}'")
        return dict(options[layer.medium])
    own = getattr(layer, "emmodel_options", None)
    return own if own else options


# ---- the flattened grid ------------------------------------------------------------------------------------------
@dataclass
class SimulationPlan:
    """sensors[sensor_index[i]] x snowpacks[snowpack_index[i]] for simulation i; `dimensions` = [(name, values), ...]
    from the outermost to the innermost nesting level (their sizes multiply to the number of simulations)."""

    sensors: list
    snowpacks: list
    sensor_index: np.ndarray
    snowpack_index: np.ndarray
    dimensions: list = field(default_factory=list)
    scalar_snowpack: bool = False   # a single snowpack was given: no snowpack dimension in the result

    def __len__(self):
        return len(self.sensor_index)

    def pairs(self):
        return [(self.sensors[i], self.snowpacks[j]) for i, j in zip(self.sensor_index, self.snowpack_index)]

    @property
    def shape(self):
        return tuple(len(values) for _, values in self.dimensions)


def nest_results(results, dimensions):
    """Fold a flat list of per-simulation results into one result with the given leading dimensions."""
    results = list(results)
    for name, values in reversed(dimensions):
        n = len(values)
        if n == 0 or len(results) % n:
            raise SMRTError(f"{len(results)} results cannot be folded along '{name}' of size {n}")
        results = [concat_results(results[k:k + n], (name, values)) for k in range(0, len(results), n)]
    if len(results) != 1:
        raise SMRTError(f"the dimensions {[d[0] for d in dimensions]} do not account for all the results")
    return results[0]


def _snowpack_axis(snowpack, snowpack_dimension, snowpack_column):
    """(list of snowpacks, (name, labels) or None, mother DataFrame or None) from the forms `run` accepts."""
    mother = None
    if isinstance(snowpack, Mapping):
        return list(snowpack.values()), ("snowpack", list(snowpack.keys())), None
    if isinstance(snowpack, pd.DataFrame):
        if snowpack_column not in snowpack.columns:
            raise SMRTError(f"the snowpack DataFrame has no column named '{snowpack_column}'.")
        mother = snowpack.drop(columns=snowpack_column)
        snowpack = snowpack[snowpack_column]
    if isinstance(snowpack, pd.Series):
        return snowpack.tolist(), (snowpack.index.name or "snowpack", snowpack.index.tolist()), mother
    if not is_sequence(snowpack):
        if snowpack_dimension is not None:
            raise SMRTError("snowpack_dimension needs a sequence of snowpacks")
        return [snowpack], None, None
    packs = list(snowpack)
    if snowpack_dimension is None:
        name, labels = "snowpack", None
    elif isinstance(snowpack_dimension, str):
        name, labels = snowpack_dimension, None
    else:
        name, labels = snowpack_dimension
        if not isinstance(name, str):
            raise SMRTError("When the 'snowpack_dimension' argument is a tuple, the first argument must be a string")
    labels = range(len(packs)) if labels is None else labels
    if len(labels) != len(packs):
        raise SMRTError("The list of snowpacks must have the same length as the snowpack_dimension")
    return packs, (name, labels), None


class Model(object):
    """An electromagnetic model + a radiative-transfer solver, ready to `run`."""

    def __init__(self, emmodel, rtsolver, emmodel_options=None, rtsolver_options=None):
        if is_sequence(emmodel):
            self.emmodel = [make_emmodel(em) for em in emmodel]
        elif isinstance(emmodel, Mapping):
            self.emmodel = {medium: make_emmodel(em) for medium, em in emmodel.items()}
        else:
            self.emmodel = None if emmodel is None else make_emmodel(emmodel)
        self.rtsolver = import_class("rtsolver", rtsolver) if isinstance(rtsolver, str) else rtsolver
        # [lock, (row blocks, stacked columns) of the last large run]: the batching rtsolver skips the concatenation of
        # thousands of row blocks when a run meets the very objects of the previous one (rtsolver/dort.py:_pack)
        self._kept_columns = [threading.Lock(), None]
        self.emmodel_options = self._checked_options(emmodel_options)
        self.rtsolver_options = dict(rtsolver_options or {})

    def set_rtsolver_options(self, options=None, **kwargs):
        self.rtsolver_options = self._merged(self.rtsolver_options, options, kwargs)

    def set_emmodel_options(self, options=None, **kwargs):
        if is_sequence(options):
            if kwargs:
                raise SMRTError("keyword options cannot be combined with a per-layer list of emmodel options")
            self.emmodel_options = self._checked_options(options)
            return
        current = self.emmodel_options if isinstance(self.emmodel_options, Mapping) else {}
        self.emmodel_options = self._merged(current, options, kwargs)

    @staticmethod
    def _checked_options(options):
        """The three forms of smrt/core/model.py:556-569: one dict for every layer, a sequence of dicts (one per layer),
        or -- with a dict of emmodels -- a dict of dicts keyed by the medium (recognised when it is used)."""
        if options is None:
            return {}
        if is_sequence(options):
            if not all(isinstance(o, Mapping) for o in options):
                raise SMRTError("a sequence of emmodel_options must hold one Mapping (eg. dict) per layer")
            return [dict(o) for o in options]
        if not isinstance(options, Mapping):
            raise SMRTError("emmodel_options must be a Mapping (eg. dict) or a sequence of them, one per layer")
        return dict(options)

    @staticmethod
    def _merged(current, options, kwargs):
        if options is not None and not isinstance(options, Mapping):
            raise SMRTError("options must be a Mapping (eg. dict)")
        return {**(current if options is None else options), **kwargs}

    # ---- planning ------------------------------------------------------------------------------------------------
    def split_axes(self, sensor):
        """The sensor axes one rtsolver call cannot broadcast, hence flattened by the model: [(axis, values), ...]."""
        broadcast = getattr(self.rtsolver, "_broadcast_capability", ())
        return [(axis, values) for axis, values in sensor.configurations() if axis not in broadcast]

    def plan(self, sensor, snowpack, snowpack_dimension=None, snowpack_column="snowpack"):
        packs, pack_dim, _ = _snowpack_axis(snowpack, snowpack_dimension, snowpack_column)
        n_packs = len(packs)
        if is_sequence(sensor):  # zip mode: sensor k looks at snowpack k
            sensors = list(sensor)
            if len(sensors) != n_packs or pack_dim is None:
                raise SMRTError("when sensor is a sequence, the length must be the same as snowpack sequence length")
            if any(self.split_axes(s) for s in sensors):
                raise SMRTError("a sequence of sensors is run pairwise with the snowpacks: every sensor must hold a "
                                "single configuration (one frequency)")
            idx = np.arange(n_packs)
            return SimulationPlan(sensors, packs, idx, idx.copy(), [pack_dim])
        axes = self.split_axes(sensor)
        if isinstance(sensor, SensorList):
            sensors = list(sensor.iterate())
            if any(self.split_axes(s) for s in sensors):
                raise SMRTError("the members of a SensorList must hold a single configuration each")
        else:
            sensors = list(sensor.split([axis for axis, _ in axes]))
        n_sens = len(sensors)
        dims = list(axes) + ([pack_dim] if pack_dim is not None else [])
        return SimulationPlan(sensors, packs, np.repeat(np.arange(n_sens), n_packs), np.tile(np.arange(n_packs), n_sens),
                              dims, scalar_snowpack=pack_dim is None)

    def prepare_simulations(self, sensor, snowpack, snowpack_dimension, snowpack_column):
        """(simulations, dimensions) in the reference's form: the flat list of (sensor, snowpack) pairs and the
        (axis, values) pairs used to nest their results."""
        plan = self.plan(sensor, snowpack, snowpack_dimension, snowpack_column)
        return plan.pairs(), plan.dimensions

    # ---- running -------------------------------------------------------------------------------------------------
    def default_runner(self, parallel_computation, progressbar=False):
        from ..runner.hip_batch_runner import HipBatchRunner
        from ..runner.sequential_runner import SequentialRunner

        if parallel_computation in ("outer", "auto", "inner", True):
            return HipBatchRunner(progressbar=progressbar)
        if parallel_computation in ("none", None, False):
            return SequentialRunner(progressbar=progressbar)
        raise SMRTError(f"parallel_computation={parallel_computation} is not valid. Must be 'outer', 'inner', 'none' "
                        "or None")

    def run(self, sensor, snowpack, atmosphere=None, snowpack_dimension=None, snowpack_column="snowpack",
            progressbar=False, parallel_computation="outer", runner=None):
        """Run the model for the sensor configuration(s) and the snowpack(s): one snowpack, a sequence, a dict, a
        pandas Series or a DataFrame holding them in `snowpack_column`.  The result gains one dimension per flattened
        sensor axis and one for the snowpacks."""
        if atmosphere is not None:
            raise DeprecationWarning("The atmosphere argument of the run method is depreciated. Use instead the "
                                     "atmosphere argument of make_snowpack (or `atmosphere + snowpack`).")
        single = isinstance(sensor, SensorBase)
        if not single and not (is_sequence(sensor) and all(isinstance(s, SensorBase) for s in sensor)):
            raise SMRTError("the first argument of 'run' must be a sensor or a sequence of sensor")
        plan = self.plan(sensor, snowpack, snowpack_dimension, snowpack_column)
        if runner is None:
            runner = self.default_runner(parallel_computation, progressbar)
        if hasattr(runner, "run_plan"):
            result = runner.run_plan(self, plan)
        else:
            results = runner(self.run_single_simulation,
                             ((pair, atmosphere, parallel_computation) for pair in plan.pairs()))
            result = nest_results(results, plan.dimensions)
        if isinstance(snowpack, pd.DataFrame):
            result.mother_df = snowpack.drop(columns=snowpack_column)
        return result

    # ---- one simulation (the unit a generic runner maps over) ----------------------------------------------------
    def emmodel_of_layer(self, index, layer, n_layers):
        """The emmodel class for one layer (`select_emmodel` on this model's configuration)."""
        return select_emmodel(self.emmodel, index, layer, n_layers)

    def emmodel_options_of_layer(self, layer, index=None, n_layers=None):
        """The options of one layer's emmodel (`select_emmodel_options` on this model's configuration)."""
        return select_emmodel_options(self.emmodel, self.emmodel_options, layer, index, n_layers)

    def prepare_emmodels(self, sensor, snowpack):
        """One emmodel instance per layer."""
        n = snowpack.nlayer
        return [make_emmodel_instance(self.emmodel_of_layer(k, layer, n), sensor, layer,
                                      **self.emmodel_options_of_layer(layer, k, n))
                for k, layer in enumerate(snowpack.layers)]

    def make_rtsolver_instance(self):
        if self.rtsolver is None:
            return None
        if inspect.isclass(self.rtsolver):
            return self.rtsolver(**self.rtsolver_options)
        if not getattr(self.rtsolver, "_reentrant", False):
            raise SMRTError("This solver can not be used with an instance")
        return self.rtsolver

    def run_single_simulation(self, simulation, atmosphere, parallel_computation):
        """One (sensor, snowpack) pair through a fresh rtsolver instance."""
        sensor, snowpack = simulation
        emmodels = self.prepare_emmodels(sensor, snowpack)
        rtsolver = self.make_rtsolver_instance()
        if rtsolver is None:
            return None
        return rtsolver.solve(snowpack, emmodels, sensor, snowpack.atmosphere or atmosphere,
                              parallel_computation=parallel_computation)
