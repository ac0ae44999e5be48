"""make_model / Model with the reference's surface (smrt/core/model.py:120-624): same constructor, `run`,
`prepare_simulations` (frequency-major flattening), `run_single_simulation`, runner protocol and result nesting.
The default runner is the batching GPU runner."""
import inspect
import itertools
from collections.abc import Mapping, Sequence

import numpy as np
import pandas as pd

from ..runner.hip_batch_runner import HipBatchRunner
from ..runner.sequential_runner import SequentialRunner
from .error import SMRTError
from .plugin import import_class
from .result import concat_results
from .sensor import SensorBase


def is_sequence(x):
    return isinstance(x, (Sequence, np.ndarray)) and not isinstance(x, str)


def _specialize(scope, cls, **options):
    if isinstance(cls, str):
        cls = import_class(scope, cls)
    if not options:
        return cls
    return type(cls.__name__, (cls,), {"__init__": (lambda self, *a, **kw: cls.__init__(self, *a, **{**options, **kw}))})


def make_rtsolver(rtsolver_class, **options):
    """make_model(..., make_rtsolver("dort", n_max_stream=128))  (model.py:180-194)."""
    return _specialize("rtsolver", rtsolver_class, **options)


def make_emmodel(emmodel_class, **options):
    return _specialize("emmodel", emmodel_class, **options)


def make_emmodel_instance(emmodel, sensor, layer, **emmodel_options):
    emmodel = make_emmodel(emmodel)
    if not isinstance(sensor, SensorBase):
        raise SMRTError("the first argument of 'run' must be a sensor")
    return emmodel(sensor, layer, **emmodel_options)


def make_model(emmodel=None, rtsolver=None, emmodel_options=None, rtsolver_options=None, emmodel_kwargs=None,
               rtsolver_kwargs=None):
    """Create a new model with a given EM model and RT solver (model.py:120-177)."""
    if emmodel_kwargs is not None or rtsolver_kwargs is not None:
        raise DeprecationWarning("Use emmodel_options / rtsolver_options")
    return Model(emmodel, rtsolver, emmodel_options=emmodel_options, rtsolver_options=rtsolver_options)


class Model(object):
    """Drive the whole calculation."""

    def __init__(self, emmodel, rtsolver, emmodel_options=None, rtsolver_options=None):
        if is_sequence(emmodel):
            self.emmodel = [make_emmodel(em) for em in emmodel]
        elif isinstance(emmodel, Mapping):
            self.emmodel = {k: make_emmodel(em) for k, em in emmodel.items()}
        else:
            self.emmodel = make_emmodel(emmodel)
        self.rtsolver = import_class("rtsolver", rtsolver) if isinstance(rtsolver, str) else rtsolver
        self.emmodel_options = emmodel_options if emmodel_options is not None else dict()
        self.rtsolver_options = rtsolver_options if rtsolver_options is not None else dict()

    def set_rtsolver_options(self, options=None, **kwargs):
        if options is not None:
            self.rtsolver_options = dict(options)
        self.rtsolver_options.update(kwargs)

    def set_emmodel_options(self, options=None, **kwargs):
        if options is not None:
            self.emmodel_options = dict(options)
        self.emmodel_options.update(kwargs)

    def run(self, sensor, snowpack, atmosphere=None, snowpack_dimension=None, snowpack_column="snowpack",
            progressbar=False, parallel_computation="outer", runner=None):
        """Run the model for the given sensor configuration(s) and snowpack(s) (model.py:310-413)."""
        if atmosphere is not None:
            raise DeprecationWarning("The atmosphere argument of the run method is depreciated.")
        if not (isinstance(sensor, SensorBase)
                or (is_sequence(sensor) and all(isinstance(s, SensorBase) for s in sensor))):
            raise SMRTError("the first argument of 'run' must be a sensor or a sequence of sensor")
        simulations, dimensions = self.prepare_simulations(sensor, snowpack, snowpack_dimension, snowpack_column)
        if runner is None:
            if parallel_computation in ("outer", "auto", True, "inner"):
                runner = HipBatchRunner(progressbar=progressbar)
            elif parallel_computation in ("none", None, False):
                runner = SequentialRunner(progressbar=progressbar)
            else:
                raise SMRTError(f"parallel_computation={parallel_computation} is not valid. "
                                "Must be 'outer', 'inner', 'none' or None")
        results = list(runner(self.run_single_simulation,
                              ((simul, atmosphere, parallel_computation) for simul in simulations)))
        for dimension in reversed(dimensions):
            n = len(dimension[1])
            assert n > 0, f"dimension={dimensions}"
            results = [concat_results(results[i:i + n], dimension) for i in range(0, len(results), n)]
        assert len(results) == 1, f"Results size is {len(results)=}"
        results = results[0]
        if isinstance(snowpack, pd.DataFrame):
            results.mother_df = snowpack.drop(snowpack_column, axis=1)
        return results

    def prepare_simulations(self, sensor, snowpack, snowpack_dimension, snowpack_column):
        """Flat list of (sensor, snowpack) pairs, frequency-major, plus the (axis, values) list used to nest the
        results (model.py:415-527)."""
        if isinstance(snowpack, Mapping):
            snowpack_dimension = "snowpack", list(snowpack.keys())
            snowpack = list(snowpack.values())
        if isinstance(snowpack, pd.DataFrame):
            try:
                snowpack = snowpack[snowpack_column]
            except KeyError:
                raise SMRTError(f"the snowpack DataFrame has no column named '{snowpack_column}'.")
        if isinstance(snowpack, pd.Series):
            name = snowpack.index.name or "snowpack"
            snowpack_dimension = name, snowpack.index.tolist()
            snowpack = snowpack.tolist()
        if is_sequence(snowpack):
            if snowpack_dimension is None:
                snowpack_dimension = "snowpack", None
            if snowpack_dimension[1] is None:
                snowpack_dimension = snowpack_dimension[0], range(len(snowpack))
            if len(snowpack) != len(snowpack_dimension[1]):
                raise SMRTError("The list of snowpacks must have the same length as the snowpack_dimension")
        if isinstance(snowpack_dimension, tuple) and not isinstance(snowpack_dimension[0], str):
            raise SMRTError("When the 'snowpack_dimension' argument is a tuple, the first argument must be a string")

        def get_sensor_configurations(sensor):
            capability = getattr(self.rtsolver, "_broadcast_capability", [])
            return [(axis, values) for (axis, values) in sensor.configurations() if axis not in capability]

        def prepare_recursive(sensor, sensor_configurations, snowpack):
            if sensor_configurations:
                axis, _ = sensor_configurations[0]
                for sensor_subset in sensor.iterate(axis):
                    yield from prepare_recursive(sensor_subset, sensor_configurations[1:], snowpack)
            elif is_sequence(snowpack):
                for sp in snowpack:
                    yield (sensor, sp)
            else:
                yield (sensor, snowpack)

        if is_sequence(sensor):
            if len(sensor) != len(snowpack):
                raise SMRTError("when sensor is a sequence, the length must be the same as snowpack sequence length")
            sensor_configurations = get_sensor_configurations(next(iter(sensor)))
            simulations = list(itertools.chain(*(prepare_recursive(se, sensor_configurations, sp)
                                                 for se, sp in zip(sensor, snowpack))))
        else:
            sensor_configurations = get_sensor_configurations(sensor)
            simulations = prepare_recursive(sensor, list(sensor_configurations), snowpack)
        dimensions = sensor_configurations
        if snowpack_dimension is not None:
            dimensions.append(snowpack_dimension)
        return simulations, dimensions

    def prepare_emmodels(self, sensor, snowpack):
        """One emmodel instance per layer (model.py:529-582)."""
        if is_sequence(self.emmodel):
            assert len(self.emmodel) == snowpack.nlayer
            emmodel_list = self.emmodel
        elif isinstance(self.emmodel, Mapping):
            emmodel_list = (self.emmodel[layer.medium] for layer in snowpack.layers)
        else:
            emmodel_list = (layer.emmodel or self.emmodel for layer in snowpack.layers)
        return [make_emmodel_instance(em, sensor, layer, **(layer.emmodel_options or self.emmodel_options))
                for em, layer in zip(emmodel_list, snowpack.layers)]

    def run_single_simulation(self, simulation, atmosphere, parallel_computation):
        """One (sensor, snowpack) through a fresh rtsolver instance (model.py:584-619)."""
        sensor, snowpack = simulation
        emmodel_instances = self.prepare_emmodels(sensor, snowpack)
        if self.rtsolver is None:
            return None
        if inspect.isclass(self.rtsolver):
            rtsolver = self.rtsolver(**self.rtsolver_options)
        else:
            if not getattr(self.rtsolver, "_reentrant", False):
                raise SMRTError("This solver can not be used with an instance")
            rtsolver = self.rtsolver
        return rtsolver.solve(snowpack, emmodel_instances, sensor, snowpack.atmosphere or atmosphere,
                              parallel_computation=parallel_computation)
