"""Batching GPU runner: the MI355X replacement of JoblibParallelRunner (smrt/runner/joblib_runner.py:15-72).

Two ways in:

* `run_plan(model, plan)` -- what `smrt_amd.Model.run` calls: the whole `SimulationPlan` goes to the rtsolver's
  `solve_plan`, which packs the distinct snowpacks once, launches once per GPU and returns the nested Result;
* `runner(function, argument_list)` -- the reference's runner protocol (smrt/core/model.py:395-398): `function` is the
  bound `Model.run_single_simulation` (so `function.__self__` is the Model), every argument is
  `((sensor_f, snowpack), atmosphere, parallel_computation)`.  Instead of mapping `function` over the list the whole
  list becomes one device batch per GPU and one Result per item comes back, in order.

Either way the emmodel configuration of the model is honoured the way the per-simulation route honours it: per-layer
emmodels and emmodel options go through `Model.emmodel_of_layer` / `emmodel_options_of_layer` and the emmodel class
validates the options; what the device path cannot do raises SMRTError instead of being ignored."""
import inspect

from ..core.error import SMRTError


class HipBatchRunner(object):
    def __init__(self, progressbar=False, devices=None, block_threads=0):
        self.progressbar = progressbar  # accepted for signature compatibility; one launch has no progress to show
        self.devices = devices
        self.block_threads = block_threads

    def _rtsolver(self, model):
        rtsolver = getattr(model, "rtsolver", None)
        if rtsolver is None:
            raise SMRTError("the model has no rtsolver")
        if inspect.isclass(rtsolver):
            options = dict(model.rtsolver_options)
            options.setdefault("devices", self.devices)
            options.setdefault("block_threads", self.block_threads)
            try:
                rtsolver = rtsolver(**options)
            except TypeError:  # an rtsolver that does not know smrt_amd's own knobs
                rtsolver = model.rtsolver(**model.rtsolver_options)
        return rtsolver

    def run_plan(self, model, plan):
        rtsolver = self._rtsolver(model)
        if not hasattr(rtsolver, "solve_plan"):  # a foreign rtsolver: fall back on the generic protocol
            from ..core.model import nest_results

            return nest_results([model.run_single_simulation(pair, None, "outer") for pair in plan.pairs()],
                                plan.dimensions)
        return rtsolver.solve_plan(model, plan)

    def __call__(self, function, argument_list):
        args = list(argument_list)
        if not args:
            return []
        model = getattr(function, "__self__", None)
        if model is None or not hasattr(model, "rtsolver"):
            raise SMRTError("HipBatchRunner must be given the bound Model.run_single_simulation method")
        for _, atmosphere, _ in args:
            if atmosphere is not None:  # Model.run's deprecated argument (model.py:345-349): use snowpack.atmosphere
                raise SMRTError("give the atmosphere to the snowpack (make_snowpack(..., atmosphere=...) or "
                                "atmosphere + snowpack), not to Model.run")
        rtsolver = self._rtsolver(model)
        if not hasattr(rtsolver, "solve_batch"):
            raise SMRTError("HipBatchRunner needs an rtsolver with a solve_batch method (smrt_amd.rtsolver.dort.DORT)")
        from ..core.model import SimulationPlan  # the same emmodel checks as run_plan
        import numpy as np

        simulations = [simul for simul, _, _ in args]
        sensors = list({id(s): s for s, _ in simulations}.values())
        packs = list({id(p): p for _, p in simulations}.values())
        probe = SimulationPlan(sensors, packs, np.zeros(0, int), np.zeros(0, int))
        emmodel = rtsolver.emmodel_names(model, probe) if hasattr(rtsolver, "emmodel_names") else model.emmodel
        if isinstance(emmodel, str):   # uniform: solve_batch takes the class (any layer's will do)
            emmodel = model.emmodel_of_layer(0, packs[0].layers[0], packs[0].nlayer)
        return rtsolver.solve_batch(simulations, emmodel)
