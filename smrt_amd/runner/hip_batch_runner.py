"""Batching GPU runner: the MI355X replacement of JoblibParallelRunner (smrt/runner/joblib_runner.py:15-72).

A runner is a callable `runner(function, argument_list) -> list[Result]` (same length and order as the input).
`function` is the bound `Model.run_single_simulation`, so `function.__self__` is the Model (emmodel, rtsolver,
options); every argument is `((sensor_f, snowpack), atmosphere, parallel_computation)` (smrt/core/model.py:395-398).
Instead of mapping `function` over the list this runner packs the WHOLE list into one device batch per GPU, launches
once and returns one Result per item; `Model.run` then nests them with concat_results as usual."""
import inspect

from ..core.error import SMRTError


class HipBatchRunner(object):
    def __init__(self, progressbar=False, devices=None, block_threads=0):
        self.progressbar = progressbar  # accepted for signature compatibility; one launch has no progress to show
        self.devices = devices
        self.block_threads = block_threads

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
        rtsolver = model.rtsolver
        if inspect.isclass(rtsolver):
            options = dict(model.rtsolver_options)
            options.setdefault("devices", self.devices)
            options.setdefault("block_threads", self.block_threads)
            rtsolver = rtsolver(**options)
        if not hasattr(rtsolver, "solve_batch"):
            raise SMRTError("HipBatchRunner needs an rtsolver with a solve_batch method (smrt_amd.rtsolver.dort.DORT)")
        emmodel = model.emmodel
        if isinstance(emmodel, (list, tuple, dict)):
            raise SMRTError("smrt_amd's DORT needs one emmodel for all the layers")
        return rtsolver.solve_batch([simul for simul, _, _ in args], emmodel)
