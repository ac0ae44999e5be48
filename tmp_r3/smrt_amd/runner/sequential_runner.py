"""Sequential runner (smrt/runner/sequential_runner.py:17-48): one `function(*args)` call per simulation, i.e. one
small device launch per (sensor, snowpack).  Mostly useful for debugging and for comparing with the batch runner."""


class SequentialRunner(object):
    def __init__(self, progressbar=False, max_numerical_threads=1):
        self.progressbar = progressbar

    def __call__(self, function, argument_list):
        return [function(*args) for args in argument_list]
