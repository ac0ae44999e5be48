"""Flat (Fresnel) interface marker -- the only interface type in scope (smrt/interface/flat.py:11-75).  The
reflection / transmission coefficients themselves are evaluated inside the HIP kernel."""


class Flat:
    args = []
    optional_args = {}

    def __eq__(self, other):
        return isinstance(other, Flat)

    def __hash__(self):
        return hash("Flat")
