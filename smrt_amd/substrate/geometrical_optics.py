"""Geometrical-optics rough substrate: smrt/substrate/geometrical_optics.py.  The physics lives in smrt_amd/interface/geometrical_optics.py; this is that model under
the last layer, against the substrate's own permittivity (substrate/rough.py)."""
from ..interface.geometrical_optics import GeometricalOptics as _Interface
from .rough import InterfaceSubstrate


class GeometricalOptics(InterfaceSubstrate):
    interface_class = _Interface
