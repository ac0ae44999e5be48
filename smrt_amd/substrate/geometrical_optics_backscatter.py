"""Geometrical optics, backscatter lobe only (usable in passive mode): smrt/substrate/geometrical_optics_backscatter.py.  The physics lives in smrt_amd/interface/geometrical_optics_backscatter.py; this is that model under
the last layer, against the substrate's own permittivity (substrate/rough.py)."""
from ..interface.geometrical_optics_backscatter import GeometricalOpticsBackscatter as _Interface
from .rough import InterfaceSubstrate


class GeometricalOpticsBackscatter(InterfaceSubstrate):
    interface_class = _Interface
