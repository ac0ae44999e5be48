"""IEM (Fung et al. 1992) rough substrate: smrt/substrate/iem_fung92.py.  The physics lives in smrt_amd/interface/iem_fung92.py; this is that model under
the last layer, against the substrate's own permittivity (substrate/rough.py)."""
from ..interface.iem_fung92 import IEM_Fung92 as _Interface
from .rough import InterfaceSubstrate


class IEM_Fung92(InterfaceSubstrate):
    interface_class = _Interface
