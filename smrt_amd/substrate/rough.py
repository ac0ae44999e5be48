"""Substrates made of a rough-interface model (the role of smrt/core/interface.py:169-240, substrate_from_interface): the
medium below is the substrate's own permittivity, the methods are the substrate protocol the DORT solver evaluates on
the streams of the last layer (rtsolver/dort.py:substrate_matrices, _substrates_on_host) -- specular_reflection_matrix,
emissivity_matrix (= the interface's coherent transmission), ft_even_diffuse_reflection_matrix."""
from ..core.error import SMRTError
from ..core.substrate import SubstrateBase


class InterfaceSubstrate(SubstrateBase):
    interface_class = None          # set by the subclasses

    def __init__(self, temperature=None, permittivity_model=None, **interface_parameters):
        super().__init__(temperature=temperature, permittivity_model=permittivity_model)
        self.interface = self.interface_class(**interface_parameters)
        for key in list(self.interface_class.args) + list(self.interface_class.optional_args):
            setattr(self, key, getattr(self.interface, key))

    def _below(self, frequency):
        eps = self.permittivity(frequency)
        if eps is None:
            raise SMRTError(f"No permittivity_model have been given to the substrate '{type(self).__name__}'")
        return eps

    def specular_reflection_matrix(self, frequency, eps_1, mu1, npol):
        return self.interface.specular_reflection_matrix(frequency, eps_1, self._below(frequency), mu1, npol)

    def emissivity_matrix(self, frequency, eps_1, mu1, npol):
        return self.interface.coherent_transmission_matrix(frequency, eps_1, self._below(frequency), mu1, npol)

    def ft_even_diffuse_reflection_matrix(self, frequency, eps_1, mu_s, mu_i, m_max, npol):
        return self.interface.ft_even_diffuse_reflection_matrix(frequency, eps_1, self._below(frequency), mu_s, mu_i, m_max, npol)
