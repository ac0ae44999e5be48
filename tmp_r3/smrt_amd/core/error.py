"""Error and warning types of the plugin surface (counterparts of smrt/core/error.py:6-29)."""
import warnings


class SMRTError(Exception):
    """Error raised by the model (same name and role as smrt.core.error.SMRTError)."""


class SMRTWarning(Warning):
    pass


def smrt_warn(message, category=SMRTWarning):
    warnings.warn(message, category, stacklevel=2)
