"""Atmospheres (counterpart of smrt/atmosphere/simple_isotropic_atmosphere.py)."""
