"""Substrates under the last layer (counterparts of smrt/substrate/flat.py and reflector.py)."""
