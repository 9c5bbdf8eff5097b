"""Shim of the reference's `videoswap` package: the sub-modules `test.py` imports, backed by videoswap_amd."""
