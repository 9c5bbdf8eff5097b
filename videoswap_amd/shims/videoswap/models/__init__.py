from videoswap_amd import build_model  # noqa: F401
