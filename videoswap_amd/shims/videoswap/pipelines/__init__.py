from videoswap_amd import build_pipeline  # noqa: F401
