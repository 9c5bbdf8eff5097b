from videoswap_amd.data import DATASET_REGISTRY, build_dataset  # noqa: F401
