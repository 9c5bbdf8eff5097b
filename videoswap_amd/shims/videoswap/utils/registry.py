from videoswap_amd.compat import MODEL_REGISTRY, PIPELINE_REGISTRY, Registry  # noqa: F401
from videoswap_amd.data import DATASET_REGISTRY, TRANSFORM_REGISTRY  # noqa: F401
