"""`from diffusers.optimization import get_scheduler` (train.py:13)"""
from videoswap_amd.runner import get_scheduler as _get


def get_scheduler(name, optimizer, num_warmup_steps=None, num_training_steps=None, **unused):
    return _get(name, optimizer, num_warmup_steps or 0, num_training_steps)
