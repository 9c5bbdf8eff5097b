"""Host-side step-invariant cache (videoswap_amd/layers.py): hit / invalidation rules, no GPU needed."""
import torch

from videoswap_amd.layers import StepInvariantCache, param_key


def test_hit_requires_same_object_version_and_params():
    cache = StepInvariantCache(limit=4)
    w = torch.nn.Parameter(torch.ones(3))
    src = torch.zeros(2)
    calls = []

    def fn():
        calls.append(1)
        return len(calls)

    assert cache.get(src, None, (w,), fn) == 1
    assert cache.get(src, None, (w,), fn) == 1          # same tensor object, same versions: hit
    src.add_(1)                                          # in-place edit of the input bumps _version: miss
    assert cache.get(src, None, (w,), fn) == 2
    with torch.no_grad():
        w.copy_(torch.zeros(3))                          # load_state_dict-style update bumps the parameter version
    assert cache.get(src, None, (w,), fn) == 3
    assert cache.get(src.clone(), None, (w,), fn) == 4   # equal values in a different tensor object: miss
    assert cache.get(src, 5, (w,), fn) == 5              # different `extra` key (ED-LoRA layer index)
    w.data = torch.ones(3)                               # .half()/.to() replace the storage
    assert cache.get(src, None, (w,), fn) == 6


def test_entries_keep_the_source_alive_and_are_bounded():
    cache = StepInvariantCache(limit=3)
    w = torch.nn.Parameter(torch.ones(1))
    ids = set()
    for i in range(10):
        t = torch.full((1,), float(i))
        ids.add(id(t))
        assert cache.get(t, None, (w,), lambda i=i: i) == i
        assert len(cache.entries) <= 3
    # a recycled id() can never alias a live entry: every entry holds a reference to its source tensor
    for (sid, _), (src, _, _) in cache.entries.items():
        assert id(src) == sid


def test_param_key_ignores_none():
    w = torch.nn.Parameter(torch.ones(2))
    assert param_key(w, None) == param_key(w)
