import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

sys.dont_write_bytecode = True


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # pragma: no cover
        return False


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line(
        'markers', 'device: model-level test of the host mirror.  On a GPU box it IS a gpu test (it gets the gpu marker '
        'and runs on the HIP kernels); without a GPU it runs on CPU tensors with videoswap_amd.ops swapped for the '
        'test-only restatement of the ops contract in tests/host_emulation.py, so the host logic stays covered by '
        '-m "not gpu"')


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    # runs before the -m expression is applied: on a GPU box `device` tests are selected by -m gpu
    has_gpu = _has_gpu()
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if has_gpu and item.get_closest_marker('device') is not None:
            item.add_marker(pytest.mark.gpu)
        if not has_gpu and item.get_closest_marker('gpu') is not None:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _device_backend(request):
    if request.node.get_closest_marker('device') is None or _has_gpu():
        yield
        return
    import host_emulation
    with host_emulation.installed():
        yield
