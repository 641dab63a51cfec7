import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vcvits_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _quiesce_gpu_between_tests(request):
    """GPU tests build and drop modules, optimizers and captured HIP graphs by the hundred in one process: let every test's
    device work finish and its objects (graphs and their memory pools included) be collected at a point where nothing is
    in flight, instead of whenever the garbage collector gets to them under a later test's launches."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
