import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """('emu', cpu): the kernel sources on the CPU emulator (tools/kernel_emu, logic check only);
    ('hip', cuda): the shipped libsbk_hip.so on the MI355X -- the parity tests proper."""
    import torch

    import emu_utils
    from speechbrain_amd import native

    if request.param == "emu":
        emu_utils.attach()
        yield native, torch.device("cpu")
        emu_utils.detach()
    else:
        assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
        emu_utils.detach()
        native.load()
        yield native, torch.device("cuda:0")
