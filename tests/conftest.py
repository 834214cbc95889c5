import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order: the driver runs `pytest -m gpu -x`, so a failure must LOCALISE, not hide -- the
# kernel-level tests come first, then the model-level goldens, streaming / Whisper, and the composite full-size file last.
_ORDER = ["test_abi.py", "test_oracle_known_answers.py", "test_oracle_golden.py", "test_host_logic.py", "test_wer.py",
          "test_resample.py", "test_kernels.py", "test_model_parity.py", "test_streaming.py", "test_whisper.py",
          "test_distributed.py", "test_full_size_gpu.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 1

    items.sort(key=rank)  # (stable: the order inside a file is kept)


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
