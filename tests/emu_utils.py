"""Attach the CPU kernel emulator (tools/kernel_emu) to speechbrain_amd.native -- tests only.

The emulator runs the SAME kernel sources (speechbrain_amd/csrc/*.hip) on host
fibers, so kernel logic can be checked without a GPU.  The product never loads it.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


_strict_dev_ok = None


def _contiguous_only(*ts):
    """Stand-in for native._dev_ok while the emulator is attached: host tensors are what it takes."""
    from speechbrain_amd import native

    for t in ts:
        if t is not None and not t.is_contiguous():
            raise native.SbkError("non-contiguous tensor passed to a kernel")


def attach():
    """Point the binding at the emulator build.  The hook lives HERE, not in the product: the shipped
    native.py has no switch that accepts host tensors."""
    global _strict_dev_ok
    from tools.kernel_emu.build_emu import build
    from speechbrain_amd import native

    native._lib = None
    lib = native.load(build())
    # SBK_TEST_KNOBS="17=1,14=0": run the CPU suite with tuning variants switched on (validation of a prepared kernel
    # against every model-level golden before it becomes a default)
    for kv in filter(None, os.environ.get("SBK_TEST_KNOBS", "").split(",")):
        key, value = kv.split("=")
        lib.sbk_prof_set_knob(int(key), int(value))
    if _strict_dev_ok is None:
        _strict_dev_ok = native._dev_ok
    native._dev_ok = _contiguous_only
    return native


def detach():
    from speechbrain_amd import native

    native._lib = None
    if _strict_dev_ok is not None:
        native._dev_ok = _strict_dev_ok
