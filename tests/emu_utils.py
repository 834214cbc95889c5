"""Attach the CPU kernel emulator (tools/kernel_emu) to speechbrain_amd.native -- tests only.

The emulator runs the SAME kernel sources (speechbrain_amd/csrc/*.hip) on host
fibers, so kernel logic can be checked without a GPU.  The product never loads it.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def attach():
    from tools.kernel_emu.build_emu import build
    from speechbrain_amd import native

    native._attach_for_tests(build())
    return native


def detach():
    from speechbrain_amd import native

    native._detach_for_tests()
