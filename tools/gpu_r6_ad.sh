#!/bin/bash
# Round 6, visit AD: the same after the compiler-visible wait behind the epilogue (no s_waitcnt vmcnt(0) behind the LDS-DMA of every K tile any more):
# bit-identity tests, per-shape times, measurement builds, the Whisper encoder at 32 layers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py -q -m gpu -x -k "lp256 or fp8a or bf16_activation or whisper" 2>&1 | tail -3
  echo "== per shape"
  timeout 300 python tools/microbench.py --lp256 2>&1 | grep -v amdgpu.ids
  echo "== modes"
  timeout 300 python tools/microbench.py --lp256-modes 2>&1 | grep -v amdgpu.ids | grep -v "mode 16\|mode 32\|mode 48"
  echo "== whisper, 32 layers"
  timeout 300 python tools/whisper_probe.py --layers 32 --prec bf16,fp8 2>&1 | grep -v amdgpu.ids | head -16
} 2>&1 | tee gpurun_out/r6_ad.log
