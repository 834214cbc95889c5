#!/bin/bash
# Round 6, visit AG: the reduced-precision epilogues' GELU as erfc by Abramowitz-Stegun 7.1.26 (sbk::gelu_erfc) instead of libm's erff:
# the tests through bf16 / fp8 activations (kernels, Whisper at 2 and 32 layers), per-shape times, the Whisper encoder at 32 layers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 1200 python -m pytest tests/test_kernels.py tests/test_whisper.py tests/test_full_size_gpu.py -q -m gpu -x -k "lp256 or fp8 or bf16 or whisper" 2>&1 | tail -3
  echo "== per shape"
  timeout 300 python tools/microbench.py --lp256 2>&1 | grep -v amdgpu.ids | head -5
  echo "== whisper, 32 layers"
  timeout 300 python tools/whisper_probe.py --layers 32 --prec fp32,bf16,fp8 2>&1 | grep -v amdgpu.ids | head -24
} 2>&1 | tee gpurun_out/r6_ag.log
