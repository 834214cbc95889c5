#!/bin/bash
# Round 3, call S: LDS-shared bf16 attention (tests, Whisper probe), Conformer encoder under bf16 with / without bf16 activations.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py -x -q -m gpu -k "bf16" 2>&1 | tail -4
  echo "== whisper probe"
  timeout 500 python tools/whisper_probe.py --prec fp32,bf16 2>&1 | grep -v amdgpu.ids
  echo "== conformer encoder, bf16"
  timeout 500 python tools/microbench.py --enc-bf16 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_s.log
