#!/bin/bash
# Round 3, call O: bf16-activation pipeline (kernel tests, microbench, Whisper probe).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py -x -q -m gpu -k "bf16_activation" 2>&1 | tail -5
  echo "== microbench --bf16a"
  timeout 600 python tools/microbench.py --bf16a 2>&1 | grep -v amdgpu.ids
  echo "== whisper probe"
  timeout 500 python tools/whisper_probe.py --prec fp32,bf16 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_o.log
