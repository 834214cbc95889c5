#!/bin/bash
# Round 3, call Q: 256-register budget (no AGPR copies around the K loop) on the GEMM kernels: bf16 microbench, decode probe, Whisper probe.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
  echo "== microbench --bf16a"
  timeout 600 python tools/microbench.py --bf16a 2>&1 | grep -v amdgpu.ids
  echo "== decode probe"
  timeout 300 python tools/decode_probe.py --steps 24 --reps 2 --report 2>&1 | grep -v amdgpu.ids | head -16
  echo "== whisper probe"
  timeout 500 python tools/whisper_probe.py --prec fp32,bf16 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_q.log
