#!/bin/bash
# Round 3, call V: fp32 contraction on the bf16 matrix pipe (three-way operand split): parity on the GPU, sweep, bench A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "f32x3 or stream_k" 2>&1 | tail -12
  echo "== microbench --x3"
  timeout 400 python tools/microbench.py --x3 2>&1 | grep -v amdgpu.ids
  echo "== bench, split-operand contractions on"
  SBK_F32X3=1 timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-roofline 2>&1 | tail -1 | cut -c1-1500
  echo "== bench, fp32-MFMA contractions"
  SBK_F32X3=0 timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-roofline 2>&1 | tail -1 | cut -c1-1500
} 2>&1 | tee gpurun_out/r3_v.log
