#!/bin/bash
# Round 3, call W: split-operand kernel with transposed accumulators (16-byte epilogue stores); parity, measurement builds, sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "f32x3 or stream_k" 2>&1 | tail -4
  echo "== microbench --x3 --x3-short --x3-modes"
  timeout 400 python tools/microbench.py --x3 --x3-short --x3-modes 2>&1 | grep -v amdgpu.ids
  echo "== microbench --x3"
  timeout 400 python tools/microbench.py --x3 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_w5.log
