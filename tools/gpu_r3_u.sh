#!/bin/bash
# Round 3, call U: LDS-DMA pieces interleaved with the MFMA groups in the fp32 persistent kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "stream_k" 2>&1 | tail -2
  echo "== microbench --sk"
  timeout 600 python tools/microbench.py --sk 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_u.log
