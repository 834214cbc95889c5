#!/bin/bash
# Round 3, call Y: DVFS probe of the split-operand kernel (random vs zero-filled operands)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 400 python tools/microbench.py --x3 --x3-short --x3-zero 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_y.log
