#!/bin/bash
# Round 6, visit AE: where gemm_nt_x3p_kernel's time goes at the encoder's shapes -- measurement builds (key 64).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 600 python tools/microbench.py --x3p-modes 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r6_ae.log
