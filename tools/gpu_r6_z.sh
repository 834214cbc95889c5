#!/bin/bash
# Round 6, visit Z: where gemm_nt_lp256_kernel's time goes -- measurement builds (no LDS-DMA / no MFMAs / no epilogue / no fragment
# fetches) and two schedule variants (MFMA phase at raised priority; LDS reads awaited behind the phase barrier), key 62.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 600 python tools/microbench.py --lp256-modes 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r6_z.log
