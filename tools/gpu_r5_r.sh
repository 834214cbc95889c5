#!/bin/bash
# Round 5, visit R: fabric-traffic counters of the kernels as shipped -- gemm_x3r with the XCDs' 2-D ownership of its tile space
# (and gemm_nt_x3p beside it: the two traffic passes of tools/run_pmc_r5.sh), then the decode step's memory-bound kernels with the
# register-ring cross-attention (tools/run_pmc_r5_decode.sh).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
  PMC_PASSES=2 bash tools/run_pmc_r5.sh 2>&1 | tail -24
  bash tools/run_pmc_r5_decode.sh 2>&1 | tail -6
} 2>&1 | tee gpurun_out/r5_r.log
