#!/bin/bash
# Round 4, visit S (last GPU seconds of the round; the change under test is NOT in the tree at HEAD -- staging/ holds it as a
# patch): sbk_gemm_ln_nt_x3r on the GPU -- its kernel test, the decoder on its route against the oracle, and its time
# against the two launches it replaces
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
  timeout 100 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -x -m gpu -k "gemm_ln_x3r or x3r_route" 2>&1 | tail -5
  timeout 60 python tools/microbench.py --x3r-ln 2>&1 | grep "x3r-ln"
} 2>&1 | tee gpurun_out/r4_s.log
