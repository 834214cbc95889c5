#!/bin/bash
# Round 6, visit AJ: gemm_nt_lp256_kernel with the touch stream (four bytes of every cache line of the K tile two ahead, key 65):
# bit-identity tests, per-shape A/B, the Whisper encoder at 32 layers either way.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py -q -m gpu -x -k "lp256 or fp8a or bf16_activation or whisper" 2>&1 | tail -3
  echo "== per shape"
  timeout 300 python tools/microbench.py --lp256 2>&1 | grep -v amdgpu.ids | head -7
  for k in 0 1; do
    echo "== whisper, 32 layers, key 65 = $k"
    timeout 300 python tools/whisper_probe.py --layers 32 --prec bf16,fp8 --knob 65=$k 2>&1 | grep -v amdgpu.ids | grep "ms per forward\|gemm_nt_bf16a\|gemm_nt_fp8a"
  done
} 2>&1 | tee gpurun_out/r6_aj.log
