#!/bin/bash
# Round 6, visit Y: csrc/gemm_lp256.hip on the GPU for the first time -- 256 x 256 tiles for the bf16 / e4m3 activation x weight
# contractions (key 61): bit-identity with the 128 x 128 kernels, per-shape times, the Whisper encoder at 32 layers either way.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py -q -m gpu -x -k "lp256 or fp8a or bf16_activation or whisper" 2>&1 | tail -3
  echo "== per shape"
  timeout 300 python tools/microbench.py --lp256 2>&1 | grep -v amdgpu.ids
  for k in 0 1; do
    echo "== whisper, 32 layers, key 61 = $k"
    timeout 300 python tools/whisper_probe.py --layers 32 --prec bf16,fp8 --knob 61=$k 2>&1 | grep -v amdgpu.ids | head -16
  done
} 2>&1 | tee gpurun_out/r6_y.log
