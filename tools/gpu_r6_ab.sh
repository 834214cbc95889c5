#!/bin/bash
# Round 6, visit AB: gemm_nt_lp256_kernel with the K tiles of consecutive output tiles as one stream (no store drain at a tile's start);
# gemm_nt_x3p's epilogue with the wave's whole residual block requested up front (key 63): tests, per-shape A/B, Whisper, headline A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6ab.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])
except Exception as e: print('no result', e)"; }
{
  echo "== tests"
  timeout 900 python -m pytest tests/test_kernels.py tests/test_whisper.py tests/test_model_parity.py -q -m gpu -x -k "lp256 or fp8a or bf16_activation or whisper or x3p or encoder" 2>&1 | tail -3
  echo "== lp256 per shape"
  timeout 300 python tools/microbench.py --lp256 2>&1 | grep -v amdgpu.ids | head -5
  echo "== x3p residual hoist"
  timeout 300 python tools/microbench.py --x3p-res 2>&1 | grep -v amdgpu.ids
  echo "== whisper, 32 layers"
  timeout 300 python tools/whisper_probe.py --layers 32 --prec bf16,fp8 2>&1 | grep -v amdgpu.ids | head -16
  for rep in 1 2; do for k in 0 1; do echo "-- headline, key 63 = $k"; bench --steps 20 --warmup 5 --knob 63=$k; done; done
} 2>&1 | tee gpurun_out/r6_ab.log
