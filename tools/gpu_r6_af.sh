#!/bin/bash
# Round 6, visit AF: gemm_nt_x3p_kernel with the hot epilogue forms as straight-line code (key 63): tests through it (kernels, encoder
# against the oracle, goldens), per-shape A/B with bit-identity of the two forms, headline A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 300 python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6af.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])
except Exception as e: print('no result', e)"; }
{
  echo "== tests"
  timeout 1200 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_full_size_gpu.py -q -m gpu -x -k "x3p or panel or encoder or golden or conformer" 2>&1 | tail -3
  echo "== per shape, generic / straight-line epilogue"
  timeout 300 python tools/microbench.py --x3p-modes --x3p-epi 2>&1 | grep -v amdgpu.ids
  for rep in 1 2; do for k in 0 1; do echo "-- headline, key 63 = $k"; bench --steps 20 --warmup 5 --knob 63=$k; done; done
} 2>&1 | tee gpurun_out/r6_af.log
