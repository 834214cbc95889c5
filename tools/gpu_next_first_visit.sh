#!/bin/bash
# First GPU visit of the next round: what was prepared without GPU minutes at the end of round 3.
#  1. the GPU tests that have not run on hardware yet (cross-stream readiness of cached weight images, the cache's view keys)
#  2. tools/microbench.py --x3: the shipped split-operand kernel next to its prepared issue-order variants (knob 38 = 1 / 2 / 3)
#  3. the bench (its new reference leg value_fp32_mfma_contractions included)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== tests"
  timeout 200 python -m pytest tests/test_kernels.py -q -m gpu -k "another_stream or cache_follows_views or f32x3 or tiled_splitk or cross_attention_lds_dma" 2>&1 | tail -4
  echo "== microbench --x3 (columns: shipped kernel at three grids, knob 38 variants, error vs fp64)"
  timeout 300 python tools/microbench.py --x3 2>&1 | grep -v amdgpu.ids
  echo "== bench"
  timeout 400 python bench.py --no-cpu-baseline 2> gpurun_out/next_bench.err | tail -1 > gpurun_out/next_bench.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/next_bench.json"))
for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "value_fp32_mfma_contractions", "p50_latency_ms", "parity_check", "roofline"):
    print(k, json.dumps(d.get(k))[:400])
print(d["config"].get("gpu_memory_reserved_gb"), d["config"].get("fp32_mfma_leg_error"))
PY
} 2>&1 | tee gpurun_out/next_first_visit.log
