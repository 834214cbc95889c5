#!/bin/bash
# Round 3, final visit 2: counters of the split-operand kernel (fabric traffic, MFMA busy / clock) and the bench with a
# trimmed allocator between its legs (CPU baseline skipped here: profiles/r03_f_bench.json has it)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== counters"
  bash tools/run_pmc_x3.sh 2>&1 | tail -14
  echo "== bench --no-cpu-baseline"
  S=$(date +%s)
  timeout 300 python bench.py --no-cpu-baseline 2> gpurun_out/r3g_bench.err | tail -1 > gpurun_out/r3g_bench.json
  echo "bench seconds: $(( $(date +%s) - S ))"
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r3g_bench.json"))
for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "bf16_vs_fp32_token_error_rate_percent", "p50_latency_ms", "parity_check"):
    print(k, json.dumps(d.get(k))[:300])
print(d["config"].get("gpu_memory_reserved_gb"))
PY
} 2>&1 | tee gpurun_out/r3_f2.log
