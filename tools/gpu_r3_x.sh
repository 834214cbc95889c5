#!/bin/bash
# Round 3, call X: GPU suite (without the 7-minute oracle test) and the driver's bench command with the split-operand contractions on
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest -m gpu (all but the oracle-at-30-s test)"
  S=$(date +%s)
  timeout 900 python -m pytest tests -q -m gpu -k "not headline_shape_beam10" -x --durations=4 2>&1 | tail -12
  echo "suite seconds: $(( $(date +%s) - S ))"
  echo "== bench (driver command)"
  S=$(date +%s)
  timeout 600 python bench.py 2> gpurun_out/r3x_bench.err | tail -1 > gpurun_out/r3x_bench.json
  echo "bench seconds: $(( $(date +%s) - S ))"
  cut -c1-400 gpurun_out/r3x_bench.json
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r3x_bench.json"))
for k in ("value", "value_batch128", "p50_latency_ms", "parity_check", "roofline", "roofline_top3", "cpu_baseline", "config1_encoder_S"):
    print(k, json.dumps(d.get(k))[:600])
print("kernel_breakdown_ms", json.dumps(d.get("kernel_breakdown_ms"))[:1500])
PY
} 2>&1 | tee gpurun_out/r3_x.log
