#!/bin/bash
# Round 3, final visit 1: GPU suite (without the 7-minute oracle-at-30-s test), the driver's bench command, kernel statistics
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest -m gpu (all but the oracle-at-30-s test)"
  S=$(date +%s)
  timeout 900 python -m pytest tests -q -m gpu -k "not headline_shape_beam10" --durations=4 2>&1 | tail -12
  echo "suite seconds: $(( $(date +%s) - S ))"
  echo "== bench (driver command)"
  S=$(date +%s)
  timeout 600 python bench.py 2> gpurun_out/r3f_bench.err | tail -1 > gpurun_out/r3f_bench.json
  echo "bench seconds: $(( $(date +%s) - S ))"
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r3f_bench.json"))
for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "p50_latency_ms", "parity_check", "roofline", "roofline_top3", "cpu_baseline", "config1_encoder_S", "config5_whisper_encoder"):
    print(k, json.dumps(d.get(k))[:700])
print("kernel_breakdown_ms", json.dumps(d.get("kernel_breakdown_ms"))[:1500])
PY
  echo "== rocprofv3 --kernel-trace --stats: bench.py --streams 1 --steps 4"
  S=$(date +%s)
  (cd /tmp && rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o s -- python $OLDPWD/bench.py --streams 1 --steps 4 --no-extras --no-cpu-baseline --no-roofline --latency-runs 0 > $OLDPWD/gpurun_out/r3f_prof.log 2>&1)
  f=$(find /tmp/prof_f -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3f_kernel_stats_single_stream.csv
  f=$(find /tmp/prof_f -name "*domain_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3f_domain_stats_single_stream.csv
  echo "profile seconds: $(( $(date +%s) - S ))"
  head -8 gpurun_out/r3f_kernel_stats_single_stream.csv | cut -c1-160
} 2>&1 | tee gpurun_out/r3_f1.log
