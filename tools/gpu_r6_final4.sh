#!/bin/bash
# Round 6, final visit 4: the full GPU suite, smoke() and the driver's own bench command at HEAD (after sbk::gelu_erfc).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== suite"; timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -4
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
  echo "== driver bench"
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final4_bench.json 2> gpurun_out/r06_final4_bench.err
  tail -c 400 gpurun_out/r06_final4_bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final4_bench.json").read().strip().splitlines()[-1])
keys = ["value", "ms_per_step", "value_batch128", "value_encoder_gemms_bf16", "value_fp32_mfma_contractions", "p50_latency_ms", "p50_latency_ms_by_mode",
        "decode_step_ms", "launches_per_decode_step"]
print({k: d.get(k) for k in keys})
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms", "launches", "share_of_gpu_time", "traffic")})
print("top3", [(e["kernel"], e["frac"], e["avg_launch_ms"], e.get("traffic_over_algorithmic")) for e in d["roofline_top3"]])
print("checks", d["parity_check"]["ids_equal"], d["determinism_check"]["ids_equal"], d.get("token_error_rate_vs_oracle_12x10s_peaked_heads", {}).get("ids_equal"))
print("cpu", d.get("cpu_baseline", {}).get("value"), "mem", d["config"]["gpu_memory_reserved_gb"], "workers", d["config"]["workers_per_gpu"], d["config"]["batches_per_grouped_search"])
print("breakdown", d.get("kernel_breakdown_ms"))
PY
} 2>&1 | tee gpurun_out/r6_final4.log
