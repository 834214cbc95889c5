#!/bin/bash
# Round 3, call H: the whole GPU suite (timed), smoke, default bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest -m gpu (all)"
  SECONDS=0
  timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -30
  echo "suite seconds: $SECONDS"
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
  echo "== bench"
  timeout 1200 python bench.py --steps 12 --warmup 2 > gpurun_out/r3_h_bench.json 2> gpurun_out/r3_h_bench.err
  tail -3 gpurun_out/r3_h_bench.err
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3_h_bench.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "p50_latency_ms", "ms_per_step", "parity_check", "cpu_baseline", "config5_whisper_encoder")}
    keep["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
    keep["top3"] = [(e["kernel"], e["achieved"], e["frac"]) for e in d.get("roofline_top3", [])]
    keep["breakdown"] = d.get("kernel_breakdown_ms")
    print(json.dumps(keep))
except Exception as e:
    print("no json:", e)
PY
} 2>&1 | tee gpurun_out/r3_h.log
