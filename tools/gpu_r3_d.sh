#!/bin/bash
# Round 3, call D: headline-shape parity tests, in-situ encoder GEMM routing A/B, bench with the new defaults.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== new full-size tests"
  timeout 1500 python -m pytest tests/test_full_size_gpu.py -q -m gpu -x -k "headline or long_utterance or encoder_vs_oracle" 2>&1 | tail -15
  echo "== encoder in situ"
  timeout 600 python tools/microbench.py --enc-layer 2>&1 | grep -v amdgpu.ids | tail -20
  echo "== bench"
  timeout 1200 python bench.py --steps 8 --warmup 2 > gpurun_out/r3_d_bench.json 2> gpurun_out/r3_d_bench.err
  tail -5 gpurun_out/r3_d_bench.err
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3_d_bench.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "p50_latency_ms", "ms_per_step", "parity_check", "cpu_baseline", "per_rank", "token_error_rate_vs_oracle")}
    keep["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
    keep["top3"] = [(e["kernel"], e["achieved"], e["frac"]) for e in d.get("roofline_top3", [])]
    keep["breakdown"] = d.get("kernel_breakdown_ms")
    print(json.dumps(keep))
except Exception as e:
    print("no json:", e)
PY
} 2>&1 | tee gpurun_out/r3_d.log
