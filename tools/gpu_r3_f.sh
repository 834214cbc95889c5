#!/bin/bash
# Round 3, call F: LDS-DMA / MFMA cross-attention step (knob 4 = 5) and the persistent GEMM on decode shapes (knob 24).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest cross attention variant + model parity"
  timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "cross_attention_lds_dma" 2>&1 | tail -4
  for K in "4=0" "4=5" "4=5 --knob 24=1024" "4=0 --knob 24=1024"; do
    echo "== decode probe, knob $K"
    timeout 300 python tools/decode_probe.py --steps 16 --report --knob $K 2>&1 | grep -v amdgpu.ids | head -12
  done
  for K in "4=0" "4=5"; do
    echo "== bench, knob $K"
    timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --latency-runs 3 --knob $K > gpurun_out/r3_f_bench_$K.json 2> gpurun_out/r3_f_bench_$K.err
    tail -2 gpurun_out/r3_f_bench_$K.err
    python - "$K" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/r3_f_bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "p50_latency_ms", "ms_per_step", "parity_check")}
    keep["top3"] = [(e["kernel"], e["achieved"], e["frac"]) for e in d.get("roofline_top3", [])]
    keep["breakdown"] = d.get("kernel_breakdown_ms")
    print(json.dumps(keep))
except Exception as e:
    print("no json:", e)
PY
  done
} 2>&1 | tee gpurun_out/r3_f.log
