#!/bin/bash
# Round 3, call B: MFMA ceiling / stream-K ablations, counters of the two big-GEMM kernels, full GPU suite, bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== mfma peak + ablations"
  timeout 300 python tools/microbench.py --mfma-peak 2>&1 | grep -v amdgpu.ids | tail -30
  echo "== counters"
  CMD="python $PWD/tools/microbench.py --sk-pmc"
  n=0
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n+1))
    (cd /tmp && rm -rf /tmp/pmcB && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcB -o m -- $CMD > $OLDPWD/gpurun_out/r3_pmc_$n.log 2>&1)
    f=$(find /tmp/pmcB -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ] && [ -s "$f" ]; then cp "$f" gpurun_out/r3_pmc_$n.csv; echo "collected: $SET"; else echo "counter set failed: $SET"; tail -3 gpurun_out/r3_pmc_$n.log; fi
  done
  python - <<'PY'
import csv, collections, re, glob
for fn in sorted(glob.glob("gpurun_out/r3_pmc_*.csv")):
    rows = list(csv.DictReader(open(fn)))
    agg = collections.OrderedDict()
    for r in rows:
        m = re.search(r"(gemm_nt_kernel<[^>]*>|gemm_nt_sk_kernel)", r["Kernel_Name"])
        if not m:
            continue
        k = (m.group(1)[:40], r["Grid_Size"])
        a = agg.setdefault(k, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0, 0.0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
        c[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for (name, grid), cs in agg.items():
        print(fn[-12:], name, "grid", grid, {c: round(v[1] / v[0]) for c, v in cs.items()}, "dur_us", round(next(iter(cs.values()))[2] / next(iter(cs.values()))[0] / 1e3, 1))
PY
  echo "== pytest -m gpu (all)"
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
  echo "== bench"
  timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/r3_b_bench.json 2> gpurun_out/r3_b_bench.err
  tail -3 gpurun_out/r3_b_bench.err
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3_b_bench.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "p50_latency_ms", "ms_per_step")}
    keep["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
    keep["top3"] = [(e["kernel"], e["achieved"], e["frac"]) for e in d.get("roofline_top3", [])]
    keep["breakdown"] = d.get("kernel_breakdown_ms")
    keep["whisper"] = d.get("config5_whisper_encoder")
    print(json.dumps(keep))
except Exception as e:
    print("no json:", e)
PY
} 2>&1 | tee gpurun_out/r3_b.log
