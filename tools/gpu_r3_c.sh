#!/bin/bash
# Round 3, call C: persistent GEMM (whole tiles + stream-K tail): parity, sweep, counters, bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest gemm"
  timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "gemm" 2>&1 | tail -4
  echo "== sweep"
  timeout 600 python tools/microbench.py --sk 2>&1 | grep -v amdgpu.ids | tail -130
  echo "== ablation"
  timeout 300 python tools/microbench.py --mfma-peak 2>&1 | grep -v amdgpu.ids | tail -16
  echo "== counters"
  CMD="python $PWD/tools/microbench.py --sk-pmc"
  n=0
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n+1))
    (cd /tmp && rm -rf /tmp/pmcB && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcB -o m -- $CMD > $OLDPWD/gpurun_out/r3c_pmc_$n.log 2>&1)
    f=$(find /tmp/pmcB -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ] && [ -s "$f" ]; then cp "$f" gpurun_out/r3c_pmc_$n.csv; echo "collected: $SET"; else echo "counter set failed: $SET"; tail -3 gpurun_out/r3c_pmc_$n.log; fi
  done
  python - <<'PY'
import csv, collections, re, glob
for fn in sorted(glob.glob("gpurun_out/r3c_pmc_*.csv")):
    rows = list(csv.DictReader(open(fn)))
    agg = collections.OrderedDict()
    for r in rows:
        m = re.search(r"(gemm_nt_kernel<[^>]*>|gemm_nt_sk_kernel)", r["Kernel_Name"])
        if not m:
            continue
        # the persistent kernel has one grid for every shape: key by launch order (3 shapes x (3 warm-up + 5 timed))
        k = (m.group(1)[:24], r["Grid_Size"], (int(r["Dispatch_Id"]) if "sk" in m.group(1) else 0))
        a = agg.setdefault(k, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0, 0.0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
        c[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    seen = 0
    for (name, grid, did), cs in agg.items():
        if "sk" in name:
            seen += 1
            if seen % 8 != 5:
                continue
        print(fn[-13:], name, "grid", grid, "dispatch", did, {c: round(v[1] / v[0]) for c, v in cs.items()}, "dur_us", round(next(iter(cs.values()))[2] / next(iter(cs.values()))[0] / 1e3, 1))
PY
  echo "== bench"
  timeout 900 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r3_c_bench.json 2> gpurun_out/r3_c_bench.err
  tail -3 gpurun_out/r3_c_bench.err
  python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3_c_bench.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "p50_latency_ms", "ms_per_step")}
    keep["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
    keep["breakdown"] = d.get("kernel_breakdown_ms")
    print(json.dumps(keep))
except Exception as e:
    print("no json:", e)
PY
} 2>&1 | tee gpurun_out/r3_c.log
