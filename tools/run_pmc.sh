#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters (separate passes, as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never combined with other traces
# than --kernel-trace).  Output: gpurun_out/pmc_fetch_*.csv, gpurun_out/pmc_write_*.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CMD="python $PWD/tools/microbench.py --pmc-workload"
(cd /tmp && rm -rf /tmp/pmc1 && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc1 -o f -- $CMD > $OLDPWD/gpurun_out/pmc_fetch.log 2>&1)
(cd /tmp && rm -rf /tmp/pmc2 && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc2 -o w -- $CMD > $OLDPWD/gpurun_out/pmc_write.log 2>&1)
find /tmp/pmc1 -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_fetch_counters.csv \;
find /tmp/pmc2 -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_write_counters.csv \;
ls -la /tmp/pmc1 /tmp/pmc2; ls -la gpurun_out | tail -5
python - <<'PY'
import csv, collections
for tag in ("fetch", "write"):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/pmc_{tag}_counters.csv")))
    except Exception as e:
        print(tag, "missing", e); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        k = r.get("Kernel_Name", "?").split("(")[0][-60:]
        agg[k][0] += 1; agg[k][1] += float(r.get("Counter_Value", 0))
    print("==", tag, list(rows[0].keys()) if rows else None)
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:62s} launches={n:6d} sum={v:14.1f} per_launch={v/n:12.2f}")
PY
