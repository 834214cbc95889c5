#!/bin/bash
# single-stream rocprofv3 kernel statistics for several knob settings of the same short run (A/B of kernel variants in situ)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && rm -rf /tmp/prof_ab && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o st -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-roofline --no-extras --latency-runs 0 --streams 1 --steps 4 --warmup 1 $line > "$OLDPWD/gpurun_out/stats_ab_run.log" 2>&1)
  f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
  echo "== $line"
  [ -n "$f" ] && cp "$f" gpurun_out/stats_ab_$i.csv && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/1e6:.1f} ms")
for r in rows[:16]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print(f"  {n:46s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.1f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
done < "${STATS_CASES:-tools/stats_cases.txt}"
