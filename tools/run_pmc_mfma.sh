#!/bin/bash
# MFMA-busy / wait-state counters of the dominant kernels (one --pmc pass; only --kernel-trace beside it).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CMD="python $PWD/tools/microbench.py --pmc-workload"
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  (cd /tmp && rm -rf /tmp/pmc3 && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc3 -o m -- $CMD > $OLDPWD/gpurun_out/pmc_mfma.log 2>&1)
  f=$(find /tmp/pmc3 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ] && [ -s "$f" ]; then cp "$f" gpurun_out/pmc_mfma_counters.csv; echo "collected: $SET"; break; fi
  echo "counter set failed: $SET"; tail -3 gpurun_out/pmc_mfma.log
done
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open("gpurun_out/pmc_mfma_counters.csv")))
agg = collections.OrderedDict()
for r in rows:
    m = re.search(r"(gemm_nt_kernel<[^>]*>|gemm_skinny_kernel<\d+>|ctc_score_step_kernel<[^>]*>)", r["Kernel_Name"])
    if not m:
        continue
    k = (m.group(1), r["Grid_Size"], r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for (name, grid, ctr), (n, v) in agg.items():
    print(f"{name[:44]:44s} grid={grid:>9s} {ctr:28s} n={n:3d} per_launch={v / n:16.1f}")
PY
