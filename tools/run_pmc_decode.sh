#!/bin/bash
# Wave-level counters + true kernel durations of the decode-step GEMM variants (one --pmc pass; only --kernel-trace beside it).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
CMD="python $PWD/tools/microbench.py --pmc-decode"
(cd /tmp && rm -rf /tmp/pmc4 && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc4 -o m -- $CMD > $OLDPWD/gpurun_out/pmc_decode.log 2>&1)
f=$(find /tmp/pmc4 -name "*counter_collection.csv" | head -1); k=$(find /tmp/pmc4 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/pmc_decode_counters.csv; [ -n "$k" ] && cp "$k" gpurun_out/pmc_decode_trace.csv
tail -3 gpurun_out/pmc_decode.log
python - <<'PY'
import csv, collections, re
agg = collections.OrderedDict()
for r in csv.DictReader(open("gpurun_out/pmc_decode_counters.csv")):
    m = re.search(r"(gemm_\w+kernel<[^>]*>|splitk_reduce_kernel)", r["Kernel_Name"])
    if not m:
        continue
    a = agg.setdefault((m.group(1)[:40], r["Grid_Size"]), collections.OrderedDict())
    c = a.setdefault(r["Counter_Name"], [0, 0.0]); c[0] += 1; c[1] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_decode_trace.csv")):
    m = re.search(r"(gemm_\w+kernel<[^>]*>|splitk_reduce_kernel)", r["Kernel_Name"])
    if m:
        dur[(m.group(1)[:40], r["Grid_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, ctrs in agg.items():
    d = dur.get(k, [0]); d = sorted(d)[len(d) // 2]
    v = {n: c[1] / c[0] for n, c in ctrs.items()}
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k[0]:40s} grid={k[1]:>8s} dur={d:7.1f}us waves={v.get('SQ_WAVES',0):7.0f} gui={v.get('GRBM_GUI_ACTIVE',0):9.0f} busy={v.get('SQ_BUSY_CYCLES',0):10.0f} "
          f"wave_cyc={wc:11.0f} wait_any={v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst={v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active={v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} mfma_busy={v.get('SQ_VALU_MFMA_BUSY_CYCLES',0):11.0f}")
PY
