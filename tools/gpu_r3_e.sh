#!/bin/bash
# Round 3, call E: what limits the decode-step kernels (event times + PMC passes over tools/decode_probe.py), and the
# concurrency of the eight-worker timed region (kernel trace).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== counters available (SQ / TA / TCP / TCC subset)"
  (cd /tmp && timeout 120 rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|GRBM|TD)_[A-Z0-9_a-z]+" | sort -u | tr '\n' ' ' | head -c 6000); echo
  echo "== decode probe, event times"
  timeout 300 python tools/decode_probe.py --steps 16 --report 2>&1 | grep -v amdgpu.ids | tail -30
  echo "== PMC passes"
  CMD="python $PWD/tools/decode_probe.py --steps 6"
  n=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
    n=$((n+1))
    (cd /tmp && rm -rf /tmp/pmcE && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcE -o m -- $CMD > $OLDPWD/gpurun_out/r3e_pmc_$n.log 2>&1)
    f=$(find /tmp/pmcE -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ] && [ -s "$f" ]; then cp "$f" gpurun_out/r3e_pmc_$n.csv; echo "collected: $SET"; else echo "counter set failed: $SET"; tail -3 gpurun_out/r3e_pmc_$n.log; fi
  done
  python - <<'PY'
import csv, collections, re, glob
# per kernel class: mean counter value per dispatch over the SECOND search (the first one is the warm-up)
out = collections.OrderedDict()
for fn in sorted(glob.glob("gpurun_out/r3e_pmc_*.csv")):
    rows = list(csv.DictReader(open(fn)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    half = ids[len(ids) // 2]
    for r in rows:
        if int(r["Dispatch_Id"]) < half:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.split(r"[<(]", name)[0][-40:]
        if name.startswith("at::") or "rocclr" in name:
            continue
        a = out.setdefault(name, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
        d = a.setdefault("dur_us", [0, 0.0])
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "TCC_HIT_sum", "WRITE_SIZE", "SQ_INSTS_VALU"):
            d[0] += 1
            d[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
for name, cs in out.items():
    print(name, {c: round(v[1] / max(v[0], 1), 1) for c, v in cs.items()})
PY
  echo "== eight-worker timed region: kernel trace"
  (cd /tmp && rm -rf /tmp/trE && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trE -o t -- python $OLDPWD/bench.py --steps 6 --warmup 1 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 > $OLDPWD/gpurun_out/r3e_trace_bench.json 2> $OLDPWD/gpurun_out/r3e_trace_bench.err)
  tail -1 gpurun_out/r3e_trace_bench.json | cut -c1-300
  f=$(find /tmp/trE -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_overlap.py "$f" 0.3 0.9 | tee gpurun_out/r3e_trace_overlap.json
} 2>&1 | tee gpurun_out/r3_e.log
