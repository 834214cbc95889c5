#!/bin/bash
# Round 6, visit AI: where the bf16 attention kernel's wave cycles go (SQ counters: active / parked / issue-stalled, VALU, MFMA, LDS),
# and the tile count from which the 256 x 256 contraction route pays (tools/microbench.py --lp256 --lp256-small).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  echo "== lp256 from how many tiles on"
  timeout 300 python tools/microbench.py --lp256 --lp256-small 2>&1 | grep -v amdgpu.ids
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES"; do
    echo "== counters: $SET"
    (cd /tmp && rm -rf /tmp/pmca && timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmca -o a -- python $R/tools/whisper_probe.py --layers 2 --prec bf16 > /tmp/pmca.log 2>&1; tail -2 /tmp/pmca.log | cut -c1-200)
    f=$(find /tmp/pmca -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for kn in ("attn_lds_bf16_kernel", "gemm_nt_lp256_kernel"):
    per = collections.OrderedDict()
    for r in rows:
        if kn in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in per.items():
        v = v[len(v) // 2:]  # (the later launches: warm)
        print(f"{kn:24s} {c:28s} per_launch={sum(v) / len(v):16.1f}  ({len(v)} launches)")
PY
  done
} 2>&1 | tee gpurun_out/r6_ai.log
