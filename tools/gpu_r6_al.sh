#!/bin/bash
# Round 6, visit AL: evidence for configs[4] at HEAD -- rocprofv3 kernel statistics of the Whisper large-v3 encoder forward (8 x 30 s, fp8 and
# bf16 pipelines) and the fabric traffic (FETCH_SIZE / WRITE_SIZE in separate passes) + MFMA busy of gemm_nt_lp256_kernel at the layer's four shapes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  for prec in fp8 bf16; do
    echo "== kernel statistics, whisper probe, 32 layers, $prec"
    (cd /tmp && rm -rf /tmp/kstw && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstw -o k -- python $R/tools/whisper_probe.py --layers 32 --prec $prec 2>&1 | grep "ms per forward" | cut -c1-160)
    f=$(find /tmp/kstw -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_al_whisper_${prec}_kernel_stats.csv && head -9 "$f" | cut -c1-230
  done
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1)); tag=$(echo fetch write mfma | cut -d' ' -f$i)
    (cd /tmp && rm -rf /tmp/pmcl && timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcl -o x -- python $R/tools/microbench.py --lp256-pmc > /tmp/pmcl.log 2>&1)
    f=$(find /tmp/pmcl -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" gpurun_out/r06_al_pmc_lp256_$tag.csv
  done
  python - <<'PY'
import csv, collections
SHAPES = ["M=12000 N=3840 K=1280 -> bf16", "M=12000 N=1280 K=1280 fp32 + residual", "M=12000 N=5120 K=1280 GELU -> bf16 / fp8", "M=12000 N=1280 K=5120 fp32 + residual"]
ALG = {  # algorithmic bytes per launch: both operands once + the result (+ the residual); bf16 then fp8
    0: [2 * (12000 * 1280 + 3840 * 1280) + 2 * 12000 * 3840, 2 * (12000 * 1280 + 1280 * 1280) + 8 * 12000 * 1280, 2 * (12000 * 1280 + 5120 * 1280) + 2 * 12000 * 5120, 2 * (12000 * 5120 + 1280 * 5120) + 8 * 12000 * 1280],
    1: [(12000 * 1280 + 3840 * 1280) + 2 * 12000 * 3840, (12000 * 1280 + 1280 * 1280) + 8 * 12000 * 1280, (12000 * 1280 + 5120 * 1280) + 1 * 12000 * 5120, (12000 * 5120 + 1280 * 5120) + 8 * 12000 * 1280]}
vals = {}
for tag in ("fetch", "write", "mfma"):
    try:
        rows = [r for r in csv.DictReader(open(f"gpurun_out/r06_al_pmc_lp256_{tag}.csv")) if "gemm_nt_lp256_kernel" in r["Kernel_Name"]]
    except Exception as e:
        print(tag, "missing", e); continue
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(r["Counter_Name"], []).append(r)
    for ctr, rs in per.items():
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        for k in range(8):  # 7 launches per (type, shape): one warm-up + six
            grp = rs[7 * k + 3: 7 * k + 7]
            if grp:
                v = sum(float(r["Counter_Value"]) for r in grp) / len(grp)
                d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp]
                vals[(ctr, k)] = (v, sum(d) / len(d) / 1e3)
for k in range(8):
    fp8, si = k // 4, k % 4
    f, w = vals.get(("FETCH_SIZE", k)), vals.get(("WRITE_SIZE", k))
    m, g = vals.get(("SQ_VALU_MFMA_BUSY_CYCLES", k)), vals.get(("GRBM_GUI_ACTIVE", k))
    line = f"{'fp8 ' if fp8 else 'bf16'} {SHAPES[si]:44s}"
    if f and w:
        traffic = (2.0 * f[0] + w[0]) * 1024.0  # bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB: profiles/pmc_traffic.json's _source and the guide's gfx950 units
        line += f" FETCH_SIZE {f[0]:9.0f} WRITE_SIZE {w[0]:9.0f} (KiB) = {traffic / ALG[fp8][si]:5.2f} x algorithmic ({ALG[fp8][si] / 1e6:6.1f} MB), {f[1]:6.1f} us"
    if m and g:
        line += f" | MFMA busy {m[0] / (g[0] / 8 * 1024):5.2f} of the launch"
    print(line)
PY
} 2>&1 | tee gpurun_out/r6_al.log
