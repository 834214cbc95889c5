#!/bin/bash
# Counters of the contraction kernels at the round-6 tree (gemm_x3r_kernel: the decode step's projections at 1 280 rows; gemm_nt_x3p_kernel:
# the encoder's at 12 800 - 24 032 rows): fabric traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes) and MFMA busy / clock (one pass).  --pmc only with --kernel-trace.  Output: gpurun_out/pmc_r6_{fetch,write,mfma}.csv
# + a per-shape summary on stdout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
# (90 s per pass: a healthy pass takes 10-25 s; in visit R rocprofv3 died at start-up and sat in its signal handler until the
#  timeout -- two 300-second waits were the round's last GPU minutes, profiles/r05_r_*)
CMD="python $PWD/tools/microbench.py --r4-pmc"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1)); tag=$(echo fetch write mfma | cut -d' ' -f$i)
  [ $i -gt ${PMC_PASSES:-3} ] && break  # (PMC_PASSES=2: the two traffic passes only)
  (cd /tmp && rm -rf /tmp/pmcx6 && timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcx6 -o x -- $CMD > $OLDPWD/gpurun_out/pmc_r6_$tag.log 2>&1)
  f=$(find /tmp/pmcx6 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/pmc_r6_$tag.csv
done
python - <<'PY'
import csv, collections
KERNELS = {"gemm_x3r_kernel": ["M=1280 N=512 K=512", "M=1280 N=2048 K=512", "M=1280 N=512 K=2048", "M=1280 N=5000 K=512"],
           "gemm_nt_x3p_kernel": ["M=12800 N=2048 K=512", "M=12800 N=512 K=2048", "M=24032 N=1536 K=512"]}  # 6 launches each
for tag in ("fetch", "write", "mfma"):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/pmc_r6_{tag}.csv")))
    except Exception as e:
        print(tag, "missing", e); continue
    print("==", tag)
    for kname, shapes in KERNELS.items():
        per = collections.OrderedDict()
        for r in rows:
            if kname in r["Kernel_Name"]:
                per.setdefault(r.get("Counter_Name", "?"), []).append(r)
        for ctr, rs in per.items():
            rs.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
            for si, shape in enumerate(shapes):
                grp = rs[6 * si + 2: 6 * si + 6]  # (the first two launches of a shape warm the caches)
                if not grp:
                    continue
                v = sum(float(r["Counter_Value"]) for r in grp) / len(grp)
                d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp if r.get("End_Timestamp")]
                dur = (sum(d) / len(d) / 1e3) if d else float("nan")
                print(f"{kname:20s} {shape:24s} {ctr:28s} per_launch={v:16.1f} avg_us={dur:8.1f}")
PY
