#!/bin/bash
# Round 6, visit I: the shared-ancestry self-attention, second form (a workgroup per (utterance, head), its four waves a quarter of the
# prefix each): GPU tests, HIP-event time per launch at 24 / 60 steps against the wave-per-(hypothesis, head) kernel (knob 55 = 0),
# the device timeline (true durations), L2 requests of both kernels (TCC_HIT_sum + TCC_MISS_sum, one --pmc pass each), headline A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6i.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "shared_ancestry or persistent_few_row or golden_model or wide_beam or grouped_search or step_protocol or ctc" 2>&1 | tail -3
  for st in 24 60; do for k in 0 1; do echo "-- steps $st knob 55=$k"; timeout 150 python tools/decode_probe.py --steps $st --reps 2 --report --knob 55=$k 2>&1 | grep -E "decode probe|self_attn"; done; done
  for k in 0 1; do
    echo "== timeline, 60 steps, knob 55=$k"
    (cd /tmp && rm -rf /tmp/ti$k && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ti$k -o t -- python $R/tools/decode_probe.py --steps 60 --reps 1 --knob 55=$k 2>&1 | grep "decode probe")
    f=$(find /tmp/ti$k -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/decode_timeline.py "$f" 20 | grep -E "steps of|self_attn" | head -8
    echo "== L2 requests, knob 55=$k"
    (cd /tmp && rm -rf /tmp/pi$k && timeout 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pi$k -o x -- python $R/tools/decode_probe.py --steps 60 --reps 1 --knob 55=$k > /dev/null 2>&1)
    f=$(find /tmp/pi$k -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" gpurun_out/r06_i_pmc_self_attn_l2_requests_knob55_$k.csv && python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "self_attn" in r["Kernel_Name"]]
per = collections.defaultdict(list)
for r in rows: per[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = min(len(v) for v in per.values()) if per else 0
for k, v in per.items():
    tail = v[-(n // 2):] if n else v
    print(f"  {k:14s} launches {len(v):5d}  mean over the second half (long prefixes) {sum(tail) / max(len(tail), 1):14.1f}")
PY
  done
  echo "== bench A/B"
  for k in 0 1 0 1; do echo "-- knob 55=$k"; bench --knob 55=$k; done
} 2>&1 | tee gpurun_out/r6_i.log
