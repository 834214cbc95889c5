#!/bin/bash
# Round 6, visit K: gemm_x3r with eight waves per workgroup (K cut eight ways; knob 57 = the widest N that takes it): tests of both
# forms, the device timeline of a decoding step launch by launch (true durations), the step's wall clock, the headline A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6k.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "x3r" 2>&1 | tail -3
  for k in 0 512 2048 1048576; do
    echo "== timeline, knob 57=$k"
    (cd /tmp && rm -rf /tmp/tk$k && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tk$k -o t -- python $R/tools/decode_probe.py --steps 24 --reps 1 --knob 57=$k 2>&1 | grep "decode probe")
    f=$(find /tmp/tk$k -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/decode_timeline.py "$f" 16 | sed -n '1p;20,30p;54,57p'
  done
  for k in 0 512 2048 1048576 0 512; do echo "-- knob 57=$k"; timeout 150 python tools/decode_probe.py --steps 40 --reps 3 --knob 57=$k 2>&1 | grep "decode probe"; done
  echo "== bench A/B"
  for k in 0 512 2048 0 512 2048; do echo "-- knob 57=$k"; bench --knob 57=$k; done
} 2>&1 | tee gpurun_out/r6_k.log
