#!/bin/bash
# Round 6, visit L: gemm_x3r with the operand loads of two k steps issued together (knob 58: bit 0 the plain kernel, bit 1 the
# LayerNorm-prologue kernel): tests, the decoding step launch by launch, the step's wall clock, the headline A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6l.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "x3r" 2>&1 | tail -3
  for k in 0 1 3; do
    echo "== timeline, knob 58=$k"
    (cd /tmp && rm -rf /tmp/tl$k && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$k -o t -- python $R/tools/decode_probe.py --steps 24 --reps 1 --knob 58=$k 2>&1 | grep "decode probe")
    f=$(find /tmp/tl$k -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/decode_timeline.py "$f" 16 | sed -n '1p;20,30p;54,57p'
  done
  for k in 0 1 3 0 1 3; do echo "-- knob 58=$k"; timeout 150 python tools/decode_probe.py --steps 40 --reps 3 --knob 58=$k 2>&1 | grep "decode probe"; done
  echo "== bench A/B"
  for k in 0 1 3 0 1 3; do echo "-- knob 58=$k"; bench --knob 58=$k; done
} 2>&1 | tee gpurun_out/r6_l.log
