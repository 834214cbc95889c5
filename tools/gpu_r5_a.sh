#!/bin/bash
# Round 5, first visit -- AFTER `git apply` of the two patches of staging/ (in order), both libraries rebuilt and the
# CPU suite green here.  Full GPU suite at the patched tree, then what the patch buys: the decoder's LayerNorms as launches
# (knob 45 = 0), inside the projections from a pre-pass (1) and from handed-over block statistics (3) -- the decoding step
# on one stream (probe + kernel trace) and the headline under the two schedules that measured best (8 x 4, 4 x 8).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 80 python bench.py --steps 16 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r5a.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  echo "== suite"; timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
  for k in 0 1 3; do echo "== decode probe, knob 45 = $k"; timeout 40 python tools/decode_probe.py --steps 16 --reps 3 --knob 45=$k 2>&1 | grep "decode probe"; done
  echo "== decode trace, knob 45 = 3"
  (cd /tmp && rm -rf /tmp/dtr && timeout 60 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $R/tools/decode_probe.py --steps 16 --reps 2 --knob 45=3 2>&1 | grep "decode probe")
  f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 32 | head -16
  echo "== the eight-worker regime from HIP events (instrumented run, 4 steps)"; timeout 80 python bench.py --steps 4 --prof-concurrent --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>>gpurun_out/r5a.err | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d[\"value\"], json.dumps(d.get(\"concurrent_kernels\")))"
  for rep in 1 2; do for k in 0 1 3; do echo "== bench 8 x 4, knob 45 = $k (run $rep)"; bench --knob 45=$k; done; done
  for k in 0 3; do echo "== bench 4 x 8, knob 45 = $k"; bench --streams 4 --group 8 --knob 45=$k; done
  for k in 0 3; do echo "== bench 8 x 4, knob 45 = $k, one cross-attention run per utterance (4 = 5, 8 = 3)"; bench --knob 45=$k --knob 4=5 --knob 8=3; done
  echo "== RelPosMHAXL on split operands (second staged patch)"; timeout 90 python tools/microbench.py --relpos-x3 2>&1 | grep "relpos attention"
  for v in 0 1; do echo "== bench 8 x 4, SBK_RELPOS_X3=$v"; SBK_RELPOS_X3=$v bench; done
  echo "== microbench"; timeout 60 python tools/microbench.py --x3r-ln 2>&1 | grep "x3r-ln" | head -12
} 2>&1 | tee gpurun_out/r5_a.log
