#!/bin/bash
# Round 6, visit D: the shared-ancestry self-attention on the GPU (tests, then per-kernel times at 24 / 60 decoding steps against the
# wave-per-(hypothesis, head) kernel, knob 55), the headline A/B, and workers x batches per grouped search again (the projections run
# at their isolated speed in the step: larger searches amortise their fixed cost).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6d.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "shared_ancestry or persistent_few_row or golden_model or wide_beam or grouped_search" 2>&1 | tail -3
  for st in 24 60; do for k in 0 1 0 1; do echo "-- steps $st knob 55=$k"; timeout 150 python tools/decode_probe.py --steps $st --reps 2 --report --knob 55=$k 2>&1 | grep -E "decode probe|self_attn"; done; done
  echo "== bench A/B"
  for k in 0 1 0 1; do echo "-- knob 55=$k"; bench --knob 55=$k; done
  echo "== workers x batches per search"
  bench --streams 4 --group 8
  bench --streams 6 --group 6
  bench --streams 8 --group 6
  bench --streams 6 --group 8
  bench --streams 8 --group 4
} 2>&1 | tee gpurun_out/r6_d.log
