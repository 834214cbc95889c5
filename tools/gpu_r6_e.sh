#!/bin/bash
# Round 6, visit E: the few-row split-operand kernel on 128 x 64 tiles (knob 56): tests of both tile heights, per-shape times in
# isolation, the decoding step (wall clock of the probe without events), the headline A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6e.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  timeout 600 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "x3r or persistent_few_row or ctc or window" 2>&1 | tail -3
  timeout 300 python tools/x3r_tiles_probe.py 2>&1 | grep -v amdgpu.ids
  for k in 0 1 0 1; do echo "-- knob 56=$k"; timeout 150 python tools/decode_probe.py --steps 40 --reps 3 --knob 56=$k 2>&1 | grep "decode probe"; done
  echo "== bench A/B"
  for k in 0 1 0 1; do echo "-- knob 56=$k"; bench --knob 56=$k; done
} 2>&1 | tee gpurun_out/r6_e.log
