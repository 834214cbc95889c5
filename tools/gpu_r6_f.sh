#!/bin/bash
# Round 6, visit F: is the 8-worker headline bound by the GPU or by the host process?  Two bench processes of eight workers each
# on the same GPU at the same time (each with half the steps) against one process; then one process with 16 workers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
val() { python -c "
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('workers_per_gpu'))
except Exception as e: print(sys.argv[1], 'no result', e)" $1; }
B="python bench.py --no-extras --no-roofline --no-cpu-baseline --latency-runs 0"
{
  echo "== one process, 12 steps"
  timeout 300 $B --steps 12 > gpurun_out/f_one.json 2>>gpurun_out/r6f.err; val gpurun_out/f_one.json
  echo "== two processes at once, 12 steps each"
  (timeout 400 $B --steps 12 > gpurun_out/f_two_a.json 2>>gpurun_out/r6f.err) &
  (timeout 400 $B --steps 12 > gpurun_out/f_two_b.json 2>>gpurun_out/r6f.err) &
  wait
  val gpurun_out/f_two_a.json; val gpurun_out/f_two_b.json
  echo "== one process, 16 workers x 4"
  timeout 300 $B --steps 12 --streams 16 --group 4 > gpurun_out/f_16.json 2>>gpurun_out/r6f.err; val gpurun_out/f_16.json
  echo "== one process, 8 workers x 4 again"
  timeout 300 $B --steps 12 > gpurun_out/f_one2.json 2>>gpurun_out/r6f.err; val gpurun_out/f_one2.json
} 2>&1 | tee gpurun_out/r6_f.log
