#!/bin/bash
# Round 6, visit M: the tree's defaults (ring / CTC loads non-temporal, CTC frame loops clipped to the utterance, paired operand loads in
# the plain gemm_x3r, wave-per-(hypothesis, head) self-attention): the whole decoding step launch by launch, the fabric counters of the
# memory-bound decode kernels again (the CTC ratio after the clip), two 12-step headline runs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6m.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  for st in 24 60; do
    echo "== timeline at HEAD, $st steps"
    (cd /tmp && rm -rf /tmp/tm$st && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tm$st -o t -- python $R/tools/decode_probe.py --steps $st --reps 1 2>&1 | grep "decode probe")
    f=$(find /tmp/tm$st -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/decode_timeline.py "$f" 16
  done
  echo "== counters: the decode step's memory-bound kernels"; bash tools/run_pmc_r6_decode.sh 2>&1 | tail -6
  cp gpurun_out/pmc_r6_decode_fetch.csv gpurun_out/r06_m_pmc_decode_fetch.csv; cp gpurun_out/pmc_r6_decode_write.csv gpurun_out/r06_m_pmc_decode_write.csv
  echo "== bench"; bench; bench
} 2>&1 | tee gpurun_out/r6_m.log
