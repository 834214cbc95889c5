#!/bin/bash
# Round 6, visit N: the CTC posteriors with frame rows padded to whole 128-byte lines: CTC tests on the GPU, the score kernel's fabric
# counters again (1.25 x its algorithmic bytes before: a 1 KB segment of a 20 000-byte row is 9 lines), the step's timeline, two headline runs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 200 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r6n.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'])
except Exception as e: print('no result', e)"; }
{
  timeout 900 python -m pytest tests/test_model_parity.py tests/test_kernels.py tests/test_full_size_gpu.py -q -m gpu -x -k "golden or ctc or window or partial or step_protocol or long_utterance or wide_beam or waveform or headline_shape or conformer_l_decoder" 2>&1 | tail -3
  echo "== counters: the decode step's memory-bound kernels"; bash tools/run_pmc_r6_decode.sh 2>&1 | tail -6
  cp gpurun_out/pmc_r6_decode_fetch.csv gpurun_out/r06_n_pmc_decode_fetch.csv; cp gpurun_out/pmc_r6_decode_write.csv gpurun_out/r06_n_pmc_decode_write.csv
  echo "== timeline, 24 steps"
  (cd /tmp && rm -rf /tmp/tn && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tn -o t -- python $R/tools/decode_probe.py --steps 24 --reps 1 2>&1 | grep "decode probe")
  f=$(find /tmp/tn -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/decode_timeline.py "$f" 16 | sed -n '1,12p;54,58p'
  echo "== bench"; bench; bench
} 2>&1 | tee gpurun_out/r6_n.log
