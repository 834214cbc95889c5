#!/bin/bash
# Round 5, visit B: the persistent few-row decoding step (csrc/decoder_persist.hip) on the GPU for the first time -- its
# parity tests and the tests whose path it changes, the single-utterance latency with and without it (grid sweep, helper
# stream), a kernel trace of the search, then the headline A/B of the one-run cross-attention (knob 4 = 5, 8 = 3) with repeats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 100 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r5b.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'))
except Exception as e: print('no result', e)"; }
{
  echo "== tests"; timeout 900 python -m pytest tests/ -q -m gpu -k "persistent_few_row or golden_model or x3r_route or conformer_l_decoder or greedy_beam1 or properties_at_bench_shape or grouped_search or run_opts_precision or panel_route_serves or abi or fused_scoring or whisper_greedy or whisper_beam or degenerate or long_utterance_search or headline_shape or wide_beam or partial_ctc or window" 2>&1 | tail -8
  echo "== latency, launch per operation"
  timeout 120 python tools/latency_probe.py --knob 47=0 --overlap 3 2>&1 | grep "latency probe"
  timeout 120 python tools/latency_probe.py --knob 47=0 --overlap 0 2>&1 | grep "latency probe"
  echo "== latency, persistent step"
  for g in 32 64 96 128 192 256; do timeout 120 python tools/latency_probe.py --knob 48=$g 2>&1 | grep "latency probe"; done
  timeout 120 python tools/latency_probe.py --overlap 3 2>&1 | grep "latency probe"
  timeout 120 python tools/latency_probe.py --knob 48=64 --overlap 3 --report 2>&1 | grep -A14 "latency probe"
  timeout 120 python tools/latency_probe.py --seconds 28 --knob 47=0 --overlap 3 2>&1 | grep "latency probe"
  timeout 120 python tools/latency_probe.py --seconds 28 2>&1 | grep "latency probe"
  echo "== kernel trace of the persistent search"
  (cd /tmp && rm -rf /tmp/ltr && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ltr -o t -- python $R/tools/latency_probe.py --runs 3 2>&1 | grep "latency probe")
  f=$(find /tmp/ltr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f"
  for rep in 1 2 3; do
    echo "== bench 8 x 4, default (run $rep)"; bench
    echo "== bench 8 x 4, one cross-attention run per utterance (run $rep)"; bench --knob 4=5 --knob 8=3
  done
  echo "== bench (python bench.py --no-extras): the line with the latency modes"
  timeout 400 python bench.py --no-extras --no-cpu-baseline 2>>gpurun_out/r5b.err | tail -1 > gpurun_out/r5b_bench.json
  python -c "
import json; d = json.load(open('gpurun_out/r5b_bench.json')); print(d['value'], d.get('p50_latency_ms'), d.get('p50_latency_ms_by_mode'), d.get('parity_check'))"
} 2>&1 | tee gpurun_out/r5_b.log
