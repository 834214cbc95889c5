#!/bin/bash
# Round 5, first visit -- the staged series applied (commit "apply the staged series").  Full GPU suite at the patched tree, the
# persistent-kernel probe (grid barriers, agent-scope hand-over: tools/persist_probe.hip), then what the series buys: the
# decoder's LayerNorms as launches (knob 45 = 0), inside the projections from a pre-pass (1) and from handed-over block
# statistics (3) -- one stream (probe + kernel trace) and the headline under eight workers; one cross-attention run per
# utterance; RelPosMHAXL on split operands.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
bench() { timeout 100 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r5a.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['config'].get('workers_per_gpu'), d['config'].get('batches_per_grouped_search'), d['config']['gpu_memory_reserved_gb'], d.get('parity_check'))
except Exception as e: print('no result', e)"; }
{
  echo "== suite"; timeout 1300 python -m pytest tests/ -q -m gpu 2>&1 | tail -12
  echo "== persistent-kernel probe"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/persist_probe tools/persist_probe.hip 2>&1 | tail -3; timeout 120 /tmp/persist_probe
  for k in 0 1 3; do echo "== decode probe, knob 45 = $k"; timeout 60 python tools/decode_probe.py --steps 16 --reps 3 --knob 45=$k 2>&1 | grep "decode probe"; done
  echo "== decode probe, knob 45 = 3, one cross-attention run per utterance"; timeout 60 python tools/decode_probe.py --steps 16 --reps 3 --knob 45=3 --knob 4=5 --knob 8=3 2>&1 | grep "decode probe"
  echo "== decode trace, knob 45 = 3"
  (cd /tmp && rm -rf /tmp/dtr && timeout 90 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $R/tools/decode_probe.py --steps 16 --reps 2 --knob 45=3 2>&1 | grep "decode probe")
  f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 32 | head -24
  for k in 0 3 1 0 3; do echo "== bench 8 x 4, knob 45 = $k"; bench --knob 45=$k; done
  echo "== bench 8 x 4, knob 45 = 3, one cross-attention run per utterance (4 = 5, 8 = 3)"; bench --knob 45=3 --knob 4=5 --knob 8=3
  echo "== RelPosMHAXL on split operands"; timeout 90 python tools/microbench.py --relpos-x3 2>&1 | grep "relpos attention"
  echo "== bench 8 x 4, knob 45 = 3, SBK_RELPOS_X3=1"; SBK_RELPOS_X3=1 bench --knob 45=3
  echo "== microbench"; timeout 60 python tools/microbench.py --x3r-ln 2>&1 | grep "x3r-ln" | head -12
} 2>&1 | tee gpurun_out/r5_a.log
