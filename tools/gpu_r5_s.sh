#!/bin/bash
# Round 5, last visit (75 GPU-seconds left after the counters run of visit R hung in rocprofv3 at start-up): the headline leg at HEAD.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 28 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>gpurun_out/r5s.err | tail -1 > gpurun_out/r05_s_bench_12_steps.json
python -c "
import json; d = json.load(open('gpurun_out/r05_s_bench_12_steps.json')); print('12 steps:', d['value'], d.get('parity_check'), d.get('determinism_check'))" 2>&1 | cut -c1-400
timeout 45 python bench.py --no-extras --no-cpu-baseline --latency-runs 0 2>>gpurun_out/r5s.err | tail -1 > gpurun_out/r05_s_bench_16_steps_roofline.json
python -c "
import json; d = json.load(open('gpurun_out/r05_s_bench_16_steps_roofline.json')); print('16 steps:', d['value'], d.get('decode_step_ms'), d.get('launches_per_decode_step')); print(d.get('roofline_top3')); print(d.get('kernel_breakdown_ms'))" 2>&1 | cut -c1-1500
