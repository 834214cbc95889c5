#!/bin/bash
# Round 4, visit Q (measurement only, no product change): rocprofv3 kernel trace of the DEFAULT bench configuration
# (8 worker streams x grouped searches of 4 batches) for tools/concurrency_trace.py -- how much of the wall clock has
# 0 / 1 / 2+ kernels resident, and which kernels run alone.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
(cd /tmp && rm -rf /tmp/ct && timeout 215 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o c -- python $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-roofline --latency-runs 0 > $R/gpurun_out/r4q_bench.log 2>&1)
f=$(find /tmp/ct -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && { wc -l "$f"; head -2 "$f" | cut -c1-400; gzip -c "$f" > gpurun_out/r4q_kernel_trace.csv.gz; ls -l gpurun_out/r4q_kernel_trace.csv.gz; }
tail -1 gpurun_out/r4q_bench.log | cut -c1-300
