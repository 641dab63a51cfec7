#!/bin/bash
# kernel-only durations of the thin convolutions (tools/thin_bench.py under rocprofv3 --kernel-trace, aggregated by grid),
# for each value of the conv_c1_fwd channel-chunk switch:  gpurun -- 'bash tools/probes/thin_trace.sh "0 64 16 8"'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/thin
for ch in ${1:-0}; do
  rm -rf gpurun_out/thin/t_$ch
  VCVITS_TUNING=c1_chunk=$ch timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/thin/t_$ch -o t -- python3 tools/thin_bench.py --reps 5 > gpurun_out/thin/bench_$ch.log 2>&1
  f=$(find gpurun_out/thin/t_$ch -name 't_kernel_trace.csv' | head -1)
  echo "== VCVITS_TUNING=c1_chunk=$ch"
  python3 tools/trace_by_grid.py "$f" 1 80 | grep -v "at::\|elementwise\|distribution\|copyBuffer\|fillBuffer" > gpurun_out/thin/by_grid_$ch.txt
  grep "c1_\|total" gpurun_out/thin/by_grid_$ch.txt | cut -c1-175
  rm -rf gpurun_out/thin/t_$ch
done
