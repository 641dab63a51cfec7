#!/bin/bash
# N consecutive runs of the whole GPU suite on one box (verdict r4 #1: stability of the recorded-batch path), one summary line
# per run in gpurun_out/<tag>_suite_runs.txt; the first run also records the parity statistics the tests observe.
#   /usr/local/graft/bin/gpurun --timeout 4500 -- 'bash tools/suite_loop.sh 15 r5'
set -u
N=${1:-10}
TAG=${2:-r5}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/suite
OUT=gpurun_out/${TAG}_suite_runs.txt
echo "# $(date -u +%FT%TZ)  $N consecutive runs of: python -m pytest tests -x -q -m gpu   (one box, one process per run)" >> $OUT
rm -f gpurun_out/${TAG}_parity_stats_raw.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  if [ $i -eq 1 ]; then export VCVITS_PARITY_STATS=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_parity_stats_raw.txt; else unset VCVITS_PARITY_STATS; fi
  timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/suite/run_$i.log 2>&1
  rc=$?
  echo "run $i: rc=$rc $(tail -1 gpurun_out/suite/run_$i.log) [$(( $(date +%s) - t0 )) s wall]" >> $OUT
  if [ $rc -ne 0 ]; then tail -60 gpurun_out/suite/run_$i.log > gpurun_out/suite/FAILED_run_$i.txt; fi
done
unset VCVITS_PARITY_STATS
[ -s gpurun_out/${TAG}_parity_stats_raw.txt ] && python3 tools/parity_summary.py gpurun_out/${TAG}_parity_stats_raw.txt gpurun_out/${TAG}_parity_stats && rm -f gpurun_out/${TAG}_parity_stats_raw.txt
tail -$((N + 1)) $OUT
