#!/bin/bash
# The training configurations with EVERY timed step a graph replay (bench.py --no-prof: no per-launch events, so no eager
# steps inside the timed region) -- what a training run sees; the committed bench lines run every fourth step eagerly.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r5}_all_replay_lines.txt
echo "# python3 bench.py <flags> --steps 20 --warmup 3 --no-prof --no-cpu-baseline --no-extra --no-host-probe   (one box, back to back)" > $OUT
run() {
  local name=$1; shift
  python3 bench.py "$@" --steps 20 --warmup 3 --no-prof --no-cpu-baseline --no-extra --no-host-probe 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-28s %9.3f %s  %7.3f ms/step  graph replays in the timed steps: %s of %s' % ('$name', d['value'], d['unit'], d['ms_per_step'], c.get('hip_graph_replays_in_timed_steps'), d['steps']))" >> $OUT
}
run "configs[1] fp32"
run "configs[1] widths, bf16 mode" --dtype bf16
run "configs[2]" --workload full --batch 32 --dtype bf16
run "configs[3], one rank" --config 48k --workload full --dtype bf16
cat $OUT
