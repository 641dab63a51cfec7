#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + the two PMC passes of the default bench (fp32, BASELINE configs[1])
# and of the bf16 variant, raw CSVs under gpurun_out/<tag>/..., then condensed by tools/profile_summary.py into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/run_profiles.sh r2'
set -u
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
run() {  # name, bench flags...
  local name=$1; shift
  local out=gpurun_out/$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out.trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" > $out.fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" > $out.write.log 2>&1
}
mkdir -p gpurun_out
run ${TAG}_f32
run ${TAG}_bf16 --dtype bf16
run ${TAG}_full_f32 --workload full
python3 bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_line_bf16.json 2>/dev/null
python3 bench.py --dtype bf16 --workload full --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_line_cfg3.json 2>/dev/null
python3 bench.py --dtype bf16 --config 48k --workload full --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_line_cfg4_1gpu.json 2>/dev/null
python3 bench.py --dtype bf16 --config 48k --workload infer --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_line_cfg5.json 2>/dev/null
python3 bench.py --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_line_cfg5_f32.json 2>/dev/null
ls -la gpurun_out/${TAG}_f32/*/ gpurun_out/${TAG}_bf16/*/ 2>/dev/null | head -40
