#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + the two HBM-traffic PMC passes (each in its own run, per
# MI355X_MICROARCH.md) of the bench workloads, an MFMA-busy PMC pass of the dominant kernels, raw CSVs under
# gpurun_out/<tag>_*/..., condensed by tools/profile_summary.py / tools/pmc_busy_summary.py into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/run_profiles.sh r6'
set -u
TAG=${1:-r6}
# every step under its own limit: SIGUSR1 first (bench.py dumps the Python stacks of all threads into the step's log),
# SIGKILL 20 s later -- one stuck pass must not eat the whole call
T="timeout -s USR1 -k 20"
keep() { cp profiles/${TAG}_* gpurun_out/ 2>/dev/null; }
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out profiles
run() {  # name, workload key, bench flags...
  local name=$1 key=$2; shift 2
  local out=gpurun_out/$name
  rm -rf $out
  $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-extra "$@" > $out.trace.log 2>&1
  # (counter passes: eager launches only -- the discriminator step's graph replay is switched off, same kernels)
  VCVITS_GRAPHS=0 $T 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py --steps ${PMC_STEPS:-2} --warmup ${PMC_WARMUP:-4} --no-cpu-baseline --no-extra --no-prof --no-host-probe "$@" > $out.fetch.log 2>&1
  VCVITS_GRAPHS=0 $T 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py --steps ${PMC_STEPS:-2} --warmup ${PMC_WARMUP:-4} --no-cpu-baseline --no-extra --no-prof --no-host-probe "$@" > $out.write.log 2>&1
  python3 tools/profile_summary.py $out profiles/$name $key > $out.summary.log 2>&1
  rm -rf $out/*/*/*.db $out/*/*_kernel_trace.csv $out/*/*/*_kernel_trace.csv $out/*/*counter_collection.csv $out/*/*/*counter_collection.csv
  keep
}
if [ -n "${ONLY:-}" ]; then
  case "$ONLY" in
    full_bf16) run ${TAG}_full_bf16 base/full/bf16 --workload full --batch 32 --dtype bf16 ;;
  esac
  $T 400 python3 bench.py --dtype bf16 --workload full --batch 32 --steps 10 --warmup 5 --no-cpu-baseline > profiles/${TAG}_bench_line_cfg2.json 2>/dev/null
  keep
  exit 0
fi
run ${TAG}_f32 base/vocoder/f32
run ${TAG}_bf16 base/vocoder/bf16 --dtype bf16
run ${TAG}_full_bf16 base/full/bf16 --workload full --batch 32 --dtype bf16
run ${TAG}_48k_full_bf16 48k/full/bf16 --config 48k --workload full --dtype bf16
run ${TAG}_48k_infer_f32 48k/infer/f32 --config 48k --workload infer
run ${TAG}_48k_infer_bf16 48k/infer/bf16 --config 48k --workload infer --dtype bf16
# MFMA pipe busy of the dominant kernels (period-discriminator layers) and of the fused attention kernels
for sp in 6 0; do
  rm -rf gpurun_out/${TAG}_busy$sp
  $T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_busy$sp -o m -- python3 tools/conv_layer_bench.py --reps 3 --only discP --split $sp > gpurun_out/${TAG}_busy$sp.log 2>&1
  python3 tools/pmc_busy_summary.py gpurun_out/${TAG}_busy$sp profiles/${TAG}_mfma_busy_split$sp.txt "python3 tools/conv_layer_bench.py --reps 3 --only discP --split $sp" > /dev/null
  rm -rf gpurun_out/${TAG}_busy$sp
done
# ... of the inference decode (bf16-io convs and the fused ResBlock pairs)
rm -rf gpurun_out/${TAG}_busyi
$T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_busyi -o m -- python3 bench.py --config 48k --workload infer --dtype bf16 --steps 1 --warmup 2 --no-cpu-baseline --no-extra --no-prof --no-host-probe > gpurun_out/${TAG}_busyi.log 2>&1
python3 tools/pmc_busy_summary.py gpurun_out/${TAG}_busyi profiles/${TAG}_infer_mfma_busy.txt "python3 bench.py --config 48k --workload infer --dtype bf16 --steps 1 --warmup 2 --no-prof" > /dev/null
rm -rf gpurun_out/${TAG}_busyi
rm -rf gpurun_out/${TAG}_busya
$T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_busya -o m -- python3 tools/attn_bench.py --reps 3 > gpurun_out/${TAG}_busya.log 2>&1
python3 tools/pmc_busy_summary.py gpurun_out/${TAG}_busya profiles/${TAG}_attn_mfma_busy.txt "python3 tools/attn_bench.py --reps 3" > /dev/null
rm -rf gpurun_out/${TAG}_busya
# per-layer table and the bench lines of every configuration (after the traffic files exist: the lines carry `traffic`)
$T 300 python3 tools/conv_layer_bench.py --reps 10 > profiles/${TAG}_conv_layers.txt 2>/dev/null
$T 300 python3 tools/conv_layer_bench.py --reps 10 --split 0 > profiles/${TAG}_conv_layers_fp32_mfma.txt 2>/dev/null
$T 300 python3 tools/attn_bench.py > profiles/${TAG}_attn_bench.txt 2>/dev/null
$T 300 python3 tools/thin_bench.py > profiles/${TAG}_thin_bench.txt 2>/dev/null
keep
# STFT launches: kernel-only durations (the segment launch is shorter than a host-side timing loop can resolve)
rm -rf gpurun_out/${TAG}_stft
( cd /tmp && $T 200 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_stft" -o stft -- python3 "$GRAFT_REPO_ROOT/tools/stft_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_stft.log" 2>&1 )
{ echo "# rocprofv3 --kernel-trace -- python3 tools/stft_bench.py: kernel-only durations by (kernel, grid); bytes: tools/stft_bench.py"; python3 tools/rocpd_kernel_times.py $(ls gpurun_out/${TAG}_stft/*.db gpurun_out/${TAG}_stft/*/*.db 2>/dev/null | head -1) stft; } > profiles/${TAG}_stft_kernels.txt 2>/dev/null
rm -rf gpurun_out/${TAG}_stft
$T 400 python3 bench.py --steps 10 --warmup 3 > profiles/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
$T 400 python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > profiles/${TAG}_bench_line_bf16.json 2>/dev/null
$T 400 python3 bench.py --dtype bf16 --workload full --batch 32 --steps 10 --warmup 5 --no-cpu-baseline > profiles/${TAG}_bench_line_cfg2.json 2>/dev/null
$T 400 python3 bench.py --dtype bf16 --config 48k --workload full --steps 10 --warmup 5 --no-cpu-baseline > profiles/${TAG}_bench_line_cfg3_1gpu.json 2>/dev/null
$T 400 python3 bench.py --dtype bf16 --workload full --batch 32 --varlen 64 --steps 16 --warmup 3 --no-cpu-baseline > profiles/${TAG}_bench_line_cfg2_varlen.json 2>/dev/null
$T 400 python3 bench.py --dtype bf16 --config 48k --workload infer --steps 5 --warmup 2 > profiles/${TAG}_bench_line_cfg4.json 2>/dev/null
$T 400 python3 bench.py --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline > profiles/${TAG}_bench_line_cfg4_f32.json 2>/dev/null
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
ls profiles/ | grep ${TAG}_ | head -80
