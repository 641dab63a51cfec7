cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out profiles
rm -rf gpurun_out/r5_busyi
timeout -s USR1 -k 20 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r5_busyi -o m -- python3 bench.py --config 48k --workload infer --dtype bf16 --steps 1 --warmup 2 --no-cpu-baseline --no-extra --no-prof --no-host-probe > gpurun_out/r5_busyi.log 2>&1
python3 tools/pmc_busy_summary.py gpurun_out/r5_busyi gpurun_out/r5_infer_mfma_busy.txt "python3 bench.py --config 48k --workload infer --dtype bf16 --steps 1 --warmup 2 --no-prof"
rm -rf gpurun_out/r5_busyi
