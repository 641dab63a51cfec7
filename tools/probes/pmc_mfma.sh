cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/mfma -o m -- python3 tools/conv_layer_bench.py --reps 3 --only discP > gpurun_out/mfma.log 2>&1
