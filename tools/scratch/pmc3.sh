cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc3/a -o a -- python3 tools/conv_layer_bench.py --reps 3 --only "discP2.conv4" > gpurun_out/pmc3_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc3/b -o b -- python3 tools/conv_layer_bench.py --reps 3 --only "discP2.conv4" > gpurun_out/pmc3_b.log 2>&1
