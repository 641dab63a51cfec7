cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc1/a -o a -- python3 tools/conv_layer_bench.py --reps 3 --only "discP2.conv4" > gpurun_out/pmc1_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc1/b -o b -- python3 tools/conv_layer_bench.py --reps 3 --only "discP2.conv4" > gpurun_out/pmc1_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc1/c -o c -- python3 tools/conv_layer_bench.py --reps 3 --only "discP2.conv4" > gpurun_out/pmc1_c.log 2>&1
ls gpurun_out/pmc1/*/
