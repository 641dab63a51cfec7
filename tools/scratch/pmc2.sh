cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc2/a -o a -- python3 tools/conv_layer_bench.py --dtype bf16 --reps 3 --only "gen.res c128 k11 d1" > gpurun_out/pmc2_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc2/b -o b -- python3 tools/conv_layer_bench.py --dtype bf16 --reps 3 --only "gen.res c128 k11 d1" > gpurun_out/pmc2_b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc2/c -o c -- python3 tools/conv_layer_bench.py --dtype bf16 --reps 3 --only "gen.res c128 k11 d1" > gpurun_out/pmc2_c.log 2>&1
