cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_bf16/trace -o t -- python3 bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/p_bf16_bench.log 2>&1
ls gpurun_out/p_bf16/trace/*/ 2>/dev/null | head
