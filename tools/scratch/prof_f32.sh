cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_f32/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/p_f32_bench.log 2>&1
