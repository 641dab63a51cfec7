set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab2
timeout 900 python -m pytest tests/test_premask_chain_gpu.py tests/test_training_step_gpu.py tests/test_full_width_step_gpu.py tests/test_determinism_gpu.py tests/test_bf16_step_gpu.py tests/test_bench_sizes_gpu.py -x -q -m gpu > gpurun_out/ab2/tests.log 2>&1; tail -3 gpurun_out/ab2/tests.log
for cfgname in cfg3 base cfg2; do
  case $cfgname in
    cfg3) FL="--config 48k --workload full --dtype bf16" ;;
    base) FL="" ;;
    cfg2) FL="--workload full --batch 32 --dtype bf16" ;;
  esac
  for v in on off on2; do
    if [ $v = off ]; then export VCVITS_PREMASK=0; else unset VCVITS_PREMASK; fi
    timeout 300 python3 bench.py $FL --steps 12 --warmup 5 --no-cpu-baseline --no-extra --no-prof --no-host-probe > gpurun_out/ab2/${cfgname}_$v.json 2> gpurun_out/ab2/${cfgname}_$v.err
    python3 -c "import json;d=json.loads(open('gpurun_out/ab2/${cfgname}_$v.json').read().strip().splitlines()[-1]);print('$cfgname $v',d['value'],d['ms_per_step'])"
  done
done
unset VCVITS_PREMASK
