set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab1
timeout 600 python -m pytest tests/test_bf16_gpu.py tests/test_bf16_step_gpu.py tests/test_elementwise_gpu.py tests/test_conv_x3_gpu.py -x -q -m gpu > gpurun_out/ab1/tests.log 2>&1; tail -3 gpurun_out/ab1/tests.log
B="python3 bench.py --config 48k --workload full --dtype bf16 --steps 12 --warmup 5 --no-cpu-baseline --no-extra --no-prof --no-host-probe"
for v in new fin_old agb_old new2; do
  case $v in
    fin_old) export VCVITS_TUNING=wgrad_finish_vec=0 ;;
    agb_old) export VCVITS_TUNING=act_grad_vec=0 ;;
    *) unset VCVITS_TUNING ;;
  esac
  timeout 300 $B > gpurun_out/ab1/$v.json 2> gpurun_out/ab1/$v.err
  python3 -c "import json;d=json.loads(open('gpurun_out/ab1/$v.json').read().strip().splitlines()[-1]);print('$v',d['value'],d['ms_per_step'])"
done
unset VCVITS_TUNING
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab1/trace -o t -- python3 bench.py --config 48k --workload full --dtype bf16 --steps 4 --warmup 4 --no-cpu-baseline --no-extra > gpurun_out/ab1/trace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ab1/trace/**/t_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows:
    if 'finish' in r['Name'] or 'act_grad' in r['Name']:
        print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, '%.2f%%'%(100*float(r['TotalDurationNs'])/tot))
PY
rm -rf gpurun_out/ab1/trace
