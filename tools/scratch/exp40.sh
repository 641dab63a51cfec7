for i in 1 2; do
for e in "VCVITS_PK_NO_X4=1" "XX=1"; do
  env $e python bench.py --dtype bf16 --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e bf16', d['value'], d['ms_per_step'], r['achieved'])"
  env $e python bench.py --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e f32', d['value'], d['ms_per_step'], r['achieved'])"
done; done
