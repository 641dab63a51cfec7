for i in 1 2; do for e in "VCVITS_PK_NO_VEC=1" "XX=1"; do env $e python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e', d['value'], d['ms_per_step'], r['achieved'], r['share_of_step_time'], r['launches_per_step'], r['avg_launch_us'])"; done; done
