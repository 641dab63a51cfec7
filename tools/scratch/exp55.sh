for i in 1 2; do for e in "VCVITS_PK_NO_VEC=1" "XX=1"; do env $e python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['share_of_step_time'], [(o['kernel'][:10], o['achieved']) for o in r['other_kernels']][-1])"; done; done
