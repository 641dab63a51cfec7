cd "$GRAFT_REPO_ROOT"
B="bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-extra --no-prof --no-host-probe"
i=0
for env in "VCVITS_BATCH_GRAPHS=0" "VCVITS_BATCH_GRAPHS=0 VCVITS_FORCE_DDP=1" "VCVITS_FORCE_DDP=1" "VCVITS_FORCE_DDP=1 VCVITS_DDP_GRAPH_LINEAR=0"; do
  i=$((i+1))
  env $env python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29520+i)) $B > gpurun_out/ddp_ab_$i.json 2> gpurun_out/ddp_ab_$i.err
  python3 -c "
import json;d=json.loads(open('gpurun_out/ddp_ab_$i.json').read().strip().splitlines()[-1]);c=d['config'];print('$env |',d['value'],d['ms_per_step'],'replays',c.get('hip_graph_replays_in_timed_steps'),'wall',c.get('host_wall_ms_per_step_in_timed_region'))"
done
