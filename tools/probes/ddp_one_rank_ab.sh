# One-rank A/B of the data-parallel forms of the training batch (forced 1-rank RCCL group: VCVITS_FORCE_DDP=1), run on the GPU box:
#   bash tools/probes/ddp_one_rank_ab.sh [bench flags, e.g. --config 48k --workload full --dtype bf16]
# Lines: eager without / with the bucket machinery, the recorded batch under DDP (three segments), the same with ONE discriminator stream, and the single-process recorded batch with one stream.
# (--no-prof: every timed step takes the same form; bench.py records the batch in its setup phase, before the warm-up steps)
cd "$GRAFT_REPO_ROOT"
B="bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-extra --no-prof $*"
i=0
for env in "VCVITS_BATCH_GRAPHS=0" "VCVITS_BATCH_GRAPHS=0 VCVITS_FORCE_DDP=1" "VCVITS_FORCE_DDP=1" "VCVITS_FORCE_DDP=1 VCVITS_STREAMS=1" "VCVITS_STREAMS=1"; do
  i=$((i+1))
  env $env timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29520+i)) $B > gpurun_out/ddp_ab_$i.json 2> gpurun_out/ddp_ab_$i.err
  python3 -c "
import json;d=json.loads(open('gpurun_out/ddp_ab_$i.json').read().strip().splitlines()[-1]);c=d['config'];print('$env |',d['value'],d['unit'],d['ms_per_step'],'ms/step; replays',c.get('hip_graph_replays_in_timed_steps'),'of',d['steps'],'; setup steps',c.get('graph_recording_steps_before_warmup'),'; host issue ms (replay / eager)',c.get('host_issue_ms_graph_replay_step'),c.get('host_issue_ms_eager_step'))" || tail -3 gpurun_out/ddp_ab_$i.err
done
