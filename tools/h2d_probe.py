"""Host -> device hand-over of one training batch (the LightningModule boundary takes host tensors from the DataLoader): bytes,
copy time from pinned memory, and the throughput with the copy counted SERIALLY in front of the step (worst case; Lightning's
loader overlaps it with the previous step).  python tools/h2d_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vcvits_amd import configs, synthetic  # noqa: E402


def measure(name, batch, step_ms, utt):
    dev = torch.device("cuda:0")
    host = {k: v.pin_memory() for k, v in batch.items() if torch.is_tensor(v)}
    nbytes = sum(v.numel() * v.element_size() for v in host.values())
    for _ in range(3):
        out = {k: v.to(dev, non_blocking=True) for k, v in host.items()}
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        out = {k: v.to(dev, non_blocking=True) for k, v in host.items()}
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del out
    print("%-34s %8.2f MB  copy %7.3f ms (%5.1f GB/s)  step %7.2f ms -> %7.1f utt/s resident, %7.1f with the copy in front (%.2f %%)" % (
        name, nbytes / 1e6, ms, nbytes / ms / 1e6, step_ms, utt / step_ms * 1e3, utt / (step_ms + ms) * 1e3, 100 * ms / step_ms))


def line(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    return d["ms_per_step"]


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = configs.base()
    m = cfg["model"]
    measure("configs[1] vocoder fp32 B=16", synthetic.vocoder_batch(16, m["inter_channels"]), line(root + "/profiles/r6_bench_line.json"), 16)
    measure("configs[2] full bf16 B=32", synthetic.full_batch(32, m["hubert_channels"]), line(root + "/profiles/r6_bench_line_cfg2.json"), 32)
    c48 = configs.base_48k()
    measure("configs[3] 48k full bf16 B=16", synthetic.full_batch(16, c48["model"]["hubert_channels"], hop=c48["data"]["hop_length"]),
            line(root + "/profiles/r6_bench_line_cfg3_1gpu.json"), 16)

if __name__ == "__main__":
    main()
