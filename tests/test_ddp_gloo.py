"""CPU, world_size 2, gloo: the data-parallel gradient path of FlatAdamW (bucket hooks fired from
backward, async all-reduce per contiguous bucket, averaging) and the rank-strided sharding of
synthetic utterances.  The HIP step itself needs a GPU; here the optimizer step is replaced by
an SGD stand-in on the averaged flat gradient, which is what DDP must deliver."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd.light.optim import FlatAdamW
    torch.manual_seed(0)  # same init on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1))
    opt = FlatAdamW(net.parameters(), 1e-2, bucket_mb=0.0003)  # ~79 floats per bucket -> several buckets
    assert len(opt._buckets) >= 2
    opt.broadcast_parameters()
    # every rank sees a different shard of the "utterances"
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 8, generator=g)
    y = torch.randn(5, 1, generator=g)
    opt.zero_grad()
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()  # hooks launch the bucket all-reduces
    opt.finish_grad_sync()
    out[rank] = (opt.grad.clone(), [p.grad.data_ptr() == opt.grad[o:o + p.numel()].data_ptr()
                                    for p, o in zip(opt.params, opt.offsets)])
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_global_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, views0 = out[0]
    g1, views1 = out[1]
    assert all(views0) and all(views1)  # .grad stayed a view of the flat buffer through backward
    assert torch.allclose(g0, g1, atol=0, rtol=0)  # both ranks hold the same averaged gradient
    # reference: gradient of the mean loss over the union of both shards
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1))
    xs, ys = [], []
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        xs.append(torch.randn(5, 8, generator=g))
        ys.append(torch.randn(5, 1, generator=g))
    loss = ((net(torch.cat(xs)) - torch.cat(ys)) ** 2).mean()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in reversed(list(net.parameters()))])
    assert torch.allclose(g0, ref, atol=1e-6, rtol=1e-5)
