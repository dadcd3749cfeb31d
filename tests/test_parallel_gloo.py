"""N > 1 path on CPU: two gloo ranks shard the rows of one view (interleaved), all_gather their tiles and rebuild the image.
The march itself has no collective (rays are independent); what is covered here is the sharding arithmetic and the gather layout
that bench.py --gpus N and vanerf_amd.parallel use with RCCL on the GPU node."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vanerf_amd.parallel import deinterleave, gather_image, shard_rows

H, W = 16, 6


def _image():
    return torch.arange(H * W * 3, dtype=torch.float32).view(H, W, 3)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    y0, ystep, ny = shard_rows(H, world, rank)
    rows = torch.arange(ny) * ystep + y0
    tile = _image()[rows].reshape(-1, 3)  # what this rank would have rendered: (rows * W, 3)
    full = gather_image(tile, H, W, world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks timing reduction
    q.put((rank, torch.equal(full, _image()), float(t)))
    dist.destroy_process_group()


def test_two_rank_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, 2.0), (1, True, 2.0)]


def test_shard_rows_cover_image_once():
    for world in (1, 2, 4, 8):
        seen = torch.zeros(512, dtype=torch.int32)
        for rank in range(world):
            y0, step, ny = shard_rows(512, world, rank)
            seen[torch.arange(ny) * step + y0] += 1
        assert (seen == 1).all()
    tiles = [(_image()[torch.arange(H // 4) * 4 + r]).reshape(-1, 3) for r in range(4)]
    assert torch.equal(deinterleave(torch.cat(tiles, 0), H, W, 4), _image())


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vanerf_amd.parallel import all_reduce_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))  # identical on both ranks
    net[2].weight.requires_grad_(True)
    x = torch.full((4, 7), float(rank + 1))
    net[:2](x).sum().backward()  # the last layer gets no gradient on any rank, like the IBR head at one source view
    calls = all_reduce_gradients(net.parameters(), world, bucket_bytes=64)  # tiny buckets: several collectives
    q.put((rank, calls, [p.grad.clone() for p in net.parameters()]))
    dist.destroy_process_group()


def test_two_rank_gradient_average():
    """all_reduce_gradients: bucketed all-reduce == the mean of the per-rank gradients, same result and same number of calls on every rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    want = None
    for r in range(2):
        net.zero_grad()
        net[:2](torch.full((4, 7), float(r + 1))).sum().backward()
        g = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in net.parameters()]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    want = [w / 2 for w in want]
    assert res[0][1] == res[1][1] and res[0][1] > 1
    for rank_res in res:
        for got, w in zip(rank_res[2], want):
            assert torch.allclose(got, w, atol=1e-6)
