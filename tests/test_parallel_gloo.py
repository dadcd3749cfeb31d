"""N > 1 path on CPU: two gloo ranks shard the rows of one view (interleaved), all_gather their tiles and rebuild the image.
The march itself has no collective (rays are independent); what is covered here is the sharding arithmetic and the gather layout
that bench.py --gpus N and vanerf_amd.parallel use with RCCL on the GPU node."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vanerf_amd.parallel import deinterleave, gather_image, rank_rows, shard_rows

H, W = 16, 6


def _image():
    return torch.arange(H * W * 3, dtype=torch.float32).view(H, W, 3)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = rank_rows(H, world, rank)
    tile = _image()[rows].reshape(-1, 3)  # what this rank would have rendered: (rows * W, 3)
    full = gather_image(tile, H, W, world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks timing reduction
    q.put((rank, torch.equal(full, _image()), float(t)))
    dist.destroy_process_group()


def _run_ranks(target, world=2):
    """Spawns `world` gloo ranks on a just-probed free port, ONCE: a hang, a crash or a wrong result of any rank fails the test (a retry
    loop here used to hide the first two failures)."""
    import queue as _queue
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=180) for _ in procs]
    except _queue.Empty:
        res = None
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert res is not None, "a gloo rank did not report within 180 s"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return sorted(res, key=lambda t: t[0])


def test_two_rank_gather():
    assert _run_ranks(_worker) == [(0, True, 2.0), (1, True, 2.0)]


def test_shard_rows_cover_image_once():
    for height in (512, 16, 12, 500, 334, 7):  # heights that are not a multiple of 8 * world are padded below the image
        for world in (1, 2, 3, 4, 8):
            seen = torch.zeros(height, dtype=torch.int32)
            sizes = set()
            for rank in range(world):
                y0, step, ny, yb = shard_rows(height, world, rank)
                assert yb == 8 and ny % 8 == 0
                sizes.add(ny)
                rows = rank_rows(height, world, rank)
                seen[rows[rows < height]] += 1
            assert (seen == 1).all() and len(sizes) == 1  # every image row once, equal tiles
    # blocks of 8 rows stay together (8x8 pixel tiles of the mesh query), dealt round robin
    assert rank_rows(512, 8, 3)[:10].tolist() == [24, 25, 26, 27, 28, 29, 30, 31, 88, 89]
    for hh, world in ((H, 2), (12, 4), (64, 4), (500, 8), (30, 3)):
        hp = max(int(rank_rows(hh, world, r).max()) for r in range(world)) + 1
        img = torch.arange(hp * W * 3, dtype=torch.float32).view(hp, W, 3)  # the padding rows hold something too
        tiles = [img[rank_rows(hh, world, r)].reshape(-1, 3) for r in range(world)]
        assert torch.equal(deinterleave(torch.cat(tiles, 0), hh, W, world), img[:hh])


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
    # plain lists, not tensors: a tensor in a queue travels as a file descriptor served by THIS process, which may be gone before the parent asks
    q.put((rank, calls, [p.grad.reshape(-1).tolist() for p in net.parameters()]))
    dist.destroy_process_group()


def test_two_rank_gradient_average():
    """all_reduce_gradients: bucketed all-reduce == the mean of the per-rank gradients, same result and same number of calls on every rank."""
    res = _run_ranks(_grad_worker)
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
            assert torch.allclose(torch.tensor(got), w.reshape(-1), atol=1e-6)


H4 = 64


def _worker4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img = torch.arange(H4 * W * 3, dtype=torch.float32).view(H4, W, 3)
    y0, step, ny, yb = shard_rows(H4, world, rank)
    tile = img[rank_rows(H4, world, rank)].reshape(-1, 3)
    full = gather_image(tile, H4, W, world)
    q.put((rank, torch.equal(full, img), (y0, step, ny, yb)))
    dist.destroy_process_group()


def test_four_rank_gather_of_row_blocks():
    """The layout bench.py --gpus 4 uses: 8-row blocks dealt round robin (block b -> rank b mod 4), one all_gather, de-interleaved by blocks."""
    res = _run_ranks(_worker4, world=4)
    assert [r[1] for r in res] == [True] * 4
    assert [r[2] for r in res] == [(8 * r, 32, 16, 8) for r in range(4)]


def _worker_dealt(rank, world, port, q):
    """8-row blocks dealt by cost (parallel.deal_blocks): every rank computes the same table, gathers equal tiles, rebuilds the image."""
    from vanerf_amd.parallel import block_rows, deal_blocks, deinterleave_blocks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hh, ww = 8 * 13, 5  # 13 blocks over 4 ranks: three padding blocks below the image
    img = torch.arange(hh * ww * 3, dtype=torch.float32).view(hh, ww, 3)
    costs = torch.tensor([1, 1, 5, 9, 9, 5, 2, 1, 1, 3, 7, 2, 1], dtype=torch.float64)
    assign = deal_blocks(costs, world)
    first = block_rows(assign, rank)
    rows = (first.long()[:, None] + torch.arange(8)[None]).reshape(-1)
    padded = torch.cat([img, torch.zeros(8 * int(assign.max() + 1) - hh, ww, 3)], 0)  # rows below the image are rendered like any other
    tile = padded[rows].reshape(-1, 3)
    parts = [torch.empty_like(tile) for _ in range(world)]
    dist.all_gather(parts, tile)
    full = deinterleave_blocks(torch.cat(parts, 0), assign, hh, ww)
    q.put((rank, torch.equal(full, img), assign.tolist()))
    dist.destroy_process_group()


def test_blocks_dealt_by_cost():
    """deal_blocks: every block exactly once, the same number of blocks per rank (equal all-gather tiles), a balanced load, the same table on
    every rank; the gathered tiles rebuild the image (four gloo ranks)."""
    from vanerf_amd.parallel import deal_blocks
    costs = torch.tensor([1, 1, 5, 9, 9, 5, 2, 1, 1, 3, 7, 2, 1], dtype=torch.float64)
    for world in (1, 2, 4, 8):
        a = deal_blocks(costs, world)
        per = (13 + world - 1) // world
        assert a.shape == (world, per)
        assert sorted(a.reshape(-1).tolist()) == list(range(world * per))            # every block (and padding block) exactly once
        loads = [sum(float(costs[b]) for b in r if b < 13) for r in a.tolist()]
        assert max(loads) <= sum(loads) / world + float(costs.max())                 # greedy bound
        assert all(r == sorted(r) for r in a.tolist())                               # image order inside a rank
    # an uneven view: round robin puts the expensive middle rows on few ranks, the deal does not
    uneven = torch.tensor([10.0, 1.0, 1.0, 1.0] * 2 + [1.0] * 8)  # (round robin gives rank 0 both expensive blocks)
    a = deal_blocks(uneven, 4)
    dealt = max(sum(float(uneven[b]) for b in r) for r in a.tolist())
    rr = max(sum(float(uneven[b]) for b in range(16) if b % 4 == r) for r in range(4))
    assert dealt < rr
    res = _run_ranks(_worker_dealt, world=4)
    assert [r[1] for r in res] == [True] * 4 and all(r[2] == res[0][2] for r in res)
