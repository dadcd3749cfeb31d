"""Ray-parallel multi-GPU rendering: one process per GPU, rays of one view sharded by interleaved blocks of rows, one RCCL
all_gather of the rendered tiles (torch.distributed backend "nccl" is RCCL on ROCm; xGMI is point-to-point, and a
512x334 fp32 RGB image is 2 MB in total, so the collective is latency- not bandwidth-bound: one call per view).

The march itself needs no exchange: rays are independent.  Rows are dealt out in blocks of 8 (block b ->
rank b mod N): interleaved because the work per ray is uneven (the top and bottom of the view are mostly samples that miss the source
image), in blocks of 8 because the mesh query works on 8x8 pixel tiles of neighbouring rays -- with single interleaved rows a tile
spans 8 N image rows and prunes worse (measured on one GPU, N = 8: 4.36 ms per rank against 3.23 ms for an eighth of the view)."""
import torch


ROW_BLOCK = 8


def padded_height(height, world):
    """Rows that are actually rendered: the image height rounded up to whole 8-row blocks per rank (all_gather needs equal tiles).  The extra
    rows (at most 8 * world - 1, below the image) are marched like any other -- their rays are ordinary rays of the same camera -- and cut
    off again by deinterleave()."""
    per = ROW_BLOCK * world
    return (height + per - 1) // per * per


def shard_rows(height, world, rank):
    """(y0, y_step, n_rows, y_block) of this rank's rows: row k of the rank is image row y0 + (k // y_block) * y_step + k % y_block.
    Every rank gets the same number of rows; for heights that are not a multiple of 8 * world the last blocks reach below the image."""
    return rank * ROW_BLOCK, world * ROW_BLOCK, padded_height(height, world) // world, ROW_BLOCK


def rank_rows(height, world, rank):
    """Rows of `rank`, in the order it renders them (long tensor; values >= height are the padding rows)."""
    y0, y_step, ny, yb = shard_rows(height, world, rank)
    k = torch.arange(ny)
    return y0 + (k // yb) * y_step + k % yb


def deinterleave(gathered, height, width, world, channels=3):
    """all_gather output [rank][row_in_rank][x][c] -> image (height, width, c) (inverse of shard_rows; the padding rows are dropped)."""
    hp = padded_height(height, world)
    g = gathered.view(world, hp // (world * ROW_BLOCK), ROW_BLOCK, width, channels)          # [rank][block][row in block]
    return g.permute(1, 0, 2, 3, 4).reshape(hp, width, channels)[:height]


def deal_blocks(costs, world):
    """Blocks of 8 rows dealt to the ranks by COST instead of round robin: `costs` (n_blocks,) -- e.g. every block's ray / valid-sample count in
    the previous frame of an orbit, identical on all ranks -- -> assign (world, per) long tensor of block indices, per = ceil(n_blocks / world):
    every rank gets the SAME number of blocks (the all-gather needs equal tiles; indices >= n_blocks are padding blocks below the image) and
    the largest load is as small as a greedy deal makes it: blocks in order of falling cost, each to the least loaded rank that still has
    room (ties by rank: deterministic, so every rank computes the same table).  Within a rank the blocks are kept in image order."""
    costs = torch.as_tensor(costs, dtype=torch.float64).reshape(-1).cpu()
    nb = costs.numel()
    per = (nb + world - 1) // world
    order = sorted(range(nb), key=lambda b: (-float(costs[b]), b))
    load, rooms, mine = [0.0] * world, [per] * world, [[] for _ in range(world)]
    for b in order:
        r = min((r for r in range(world) if rooms[r] > 0), key=lambda r: (load[r], r))
        mine[r].append(b)
        load[r] += float(costs[b])
        rooms[r] -= 1
    pad = nb
    for r in range(world):
        while len(mine[r]) < per:
            mine[r].append(pad)
            pad += 1
        mine[r].sort()
    return torch.tensor(mine, dtype=torch.long)


def block_rows(assign, rank):
    """First image row of each of `rank`'s blocks (int32, for renderer.render_pass(row_blocks=...)) -- padding blocks start below the image."""
    return (assign[rank] * ROW_BLOCK).to(torch.int32)


def deinterleave_blocks(gathered, assign, height, width, channels=3):
    """all_gather output [rank][block of the rank][row in block][x][c] -> image (height, width, c) for a deal_blocks table."""
    world, per = assign.shape
    g = gathered.view(world * per, ROW_BLOCK, width, channels)
    n_slots = world * per
    full = torch.empty(n_slots, ROW_BLOCK, width, channels, dtype=gathered.dtype, device=gathered.device)
    full[assign.reshape(-1).to(gathered.device)] = g
    return full.view(n_slots * ROW_BLOCK, width, channels)[:height]


def gather_image(tile, height, width, world, group=None):
    """tile: this rank's (rows*width, C) tensor -> full (height, width, C) image on every rank."""
    if world == 1:
        return tile.view(height, width, -1)
    import torch.distributed as dist
    full = torch.empty(world * tile.shape[0], tile.shape[1], dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(full, tile.contiguous(), group=group)
    return deinterleave(full, height, width, world, tile.shape[1])


def all_reduce_gradients(parameters, world, bucket_bytes=32 << 20, group=None):
    """Data-parallel training (BASELINE config 5: one process per GPU, each with its own source frame / patch): average the gradients
    over the ranks after backward().  Gradients are flattened into a few large buckets -- xGMI is point-to-point, a ring all-reduce is
    bound per link, so few large calls beat one call per parameter (the renderer's hot parameters are ~100 small tensors; the two
    encoders ~200 more) -- reduced with one all_reduce each and copied back.  Parameters without a gradient on this rank (e.g. the
    IBR head at one source view) take part as zeros so that every rank issues the same collectives."""
    if world == 1:
        return 0
    import torch.distributed as dist
    params = [p for p in parameters if p.requires_grad]
    calls, bucket, size = 0, [], 0

    def flush():
        nonlocal calls, bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, group=group)
        flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        calls += 1
        bucket, size = [], 0

    for p in params:
        nbytes = p.numel() * p.element_size()
        if bucket and (size + nbytes > bucket_bytes or p.dtype != bucket[0].dtype):
            flush()
        bucket.append(p)
        size += nbytes
    flush()
    return calls
