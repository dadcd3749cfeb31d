"""Ray-parallel multi-GPU rendering: one process per GPU, rays of one view sharded by interleaved rows, one RCCL
all_gather of the rendered tiles (torch.distributed backend "nccl" is RCCL on ROCm; xGMI is point-to-point, and a
512x334 fp32 RGB image is 2 MB in total, so the collective is latency- not bandwidth-bound: one call per view).

The march itself needs no exchange: rays are independent.  Rows are interleaved
(row r -> rank r mod N) because the work per ray is uneven (rays that miss the hand's bounding box still march)."""
import torch


def shard_rows(height, world, rank):
    """(y0, y_step, n_rows) of this rank's rows; every rank gets the same number of rows (all_gather needs equal tiles)."""
    if height % world != 0:
        raise ValueError(f"image height {height} must be divisible by the number of ranks {world}")
    return rank, world, height // world


def deinterleave(gathered, height, width, world, channels=3):
    """all_gather output [rank][row_in_rank][x][c] -> image (height, width, c)."""
    return gathered.view(world, height // world, width, channels).permute(1, 0, 2, 3).reshape(height, width, channels)


def gather_image(tile, height, width, world, group=None):
    """tile: this rank's (rows*width, C) tensor -> full (height, width, C) image on every rank."""
    if world == 1:
        return tile.view(height, width, -1)
    import torch.distributed as dist
    full = torch.empty(world * tile.shape[0], tile.shape[1], dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(full, tile.contiguous(), group=group)
    return deinterleave(full, height, width, world, tile.shape[1])
