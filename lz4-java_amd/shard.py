"""Multi-GPU sharding of a batch of independent LZ4 blocks.

Every block is self-contained (the reference's one-shot calls never use a dictionary and its frame
streams enforce block independence, LZ4FrameOutputStream.java:361-363), so a batch shards by
contiguous block-index ranges with NO data exchange: rank r owns [r*n/W, (r+1)*n/W).  The only
collective is the all-gather of the int32 per-block result sizes (4 bytes per block) -- RCCL over
xGMI under the "nccl" backend on GPUs, gloo in the CPU test-suite.  One process per GPU.
"""
import torch
import torch.distributed as dist


def block_range(n_blocks, world, rank):
    """contiguous, balanced: sizes differ by at most one block"""
    return (n_blocks * rank) // world, (n_blocks * (rank + 1)) // world


def gather_sizes(local_sizes, n_blocks, group=None):
    """all ranks' int32 per-block sizes, concatenated in block order (length n_blocks).
    `local_sizes` is this rank's slice, on the device the process group expects."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_sizes
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-n_blocks // world)  # ceil: equal-sized contributions for all_gather_into_tensor
    b0, b1 = block_range(n_blocks, world, rank)
    assert local_sizes.numel() == b1 - b0 and local_sizes.dtype == torch.int32
    padded = torch.zeros(per, dtype=torch.int32, device=local_sizes.device)
    padded[: b1 - b0] = local_sizes
    out = torch.empty(per * world, dtype=torch.int32, device=local_sizes.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    parts = []
    for r in range(world):
        r0, r1 = block_range(n_blocks, world, r)
        parts.append(out[r * per: r * per + (r1 - r0)])
    return torch.cat(parts)


def compress_sharded(codec, n_blocks, group=None):
    """Runs `codec(b0, b1) -> int32 tensor of sizes for blocks [b0, b1)` on this rank's range and
    returns (my_range, all_sizes).  On a GPU rank `codec` launches DeviceBatch.compress_fast on the
    rank's slice; the function itself is device-agnostic."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    b0, b1 = block_range(n_blocks, world, rank)
    local = codec(b0, b1)
    return (b0, b1), gather_sizes(local, n_blocks, group)
