"""Multi-GPU: frame-parallel sharding and the one exchange step.

A frame's contribution to the grid is an additive delta that never reads the grid
(src/mapping.py:424,437), so N ranks can each map a disjoint set of frames (one camera stream per
GPU) into a private grid and the shared grid is the element-wise sum: ONE all-reduce (or reduce to a
root) over RCCL.  Nothing else on the path communicates; segmentation replicas need no exchange.
"""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world_size):
    """Indices of the frames rank `rank` maps: stream k goes to rank k % world_size."""
    return list(range(rank, n_frames, world_size))


def reduce_grids(private_grid, group=None, dst=None, inplace=False, exchange_dtype=None):
    """Sum of every rank's private grid.  Returns a new tensor unless ``inplace`` (the private grid
    normally keeps accumulating).  With ``dst`` the sum is only valid on that rank.  Works on CUDA
    tensors (backend nccl = RCCL over xGMI) and CPU tensors (gloo, used by the CPU tests).
    ``exchange_dtype`` (e.g. torch.float32): the copy that travels is cast to it first -- 80 MB instead of 160 MB
    for the 2000 x 2000 x 5 grid; identity-CM grids are small integers and stay exact, log-CM grids
    round at 6e-8 relative.  The result has that dtype."""
    if exchange_dtype is not None and exchange_dtype != private_grid.dtype:
        assert not inplace, "an exchange copy of another dtype cannot be in place"
        total = private_grid.to(exchange_dtype)
    else:
        total = private_grid if inplace else private_grid.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dst is None:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(total, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return total
