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


SPARSE_MAX_FRACTION = 0.05      # "auto": the record exchange is used while fewer than 5 % of the cells are touched on every rank


def touched_records(private_grid, value_dtype=torch.float32):
    """(cell int32 [n], delta value_dtype [n][C]) of the grid's non-zero cells (a cell = one [C] row of [H][W][C]): what a rank has to
    say about the shared grid.  One pass over the private grid (160 MB at HBM rate for 2000 x 2000 x 5 float64: ~50 us)."""
    flat = private_grid.reshape(-1, private_grid.shape[-1])
    cells = torch.nonzero((flat != 0).any(dim=1)).squeeze(1)
    return cells.to(torch.int32), flat.index_select(0, cells).to(value_dtype)


def reduce_grids_sparse(private_grid, group=None, value_dtype=torch.float32, records=None):
    """The shared grid from an ALL-GATHER of every rank's (cell, delta[C]) records instead of an all-reduce of the dense grid (SURVEY 8e:
    a camera frustum touches ~1 % of a 2000 x 2000 grid -- config C: 26 k cells x (4 + 5 x 4) B = 0.6 MB per GPU against 80 MB dense in
    float32).  Record lists are padded to the longest rank's length (one small all-gather of the counts first), gathered, and summed into
    a zero grid of the private grid's dtype in rank order -- for identity-CM grids (small integers) exactly the dense result, for log-CM
    grids the float32 rounding of each rank's addend, as the dense float32 exchange.  Returns (grid, bytes this rank sent)."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    cells, vals = records if records is not None else touched_records(private_grid, value_dtype)
    C_ = private_grid.shape[-1]
    total = torch.zeros_like(private_grid)
    flat = total.reshape(-1, C_)
    if world == 1:
        flat.index_add_(0, cells.long(), vals.to(total.dtype))
        return total, 0
    n = torch.tensor([cells.numel()], dtype=torch.int64, device=private_grid.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    nmax = int(max(int(c.item()) for c in counts))
    pc = torch.zeros(nmax, dtype=torch.int32, device=private_grid.device)
    pv = torch.zeros((nmax, C_), dtype=value_dtype, device=private_grid.device)
    pc[:cells.numel()] = cells
    pv[:cells.numel()] = vals
    all_c = [torch.empty_like(pc) for _ in range(world)]
    all_v = [torch.empty_like(pv) for _ in range(world)]
    dist.all_gather(all_c, pc, group=group)
    dist.all_gather(all_v, pv, group=group)
    for r in range(world):
        k = int(counts[r].item())
        flat.index_add_(0, all_c[r][:k].long(), all_v[r][:k].to(total.dtype))
    return total, int(nmax * (4 + C_ * pv.element_size()))


def reduce_grids_auto(private_grid, group=None, exchange_dtype=torch.float32, max_fraction=SPARSE_MAX_FRACTION):
    """Sparse record exchange when EVERY rank touched fewer than `max_fraction` of the cells, the dense all-reduce otherwise (the choice
    must be the same on all ranks: it is made on the all-reduced maximum of the touched counts).  Returns (grid, mode, bytes sent)."""
    cells, vals = touched_records(private_grid, exchange_dtype)
    n = torch.tensor([cells.numel()], dtype=torch.int64, device=private_grid.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    ncells = private_grid.numel() // private_grid.shape[-1]
    if int(n.item()) < max_fraction * ncells:
        total, sent = reduce_grids_sparse(private_grid, group=group, value_dtype=exchange_dtype, records=(cells, vals))
        return total, "sparse", sent
    total = reduce_grids(private_grid, group=group, exchange_dtype=exchange_dtype)
    return total.to(private_grid.dtype), "dense", int(total.numel() * total.element_size())
