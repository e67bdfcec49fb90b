"""The N > 1 path on CPU: 2 processes, gloo, frames sharded per rank, private grids summed by the same
reduce_grids() the GPU path uses.  The private grids are produced by the oracle here (there is no GPU
in this container); what is under test is the sharding + the exchange step."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _frames(n):
    sys.path.insert(0, ROOT)
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    cam = mo.camera_matrices(1)
    H, W = 240, 320
    K = cam["K"].copy()
    K[0] *= W / 1920.0
    K[1] *= H / 1440.0
    P = K @ np.concatenate([cam["R"], cam["t"]], axis=1)
    out = []
    for k in range(n):
        rng = np.random.default_rng(100 + k)
        out.append((syn.make_cloud(rng, 3000, K, cam["R"], cam["t"], W, H), syn.colorize(syn.make_label_map(rng, H, W))))
    return out, P


def _cfg(cm):
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    return dict(range_max=100.0, boundary=syn.centred_boundary(mo.PCD_ORIGIN_OFFSET[:2], 60.0), resolution=0.5,
                label_names=mo.LABELS_NAMES, label_colors=mo.LABEL_COLORS, confusion_matrix=cm, use_pcd_intensity=True)


def _worker(rank, world, port, cm_kind, q):
    try:
        _worker_body(rank, world, port, cm_kind, q)
    except Exception as e:            # the parent must not wait for a result that never comes
        import traceback
        q.put(("error", "rank %d: %s" % (rank, traceback.format_exc(limit=3))))
        raise


def _worker_body(rank, world, port, cm_kind, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.distributed import reduce_grids, reduce_grids_auto, reduce_grids_sparse, shard_frames
    frames, P = _frames(6)
    cm = np.eye(5) if cm_kind == "eye" else syn.log_confusion(5)
    cfg = _cfg(cm)
    grid = np.zeros((240, 240, 5))
    mine = shard_frames(len(frames), rank, world)
    for k in mine:
        mo.mapping_frame(grid, frames[k][0], "velodyne", frames[k][1], None, P, cfg)
    private = torch.from_numpy(grid)
    total = reduce_grids(private)
    assert torch.equal(private, torch.from_numpy(grid))          # the private grid is left alone
    root_only = reduce_grids(private, dst=0)
    # the float32 exchange copy (what bench.py --gpus N sends: half the payload)
    total32 = reduce_grids(private, exchange_dtype=torch.float32)
    assert total32.dtype == torch.float32 and torch.equal(private, torch.from_numpy(grid))
    # round 5: the record exchange -- all-gather of (cell, delta[C]) -- gives the same grid
    sparse, sent = reduce_grids_sparse(private)
    assert sparse.dtype == private.dtype and torch.equal(private, torch.from_numpy(grid))
    # the ranks' float32 addends summed in the grid's own float64: identity-CM grids (small integers) equal the dense result exactly,
    # log-CM grids differ from the float32 all-reduce by its one rounding of the SUM (the record exchange is the more accurate one)
    if cm_kind == "eye":
        assert torch.equal(sparse, total32.to(torch.float64)) and torch.equal(sparse, total)
    else:
        assert float((sparse - total).abs().max()) <= 2e-7 * float(total.abs().max())
    touched = int((grid != 0).any(axis=2).sum())
    assert 0 < sent < 0.2 * total32.numel() * 4 and sent >= touched * 24
    auto, used, _ = reduce_grids_auto(private, max_fraction=0.5)
    assert used == "sparse" and torch.equal(auto, sparse)
    auto_d, used_d, sent_d = reduce_grids_auto(private, max_fraction=1e-6)
    assert used_d == "dense" and torch.equal(auto_d, total32.to(torch.float64)) and sent_d == total32.numel() * 4
    assert float((auto_d - sparse).abs().max()) <= 2e-7 * max(1.0, float(total.abs().max()))
    if rank == 0:
        assert torch.equal(root_only, total)
        q.put(("ok", (mine, total.numpy(), total32.numpy())))
    dist.barrier()
    dist.destroy_process_group()


def _run(cm_kind, port=None):
    import queue
    import socket

    import pytest
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:            # a free port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cm_kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        status, payload = q.get(timeout=240)
    except queue.Empty:
        status, payload = "error", "no result within 240 s (exit codes %r)" % [p.exitcode for p in procs]
    if status != "ok":
        for p in procs:
            p.terminate()
        pytest.fail(payload)
    mine, total, total32 = payload
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # float32 exchange: exact for the integer (identity-CM) grids, one float32 rounding per rank's addend otherwise
    assert total32.dtype == np.float32
    assert np.max(np.abs(total32.astype(np.float64) - total)) <= 2e-7 * max(1.0, float(np.abs(total).max()))
    if cm_kind == "eye":
        assert np.array_equal(total32.astype(np.float64), total)
    return mine, total


def test_sharding_covers_every_frame_once():
    from vision_semantic_segmentation_amd.distributed import shard_frames
    for world in (1, 2, 3, 8):
        got = sorted(sum((shard_frames(13, r, world) for r in range(world)), []))
        assert got == list(range(13))


def test_two_ranks_identity_cm_sum_is_exact():
    from oracle import mapping_oracle as mo
    mine, total = _run("eye", 29611)
    assert mine == [0, 2, 4]
    frames, P = _frames(6)
    seq = np.zeros((240, 240, 5))
    for pcd, img in frames:
        mo.mapping_frame(seq, pcd, "velodyne", img, None, P, _cfg(np.eye(5)))
    assert np.array_equal(total, seq)                               # integers: any summation order is exact


def test_two_ranks_log_cm_within_tolerance():
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    _, total = _run("log", 29612)
    frames, P = _frames(6)
    seq = np.zeros((240, 240, 5))
    for pcd, img in frames:
        mo.mapping_frame(seq, pcd, "velodyne", img, None, P, _cfg(syn.log_confusion(5)))
    assert np.max(np.abs(total - seq)) <= 1e-9                      # re-association only; bar is 1e-3
