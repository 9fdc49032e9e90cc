"""Image-level data parallelism: one process per GPU, every rank a full replica.

The path has no exchange step, so there is no collective on the data path.  The only communication is
ONE fixed-size gather of per-image result rows after the loop (RCCL `all_gather_into_tensor` when the
backend is nccl, i.e. over xGMI inside a node; gloo on CPU for the tests).  Rows are re-ordered by image
index and averaged in index order in float64, so the reported means are bit-identical for every world size.
"""
import math
import os

import numpy as np
import torch

ROW = 4   # [index, runtime_ms, psnr, ssim]


def init_from_env(use_cuda=True):
    """Returns (rank, world, local_rank).  Initialises torch.distributed only when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if use_cuda:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local_rank


def shard(n_items, rank, world):
    """Round-robin: image i -> rank i mod W (balances DIV2K's varying image sizes)."""
    return list(range(rank, n_items, world))


def gather_rows(local_rows, n_items, rank, world, device):
    """local_rows: list of (index, runtime_ms, psnr, ssim) of this rank.  Returns an [n_items, 4] float64
    array ordered by image index on every rank."""
    per = math.ceil(n_items / world) if n_items else 0
    buf = torch.full((max(per, 1), ROW), float("nan"), dtype=torch.float64)
    for k, row in enumerate(local_rows):
        buf[k] = torch.tensor(row, dtype=torch.float64)
    if world == 1:
        allrows = buf
    else:
        import torch.distributed as dist
        buf = buf.to(device)
        out = torch.empty((world * buf.shape[0], ROW), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, buf)
        allrows = out.cpu()
    a = allrows.numpy()
    a = a[~np.isnan(a[:, 0])]
    a = a[np.argsort(a[:, 0], kind="stable")]
    if len(a) != n_items or (n_items and not np.array_equal(a[:, 0], np.arange(n_items, dtype=np.float64))):
        raise RuntimeError(f"gather_rows: expected indices 0..{n_items - 1}, got {len(a)} rows")
    return a


def ordered_mean(values):
    """Sum in index order in float64 (python floats), like `sum(list)/len(list)` at test_demo.py:468-469."""
    s = 0.0
    for v in values:
        s += float(v)
    return s / len(values)
