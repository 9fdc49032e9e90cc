"""One rank of the harness under `python -m torch.distributed.run` on CPU (gloo): what tests/test_harness.py launches to cover the
N > 1 path end to end exactly as a multi-GPU job is started (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the launcher's
environment).  The pinned oracle port stands in for the GPU module -- host logic, sharding and the result gather are the product's.
usage: _harness_rank.py <out.json> <save_dir>"""
import json
import logging
import os
import sys
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLD, load_sd_torch  # noqa: E402
from ntire2022_esr_amd import dist as D  # noqa: E402
from ntire2022_esr_amd import harness as H  # noqa: E402


class OracleModel:
    def __init__(self):
        from oracle import torch_port as TP
        self.sd, self.f = load_sd_torch("imdn_baseline"), TP.imdn

    def __call__(self, x):
        with torch.no_grad():
            return self.f(self.sd, x)


def main():
    out, save_dir = sys.argv[1:3]
    torch.set_num_threads(2)
    rank, world, _ = D.init_from_env(use_cuda=False)
    args = types.SimpleNamespace(data_dir=os.path.join(GOLD, "mini_div2k"), save_dir=save_dir, rank=rank, world=world)
    pairs = H.select_dataset(args.data_dir, "valid")[:3]
    res = H.run(OracleModel(), "imdn", 1.0, None, logging.getLogger(f"t{rank}"), torch.device("cpu"), args, mode="valid", pairs=pairs)
    if rank == 0:
        res["_world"] = world
        json.dump(res, open(out, "w"))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
