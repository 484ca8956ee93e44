"""Worker for tests/test_distributed_cpu.py (spawned once per rank)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, world, port, spp, steps, out_path, strong=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as orc
    pkg = orc.pkg
    from importlib import import_module
    mg = import_module("mcpt_amd.multigpu")
    scene = pkg.scenes.open_box(16, 16)
    o = orc.Oracle(scene, max_depth=4)
    film = torch.zeros(16 * 16 * 4, dtype=torch.float32)
    for s in range(steps):
        acc = np.zeros((16, 16, 4), np.float32)
        # weak: every rank renders spp samples per step; strong (bench.py's default): the step is ONE job of spp samples split over the ranks
        first, n = mg.sample_share(s, rank, world, spp) if strong else (mg.first_sample(s, rank, world, spp), spp)
        if n: o.render(n, seed=11, first_sample=first, accum=acc, threads=1)
        local = torch.from_numpy(acc.reshape(-1).copy())
        mg.all_reduce_film(local)
        film += local
    if rank == 0:
        np.save(out_path, film.numpy().reshape(16, 16, 4))
    dist.barrier()
    dist.destroy_process_group()
