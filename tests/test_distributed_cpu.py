"""CPU test of the N > 1 path (gloo, world_size 2 and 4): sample-range sharding + one film all-reduce per step reproduces the
single-process render of the union of the sample ranges.  The renderer here is the CPU oracle (no GPU in this container);
bench.py runs the same two helpers (multigpu.first_sample / all_reduce_film) around the HIP renderer with backend nccl."""
import os
import socket

import numpy as np


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_ranks_equal_one_process(pkg, orc, tmp_path):
    import torch.multiprocessing as mp
    from tests import dist_worker
    world, spp, steps = 2, 3, 2
    out = str(tmp_path / "film.npy")
    mp.spawn(dist_worker.run, args=(world, _free_port(), spp, steps, out), nprocs=world, join=True)
    film = np.load(out)
    o = orc.Oracle(pkg.scenes.open_box(16, 16), max_depth=4)
    want, _, _ = o.render(world * spp * steps, seed=11, first_sample=0, threads=1)
    assert np.array_equal(film[..., 3], want[..., 3])                 # every pixel got world*spp*steps samples, counts reduce too
    assert np.allclose(film[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)   # same samples, different fp32 summation order


def test_two_ranks_split_one_job(pkg, orc, tmp_path):
    """bench.py's default since round 4 (strong scaling): every step is ONE job of job_spp samples per pixel and the ranks split its sample
    range (multigpu.sample_share; 5 samples over 2 ranks = 2 + 3) -- the reduced film is the single-process render of the same job."""
    import torch.multiprocessing as mp
    from tests import dist_worker
    world, job_spp, steps = 2, 5, 2
    out = str(tmp_path / "film.npy")
    mp.spawn(dist_worker.run, args=(world, _free_port(), job_spp, steps, out, True), nprocs=world, join=True)
    film = np.load(out)
    o = orc.Oracle(pkg.scenes.open_box(16, 16), max_depth=4)
    want, _, _ = o.render(job_spp * steps, seed=11, first_sample=0, threads=1)
    assert np.array_equal(film[..., 3], want[..., 3])
    assert np.allclose(film[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)


def test_four_ranks_split_a_job_smaller_than_the_world(pkg, orc, tmp_path):
    """A strong split with fewer samples than ranks (3 samples over 4 ranks: one rank renders nothing in a step and still takes part in the
    film all-reduce) -- what `bench.py --gpus 8` does to a small job; the reduced film is the single-process render of the job."""
    import torch.multiprocessing as mp
    from tests import dist_worker
    world, job_spp, steps = 4, 3, 2
    out = str(tmp_path / "film.npy")
    mp.spawn(dist_worker.run, args=(world, _free_port(), job_spp, steps, out, True), nprocs=world, join=True)
    film = np.load(out)
    o = orc.Oracle(pkg.scenes.open_box(16, 16), max_depth=4)
    want, _, _ = o.render(job_spp * steps, seed=11, first_sample=0, threads=1)
    assert np.array_equal(film[..., 3], want[..., 3])
    assert np.allclose(film[..., :3], want[..., :3], rtol=1e-5, atol=1e-6)


def test_sample_shares_cover_the_job_exactly(pkg):
    from importlib import import_module
    mg = import_module("mcpt_amd.multigpu")
    for world in (1, 2, 3, 4, 8):
        for job in (1, 5, 1024, 1000):
            seen = []
            for step in range(3):
                sizes = []
                for r in range(world):
                    f, n = mg.sample_share(step, r, world, job)
                    seen.extend(range(f, f + n)); sizes.append(n)
                assert max(sizes) - min(sizes) <= 1
            assert seen == list(range(3 * job))


def test_sample_ranges_are_disjoint_and_contiguous(pkg):
    from importlib import import_module
    mg = import_module("mcpt_amd.multigpu")
    for world in (1, 2, 4, 8):
        seen = []
        for step in range(3):
            for r in range(world):
                f = mg.first_sample(step, r, world, 1024)
                seen.extend(range(f, f + 1024, 512))
        assert len(seen) == len(set(seen)) and min(seen) == 0 and max(seen) == 3 * world * 1024 - 512
