"""bench.py's own N-rank launcher (`--gpus N` without torch.distributed.run), rehearsed on CPU: `--backend gloo --dry` goes through
the same spawn -> rendezvous -> sample-range sharding -> film all-reduce -> max-over-ranks clock as the GPU run, renders nothing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_launches_that_many_ranks():
    p = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "1", "--spp", "8"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                   # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry"] is True
    assert out["count_plane_ok"] is True                               # the reduced count plane holds the JOB's spp: the ranks split it (strong scaling, the default)
    assert out["scaling"] == "strong" and out["config"]["spp_per_rank"] == 4
    assert out["first_samples_rank0"] == [0, 8, 16]                    # rank 0 of steps 0, 1, 2: every step is one 8-spp job
    p = _run(["--gpus", "2", "--backend", "gloo", "--dry", "--steps", "2", "--warmup", "1", "--spp", "8", "--weak"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "weak" and out["count_plane_ok"] is True and out["first_samples_rank0"] == [0, 16, 32]   # rounds 1-3's mode: 8 spp per rank


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "4", "--backend", "gloo", "--dry"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_under_an_external_launcher_no_second_spawn():
    """With RANK / WORLD_SIZE already set (torch.distributed.run, the driver's way) bench.py must not spawn again."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry", "--steps", "1", "--warmup", "0"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    lines = [l for o in outs for l in o[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_tile_shard_helper_partitions_tiles(pkg):
    from importlib import import_module
    mg = import_module("mcpt_amd.multigpu")
    for world in (1, 2, 8):
        seen = set()
        for rank in range(world):
            mod, rem = mg.tile_shard(rank, world)
            mine = {t for t in range(1000) if t % mod == rem}
            assert not (seen & mine)
            seen |= mine
        assert seen == set(range(1000))
