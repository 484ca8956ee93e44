#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing hot path.

Metric (BASELINE.json): Mray/s (primary + continuation + shadow rays actually traversed).  Default workload = configs[1]:
S-cornell, the synthetic stand-in for cornell-box (the cg24 scene files are not in the reference repo), 800x800, 1024 spp,
depth 8.  `--config c3|c4|c5` selects the other BASELINE.json configurations (veach-mis 1280x720, bathroom2 1920x1080,
bathroom2 3840x2160 depth 16 with >= 4 M triangles); their spp per step is reduced (named in config.workload) unless --spp
says otherwise -- they are parity / roofline cases, the headline number is c2.

One "step" = one complete render job of the scene already resident in HBM: clear the film, one mcpt_render call (a stream of
[shade, trace] kernel launches over the HBM path pool), and -- with N > 1 ranks -- one RCCL all-reduce of the fp32
films inside the timed region, the path's only exchange step.  The JOB is fixed (the config's spp per step: cornell-box 1024 spp) and
N ranks split it => "scaling": "strong" (round 4; the metric is quoted on a fixed job).  Default split = sample ranges: rank r renders
its contiguous share of the step's sample indices for every pixel.  `--shard tiles` is BASELINE.json's "pixel-tile shard" instead: rank r
renders ALL samples of the 8x8 pixel tiles t with t % N == r (mcpt_render_tiles); the films are disjoint and the same all-reduce assembles
the image.  `--weak` keeps rounds 1-3's mode: every rank renders the config's spp per step (work per GPU fixed => "scaling": "weak").

`--gpus N` without a launcher (WORLD_SIZE unset) starts N worker processes itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_* set, 127.0.0.1 rendezvous) BEFORE anything touches the GPU or imports torch; under torch.distributed.run the
environment is taken as given.  `n_gpus` in the line is the size of the process group that actually ran.

Prints ONE JSON line on rank 0.  `roofline` is computed from device counters (algorithmic bytes) and the
HIP-event kernel time the library records around launches of the timed region; `cpu_baseline` times the REAL reference
(oracle/_ref, built from /root/reference by oracle/build_ref.sh) -- or, for the multi-million-triangle configs whose OBJ text the
reference's regex parser would take minutes to read, the oracle restatement -- on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0            # aggregate L2 bandwidth (same guide): the relevant ceiling while the scene is cache-resident
L2_BYTES, MALL_BYTES = 32 << 20, 256 << 20   # aggregate L2, Infinity Cache (same guide): where the traversal data (8-wide nodes + triangle records) can live
# algorithmic bytes per unit of work with THIS build's layouts
B_BOX, B_TRI = 10, 48            # an eighth of an 80-B eight-child node per child-box test; one 48-B {v0,e1,e2} record per triangle test
B_RAY = 32 + 16                  # trace kernel: ray fetch (origin + direction records) + 16-B result write-back per ray

CONFIGS = {
    # name: scene generator, its arguments, resolution, the config's spp, spp per bench step, depth, BASELINE.json wording
    "c2": dict(scene="cornell-box", kw={}, res=(800, 800), spp=1024, step_spp=1024, depth=8,
               label="S-cornell (synthetic cornell-box.obj stand-in, 39612 tris) 800x800, depth=8"),
    "c3": dict(scene="veach-mis", kw={}, res=(1280, 720), spp=2048, step_spp=256, depth=0,
               label="S-veach (synthetic veach-mis.obj stand-in, 3840 light tris) 1280x720, unbounded depth + RR"),
    "c4": dict(scene="bathroom2", kw={"detail": 160}, res=(1920, 1080), spp=4096, step_spp=128, depth=0,
               label="S-bath (synthetic bathroom2.obj stand-in, 0.59 M tris, 4 textures, mirror) 1920x1080, unbounded depth + RR"),
    "c5": dict(scene="bathroom2", kw={"detail": 420}, res=(3840, 2160), spp=16384, step_spp=32, depth=16,
               label="S-bath stress (synthetic bathroom2.obj stand-in, 4.05 M tris) 3840x2160, depth=16"),
}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(n: int, argv) -> int:
    """Start n worker processes of this script (one per GPU) and wait for them.  Called before torch is imported: the parent
    never touches the GPU, and nothing that has is ever exec'ed."""
    port = str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "MCPT_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # wait for all of them; a rank that dies (before or at the rendezvous, say) takes the others with it instead of leaving them blocked in
    # init_process_group, and the whole launch is bounded
    deadline = time.time() + float(os.environ.get("MCPT_BENCH_LAUNCH_TIMEOUT", "3000"))
    rc = 0
    while procs:
        alive = []
        for p in procs:
            r = p.poll()
            if r is None: alive.append(p)
            else: rc = max(rc, abs(r))
        procs = alive
        if procs and (rc != 0 or time.time() > deadline):
            for p in procs: p.terminate()
            for p in procs:
                try: p.wait(timeout=20)
                except subprocess.TimeoutExpired: p.kill()
            return rc or 124
        time.sleep(0.2)
    return rc


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"): return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def omp_threads(n):
    """Thread count of the OpenMP runtime the reference / oracle libraries already loaded (libgomp is one per process)."""
    import ctypes
    try: ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError: pass


def cpu_baseline(pkg, cfg, scene, rays_per_path_ref, budget_s=20.0):
    """Time the reference's Render::render on the host cores over a bounded sample of the same workload: all cores of the job's share,
    then one thread (SURVEY section 8d asks for both, with core count and CPU model)."""
    # the GPU box exposes every host core in the affinity mask but a 1-GPU job's CPU share is 16 cores; the reference's
    # OpenMP loop also serialises on one shared mt19937 (utils.h:23-28), so more threads than that only add contention
    ncores = min(16, len(os.sched_getaffinity(0)))
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    W, H = cfg["res"]; depth = cfg["depth"]
    note = ("rays = paths x %.3f rays/path (GPU counters incl. self-shadowed light samples, which the reference traverses); "
            "compare mpath_per_s with the GPU line's mpath_per_s for the like-for-like pair" % rays_per_path_ref)
    try:
        if scene.n_faces > 200_000:
            raise orc.ReferenceUnavailable("scene too large for the reference's OBJ parser within the bench budget")
        ref = orc.Reference(depth_variant=True)
        tmp = tempfile.mkdtemp(prefix="mcpt_bench_")
        obj = scene.write(tmp)
        sys.stdout.flush()
        saved = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)   # the reference prints "[Model] <path>" (model.cpp:46)
        try:
            ref.load(obj)
        finally:
            os.dup2(saved, 1); os.close(saved); os.close(devnull)
        ref.set_max_bounces(depth)
        ref.stream_mode()
        t1 = ref.render(1)                                   # one frame = one spp for all pixels
        frames = max(1, min(64, int(budget_s / max(t1, 1e-3)) - 1))
        t = ref.render(frames)
        paths = frames * W * H
        omp_threads(1)
        f1 = max(1, min(frames, int(6.0 / max(t1 * ncores / 4.0, 1e-3))))       # ~6 s of single-thread work (the loop scales ~4x on 16 threads, not 16x)
        tt1 = ref.render(f1)
        omp_threads(ncores)
        return {"value": round(paths * rays_per_path_ref / t / 1e6, 4), "unit": "Mray/s", "cores": ncores, "kind": "reference", "cpu_model": cpu_model(),
                "threads_1": {"value": round(f1 * W * H * rays_per_path_ref / tt1 / 1e6, 4), "mpath_per_s": round(f1 * W * H / tt1 / 1e6, 4), "frames": f1, "seconds": round(tt1, 2)},
                "sample": "%d frame(s) (=spp) of %dx%d %s depth %d through the real reference's Render::render (OpenMP, %d threads, "
                          "%.2f s); %s" % (frames, W, H, scene.name, depth, ncores, t, note),
                "mpath_per_s": round(paths / t / 1e6, 4)}
    except orc.ReferenceUnavailable as why:
        w, h = max(16, W // 8), max(16, H // 8)
        o = orc.Oracle(scene.with_resolution(w, h), max_depth=depth)
        _, c, t1 = o.render(1, seed=1)
        spp = max(1, min(64, int(budget_s / max(t1, 1e-3))))
        _, c, t = o.render(spp, seed=2)
        rays = c["rays_primary"] + c["rays_continuation"] + c["rays_shadow"]
        omp_threads(1)
        # ~6 s of single-thread work (the oracle's OpenMP loop scales almost linearly: r03's / 8 guess cost 92 s on c5); where even ONE sample of the
        # sampled film would take longer (c5: 33 s), the one-thread leg renders a quarter of its pixels
        pred1 = t1 * ncores * 0.9
        o1, w1, h1 = o, w, h
        if pred1 > 8.0:
            w1, h1 = max(16, w // 2), max(16, h // 2)
            o1 = orc.Oracle(scene.with_resolution(w1, h1), max_depth=depth); pred1 /= 4.0
        s1 = max(1, min(spp, int(6.0 / max(pred1, 1e-3))))
        _, c1, tt1 = o1.render(s1, seed=3)
        omp_threads(ncores)
        rays1 = c1["rays_primary"] + c1["rays_continuation"] + c1["rays_shadow"]
        return {"value": round(rays / t / 1e6, 4), "unit": "Mray/s", "cores": ncores, "kind": "port", "cpu_model": cpu_model(),
                "threads_1": {"value": round(rays1 / tt1 / 1e6, 4), "mpath_per_s": round(c1["paths"] / tt1 / 1e6, 4), "spp": s1, "film": "%dx%d" % (w1, h1), "seconds": round(tt1, 2)},
                "sample": "%dx%dx%d spp of %s depth %d through oracle/mcpt_oracle.cpp (OpenMP, %d threads, %.2f s) [%s]" % (
                    w, h, spp, scene.name, depth, ncores, t, why),
                "mpath_per_s": round(c["paths"] / t / 1e6, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json configuration (default c2 = the headline)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per step")
    ap.add_argument("--shard", choices=["samples", "tiles"], default="samples",
                    help="N > 1: each rank renders its share of the step's sample range for every pixel (default) or its interleaved share of the "
                         "8x8 pixel tiles for all samples (BASELINE.json's 'pixel-tile shard'); either way the job is fixed (strong scaling)")
    ap.add_argument("--weak", action="store_true", help="N > 1, --shard samples: every rank renders the config's spp per step (rounds 1-3's mode; weak scaling)")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="N",
                    help="one GPU renders one rank's share of an N-way strong-scaled split of the config's job (--shard samples: its share of the samples "
                         "of every pixel; --shard tiles: all samples of every N-th 8x8 tile): per-rank time of an N-GPU run without the N GPUs "
                         "(tools/scaling_emulation.py turns the sweep into profiles/r04_scaling_emulation.json)")
    ap.add_argument("--emulate-rank", default="0", metavar="R|all", help="which rank of the emulated split to render; 'all' = every rank in turn, "
                    "ms_per_step = the slowest (what an N-GPU step would take), per-rank times in emulated_ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--dry", action="store_true", help="launcher / collective rehearsal without a GPU: no rendering, films are synthetic")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))           # the parent does no GPU work and imports no torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)

    import numpy as np
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    mg = __import__("importlib").import_module("mcpt_amd.multigpu")
    cfg = CONFIGS[args.config]
    W, H = cfg["res"]; depth = cfg["depth"]
    spp = args.spp or cfg["step_spp"]

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
    n_ranks = dist.get_world_size() if dist is not None else 1

    if args.dry:
        # Rehearsal of everything around the renderer (launcher, rendezvous, sample-range sharding, film all-reduce, the
        # max-over-ranks clock) on CPU tensors.  It renders nothing and reports no throughput.
        film = torch.zeros(H * W * 4, dtype=torch.float32)
        seen = []
        weak = args.weak and args.shard == "samples"
        t0 = time.perf_counter()
        for s in range(args.warmup + args.steps):
            film.zero_()
            if weak: first, n = mg.first_sample(s, rank, world, spp), spp
            else: first, n = mg.sample_share(s, rank, world, spp)
            film.view(-1, 4)[:, 3] = float(n)                        # what a render of n samples leaves in the count plane
            mg.all_reduce_film(film)
            seen.append(first)
        dt = time.perf_counter() - t0
        ok = bool((film.view(-1, 4)[:, 3] == float(spp * n_ranks if weak else spp)).all())
        t_all = torch.tensor([dt], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "Mray/s (primary+secondary), %s %dspp" % (cfg["scene"], cfg["spp"]), "value": 0.0, "unit": "Mray/s", "dry": True,
                              "n_gpus": n_ranks, "rccl_ranks": n_ranks, "backend": args.backend, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(float(t_all.item()) / max(1, args.steps) * 1e3, 3), "count_plane_ok": ok,
                              "first_samples_rank0": seen, "scaling": "weak" if weak else "strong",
                              "config": {"workload": cfg["label"], "spp_per_step": spp, "spp_per_rank": spp if weak else mg.sample_share(0, rank, world, spp)[1]}}), flush=True)
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # HIP events around both kernels of every 8th pipeline iteration, recorded on the launch stream inside the timed region (every
    # iteration costs ~3 % of the step in event packets; the sampled mean agrees with rocprofv3's all-launch average, see profiles/)
    os.environ.setdefault("MCPT_TIME_KERNELS", "8")
    scene = pkg.scenes.SCENES[cfg["scene"]](W, H, **cfg["kw"])
    r = pkg.Renderer(scene, max_depth=depth, device=local)
    info = r.info()
    accum = torch.zeros(H * W * 4, dtype=torch.float32, device=dev)      # torch lends memory + stream + RCCL
    r.bind_accum(accum.data_ptr())
    # ONE real side stream carries the whole step -- film clear, the render's fork/join, the RCCL all-reduce -- so consecutive steps
    # are stream-ordered (the library's sub-pipeline streams fork from and join this one).  torch's default stream has handle 0,
    # which mcpt_set_stream reads as "the context's own stream": never pass that by accident.
    side = torch.cuda.Stream(device=dev)
    r.set_torch_stream(side)

    emu = args.emulate_world if world == 1 else 0
    job_spp = spp                           # samples per pixel of ONE step's job (the config's spp per step)
    split = emu if emu > 1 else n_ranks     # ranks the job is divided over
    strong = not (args.weak and args.shard == "samples")

    def share(rk):                          # (first sample offset inside a step, samples) of rank rk under the sample-range split
        if not strong: return rk * job_spp, job_spp
        return mg.sample_share(0, rk, split, job_spp)

    def step(s, rk):                        # one step = one complete render job: clear the film, render rank rk's share, sum the films
        step_base = s * job_spp * (1 if strong else split)
        with torch.cuda.stream(side):
            accum.zero_()
            if args.shard == "tiles":
                r.render_tiles(job_spp, 20251004, step_base, *mg.tile_shard(rk, split))
            else:
                lo, n = share(rk)
                if n: r.render(n, seed=20251004, first_sample=step_base + lo)
            mg.all_reduce_film(accum)       # RCCL sum over xGMI, ordered after the render on `side` (no-op for one rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(rk):
        for s in range(args.warmup):
            step(s, rk)
        fence()
        r.reset_counters()
        fence()
        t0 = time.perf_counter()
        for s in range(args.steps):
            step(args.warmup + s, rk)
        fence()
        return time.perf_counter() - t0, r.counters()

    emu_ranks = None
    if emu > 1 and args.emulate_rank == "all":        # every rank of the split in turn, on this one GPU; the slowest is what a step would take
        emu_ranks = []
        dt, c = 0.0, None
        for rk in range(emu):
            dt_k, c_k = timed(rk)
            emu_ranks.append({"rank": rk, "ms_per_step": round(dt_k / args.steps * 1e3, 3), "rays": int(c_k.rays)})
            if dt_k > dt: dt, c = dt_k, c_k
        my_rank = max(range(emu), key=lambda k: emu_ranks[k]["ms_per_step"])
    else:
        my_rank = int(args.emulate_rank) if emu > 1 else rank
        dt, c = timed(my_rank)
    spp = job_spp if args.shard == "tiles" else share(my_rank)[1]          # samples per pixel this rank renders per step
    want_count = float(job_spp if strong else job_spp * n_ranks)
    if emu > 1 and args.shard == "tiles":   # one rank of an emulated split owns every emu-th tile only
        cnt = accum.view(H, W, 4)[:, :, 3]
        film_ok = bool(((cnt == 0) | (cnt == float(job_spp))).all().item()) and abs(float((cnt > 0).float().mean().item()) - 1.0 / emu) < 0.02
    elif emu > 1:
        film_ok = bool((accum.view(-1, 4)[:, 3] == float(share(emu - 1 if emu_ranks else my_rank)[1])).all().item())
    else:
        film_ok = bool((accum.view(-1, 4)[:, 3] == want_count).all().item())   # every pixel got all the samples of the last step, from every rank
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    rays = torch.tensor([float(c.rays)], dtype=torch.float64, device=dev)
    paths = torch.tensor([float(c.paths)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX); dist.all_reduce(rays); dist.all_reduce(paths)
    dt = float(t_all.item()); total_rays = float(rays.item()); total_paths = float(paths.item())

    out = None
    if rank == 0:
        # ---- kernel durations: the library brackets the launches of every MCPT_TIME_KERNELS-th iteration of the timed region with HIP
        # events recorded on the launch stream (a sub-pipeline stream forked from `side`); *_ms_total = sampled mean x launches
        # since reset_counters().  Dominant kernel = wf_trace8_kernel (BVH traversal); one launch of it per pipeline iteration.
        launches = max(1, c.iterations)
        trace_ms = c.trace_ms_total / launches
        shade_ms = c.shade_ms_total / launches
        rays_per_launch = c.rays / launches
        # ---- algorithmic bytes per ray from an instrumented pass (same scene / depth / seed, a few spp)
        ri = pkg.Renderer(scene, max_depth=depth, device=local, flags=pkg.FLAG_COUNT_TRAVERSAL)
        ri.render(32 if W * H <= 1 << 20 else 4, seed=20251004); ci = ri.counters(); ri.close()
        trav_bytes_per_ray = (B_BOX * ci.box_tests + B_TRI * ci.tri_tests) / max(1, ci.rays)
        trace_bytes_per_ray = trav_bytes_per_ray + B_RAY
        algo_bytes = trace_bytes_per_ray * rays_per_launch
        achieved = algo_bytes / (trace_ms * 1e-3) / 1e9
        scene_bytes = int(info.traversal_bytes)                       # wide nodes + triangle test records, from the library
        resident = scene_bytes <= L2_BYTES
        residency = "L2" if scene_bytes <= L2_BYTES else "Infinity Cache (MALL)" if scene_bytes <= MALL_BYTES else "HBM"
        # ---- measured HBM traffic / VALU issue / lane utilisation: PMC counters cannot be read inside this process; they come from the
        # committed rocprofv3 --pmc passes of the same workload and build (tools/r03_profile.sh -> profiles/r03_traffic.json), per traced ray,
        # x this run's rays per launch / rays per second
        traffic = None; traffic_source = None; pmc = {}
        tpath = next((q for q in (os.path.join(ROOT, "profiles", "r04_traffic.json"), os.path.join(ROOT, "profiles", "r03_traffic.json")) if os.path.exists(q)), None)
        tname = "profiles/" + os.path.basename(tpath) if tpath else None
        if tpath:
            pmc = json.load(open(tpath)).get(args.config, {})
            if pmc.get("wf_trace_kernel_hbm_bytes_per_ray"):
                traffic = int(pmc["wf_trace_kernel_hbm_bytes_per_ray"] * rays_per_launch)
                traffic_source = "%s[%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (not this run), bytes per ray x this run's rays per launch" % (tname, args.config)
        rays_per_s = total_rays / dt / max(1, n_ranks)                     # this GPU's share of the job's ray rate
        hbm_measured = None
        if pmc.get("wf_trace_kernel_hbm_bytes_per_ray") and pmc.get("wf_shade_kernel_hbm_bytes_per_ray"):
            tb, sb = pmc["wf_trace_kernel_hbm_bytes_per_ray"], pmc["wf_shade_kernel_hbm_bytes_per_ray"]
            hbm_measured = {"level": "step (bytes per traced ray from the PMC passes of this build x this run's rays/s per GPU)",
                            "trace_bytes_per_ray": tb, "shade_bytes_per_ray": sb,
                            "trace_GBps": round(tb * rays_per_s / 1e9, 1), "shade_GBps": round(sb * rays_per_s / 1e9, 1),
                            "total_GBps": round((tb + sb) * rays_per_s / 1e9, 1), "frac_of_hbm_peak": round((tb + sb) * rays_per_s / 1e9 / HBM_PEAK_GBS, 4),
                            "fetch_correction": pmc.get("fetch_correction_note"),
                            "source": "%s[%s]" % (tname, args.config)}
        n_streams = 2 if (c.shade_ms_total > 0 and launches >= 2) else 1
        per_stream_ms = launches / max(1, args.steps) / n_streams * (trace_ms + shade_ms)
        frac_alg = round(achieved / HBM_PEAK_GBS, 4)
        # what binds, decided from the counters the line carries (not from the scene's size): the memory system when the measured HBM-side traffic is
        # at least half of the peak; else VALU issue when the two kernels keep the SIMDs' vector ALUs busy most of the time (SQ_ACTIVE_INST_VALU over
        # the SIMD-cycles of the render: `valu_busy_frac`); else latency / occupancy
        hbm_frac = hbm_measured["frac_of_hbm_peak"] if hbm_measured else None
        valu_busy = pmc.get("valu_busy_frac", pmc.get("valu_issue_frac"))
        bound = "hbm" if (hbm_frac is not None and hbm_frac >= 0.5) else "valu-issue" if (valu_busy is not None and valu_busy >= 0.6) else "latency" if (hbm_frac is not None or valu_busy is not None) else ("valu-issue" if resident else "hbm")
        roofline = {"bound": bound, "bound_rule": "hbm if measured HBM-side traffic >= 0.5 of peak, else valu-issue if valu_busy_frac >= 0.6, else latency (counters: %s)" % tname,
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": frac_alg, "frac_algorithmic": frac_alg,
                    "frac_note": "ALGORITHMIC bytes of the dominant kernel / its launch time / HBM peak (the contract's definition); the bytes the "
                                 "memory system really moved are in hbm_measured and traffic",
                    "traffic": traffic, "traffic_source": traffic_source, "hbm_measured": hbm_measured,
                    "kernel": "wf_trace8_kernel<false>", "kernel_ms": round(trace_ms, 4), "launches_per_step": round(launches / max(1, args.steps), 1),
                    "rays_per_launch": int(rays_per_launch),
                    "algorithmic_bytes_per_ray": round(trace_bytes_per_ray, 1),
                    "box_tests_per_ray": round(ci.box_tests / max(1, ci.rays), 2), "tri_tests_per_ray": round(ci.tri_tests / max(1, ci.rays), 2),
                    "traversal_data_bytes": int(scene_bytes), "cache_resident": resident, "traversal_data_lives_in": residency,
                    "l2_relative": {"traversal_GBps": round(trav_bytes_per_ray * rays_per_launch / (trace_ms * 1e-3) / 1e9, 1), "l2_peak_GBps": L2_PEAK_GBS,
                                    "frac": round(trav_bytes_per_ray * rays_per_launch / (trace_ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4)},
                    "valu_busy_frac": valu_busy, "valu_issue_note": pmc.get("valu_issue_note"),
                    "valu_lane_utilisation": pmc.get("wf_trace_kernel_valu_lane_utilisation"),
                    "salu_to_valu_instructions": pmc.get("wf_trace_kernel_salu_to_valu"),
                    "inner_steps_per_ray": pmc.get("wf_trace_kernel_inner_steps_per_ray"),
                    "second_kernel": {"kernel": "wf_shade_kernel<false>", "kernel_ms": round(shade_ms, 4),
                                      "valu_lane_utilisation": pmc.get("wf_shade_kernel_valu_lane_utilisation")},
                    "streams": n_streams,
                    "per_stream_ms_per_step": round(per_stream_ms, 2),
                    "reconciliation": "each of the %d sub-pipeline streams runs launches_per_step / %d x (trace %.4f + shade %.4f ms) = %.1f ms of "
                                      "back-to-back kernels per step; ms_per_step = %.1f (the two streams overlap: a kernel's duration is that of a kernel "
                                      "sharing the GPU with the other stream's)" % (n_streams, n_streams, trace_ms, shade_ms, per_stream_ms, dt / args.steps * 1e3),
                    "note": ("%.1f MB of nodes + triangle records live in %s%s; `frac` (vs HBM peak) is the contract's algorithmic figure -- for cache-resident data a cache "
                             "bandwidth --, `hbm_measured` what the L2's memory side moved (FETCH_SIZE counts Infinity-Cache hits too: for data that fits the "
                             "256-MB MALL that is not all HBM)") % (scene_bytes / 1e6, residency, ": HBM carries only the path-pool stream" if resident else "")}
        rpp_ref = (c.rays_primary + c.rays_continuation + c.self_shadow_tests) / max(1, c.paths)
        out = {
            "metric": "Mray/s (primary+secondary), %s %dspp" % (cfg["scene"], cfg["spp"]), "value": round(total_rays / dt / 1e6, 2), "unit": "Mray/s",
            "n_gpus": n_ranks, "rccl_ranks": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"name": args.config,
                       "workload": "%s, %d spp per step%s%s, MIS integrator, reference-faithful shadow rays" % (
                           cfg["label"], job_spp, "" if job_spp == cfg["spp"] else " (of the config's %d)" % cfg["spp"],
                           "" if split == 1 else (", EVERY one of %d ranks renders that (weak)" % split if not strong else
                                                  ", split over %d ranks by %s (%d spp per rank)" % (split, "8x8 tiles" if args.shard == "tiles" else "sample range", spp))),
                       "spp_per_step": job_spp, "spp_per_rank": spp,
                       "n_tris": int(info.n_tris), "wide_bvh": "%d-wide, %d nodes, depth %d" % (info.wide_width, info.wide_nodes, info.wide_depth), "scene_device_bytes": int(info.device_bytes),
                       "parallelism": "%s shard x%d + RCCL all-reduce of the %dx%dx4 fp32 film" % (
                           "interleaved 8x8 pixel-tile" if args.shard == "tiles" else "sample-range", n_ranks, W, H)},
            "mpath_per_s": round(total_paths / dt / 1e6, 2), "rays_per_path": round(total_rays / max(1.0, total_paths), 3),
            "ray_definition": "value counts rays actually traversed (shadow rays rejected by the reference's light self-occlusion are decided "
                              "without traversal and NOT counted: %.3f rays/path); the CPU reference traverses those too (%.3f rays/path), "
                              "so mpath_per_s is the like-for-like pair" % (total_rays / max(1.0, total_paths), rpp_ref),
            "self_shadow_rate": round(c.self_shadow_hits / max(1, c.self_shadow_tests), 4),
            "film_count_plane_ok": film_ok,
            "emulated_world": emu if emu > 1 else None, "emulated_rank": (my_rank if emu > 1 else None), "emulated_ranks": emu_ranks,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, cfg, scene, rpp_ref)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if not film_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
