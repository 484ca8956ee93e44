#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing hot path.

Metric (BASELINE.json): Mray/s (primary + continuation + shadow rays actually traversed) on S-cornell, the synthetic
stand-in for cornell-box (the cg24 scene files are not in the reference repo), 800x800, 1024 spp, depth 8 = configs[1].

One "step" = one full 800x800x1024-spp render of the scene already resident in HBM (one mcpt_render call = a stream of
[shade, trace] kernel launches over the HBM path pool).  With N > 1
ranks (one process per GPU, launched by torch.distributed.run) every rank renders the full 1024 spp of ITS OWN sample
range (rank r, step s -> samples [(s*N + r)*1024, ...)), then the fp32 accumulators are summed with one RCCL all-reduce
inside the timed region -- the path's only exchange step.  Work per GPU is fixed => "scaling": "weak".

Prints ONE JSON line on rank 0.  `roofline` is computed from device counters (algorithmic bytes) and
the HIP-event kernel time the library records around each launch; `cpu_baseline` times the REAL reference
(oracle/_ref/libmcpt_ref_depth.so, built from /root/reference by oracle/build_ref.sh) on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH = 800, 800, 1024, 8
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per unit of work with THIS build's layouts
B_BOX, B_TRI = 16, 48            # a quarter of a 64-B four-child node per slab test; one 48-B {v0,e1,e2} record per triangle test
B_RAY = 32 + 16                  # trace kernel: ray fetch (origin + direction records) + 16-B result write-back per ray
B_SHADED, B_TEXEL, B_LIGHT = 64 + 72, 16, 72 + 64     # shade kernel: shading record + fp64 corners; texel; light corners + light record
B_SLOT = 7 * 16 + 4 * 16         # shade kernel: slot state read (7 records) + written back (4 records) per visited slot


def cpu_baseline(pkg, scene, rays_per_path_ref, budget_s=20.0):
    """Time the reference's Render::render on the host cores over a bounded sample of the same workload."""
    # the GPU box exposes every host core in the affinity mask but a 1-GPU job's CPU share is 16 cores; the reference's
    # OpenMP loop also serialises on one shared mt19937 (utils.h:23-28), so more threads than that only add contention
    ncores = min(16, len(os.sched_getaffinity(0)))
    os.environ["OMP_NUM_THREADS"] = str(ncores)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    try:
        ref = orc.Reference(depth_variant=True)
        tmp = tempfile.mkdtemp(prefix="mcpt_bench_")
        obj = scene.write(tmp)
        sys.stdout.flush()
        saved = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)   # the reference prints "[Model] <path>" (model.cpp:46)
        try:
            ref.load(obj)
        finally:
            os.dup2(saved, 1); os.close(saved); os.close(devnull)
        ref.set_max_bounces(DEPTH)
        ref.stream_mode()
        t1 = ref.render(1)                                   # one frame = one spp for all 640 000 pixels
        frames = max(1, min(64, int(budget_s / max(t1, 1e-3)) - 1))
        t = ref.render(frames)
        paths = frames * WIDTH * HEIGHT
        return {"value": round(paths * rays_per_path_ref / t / 1e6, 4), "unit": "Mray/s", "cores": ncores, "kind": "reference",
                "sample": "%d frame(s) (=spp) of %dx%d S-cornell depth %d through the real reference's Render::render (OpenMP, %d threads, "
                          "%.2f s); rays = paths x %.3f rays/path (GPU counters incl. self-shadowed light samples, which the reference traverses)"
                          % (frames, WIDTH, HEIGHT, DEPTH, ncores, t, rays_per_path_ref),
                "mpath_per_s": round(paths / t / 1e6, 4)}
    except orc.ReferenceUnavailable:
        o = orc.Oracle(scene.with_resolution(200, 200), max_depth=DEPTH)
        _, c, t = o.render(4, seed=1)
        rays = c["rays_primary"] + c["rays_continuation"] + c["rays_shadow"]
        return {"value": round(rays / t / 1e6, 4), "unit": "Mray/s", "cores": ncores, "kind": "port",
                "sample": "200x200x4spp S-cornell depth %d through oracle/mcpt_oracle.cpp (OpenMP, %d threads, %.2f s)" % (DEPTH, ncores, t)}


def ri_info_nodes(r):
    return r.info().n_nodes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="override samples per step (default = the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # HIP events around both kernels of every 8th pipeline iteration, recorded on the launch stream inside the timed region (every
    # iteration costs ~3 % of the step in event packets; the sampled mean agrees with rocprofv3's all-launch average, see profiles/)
    os.environ.setdefault("MCPT_TIME_KERNELS", "8")
    mg = __import__("importlib").import_module("mcpt_amd.multigpu")
    scene = pkg.scenes.cornell_box(WIDTH, HEIGHT)
    r = pkg.Renderer(scene, max_depth=DEPTH, device=local)
    accum = torch.zeros(HEIGHT * WIDTH * 4, dtype=torch.float32, device=dev)      # torch lends memory + stream + RCCL
    r.bind_accum(accum.data_ptr())
    stream = torch.cuda.current_stream(dev)
    r.set_stream(stream.cuda_stream)

    def step(s):                            # one step = one complete render job: clear the film, render this rank's sample range, sum the films
        accum.zero_()
        r.render(args.spp, seed=20251004, first_sample=mg.first_sample(s, rank, world, args.spp))
        mg.all_reduce_film(accum)           # RCCL sum over xGMI on the same stream (no-op for one rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for s in range(args.warmup):
        step(s)
    fence()
    r.reset_counters()
    fence()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    fence()
    dt = time.perf_counter() - t0
    c = r.counters()
    t_all = torch.tensor([dt], dtype=torch.float64, device=dev)
    rays = torch.tensor([float(c.rays)], dtype=torch.float64, device=dev)
    paths = torch.tensor([float(c.paths)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX); dist.all_reduce(rays); dist.all_reduce(paths)
    dt = float(t_all.item()); total_rays = float(rays.item()); total_paths = float(paths.item())

    out = None
    if rank == 0:
        # ---- kernel durations: the library brackets the launches of every MCPT_TIME_KERNELS-th iteration of the timed region with HIP
        # events recorded on the launch stream (forked from torch's current stream, bound above); *_ms_total = sampled mean x launches
        # since reset_counters().  Dominant kernel =
        # wf_trace_kernel (BVH traversal); one launch of it per pipeline iteration.
        launches = max(1, c.iterations)
        trace_ms = c.trace_ms_total / launches
        shade_ms = c.shade_ms_total / launches
        rays_per_launch = c.rays / launches
        # ---- algorithmic bytes per ray from an instrumented pass (same scene / depth / seed, 32 spp)
        ri = pkg.Renderer(scene, max_depth=DEPTH, device=local, flags=pkg.FLAG_COUNT_TRAVERSAL)
        ri.render(32, seed=20251004); ci = ri.counters(); ri.close()
        trace_bytes_per_ray = (B_BOX * ci.box_tests + B_TRI * ci.tri_tests) / max(1, ci.rays) + B_RAY
        algo_bytes = trace_bytes_per_ray * rays_per_launch
        achieved = algo_bytes / (trace_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):            # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_passes.sh), same workload
            tj = json.load(open(tpath))            # PMC counters cannot be collected inside this process: measured per ray by the
            if tj.get("wf_trace_kernel_hbm_bytes_per_ray"):   # committed rocprofv3 passes, scaled by this run's rays per launch
                traffic = int(tj["wf_trace_kernel_hbm_bytes_per_ray"] * rays_per_launch)
            else:
                traffic = tj.get("wf_trace_kernel_hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel": "wf_trace_kernel<false>", "kernel_ms": round(trace_ms, 4), "launches_per_step": round(launches / max(1, args.steps), 1),
                    "algorithmic_bytes_per_ray": round(trace_bytes_per_ray, 1),
                    "box_tests_per_ray": round(ci.box_tests / max(1, ci.rays), 2), "tri_tests_per_ray": round(ci.tri_tests / max(1, ci.rays), 2),
                    "second_kernel": {"kernel": "wf_shade_kernel<false>", "kernel_ms": round(shade_ms, 4)},
                    "concurrency": "two sub-pipelines run concurrently (shade of one overlaps trace of the other), so per-launch durations are "
                                   "those of kernels sharing the GPU and their sum exceeds the step time",
                    "note": "algorithmic bytes of BVH traversal are mostly served by L1/L2 (scene = %.1f MB of nodes+triangles); the HBM "
                            "traffic of this kernel is the ray/hit stream of the path pool -- see DESIGN.md" % (
                                (ri_info_nodes(r) * 64 + r.info().n_tris * 48) / 1e6)}
        rpp_ref = (c.rays_primary + c.rays_continuation + c.self_shadow_tests) / max(1, c.paths)
        out = {
            "metric": "Mray/s (primary+secondary), cornell-box 1024spp", "value": round(total_rays / dt / 1e6, 2), "unit": "Mray/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S-cornell (synthetic cornell-box.obj stand-in, 39612 tris) %dx%d, %d spp/step/GPU, depth=%d, MIS integrator, "
                                   "reference-faithful shadow rays" % (WIDTH, HEIGHT, args.spp, DEPTH),
                       "parallelism": "sample-range shard x%d + RCCL all-reduce of the %dx%dx4 fp32 film" % (world, WIDTH, HEIGHT)},
            "mpath_per_s": round(total_paths / dt / 1e6, 2), "rays_per_path": round(total_rays / max(1.0, total_paths), 3),
            "self_shadow_rate": round(c.self_shadow_hits / max(1, c.self_shadow_tests), 4),
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, scene, rpp_ref)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
