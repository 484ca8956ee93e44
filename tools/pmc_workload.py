#!/usr/bin/env python3
"""Developer tool (GPU box): ONE render of a BASELINE.json configuration, for rocprofv3 --pmc passes (tools/r02_profile.sh).
usage: python3 tools/pmc_workload.py <c2|c3|c4|c5> <spp> [out.json]   -- writes the ray / path / launch counts of exactly what ran"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = bench.CONFIGS[sys.argv[1]]; spp = int(sys.argv[2])
W, H = cfg["res"]
scene = pkg.scenes.SCENES[cfg["scene"]](W, H, **cfg["kw"])
r = pkg.Renderer(scene, max_depth=cfg["depth"])
r.render(spp, seed=20251004); r.sync(); c = r.counters()
out = {"config": sys.argv[1], "spp": spp, "rays": int(c.rays), "paths": int(c.paths), "iterations": int(c.iterations), "kernel_ms": c.kernel_ms,
       "mray_per_s": c.rays / c.kernel_ms / 1e3}
print(json.dumps(out), flush=True)
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"))
r.close()
