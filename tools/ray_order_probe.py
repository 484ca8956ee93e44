#!/usr/bin/env python3
"""Developer tool (GPU box, under `rocprofv3 --kernel-trace`): does the ORDER of a ray list matter to wf_trace8_kernel?  The same N random rays
(origins uniform in the scene's box, directions uniform on the sphere -- what bounce rays of a closed scene look like to the tree) are traced
through mcpt_probe_trace4 four times: as drawn, grouped by direction octant, sorted by a Morton code of the origin, and by octant then Morton
code.  The kernel durations come from the profiler's trace (tools/_sNN.sh prints them in launch order); the results must be identical sets."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
if name == "cornell-box": scene = pkg.scenes.cornell_box(64, 64)
else: scene = pkg.scenes.bathroom_stress(64, 36, detail=int(name.split(":")[1]))
v = np.asarray(scene.vertex, np.float64); lo, hi = v.min(0), v.max(0)
rng = np.random.RandomState(3)
o = lo + (0.02 + 0.96 * rng.uniform(size=(n, 3))) * (hi - lo)
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
q = np.minimum(31, ((o - lo) / (hi - lo) * 32).astype(np.int64))
morton = np.zeros(n, np.int64)
for b in range(5):
    for a in range(3): morton |= ((q[:, a] >> b) & 1) << (3 * b + a)
orders = {"as drawn": np.arange(n), "by octant": np.argsort(octant, kind="stable"), "by origin (Morton, 32^3 cells)": np.argsort(morton, kind="stable"),
          "by octant, then origin": np.argsort((octant << 15) | morton, kind="stable")}
r = pkg.Renderer(scene)
r.probe_trace4(o[:4096], d[:4096])                                   # warm-up launch (first in the trace)
ref = None
for label, idx in orders.items():
    t, tri, u, vv = r.probe_trace4(o[idx], d[idx])
    back = np.empty(n, np.int64); back[idx] = np.arange(n)
    tri0 = tri[back]
    if ref is None: ref = tri0
    print("%-34s hit rate %.4f  same triangles as the first order: %s" % (label, (tri >= 0).mean(), bool(np.array_equal(tri0, ref))), flush=True)
r.close()
