#!/usr/bin/env python3
"""Experiment: N contexts (own stream + own pool) rendering concurrently from N host threads vs one context."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
scene = pkg.scenes.cornell_box(800, 800)
rs = [pkg.Renderer(scene, max_depth=8) for _ in range(n)]
for r in rs: r.render(8, seed=1); r.sync()
def work(i):
    rs[i].render(spp // n, seed=2, first_sample=i * (spp // n)); rs[i].sync()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
rays = sum(r.counters().rays for r in rs) - sum(0 for _ in rs)
c = [r.counters() for r in rs]
rays = sum(x.rays for x in c)
print("contexts=%d pool_log2=%s  wall %.1f ms  %.1f Mray/s (incl. warm-up rays in counter: ignore ~1%%)" % (n, os.environ.get("MCPT_WF_POOL_LOG2", "22"), dt * 1e3, (rays * spp / (spp + 8)) / dt / 1e6))
