#!/usr/bin/env python3
"""Where a small mcpt_render call spends its time (the reference's per-frame loop: one sample per pixel per call).

    python tools/frame_mode_probe.py [--size 800] [--depth 8]
Prints, per batch size: ms per call (render + sync), iterations per call, and the film read-back time.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--calls", type=int, default=40)
    a = ap.parse_args()
    pkg = ge.load_package()
    scene = pkg.scenes.cornell_box(a.size, a.size)
    r = pkg.Renderer(scene, max_depth=a.depth)
    r.render(4, 1, 0); r.sync()
    first = 4
    for spp in (1, 2, 4, 16, 64):
        r.reset_counters()
        t0 = time.perf_counter()
        for _ in range(a.calls):
            r.render(spp, 1, first); first += spp
            r.sync()
        dt = (time.perf_counter() - t0) / a.calls
        c = r.counters()
        print(f"spp {spp:3d}: {dt * 1e3:7.3f} ms / call  {dt * 1e3 / spp:7.3f} ms / sample  iterations / call {c.iterations / a.calls:6.1f}  "
              f"Mpath/s {c.paths / (dt * a.calls) / 1e6:8.1f}  GPU span {c.kernel_ms_total / a.calls:7.3f} ms  shade {c.shade_ms_total / a.calls:6.3f} trace {c.trace_ms_total / a.calls:6.3f}", flush=True)
    t0 = time.perf_counter()
    for _ in range(10):
        r.read_accum()
    print(f"read_accum (numpy alloc + D2H): {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms", flush=True)
    # back-to-back calls without a sync in between (what a device-resident film allows)
    for spp in (1, 4):
        t0 = time.perf_counter()
        for _ in range(a.calls):
            r.render(spp, 1, first); first += spp
        r.sync()
        dt = (time.perf_counter() - t0) / a.calls
        print(f"spp {spp:3d}, no sync between calls: {dt * 1e3:7.3f} ms / call", flush=True)


if __name__ == "__main__":
    main()
