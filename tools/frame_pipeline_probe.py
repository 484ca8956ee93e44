#!/usr/bin/env python3
"""Developer tool (GPU box): what overlapping consecutive one-sample-per-pixel calls would buy.  K contexts for the same scene on one GPU
(mcpt_clone_to_device) take the frames in turn, nobody waits between calls: K frames are in flight at any time, each on its own streams and
pools -- the emulation of a K-deep frame pipeline without any library change.   python tools/frame_pipeline_probe.py [--size 800] [--depth 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=800); ap.add_argument("--depth", type=int, default=8); ap.add_argument("--frames", type=int, default=240)
    a = ap.parse_args()
    pkg = ge.load_package()
    scene = pkg.scenes.cornell_box(a.size, a.size)
    first = pkg.Renderer(scene, max_depth=a.depth)
    ctxs = [first] + [first.clone(0) for _ in range(3)]
    for r in ctxs: r.render(2, 1, 0); r.sync()
    for k in (1, 2, 3, 4):
        use = ctxs[:k]
        t0 = time.perf_counter()
        for f in range(a.frames): use[f % k].render(1, 1, 10 + f)
        for r in use: r.sync()
        dt = (time.perf_counter() - t0) / a.frames
        print("%d context(s) in rotation, no wait between calls: %.3f ms per frame" % (k, dt * 1e3), flush=True)
    for r in ctxs[1:]: r.close()
    first.close()


if __name__ == "__main__":
    main()
