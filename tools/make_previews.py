#!/usr/bin/env python3
"""Developer tool (GPU box): side-by-side previews for profiles/previews/ -- the three synthetic scenes through libmcpt_hip.so, and
S-cornell through the REAL reference (oracle/_ref, CPU) with the reference's own tonemap (Scene::getPixelsColor), same size and depth.
usage: python tools/make_previews.py [outdir=gpurun_out/previews]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import __graft_entry__ as ge
pkg = ge.load_package()
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/previews"
os.makedirs(out, exist_ok=True)
W = 320
for name, scene, spp, depth in (("cornell", pkg.scenes.cornell_box(W, W), 1024, 8), ("veach", pkg.scenes.veach_mis(480, 270), 1024, 8),
                                ("bath", pkg.scenes.bathroom_stress(480, 270, detail=64), 1024, 8)):
    r = pkg.Renderer(scene, max_depth=depth); t = time.time(); r.render(spp, seed=7); r.sync(); dt = time.time() - t
    Image.fromarray(r.tonemap(flip_y=True)).save(os.path.join(out, name + "_mi355x_%dspp.png" % spp)); r.close()
    print("%s %dx%d %d spp on the GPU: %.2f s" % (name, scene.camera.width, scene.camera.height, spp, dt), flush=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as orc
os.environ["OMP_NUM_THREADS"] = str(min(16, len(os.sched_getaffinity(0))))
ref = orc.Reference(depth_variant=True)
obj = pkg.scenes.cornell_box(W, W).write(tempfile.mkdtemp(prefix="mcpt_prev_"))
ref.load(obj); ref.set_max_bounces(8); ref.stream_mode()
t = time.time(); ref.render(96); dt = time.time() - t
img = ref.pixels_u8()[::-1]                                                   # film row 0 = bottom of the image (Scene.cpp:40-46)
Image.fromarray(np.ascontiguousarray(img)).save(os.path.join(out, "cornell_reference_cpu_96spp.png"))
print("cornell %dx%d 96 spp through the real reference on %s host threads: %.1f s" % (W, W, os.environ["OMP_NUM_THREADS"], dt), flush=True)
