import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
scene = pkg.scenes.SCENES["veach-mis"](64, 36, light_lon=12, light_lat=6, plate_cells=4)
for spp in (16, 64):
    imgs = []
    for gpu_tree in (False, True):
        r = pkg.Renderer(scene, max_depth=0, flags=pkg.FLAG_DETERMINISTIC | (pkg.FLAG_GPU_BVH_BUILD if gpu_tree else 0))
        r.render(spp, seed=21); imgs.append(r.read_accum()); r.close()
    a, b = imgs
    differ = np.any(a[..., :3] != b[..., :3], axis=-1)
    rel = np.abs(a[..., :3] - b[..., :3]).max(-1) / np.maximum(1e-6, np.abs(a[..., :3]).max(-1))
    print(os.environ.get("MCPT_NO_RECENTRE"), "spp", spp, "differ %.4f" % differ.mean(), "rel>1e-3: %.4f" % (rel > 1e-3).mean(), "mean", a[..., :3].mean(), b[..., :3].mean())
