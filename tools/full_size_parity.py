#!/usr/bin/env python3
"""Developer tool (GPU box): statistical parity at the BENCH configuration -- S-cornell 800x800 depth 8 -- between libmcpt_hip.so
(8 batches of 128 spp) and the REAL reference renderer (oracle/_ref, CPU, 8 batches of `ref_spp` frames; ~20 s per frame-batch
of 4).  Prints the image means and the share of pixels whose means differ by more than 4 sigma (batch-to-batch variance).
usage: python tools/full_size_parity.py [ref_spp_per_batch=4] [scene=cornell-box|veach-mis|bathroom2] [width height] [depth] > profiles/<name>.txt"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
ref_spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
name = sys.argv[2] if len(sys.argv) > 2 else "cornell-box"
W = int(sys.argv[3]) if len(sys.argv) > 3 else 800
H = int(sys.argv[4]) if len(sys.argv) > 4 else 800
DEPTH = int(sys.argv[5]) if len(sys.argv) > 5 else 8
scene = pkg.scenes.SCENES[name](W, H, **({"detail": 64} if name == "bathroom2" else {}))
r = pkg.Renderer(scene, max_depth=DEPTH)
gm = []
t = time.time()
for b in range(8):
    r.clear(); r.render(128, seed=321, first_sample=b * 128); a = r.read_accum(); gm.append(a[..., :3] / a[..., 3:])
r.close()
g = np.stack(gm); g_mean, g_var = g.mean(0), g.var(0, ddof=1) / 8
print("GPU: 8 x 128 spp in %.1f s" % (time.time() - t), flush=True)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle as orc
os.environ["OMP_NUM_THREADS"] = str(min(16, len(os.sched_getaffinity(0))))
ref = orc.Reference(depth_variant=True)
ref.load(scene.write(tempfile.mkdtemp(prefix="mcpt_fsp_"))); ref.set_max_bounces(DEPTH); ref.stream_mode()
rm = []
t = time.time()
for b in range(8):
    ref.clear(); ref.render(ref_spp); a = ref.accum(); rm.append(a[..., :3] / a[..., 3:])
    print("reference batch %d (%d spp): %.0f s elapsed" % (b, ref_spp, time.time() - t), flush=True)
q = np.stack(rm); q_mean, q_var = q.mean(0), q.var(0, ddof=1) / 8
z = np.abs(g_mean - q_mean) / np.sqrt(g_var + q_var + 1e-12)
print("%s %dx%d depth %d (%d triangles), reference-faithful shadow rays" % (name, W, H, DEPTH, scene.n_faces))
print("image mean  GPU (1024 spp)       %s" % g_mean.mean((0, 1)))
print("image mean  reference (%d spp)   %s" % (8 * ref_spp, q_mean.mean((0, 1))))
print("relative difference of the means %s" % ((g_mean.mean((0, 1)) - q_mean.mean((0, 1))) / q_mean.mean((0, 1))))
print("pixels beyond 4 sigma: %.4f %%   beyond 3 sigma: %.3f %% (Student t, 7 degrees of freedom -- the reference variance comes from 8 batches: 0.52 %% / 1.99 %%)" % (
    100 * float((z > 4).mean()), 100 * float((z > 3).mean())))
print("median |z| %.3f (t7: 0.711)" % float(np.median(z)))
# the same on 4x4-pixel block means: closer to normal than single pixels fed by a few dozen heavy-tailed samples
def blocks(x): return x[:, :H // 4 * 4, :W // 4 * 4].reshape(x.shape[0], H // 4, 4, W // 4, 4, 3).mean((2, 4))
gb, qb = blocks(g), blocks(q)
zb = np.abs(gb.mean(0) - qb.mean(0)) / np.sqrt(gb.var(0, ddof=1) / 8 + qb.var(0, ddof=1) / 8 + 1e-12)
print("4x4 blocks beyond 4 sigma: %.4f %%   beyond 3 sigma: %.3f %%   median |z| %.3f" % (100 * float((zb > 4).mean()), 100 * float((zb > 3).mean()), float(np.median(zb))))
# reference against itself (first half of the batches vs second half) with the same statistic: the yardstick for the numbers above
h0, h1 = q[:4], q[4:]
zs = np.abs(h0.mean(0) - h1.mean(0)) / np.sqrt(h0.var(0, ddof=1) / 4 + h1.var(0, ddof=1) / 4 + 1e-12)
print("reference half vs half, pixels beyond 4 sigma: %.4f %%   beyond 3 sigma: %.3f %%   median |z| %.3f" % (100 * float((zs > 4).mean()), 100 * float((zs > 3).mean()), float(np.median(zs))))
if os.environ.get("MCPT_FSP_SAVE"):
    np.savez_compressed(os.environ["MCPT_FSP_SAVE"], g_mean=g_mean.astype(np.float32), g_var=g_var.astype(np.float32), q_mean=q_mean.astype(np.float32), q_var=q_var.astype(np.float32))
