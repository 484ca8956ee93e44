import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
pkg = ge.load_package()
d = int(sys.argv[1]) if len(sys.argv) > 1 else 420
scene = pkg.scenes.bathroom_stress(1920, 1080, detail=d)
for kind, env in (("host-sah", {}), ("ploc+host-collapse", {"MCPT_HOST_COLLAPSE": "1"}), ("ploc+device-collapse", {})):
    for k, v in env.items(): os.environ[k] = v
    t0 = time.time()
    r = pkg.Renderer(scene, max_depth=8, flags=0 if kind == "host-sah" else pkg.FLAG_GPU_BVH_BUILD)
    t = time.time() - t0; i = r.info()
    r.render(8, seed=1); r.sync(); r.reset_counters(); r.render(32, seed=2); r.sync(); c = r.counters()
    print("%-22s tris=%d create %.2f s (bvh %.0f ms, upload-phase %.0f ms) nodes %d depth %d  render %.1f ms %.0f Mray/s" % (kind, i.n_tris, t, i.bvh_build_ms, i.upload_ms, i.n_nodes, i.bvh_depth, c.kernel_ms, c.rays / c.kernel_ms / 1e3), flush=True)
    r.close()
    for k in env: os.environ.pop(k)
