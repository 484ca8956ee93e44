#!/usr/bin/env python3
"""Developer tool: write the three synthetic scenes to /tmp and render previews with mcpt_cli (PNG into gpurun_out/)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
cli = os.path.join(ge.PKG_DIR, "csrc", "mcpt_cli")
os.makedirs("gpurun_out", exist_ok=True)
for name, scene, spp in (("cornell", pkg.scenes.cornell_box(400, 400), 256), ("veach", pkg.scenes.veach_mis(640, 360), 256),
                         ("bath", pkg.scenes.bathroom_stress(640, 360, detail=64), 256)):
    obj = scene.write("/tmp/prev_" + name)
    out = subprocess.check_output([cli, obj, "--spp", str(spp), "--depth", "8", "--out", "gpurun_out/prev_" + name + "_"]).decode()
    print(name, out.strip().splitlines()[-2])
