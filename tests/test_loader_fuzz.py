"""Sanitizer fuzz of the host loader: a short fixed-seed run of tools/fuzz_loader.py in the CPU suite.

The loader reads untrusted OBJ / MTL / XML / PNG / JPEG files (the reference's model.cpp:19-196 trusts them).  It may
reject a file but must not crash, hang or touch memory out of bounds; the build is ASan + UBSan on the CPU
(the GPU pool runs no sanitizers).  `python tools/fuzz_loader.py --cases 2500` is the long form.
"""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_loader", os.path.join(ROOT, "tools", "fuzz_loader.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_loader_survives_mutated_scenes(tmp_path):
    findings = _tool().run(cases=60, seed=3, workdir=str(tmp_path), verbose=False)
    assert findings == []


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_scene_builder_survives_hostile_descriptions(tmp_path):
    """tools/fuzz_scene_build.cpp under ASan + UBSan: NaN / inf / huge coordinates, indices out of range, degenerate geometry,
    zero-sized films and textures.  Rejected or accepted, never a crash or a hang; an accepted scene's 8-wide tree must be sound."""
    import subprocess
    env = dict(os.environ, TMPDIR=str(tmp_path))
    p = subprocess.run([os.path.join(ROOT, "tools", "fuzz_scene_build.sh"), "40", "9"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "0 unsound" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
