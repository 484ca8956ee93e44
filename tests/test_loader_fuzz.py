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
