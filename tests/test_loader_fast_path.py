"""CPU test: the loader's parallel fast path (std::from_chars on plain decimal tokens, chunks parsed side by side and stitched in file
order) against its own stream-extraction path (`MCPT_LOADER_SLOW=1`: every record through istringstream, the code the reference-parser
goldens of test_loader_vs_reference.py were pinned with).  The dumps must be identical byte for byte -- on a file that is large enough
for several chunks and full of tokens the two number parsers could disagree on."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")

AWKWARD = """v 1 2 3
v +1 2 3
v .5 5. -0
v 1e3 1E-3 1.5e+2
v 1e-320 4.9e-324 1e-400
v 1e400 -1e400 0
v nan 1 2
v inf -inf 3
v 0x10 1 2
v 1.5abc 2 3
v 1 2
v\t7\t8\t9
v   10    11   12   trailing words
v 1,5 2 3
v 0.1 0.2 0.30000000000000004
v 123456789012345678901234567890 1e22 1e23
v -.5e-1 1.e1 00012.5000
vn 0 1 0
vn
vnx 1 2 3
vn 1e0 -1e0 +0
vt 0.25 0.75
vt .5
vt\t0.125 0.5 0.0
f 1/1/1 2/2/2 3/3/3
f 1/1/1 2/2/2 3/3/3 4/4/4
f 1 / 1 / 1 2/2/2 3/3/3
f 1//1 2//2 3//3
f 1/1 2/2 3/3
f -1/-1/-1 -2/-2/-2 -3/-3/-3
f +1/1/1 2/2/2 3/3/3
f 1/1/1\t2/2/2\t3/3/3
f 99999999999/1/1 2/2/2 3/3/3
f 1/1/1 2/2/2
f 1/1/1  2/2/2   3/3/3   
"""


def test_fast_path_equals_stream_path(tmp_path):
    rng = np.random.RandomState(5)
    d = tmp_path / "scene"; d.mkdir()
    with open(d / "s.mtl", "w") as f:
        f.write("newmtl a\nKd 0.5 0.5 0.5\nnewmtl b\nKd 0.1 0.2 0.3\nKs 0.5 0.5 0.5\nNs 20\nnewmtl light\nKd 1 1 1\n")
    with open(d / "s.xml", "w") as f:
        f.write('<camera type="perspective" width="32" height="24" fovy="40">\n<eye x="0" y="0" z="5"/>\n<lookat x="0" y="0" z="0"/>\n<up x="0" y="1" z="0"/>\n</camera>\n'
                '<light mtlname="light" radiance="5,5,5"/>\n')
    mats = ["a", "b", "light", "missing"]
    with open(d / "s.obj", "w", newline="") as f:
        f.write("usemtl b\n")                                      # before its mtllib line: still resolves (the reference resolves usemtl after reading the whole file)
        f.write("f 1/1/1 2/2/2 3/3/3\n")
        f.write("mtllib s.mtl\n")
        f.write(AWKWARD)
        n = 60000                                                  # ~5 MB: several chunks on any multi-core host
        fmt = ["%.17g", "%.9g", "%.3f", "%e", "%g"]
        for i in range(n):
            v = rng.normal(size=3) * 10.0 ** rng.randint(-8, 9)
            f.write("v " + " ".join(fmt[(i + k) % 5] % x for k, x in enumerate(v)) + ("\r\n" if i % 7 == 0 else "\n"))
            if i % 3 == 0: f.write("vn %.17g %.17g %.17g\n" % tuple(rng.normal(size=3)))
            if i % 3 == 1: f.write("vt %.9g %.9g\n" % tuple(rng.uniform(-2, 2, 2)))
            if i % 997 == 0: f.write("usemtl %s\n" % mats[(i // 997) % 4])
            if i % 2 == 0 and i > 10:
                a, b, c = rng.randint(1, i, 3)
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, 1 + a % 7, 1 + a % 5, b, 1 + b % 7, 1 + b % 5, c, 1 + c % 7, 1 + c % 5))
            if i % 5000 == 0: f.write("# comment\ng group\ns off\n\n")
    assert os.path.getsize(d / "s.obj") > 4 << 20
    outs = []
    for slow in (False, True):
        out = str(tmp_path / ("slow.txt" if slow else "fast.txt"))
        env = dict(os.environ)
        env.pop("MCPT_LOADER_SLOW", None)
        if slow: env["MCPT_LOADER_SLOW"] = "1"
        subprocess.check_call([CLI, str(d / "s.obj"), "--dump-model", out, "--ref-index-order"], stdout=subprocess.DEVNULL, env=env)
        outs.append(open(out, "rb").read())
    assert len(outs[0]) > 1 << 20
    assert outs[0] == outs[1]
    head = outs[0].split(b"\n", 1)[0].split()
    assert head[0] == b"counts" and int(head[1]) == 60000 + 16    # every "v " record, parsable or not, is a vertex ("v<TAB>" is not a record: model.cpp:75-77)
