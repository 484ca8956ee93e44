#!/usr/bin/env python3
"""Sanitizer fuzz of the host loader (OBJ / MTL / XML text; PNG incl. Adam7, baseline and progressive JPEG, BMP incl. palettised and RLE8,
TGA incl. colour-mapped, Radiance RGBE textures).

Builds host/Model.cpp + host/Jpeg.cpp with -fsanitize=address,undefined (CPU only; the GPU pool has no
sanitizer runs), then feeds it mutated copies of tests/golden/loader_quirks/*: truncations, byte
flips, deletions, spliced garbage, shuffled lines and -- for the binary containers -- HEADER-FIELD mutations: a 16- or 32-bit
little-endian field in the first 64 bytes overwritten with a boundary value (0, -1, -25000, 65535, 100000, INT_MAX, INT_MIN ...), which
is what random byte flips almost never produce (the r03 advisor's BMP finding: biClrUsed < 0 with a huge header size).  A finding is any
sanitizer report, a crash, or a hang.  The loader may reject a file (ok=0) but must not read or write out of bounds.

    python tools/fuzz_loader.py [--cases N] [--seed S]
"""
import argparse
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "host")
SRC = os.path.join(ROOT, "tests", "golden", "loader_quirks")

MAIN = r"""
#include "Model.h"
#include <iostream>
int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        Model m(argv[i], true);
        size_t texels = 0;
        for (auto& mt : m.materials) texels += mt.Map_Kd->image_color.size();
        std::cout << "ok=" << m.ok << " faces=" << m.face.size() << " texels=" << texels << std::endl;
    }
}
"""


def build(workdir):
    main = os.path.join(workdir, "fuzz_main.cpp")
    with open(main, "w") as f:
        f.write(MAIN)
    exe = os.path.join(workdir, "fuzz_loader")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I", HOST, main, os.path.join(HOST, "Model.cpp"), os.path.join(HOST, "Jpeg.cpp"), "-lz", "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


BOUNDARY = [0, 1, -1, -2, -25000, 255, 256, 32767, 32768, 65535, 65536, 100000, 0x7fffffff, -0x80000000, 0x10000000]


def mutate(data, rng, text):
    d = bytearray(data)
    mode = rng.randrange(4) if text else rng.randrange(6)
    if mode >= 4 and len(d) >= 8:                                          # header-field mutation (binary containers)
        for _ in range(rng.randint(1, 3)):
            at = rng.randrange(0, min(len(d), 64) - 3)
            v = rng.choice(BOUNDARY)
            if rng.random() < 0.5: d[at:at + 4] = (v & 0xffffffff).to_bytes(4, "little")
            else: d[at:at + 2] = (v & 0xffff).to_bytes(2, "little")
        return bytes(d)
    mode %= 4
    if mode == 0:
        d = d[: rng.randint(1, max(1, len(d) - 1))]
    elif mode == 1:
        alphabet = b' /\n-0123456789.eE"<>=x\x00\xff' if text else bytes(range(256))
        for _ in range(rng.randint(1, 8)):
            d[rng.randrange(len(d))] = rng.choice(alphabet)
    elif mode == 2:
        i = rng.randrange(len(d))
        del d[i: min(len(d), i + rng.randint(1, 30))]
    elif text:
        lines = d.split(b"\n")
        rng.shuffle(lines)
        d = b"\n".join(lines)
    else:
        i, j = rng.randrange(len(d)), rng.randrange(len(d))
        d = d[:i] + bytes(rng.randrange(256) for _ in range(rng.randint(1, 40))) + d[j:]
    return bytes(d)


def run(cases, seed, workdir=None, verbose=True):
    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="mcpt_fuzz_")
    findings = []
    try:
        exe = build(workdir)
        rng = random.Random(seed)
        # (model stem, file to corrupt, is text)
        targets = [("quirk", "quirk.obj", True), ("quirk", "quirk.mtl", True), ("quirk", "quirk.xml", True),
                   ("quirk", "tex.png", False), ("jpeg", "tex.jpg", False), ("order", "order.obj", True), ("order", "order.mtl", True),
                   ("formats", "tex_prog.jpg", False), ("formats", "tex_i.png", False), ("formats", "tex.bmp", False), ("formats", "tex.tga", False),
                   ("formats2", "tex8.bmp", False), ("formats2", "tex_rle.bmp", False), ("formats2", "tex_map.tga", False), ("formats2", "tex_map_rle.tga", False),
                   ("formats2", "tex.hdr", False)]
        scene = os.path.join(workdir, "scene")
        for k in range(cases):
            stem, victim, text = targets[k % len(targets)]
            shutil.rmtree(scene, ignore_errors=True)
            shutil.copytree(SRC, scene)
            path = os.path.join(scene, victim)
            with open(path, "rb") as f:
                data = f.read()
            with open(path, "wb") as f:
                f.write(mutate(data, rng, text))
            try:
                p = subprocess.run([exe, os.path.join(scene, stem + ".obj")], capture_output=True, text=True,
                                   errors="replace", timeout=60)
            except subprocess.TimeoutExpired:
                findings.append((k, victim, "timeout"))
                continue
            if "AddressSanitizer" in p.stderr or "runtime error" in p.stderr or p.returncode not in (0, 1):
                findings.append((k, victim, p.stderr[-600:]))
        if verbose:
            print(f"{cases} mutated scenes, {len(findings)} findings")
            for f in findings[:5]:
                print(f)
    finally:
        if own:
            shutil.rmtree(workdir, ignore_errors=True)
    return findings


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    sys.exit(1 if run(a.cases, a.seed) else 0)
