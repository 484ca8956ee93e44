"""CPU tests of the drop-in boundary and the host-side logic of libmcpt_hip.so (no compute call needs a GPU here)."""
import ctypes as C
import os
import re
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcpt.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcpt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    declared = _declared_functions()
    assert len(declared) >= 20
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(pkg.EXPORTED_SYMBOLS) == declared           # the ctypes plumbing binds exactly the header
    assert lib.mcpt_abi_version() == 4


def test_ctypes_structs_match_the_header_layout(pkg):
    """Compile a 20-line C program against include/mcpt.h and compare sizeof/offsetof with the ctypes mirrors."""
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "mcpt.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(mcpt_texture), sizeof(mcpt_material), sizeof(mcpt_camera), sizeof(mcpt_scene_desc),
         sizeof(mcpt_opts), sizeof(mcpt_counters), sizeof(mcpt_scene_info), offsetof(mcpt_scene_desc, camera));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c"); exe = os.path.join(d, "t")
        open(src, "w").write(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])   # the header is plain C
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(pkg.Texture), C.sizeof(pkg.MaterialC), C.sizeof(pkg.CameraC), C.sizeof(pkg.SceneDesc), C.sizeof(pkg.Opts),
            C.sizeof(pkg.Counters), C.sizeof(pkg.SceneInfo), pkg.SceneDesc.camera.offset]
    assert got == want


def test_no_cpu_fallback(pkg):
    """Without a HIP device mcpt_create must fail loudly (MCPT_ERR_NO_DEVICE) -- there is no CPU path to fall back to."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.McptError) as e:
        pkg.Renderer(pkg.scenes.open_box(8, 8))
    assert "status 2" in str(e.value) and "no CPU fallback" in str(e.value)


def test_device_bvh_build_flag_also_needs_a_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.McptError) as e:
        pkg.Renderer(pkg.scenes.open_box(8, 8), flags=pkg.FLAG_GPU_BVH_BUILD)
    assert "status 2" in str(e.value)


def test_reference_tie_order_flag_is_refused_where_no_kernel_implements_it(pkg):
    """MCPT_FLAG_REFERENCE_TIE_ORDER (and the defined tie winner in general) lives in the production trace kernel only: a context whose kernels
    traverse the binary tree -- the recursive integrator, the cross-check megakernel -- refuses the flag (MCPT_ERR_UNSUPPORTED = status 6)
    instead of silently ignoring it.  Decided before any device is touched: runs without a GPU."""
    scene = pkg.scenes.open_box(8, 8)
    with pytest.raises(pkg.McptError) as e:
        pkg.Renderer(scene, integrator=pkg.INTEGRATOR_RECURSIVE_NEE, flags=pkg.FLAG_REFERENCE_TIE_ORDER)
    assert "status 6" in str(e.value) and "tie" in str(e.value).lower()
    os.environ["MCPT_PIPELINE"] = "mega"
    try:
        with pytest.raises(pkg.McptError) as e:
            pkg.Renderer(scene, flags=pkg.FLAG_REFERENCE_TIE_ORDER)
        assert "status 6" in str(e.value)
    finally:
        os.environ.pop("MCPT_PIPELINE", None)


def test_product_library_does_not_reference_the_oracle(pkg):
    """The shipped library must not link, load or embed anything under oracle/."""
    out = subprocess.check_output(["ldd", pkg.LIB_PATH]).decode()
    assert "oracle" not in out and "mcpt_ref" not in out
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"liboracle" not in blob and b"libmcpt_ref" not in blob and b"oracle_render" not in blob
    for root, _, files in os.walk(os.path.join(ROOT, "monte-carlo-path-tracer_amd")):
        for f in files:
            if f.endswith((".cpp", ".h", ".hip", ".py")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in txt and "mcpt_oracle" not in txt and "import oracle" not in txt, f


# ------------------------------------------------------------------ host logic: validation + flatten + BVH (mcpt_check_scene)
def test_check_scene_on_every_generator(pkg):
    for name, kw in [("open-box", {}), ("cornell-box-small", {}), ("cornell-box", {}), ("veach-mis", {}), ("bathroom2", {"detail": 24})]:
        s = pkg.scenes.SCENES[name](**kw)
        st, info, msg = pkg.check_scene(s)
        assert st == 0, (name, msg)
        assert info.n_tris == s.n_faces and info.n_lights >= 1
        assert info.max_leaf <= 4 and info.bvh_depth <= 30        # LDS traversal stack holds 32 entries
        assert info.n_nodes <= max(1, s.n_faces)                  # a binary tree over <= 4-triangle leaves


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_work_item_decode_divisions_are_exact(pkg, tmp_path):
    """The shade kernel decodes a work item with three divisions by run-time values (tiles of the call, tiles per row, film width) done as
    multiply + shift with constants the host computes per launch (wavefront.hip: wf_make_fastdiv): tests/fastdiv_check.cpp links the
    library and checks x / d for 26 000 divisors against 4 million dividends below the 2^30 bound, multiples and their neighbours first."""
    csrc = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc")
    exe = str(tmp_path / "fastdiv_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "fastdiv_check.cpp"), "-o", exe, "-L" + csrc, "-lmcpt_hip",
                           "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    # and the film-size bound the scheme rests on is enforced where scenes come in
    s = pkg.scenes.open_box(8, 8)
    st, _, msg = pkg.check_scene(s.with_resolution(32768, 32768))
    assert st != 0 and "film too large" in msg, (st, msg)


def test_leaf_formation_inside_the_collapse_keeps_the_tree_sound(pkg, monkeypatch):
    """Developer knobs of the 8-wide collapse (scene_build.cpp, LeafCosts; DESIGN 5.0 'fuller nodes'): one-triangle binary leaves, and the
    dynamic programme decides which subtrees of <= 3 triangles become ONE leaf child.  Measured: no faster than the default -- kept as a
    knob, so it is kept sound: every triangle reachable exactly once inside every quantised box on its path (mcpt_check_scene's walk)."""
    s = pkg.scenes.cornell_box(64, 64)
    st, base, msg = pkg.check_scene(s)
    assert st == 0, msg
    for visit, tri in (("0", "0.3"), ("0.1", "0.5"), ("0", "1")):
        monkeypatch.setenv("MCPT_BIN_LEAF", "1"); monkeypatch.setenv("MCPT_DP_LEAF_VISIT", visit); monkeypatch.setenv("MCPT_DP_LEAF_TRI", tri)
        st, info, msg = pkg.check_scene(s)
        assert st == 0, (visit, tri, msg)
        assert info.n_tris == base.n_tris and info.n_nodes > base.n_nodes                   # the binary tree goes down to single triangles
        assert 0.9 * base.wide_nodes <= info.wide_nodes <= 1.3 * base.wide_nodes           # ... and the collapse forms the leaves again
    monkeypatch.setenv("MCPT_BIN_LEAF", "1"); monkeypatch.setenv("MCPT_DP_LEAF_VISIT", "0"); monkeypatch.setenv("MCPT_DP_LEAF_TRI", "0")
    st, info, msg = pkg.check_scene(s)                                                      # free leaves: everything that fits is merged
    assert st == 0 and info.wide_nodes <= base.wide_nodes * 1.02, msg


def test_reference_undefined_behaviour_becomes_error_codes(pkg):
    s = pkg.scenes.open_box(8, 8)
    # no emissive triangle: the reference indexes lights[-1] (Render.cpp:204-206)
    mats = [pkg.scenes.Material(m.name, m.kd, m.ks, m.ns, (0, 0, 0)) for m in s.materials]
    dark = pkg.scenes.SceneData(s.name, s.vertex, s.normal, s.texcoord, s.face, mats, s.camera)
    st, _, msg = pkg.check_scene(dark)
    assert st == 4 and "emissive" in msg
    # out-of-range indices: the reference reads past the vector
    bad = s.face.copy(); bad[0, 1, 0] = 10 ** 6
    st, _, msg = pkg.check_scene(pkg.scenes.SceneData(s.name, s.vertex, s.normal, s.texcoord, bad, s.materials, s.camera))
    assert st == 1 and "index out of range" in msg
    bad = s.face.copy(); bad[0, 0, 3] = 99
    st, _, msg = pkg.check_scene(pkg.scenes.SceneData(s.name, s.vertex, s.normal, s.texcoord, bad, s.materials, s.camera))
    assert st == 1 and "material" in msg
    # empty / degenerate image
    st, _, _ = pkg.check_scene(s.with_resolution(0, 8))
    assert st == 1
    st, _, _ = pkg.check_scene(pkg.scenes.SceneData(s.name, s.vertex, s.normal, s.texcoord, s.face[:0], s.materials, s.camera))
    assert st == 1
    # coordinates without an order (NaN, inf) or past the fp32 boxes' range: the builders' partitions would never end
    for poison in (np.nan, np.inf, -np.inf, 1e19):
        v = s.vertex.copy(); v[int(s.face[0, 1, 0]), 2] = poison
        st, _, msg = pkg.check_scene(pkg.scenes.SceneData(s.name, v, s.normal, s.texcoord, s.face, s.materials, s.camera))
        assert st == 1 and "vertex coordinate" in msg, (poison, st, msg)
    v = s.vertex.copy(); v = np.vstack([v, [[np.nan, 0, 0]]])           # an unreferenced vertex is nobody's business
    st, _, _ = pkg.check_scene(pkg.scenes.SceneData(s.name, v, s.normal, s.texcoord, s.face, s.materials, s.camera))
    assert st == 0


def test_quantised_bvh8_is_conservative_on_awkward_geometry(pkg):
    """mcpt_check_scene walks the 8-wide tree as the kernel dequantises it and fails if any triangle sticks out of a box on its path.
    Feed it geometry that stresses the 8-bit frames: huge coordinate offsets, tiny and huge triangles side by side, flat boxes."""
    rng = np.random.RandomState(11)
    base = pkg.scenes.open_box(8, 8)
    n = 3000
    centres = rng.uniform(-1, 1, (n, 3)) * np.array([1e3, 1.0, 1e-3]) + np.array([5e4, -3.0, 0.25])
    size = 10.0 ** rng.uniform(-5, 1, (n, 1, 1))
    tri = centres[:, None, :] + size * rng.normal(size=(n, 3, 3))
    tri[::7, :, 1] = tri[::7, :1, 1]                                    # axis-aligned flat triangles
    v = np.concatenate([base.vertex, tri.reshape(-1, 3)])
    nrm = np.concatenate([base.normal, np.tile([[0.0, 1.0, 0.0]], (3 * n, 1))])
    tc = np.concatenate([base.texcoord, np.zeros((3 * n, 2))])
    off = base.vertex.shape[0]
    f = np.zeros((n, 3, 4), np.int32)
    for k in range(3):
        f[:, k, 0] = f[:, k, 1] = f[:, k, 2] = off + 3 * np.arange(n) + k
    face = np.concatenate([base.face, f])
    st, info, msg = pkg.check_scene(pkg.scenes.SceneData("stress", v, nrm, tc, face, base.materials, base.camera))
    assert st == 0, msg
    assert info.n_tris == base.n_faces + n and info.bvh_depth <= 30


def test_degenerate_geometry_builds(pkg):
    """All centroids equal (SAH cannot split) and a single-leaf scene: the builder must still terminate with a valid tree."""
    s = pkg.scenes.open_box(8, 8)
    face = np.repeat(s.face[-2:-1], 300, axis=0)                  # 300 copies of one light triangle
    st, info, msg = pkg.check_scene(pkg.scenes.SceneData(s.name, s.vertex, s.normal, s.texcoord, face, s.materials, s.camera))
    assert st == 0 and info.n_tris == 300 and info.max_leaf <= 4 and info.bvh_depth <= 30, msg
    st, info, _ = pkg.check_scene(pkg.scenes.SceneData(s.name, s.vertex, s.normal, s.texcoord, s.face[-2:], s.materials, s.camera))
    assert st == 0 and info.n_tris == 2 and info.n_nodes == 1


def test_scene_files_round_trip_through_the_reference_parser_layout(pkg, tmp_path):
    """OBJ/MTL/XML writers: what is written parses back to the arrays handed to the library (the reference's Model reads the
    same text with stringstream/stod, so both see identical doubles)."""
    s = pkg.scenes.cornell_box_small(16, 16)
    obj = s.write(str(tmp_path))
    v, n, t, f = [], [], [], []
    for line in open(obj):
        k = line.split()
        if not k: continue
        if k[0] == "v": v.append([float(x) for x in k[1:4]])
        elif k[0] == "vn": n.append([float(x) for x in k[1:4]])
        elif k[0] == "vt": t.append([float(x) for x in k[1:3]])
        elif k[0] == "f": f.append([[int(i) - 1 for i in c.split("/")] for c in k[1:4]])
    assert np.array_equal(np.array(v), s.vertex) and np.array_equal(np.array(n), s.normal) and np.array_equal(np.array(t), s.texcoord)
    assert np.array_equal(np.array(f), s.face[:, :, :3])
    xml = open(obj[:-3] + "xml").read()
    assert 'width="16"' in xml and "<light mtlname=\"light\" radiance=\"17,12,4\"/>" in xml


def test_cpp_host_loader_reads_the_reference_file_formats(pkg, tmp_path):
    """host/Model.cpp (own OBJ + MTL + XML + PPM reader behind the reference's `Model(filename)`) against the generator's
    arrays, through `mcpt_cli --check` (host-only: loader -> model_to_desc -> mcpt_check_scene)."""
    import json
    cli = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    assert os.path.exists(cli), "build() compiles mcpt_cli"
    s = pkg.scenes.bathroom_stress(64, 36, detail=8, tex_size=16)
    obj = s.write(str(tmp_path))
    out = subprocess.check_output([cli, obj, "--check"]).decode().strip().splitlines()
    j = json.loads(out[-1])
    assert j["status"] == 0 and j["faces"] == s.n_faces and j["materials"] == len(s.materials)
    assert (j["width"], j["height"]) == (64, 36) and j["fovy"] == s.camera.fovy
    w3 = np.array([1.0, 2.0, 3.0])
    assert np.isclose(j["sum_v"], (s.vertex * w3).sum(), rtol=1e-12) and np.isclose(j["sum_vn"], (s.normal * w3).sum(), rtol=1e-12)
    assert np.isclose(j["sum_vt"], (s.texcoord * w3[:2]).sum(), rtol=1e-12)
    f = s.face.astype(np.int64)
    assert j["sum_f"] == int((f[:, :, 0] + 3 * f[:, :, 1] + 5 * f[:, :, 2] + 7 * f[:, :, 3]).sum())
    tex = sum(float(pkg.texture_to_float(m.texture).sum()) if m.texture is not None else float(np.asarray(m.kd, np.float32).sum()) for m in s.materials)
    assert np.isclose(j["sum_tex"], tex, rtol=1e-4)               # (c/255)^2.2 texels, constant Kd for untextured materials
    st, info, _ = pkg.check_scene(s)
    assert (j["n_tris"], j["n_lights"], j["n_nodes"], j["bvh_depth"]) == (info.n_tris, info.n_lights, info.n_nodes, info.bvh_depth)


def test_cpp_host_loader_decodes_jpeg_textures(pkg, tmp_path):
    """host/Jpeg.cpp (baseline and progressive JPEG written from T.81) against libjpeg-turbo (Pillow): 4:4:4 / 4:2:2 / 4:2:0, odd sizes,
    grey, restart intervals, optimised Huffman tables; progressive files (spectral selection + successive approximation, the scan script
    libjpeg writes) with and without restart intervals.  `mcpt_cli --decode-image` is host-only."""
    Image = pytest.importorskip("PIL.Image")
    cli = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    rng = np.random.RandomState(3)

    def picture(w, h):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 9.0 + y / 23.0), 127 + 120 * np.cos(x / 17.0 - y / 5.0), (x * 3 + y * 5) % 256], -1)
        return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)

    def read_ppm(path):
        d = open(path, "rb").read(); parts = d.split(b"\n", 3); w, h = map(int, parts[1].split())
        return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)

    cases = [("444", (64, 48), dict(quality=92, subsampling=0)), ("420", (67, 45), dict(quality=90, subsampling=2)),
             ("422", (130, 33), dict(quality=85, subsampling=1)), ("grey", (40, 40), dict(quality=90)),
             ("rst", (100, 60), dict(quality=90, subsampling=2, restart_marker_blocks=3)), ("opt", (96, 96), dict(quality=95, subsampling=2, optimize=True)),
             ("p444", (64, 48), dict(quality=92, subsampling=0, progressive=True)), ("p420", (67, 45), dict(quality=90, subsampling=2, progressive=True)),
             ("p422", (130, 33), dict(quality=75, subsampling=1, progressive=True)), ("pgrey", (41, 39), dict(quality=90, progressive=True)),
             ("prst", (100, 60), dict(quality=90, subsampling=2, progressive=True, restart_marker_blocks=2)), ("plow", (97, 71), dict(quality=30, subsampling=2, progressive=True))]
    for name, (w, h), kw in cases:
        a = picture(w, h)
        jpg = str(tmp_path / (name + ".jpg")); out = str(tmp_path / (name + ".ppm"))
        Image.fromarray(a[..., 0] if name.endswith("grey") else a).save(jpg, **kw)
        if name.startswith("p"): assert b"\xff\xc2" in open(jpg, "rb").read()          # really a progressive (SOF2) file
        subprocess.check_call([cli, "--decode-image", jpg, out])
        mine = read_ppm(out).astype(int); ref = np.asarray(Image.open(jpg).convert("RGB")).astype(int)
        assert mine.shape == ref.shape
        d = np.abs(mine - ref)
        assert d.max() <= 3 and d.mean() <= 0.5, (name, d.max(), d.mean())
    # arithmetic-coded / lossless files stay refused: a progressive file whose SOF2 marker is rewritten to SOF10 (progressive, arithmetic)
    Image.fromarray(picture(32, 32)).save(str(tmp_path / "p.jpg"), progressive=True)
    open(str(tmp_path / "a.jpg"), "wb").write(open(str(tmp_path / "p.jpg"), "rb").read().replace(b"\xff\xc2", b"\xff\xca", 1))
    assert subprocess.call([cli, "--decode-image", str(tmp_path / "a.jpg"), str(tmp_path / "a.ppm")], stderr=subprocess.DEVNULL) == 1
    # and through the scene loader: a JPEG map_Kd ends up as (c/255)^2.2 texels (MTL: one `newmtl` block per material, in order)
    import json
    s = pkg.scenes.bathroom_stress(32, 18, detail=4, tex_size=16)
    obj = s.write(str(tmp_path))
    mtl = obj[:-3] + "mtl"
    lines = open(mtl).read().splitlines()
    expected = 0.0; n_jpeg = 0
    for i, l in enumerate(lines):
        if l.startswith("map_Kd"):
            src = os.path.join(str(tmp_path), l.split()[1]); jpg = os.path.splitext(src)[0] + ".jpg"
            Image.open(src).convert("RGB").save(jpg, quality=95, subsampling=0)
            lines[i] = "map_Kd " + os.path.basename(jpg); n_jpeg += 1
            expected += float(pkg.texture_to_float(np.asarray(Image.open(jpg).convert("RGB"))).sum())
    assert n_jpeg >= 1
    open(mtl, "w").write("\n".join(lines) + "\n")
    expected += sum(float(np.asarray(m.kd, np.float32).sum()) for m in s.materials if m.texture is None)
    j = json.loads(subprocess.check_output([cli, obj, "--check"]).decode().strip().splitlines()[-1])
    assert j["status"] == 0 and np.isclose(j["sum_tex"], expected, rtol=1e-2), (j["sum_tex"], expected)


def test_cpp_host_loader_decodes_every_png_flavour(pkg, tmp_path):
    """PNG colour types 0/2/3/4/6, bit depths 1-16 (host/Model.cpp, zlib inflate + own unfiltering) -- exact against Pillow."""
    Image = pytest.importorskip("PIL.Image")
    cli = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    rng = np.random.RandomState(5)
    a = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    cases = {"rgb": Image.fromarray(a), "rgba": Image.fromarray(np.dstack([a, rng.randint(0, 256, (37, 53, 1)).astype(np.uint8)])),
             "grey": Image.fromarray(a[..., 0]), "la": Image.fromarray(np.dstack([a[..., 0], a[..., 1]]), "LA"), "pal": Image.fromarray(a).quantize(200),
             "pal4": Image.fromarray(a).quantize(13), "bw": Image.fromarray(a[..., 0] > 128), "g16": Image.fromarray(a[..., 0].astype(np.uint16) * 257)}
    for k, im in cases.items():
        src = str(tmp_path / (k + ".png")); out = str(tmp_path / (k + ".ppm"))
        im.save(src)
        subprocess.check_call([cli, "--decode-image", src, out])
        d = open(out, "rb").read(); parts = d.split(b"\n", 3); w, h = map(int, parts[1].split())
        mine = np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)
        want = np.repeat(a[..., :1], 3, axis=2) if k == "g16" else np.asarray(Image.open(src).convert("RGB"))   # 16-bit: high byte
        assert np.array_equal(mine, want), k


def _adam7_png(img, depth=8):
    """An Adam7-interlaced PNG of an (h, w[, c]) uint8 / uint16 array, written here (Pillow cannot write interlaced files): seven passes,
    filter type 0 rows, one zlib stream -- PNG specification sections 8.2 and 9."""
    import struct, zlib
    a = np.asarray(img); h, w = a.shape[:2]; c = 1 if a.ndim == 2 else a.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    a = a.reshape(h, w, c)
    raw = b""
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0: continue
        for row in sub:
            if depth == 16: body = row.astype(">u2").tobytes()
            elif depth == 8: body = row.astype(np.uint8).tobytes()
            else: body = np.packbits(np.unpackbits(row.astype(np.uint8).reshape(-1, 1), axis=1)[:, 8 - depth:].reshape(-1)).tobytes()   # depth 1, 2, 4: grey only
            raw += b"\0" + body
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def test_cpp_host_loader_decodes_interlaced_png_bmp_and_tga(pkg, tmp_path):
    """The other containers stbi_loadf reads for the reference (model.cpp:8-23) that real scene folders may hold: Adam7-interlaced PNG
    (RGB, RGBA, grey, 16-bit, 4-bit grey; odd sizes down to 1 x 1 so that some passes are empty), BMP (24-bit, 8-bit palettised) and TGA
    (true colour and grey, raw and run-length coded) -- exact against Pillow's decode of the same files."""
    Image = pytest.importorskip("PIL.Image")
    cli = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    rng = np.random.RandomState(9)

    def decoded(path):
        out = path + ".ppm"
        subprocess.check_call([cli, "--decode-image", path, out])
        d = open(out, "rb").read(); parts = d.split(b"\n", 3); w, h = map(int, parts[1].split())
        return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)

    for name, shape, depth in (("rgb", (37, 53, 3), 8), ("rgba", (9, 20, 4), 8), ("grey", (33, 31), 8), ("g16", (17, 19), 16), ("g4", (21, 13), 4),
                               ("one", (1, 1, 3), 8), ("thin", (1, 5, 3), 8), ("tall", (7, 1, 3), 8), ("rgb16", (12, 11, 3), 16)):
        hi = 65536 if depth == 16 else (1 << depth)
        a = rng.randint(0, hi, shape).astype(np.uint16 if depth == 16 else np.uint8)
        path = str(tmp_path / ("i_%s.png" % name)); open(path, "wb").write(_adam7_png(a, depth))
        ref = Image.open(path); assert ref.info.get("interlace") == 1
        want = np.asarray(ref.convert("RGB")) if depth != 16 else None
        if depth == 16:                                                   # the loader keeps the high byte of a 16-bit sample (what an 8-bit RGB request gives)
            hi8 = (a >> 8).astype(np.uint8); want = np.repeat(hi8[..., None], 3, -1) if hi8.ndim == 2 else hi8[..., :3]
        assert np.array_equal(decoded(path), want), name
    a = rng.randint(0, 256, (23, 31, 3)).astype(np.uint8)
    for name, im, kw in (("bmp24", Image.fromarray(a), {}), ("bmp8", Image.fromarray(a).quantize(100), {}), ("bmp8g", Image.fromarray(a[..., 0]), {})):
        path = str(tmp_path / (name + ".bmp")); im.save(path, **kw)
        assert np.array_equal(decoded(path), np.asarray(Image.open(path).convert("RGB"))), name
    runs = np.repeat(rng.randint(0, 256, (23, 4, 3)).astype(np.uint8), 8, axis=1)[:, :31]        # long runs: the RLE packets get used
    for name, arr, kw in (("tga", a, {}), ("tga_rle", runs, {"compression": "tga_rle"}), ("tga_grey", a[..., 0], {}), ("tga_grey_rle", runs[..., 0], {"compression": "tga_rle"}),
                          ("tga32", np.dstack([a, a[..., :1]]), {})):
        path = str(tmp_path / (name + ".tga")); Image.fromarray(arr).save(path, **kw)
        assert np.array_equal(decoded(path), np.asarray(Image.open(path).convert("RGB"))), name


def _rgbe(a):
    """float RGB -> Radiance RGBE bytes (largest component's exponent; the classic float2rgbe)."""
    a = np.asarray(a, np.float64); m = a.max(-1)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-300))) + 1, 0).astype(np.int64)
    sc = np.where(m > 1e-32, np.ldexp(1.0, 8 - e), 0.0)
    out = np.zeros(a.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(np.floor(a * sc[..., None]), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    return out


def _hdr_rle_scanline(row):
    """One "new RLE" scanline: 2, 2, width hi, width lo, then the four component planes as runs (count > 128) and dumps."""
    w = row.shape[0]; out = bytearray([2, 2, w >> 8, w & 255])
    for k in range(4):
        comp = row[:, k]; i = 0
        while i < w:
            j = i
            while j < w and j - i < 127 and comp[j] == comp[i]: j += 1
            if j - i >= 3: out += bytes([128 + (j - i), int(comp[i])]); i = j; continue
            j = i
            while j < w and j - i < 128 and not (j + 2 < w and comp[j] == comp[j + 1] == comp[j + 2]): j += 1
            j = max(j, i + 1)
            out += bytes([j - i]) + comp[i:j].tobytes(); i = j
    return bytes(out)


def test_cpp_host_loader_decodes_radiance_hdr_rle_bmp_and_mapped_tga(pkg, tmp_path):
    """Round 4: the rest of what stbi_loadf reads for the reference (model.cpp:8-23).  Radiance .hdr is the one format stbi_loadf returns
    LINEAR (stb_image.h stbi__hdr_load / stbi__hdr_convert: mantissa * 2^(e - 136), 0 where e == 0; no (c/255)^2.2) -- flat files, the
    per-scanline run-length form, and stb's fall-back when the first scanline carries no RLE marker; fixtures written here, expected
    values from that formula (parity unpinned against a third decoder: none is installed).  RLE8 BMP and colour-mapped TGA: exact against
    Pillow's decode of the same files."""
    cli = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "csrc", "mcpt_cli")
    rng = np.random.RandomState(5)

    def decoded_hdr(path):
        out = path + ".pfm"
        subprocess.check_call([cli, "--decode-image", path, out])
        d = open(out, "rb").read(); parts = d.split(b"\n", 3); assert parts[0] == b"PF"; w, h = map(int, parts[1].split())
        return np.frombuffer(parts[3], "<f4").reshape(h, w, 3)

    def expect(q):
        return np.where(q[..., 3:] == 0, 0.0, q[..., :3].astype(np.float64) * np.ldexp(1.0, q[..., 3:].astype(np.int64) - 136)).astype(np.float32)

    for name, (h, w), rle, sig in (("flat_small", (5, 7), False, b"#?RADIANCE"), ("rle", (9, 40), True, b"#?RADIANCE"), ("rle_rgbe_sig", (3, 8), True, b"#?RGBE"),
                                   ("flat_wide", (4, 33), False, b"#?RADIANCE")):
        img = rng.uniform(0, 1, (h, w, 3)) ** 3 * 50.0
        img[:, : w // 2] = img[:, :1]                                      # long runs
        img[0, -1] = 0.0                                                   # e == 0 -> black
        q = _rgbe(img)
        body = b"".join(_hdr_rle_scanline(q[y]) for y in range(h)) if rle else q.tobytes()
        path = str(tmp_path / (name + ".hdr"))
        open(path, "wb").write(sig + b"\n# made by the test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w) + body)
        got = decoded_hdr(path)
        assert got.shape == (h, w, 3) and np.array_equal(got, expect(q)), name
        assert got.max() > 1.5                                             # really linear radiance, not an LDR range
    bad = str(tmp_path / "bad.hdr"); open(bad, "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 2 +X 2\n" + bytes(16))
    assert subprocess.call([cli, "--decode-image", bad, bad + ".out"], stderr=subprocess.DEVNULL) != 0
    cut = str(tmp_path / "cut.hdr"); open(cut, "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 64 +X 64\n" + bytes([2, 2, 0, 64, 200, 7]))
    assert subprocess.call([cli, "--decode-image", cut, cut + ".out"], stderr=subprocess.DEVNULL) != 0
    # through the Texture class: an .hdr map_Kd reaches the scene as linear texels (sum_tex of --check), no gamma
    s = pkg.scenes.bathroom_stress(64, 36, detail=8, tex_size=16)
    obj = s.write(str(tmp_path))
    import json
    texs = [m for m in s.materials if m.texture is not None]
    assert texs
    mtl = open(obj[:-4] + ".mtl").read()
    lin_sum = 0.0
    for k, m in enumerate(texs):
        hp = "lin%d.hdr" % k
        q = _rgbe(rng.uniform(0, 4, (16, 16, 3)))
        open(os.path.join(str(tmp_path), hp), "wb").write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 16 +X 16\n" + b"".join(_hdr_rle_scanline(q[y]) for y in range(16)))
        lin_sum += float(expect(q).astype(np.float64).sum())
    import re
    names = re.findall(r"map_Kd\s+(\S+)", mtl)
    assert len(names) == len(texs)
    for k, nm in enumerate(names): mtl = mtl.replace("map_Kd " + nm, "map_Kd lin%d.hdr" % k)
    open(obj[:-4] + ".mtl", "w").write(mtl)
    j = json.loads(subprocess.check_output([cli, obj, "--check"]).decode().strip().splitlines()[-1])
    const = sum(float(np.asarray(m.kd, np.float32).sum()) for m in s.materials if m.texture is None)
    assert j["status"] == 0 and np.isclose(j["sum_tex"], lin_sum + const, rtol=1e-5)

    Image = pytest.importorskip("PIL.Image")

    def decoded(path):
        out = path + ".ppm"
        subprocess.check_call([cli, "--decode-image", path, out])
        d = open(out, "rb").read(); parts = d.split(b"\n", 3); w, h = map(int, parts[1].split())
        return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)

    # ---- RLE8 BMP written here: encoded runs, absolute runs (odd length -> padded), end-of-line, a delta, end-of-bitmap
    w, h = 13, 6
    idx = np.repeat(rng.randint(0, 40, (h, 4)).astype(np.uint8), 4, axis=1)[:, :w]
    idx[2, 3:10] = np.arange(7)                                            # an absolute run
    pal = rng.randint(0, 256, (40, 3)).astype(np.uint8)
    data = bytearray()
    for y in range(h - 1, -1, -1):                                         # bottom-up
        x = 0
        while x < w:
            r = 1
            while x + r < w and idx[y, x + r] == idx[y, x] and r < 255: r += 1
            if r >= 2: data += bytes([r, int(idx[y, x])]); x += r; continue
            n = 1
            while x + n < w and n < 255 and not (x + n + 1 < w and idx[y, x + n] == idx[y, x + n + 1]): n += 1
            if n >= 3: data += bytes([0, n]) + idx[y, x:x + n].tobytes() + (b"\0" if n & 1 else b""); x += n
            else: data += bytes([1, int(idx[y, x])]); x += 1
        data += b"\0\0" if y else b"\0\1"
    palb = b"".join(bytes([int(c[2]), int(c[1]), int(c[0]), 0]) for c in pal)
    off = 14 + 40 + len(palb)
    import struct
    bmp = b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 8, 1, len(data), 2835, 2835, 40, 0) + palb + bytes(data)
    path = str(tmp_path / "rle8.bmp"); open(path, "wb").write(bmp)
    want = pal[idx]
    assert np.array_equal(decoded(path), want)
    try: assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), want)          # (Pillow >= 9.1 reads RLE8; it agrees with the fixture)
    except (OSError, NotImplementedError): pass
    # header fields that used to index outside the file: biClrUsed < 0, a header size beyond the file -- rejected, not read
    for field, val in ((46, -25000), (14, 100000), (10, 1 << 30)):
        b2 = bytearray(bmp); b2[field:field + 4] = struct.pack("<i", val)
        bp = str(tmp_path / ("bad%d.bmp" % field)); open(bp, "wb").write(bytes(b2))
        assert subprocess.call([cli, "--decode-image", bp, bp + ".out"], stderr=subprocess.DEVNULL) == 1
    fixture = os.path.join(ROOT, "tests", "golden", "loader_quirks", "bad_clrused.bmp")      # the advisor's 64-byte repro, kept as a regression fixture
    assert subprocess.call([cli, "--decode-image", fixture, str(tmp_path / "x.out")], stderr=subprocess.DEVNULL) == 1
    # an 18-byte "TGA" that claims 65535 x 65535 pixels: refused before anything is allocated
    tp = str(tmp_path / "huge.tga"); open(tp, "wb").write(bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 255, 255, 255, 255, 24, 0]))
    assert subprocess.call([cli, "--decode-image", tp, tp + ".out"], stderr=subprocess.DEVNULL) == 1
    # ---- colour-mapped TGA (types 1 / 9) as Pillow writes them
    a = rng.randint(0, 256, (23, 31, 3)).astype(np.uint8)
    runs = np.repeat(rng.randint(0, 256, (23, 4, 3)).astype(np.uint8), 8, axis=1)[:, :31]
    for name, arr, kw in (("tga_map", a, {}), ("tga_map_rle", runs, {"compression": "tga_rle"})):
        path = str(tmp_path / (name + ".tga")); Image.fromarray(arr).quantize(64).save(path, **kw)
        assert open(path, "rb").read()[2] in (1, 9)
        assert np.array_equal(decoded(path), np.asarray(Image.open(path).convert("RGB"))), name


def test_scene_survives_two_sources_and_dies_first(tmp_path):
    """host/Scene.cpp's source protocol under AddressSanitizer (tests/scene_sources_main.cpp): a source that another one displaces is told
    so (FilmSource::displaced) and no longer detaches from a Scene that may be gone by then -- two Renders sharing one Scene, the Scene
    destroyed before them (host/Render.cpp)."""
    host = os.path.join(ROOT, "monte-carlo-path-tracer_amd", "host")
    exe = str(tmp_path / "scene_sources")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", host,
                           os.path.join(ROOT, "tests", "scene_sources_main.cpp"), os.path.join(host, "Scene.cpp"), "-o", exe, "-lz"])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", (p.returncode, p.stdout, p.stderr[-2000:])
