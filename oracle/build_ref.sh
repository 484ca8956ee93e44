#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds the REAL reference hot path (laizesheng1/Monte-Carlo-Path-Tracer,
# /root/reference/src/{AABB,BSDF,BVH,Render,Scene,Triangle,model}.cpp + packages/xml/pugixml.cpp) with
# plain g++ into oracle/_ref/ (git-ignored, travels to the GPU box as a binary only).  main.cpp (GLFW /
# OpenGL window loop, Windows import libs) is not built; oracle/ref/ref_driver.cpp replaces it.
#
# The reference's own build system (CMake + MSVC + glfw3.lib/opengl32.lib) is NOT run.  Two things stop
# its sources from compiling with g++ as they lie, and both are handled here without touching
# /root/reference and without keeping any copy of its text:
#   * BVH.cpp:23 binds an rvalue to `AABB&` (MSVC extension).  The sources are copied to a mktemp dir,
#     the parameter of AABB::Union(AABB&) is made `const AABB&` there (2 tokens, AABB.h:17 + AABB.cpp:4),
#     compiled, and the temp dir is deleted.
#   * Render.cpp:73 uses the MSVC-internal std::_Pi_val; oracle/ref/ref_shim.h (force-included) defines it.
# Variant "depth" additionally bounds the integrator's `for (int bounces = 0; ; bounces++)`
# (Render.cpp:116) by the driver-settable mcpt_ref_max_bounces -- BASELINE.json's `depth=` has no
# counterpart in the reference, see DESIGN.md.  The default variant has no such edit.
#
# Usage: oracle/build_ref.sh            (no-op with a message when /root/reference is absent)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${MCPT_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
    echo "[build_ref] $REF/src not present (GPU box?) -- keeping prebuilt $OUT"; exit 0
fi
mkdir -p "$OUT"
CXX="${CXX:-g++}"
FLAGS="-std=c++17 -O2 -fopenmp -fPIC -w -include $HERE/ref/ref_shim.h -I$REF/packages"

build_variant() {   # $1 = variant name ("plain" | "depth"), $2 = output .so
    local T; T="$(mktemp -d /tmp/mcpt_refbuild.XXXXXX)"
    trap 'rm -rf "$T"' RETURN
    cp -r "$REF/src" "$T/src"; chmod -R u+w "$T/src"
    sed -i 's/AABB Union(AABB& box) const/AABB Union(const AABB\& box) const/' "$T/src/AABB.h"
    sed -i 's/AABB AABB::Union(AABB& box) const/AABB AABB::Union(const AABB\& box) const/' "$T/src/AABB.cpp"
    if [ "$1" = "depth" ]; then
        sed -i 's/for (int bounces = 0; ; bounces++)/for (int bounces = 0; bounces < mcpt_ref_max_bounces; bounces++)/' "$T/src/Render.cpp"
        grep -q 'bounces < mcpt_ref_max_bounces' "$T/src/Render.cpp" || { echo "[build_ref] depth edit did not apply"; exit 1; }
    fi
    local objs=()
    for f in AABB BSDF BVH Render Scene Triangle model; do
        $CXX $FLAGS -I"$T/src" -c "$T/src/$f.cpp" -o "$T/$f.o" & objs+=("$T/$f.o")
    done
    $CXX -std=c++17 -O2 -fPIC -w -c "$REF/packages/xml/pugixml.cpp" -o "$T/pugixml.o" & objs+=("$T/pugixml.o")
    $CXX $FLAGS -fno-access-control -I"$T/src" -c "$HERE/ref/ref_driver.cpp" -o "$T/ref_driver.o" & objs+=("$T/ref_driver.o")
    wait
    $CXX -shared -fopenmp -o "$2" "${objs[@]}"
    rm -rf "$T"
    echo "[build_ref] built $2"
}

build_variant plain "$OUT/libmcpt_ref.so"
build_variant depth "$OUT/libmcpt_ref_depth.so"
