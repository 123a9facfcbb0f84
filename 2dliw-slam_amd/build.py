"""Build recipe of libliw_window.so (hipcc, gfx950 only, in-tree so the .so travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libliw_window.so")
SOURCES = ["k_linearize.hip", "k_laser_slab.hip", "k_lm.hip", "k_lm_quad.hip", "k_preint.hip", "k_posegraph.hip", "liw_capi.hip", "liw_preint.cpp", "liw_laser.cpp", "liw_io.cpp", "liw_lie.cpp"]
HEADERS = ["liw_dual.hpp", "liw_kernels.hpp", "k_lm_common.hpp", "k_lin_laser_body.inc", os.path.join("..", "..", "include", "liw_window.h"), os.path.join("..", "..", "include", "liw_laser.h"), os.path.join("..", "..", "include", "liw_io.h"), os.path.join("..", "..", "include", "liw_posegraph.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
if os.environ.get("LIW_QUAD_OCC"):   # A/B aid: waves per SIMD the quad step kernel is compiled for
    FLAGS.append("-DLIW_QUAD_OCC=" + os.environ["LIW_QUAD_OCC"])
if os.environ.get("LIW_SMALL_OCC"):   # A/B aid: waves per SIMD k_lin_small is compiled for (3: <= 168 registers, co-resident with the lane-per-group laser kernel)
    FLAGS.append("-DLIW_SMALL_OCC=" + os.environ["LIW_SMALL_OCC"])
if os.environ.get("LIW_QUAD_BSD"):   # second-sweep staging depth of the quad step kernel (2 / 3)
    FLAGS.append("-DLIW_QUAD_BSD=" + os.environ["LIW_QUAD_BSD"])
if os.environ.get("LIW_SLAB_ROWS"):   # A/B aid: rows of end points in flight per wave of k_lin_laser_slab (2 / 3 / 4)
    FLAGS.append("-DLIW_SLAB_ROWS=" + os.environ["LIW_SLAB_ROWS"])
if os.environ.get("LIW_MARG_OCC"):   # A/B aid: waves per SIMD k_marg_schur (one wave per window) is compiled for
    FLAGS.append("-DLIW_MARG_OCC=" + os.environ["LIW_MARG_OCC"])
if os.environ.get("LIW_EXTRA_FLAGS"):   # A/B aid: any further -D... for a probe build (part of the source hash like every flag)
    FLAGS += os.environ["LIW_EXTRA_FLAGS"].split()
if os.environ.get("LIW_QUAD_TILE_ALIAS"):
    FLAGS.append("-DLIW_QUAD_TILE_ALIAS")
if os.environ.get("LIW_CLK"):   # phase-timing build for tools/clk_probe_*.py
    FLAGS.append("-DLIW_CLK")
    if os.environ.get("LIW_CLK_IT"):
        FLAGS.append("-DLIW_CLK_IT=" + os.environ["LIW_CLK_IT"])


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c
    return "hipcc"


STAMP = LIB + ".srchash"   # content hash of every source / header / flag the library was built from (git-ignored, travels with the .so)


def source_hash():
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]:
        h.update(os.path.basename(d).encode())
        if not os.path.exists(d):          # a missing file changes the hash (-> rebuild, which then names it) instead of raising at import
            h.update(b"<missing>")
            continue
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    """Content-based, not mtime-based: a snapshot copied to another box (fresh mtimes) is rebuilt exactly when its sources differ
    from the ones the shipped library was compiled from."""
    if not os.path.exists(LIB):
        return True
    if not os.path.exists(STAMP):
        # a shipped library without its stamp: rebuild if a compiler is here, else use it as it is (a box without hipcc must not fail)
        import shutil
        have = os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")
        if not have:
            print("2dliw-slam_amd.build: %s has no source stamp and hipcc is not available: using the library as shipped" % LIB, file=sys.stderr)
        return bool(have)
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    odir = os.path.join(HERE, "build")
    os.makedirs(odir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(odir, s.rsplit(".", 1)[0] + ".o")
        cmd = [_hipcc()] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), cmd))
        objs.append(o)
    for p, cmd in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
