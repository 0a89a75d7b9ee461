"""Builds libngp_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to the
sources so that it travels with the repository snapshot to the GPU box.

Staleness is decided by CONTENT, not by mtime (a snapshot copy does not keep mtimes in order): every
object remembers the digest of (source + headers + flags) it was compiled from, and the library carries the
digest of all of them (`ngp_build_id()`, also written to libngp_hip.so.id) which `_lib.load()` checks against
the sources it finds — a stale library is rebuilt when hipcc is there and refused when it is not.

NGP_AB_VARIANTS=1 in the environment compiles the superseded kernel variants in as well (tools/*microbench*)."""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libngp_hip.so")
LIB_ID = LIB + ".id"
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "ngp_hip.h")]

# (source, extra flags).  ray_kernels.hip is compiled without FMA contraction so that the
# marcher / intersector are bit-identical to the CPU oracle (see the file header).
SOURCES = [
    ("ray_kernels.hip", ["-ffp-contract=off"]),
    ("composite_kernels.hip", []),
    ("grid_kernels.hip", []),
    ("mlp_kernels.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _flags():
    # NGP_EXTRA_DEFS="-DNAME=VALUE ...": compile-time tuning constants for experiments (part of the build id)
    extra = os.environ.get("NGP_EXTRA_DEFS", "").split()
    return COMMON + (["-DNGP_AB_VARIANTS"] if os.environ.get("NGP_AB_VARIANTS") else []) + extra


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libngp_hip.so")
    return exe


def have_hipcc():
    return bool(shutil.which("hipcc")) or os.path.exists("/opt/rocm/bin/hipcc")


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(e.encode())
    return h.hexdigest()[:16]


def source_id():
    """digest of everything the library is built from (what ngp_build_id() of an up-to-date library returns)"""
    srcs = [os.path.join(CSRC, s) for s, _ in SOURCES] + [os.path.join(CSRC, "build_id.cpp")]
    extra = _flags() + [f for _, fl in SOURCES for f in fl]
    return _digest(srcs + HEADERS, extra)


def built_id():
    try:
        return open(LIB_ID).read().strip()
    except OSError:
        return None


def build(force=False, verbose=False):
    want = source_id()
    if not force and os.path.exists(LIB) and built_id() == want:
        return LIB
    # several ranks of one launch may find a stale library at once: one of them builds, the others wait on the lock
    # and find the work done; objects and the library are written under temporary names and renamed into place
    import fcntl
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and os.path.exists(LIB) and built_id() == want:
                return LIB
            return _build_locked(want, objdir, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(want, objdir, force, verbose):
    hipcc = _hipcc()
    objs = []
    units = [(s, fl, []) for s, fl in SOURCES] + [("build_id.cpp", [], [f'-DNGP_BUILD_ID="{want}"'])]
    for src, extra, defs in units:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        d = _digest([s] + HEADERS, _flags() + extra + defs)
        stamp = o + ".sha"
        have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(o) else None
        if force or have != d:
            tmp_o = o + f".{os.getpid()}.tmp"
            cmd = [hipcc] + _flags() + extra + defs + ["-c", s, "-o", tmp_o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp_o, o)
            with open(stamp, "w") as f:
                f.write(d)
        objs.append(o)
    tmp_lib = LIB + f".{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if os.path.exists(LIB_ID):
        os.remove(LIB_ID)           # never a new library beside an old id, or the reverse
    os.replace(tmp_lib, LIB)
    with open(LIB_ID + ".tmp", "w") as f:
        f.write(want)
    os.replace(LIB_ID + ".tmp", LIB_ID)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
