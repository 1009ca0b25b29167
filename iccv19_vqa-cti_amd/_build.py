"""Build of libcti_hip.so (hipcc, gfx950 only) -- used by __graft_entry__.build() and on first import when the
shared object is missing or older than its sources."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcti_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    root = os.path.dirname(HERE)
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(root, "include", "cti_hip.h")]


def stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _deps())


FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result"]


def build(force=False, verbose=False, jobs=None):
    """hipcc --offload-arch=gfx950: csrc/*.hip -> lib/obj/*.o (only what is older than its sources; up to `jobs` compilers at once) -> lib/libcti_hip.so
    (in-tree, so it travels with gpurun).  One object per source: a one-file change rebuilds in ~20 s instead of ~80."""
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libcti_hip.so cannot be built (and no prebuilt copy is present)")
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _deps() if not h.endswith(".hip"))
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj))
    jobs = jobs or min(8, os.cpu_count() or 1)
    running, failed = [], []

    def reap(block):
        for item in list(running):
            proc, src, obj, tmp = item
            if block or proc.poll() is not None:
                out = proc.communicate()[0]
                running.remove(item)
                if proc.returncode != 0:
                    if os.path.exists(tmp):
                        os.remove(tmp)
                    failed.append("hipcc failed on %s:\n%s" % (os.path.basename(src), out))
                else:
                    os.replace(tmp, obj)
                if block:
                    return

    for src, obj in todo:
        while len(running) >= jobs:
            reap(True)
        tmp = obj + ".tmp.%d" % os.getpid()
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        running.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), src, obj, tmp))
    while running:
        reap(True)
    if failed:
        raise RuntimeError("\n".join(failed))
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc (link) failed:\n" + r.stdout)
    os.replace(tmp, LIB)
    return LIB
