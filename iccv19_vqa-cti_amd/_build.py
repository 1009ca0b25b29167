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


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/*.hip -> lib/libcti_hip.so (in-tree, so it travels with gpurun)."""
    if not force and not stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libcti_hip.so cannot be built (and no prebuilt copy is present)")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wno-unused-result",
           "-o", tmp] + sources()
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    os.replace(tmp, LIB)
    return LIB
