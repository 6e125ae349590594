"""Build the in-tree native pieces: libdifacto_hip.so (HIP, gfx950) and the
C++ host binaries.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdifacto_hip.so")
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hip_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "difacto_hip.h"))
    return srcs


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> difacto_amd/libdifacto_hip.so"""
    if not force and not _newer(LIB, hip_sources()):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB, os.path.join(CSRC, "dfh_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_host(force=False, verbose=False):
    """the C++ host side (Learner/Loss/Store adaptors + the difacto CLI), if present"""
    mk = os.path.join(HERE, "host", "Makefile")
    if os.path.exists(mk):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(mk)] + (["-B"] if force else []))


if __name__ == "__main__":
    build_hip(force=True, verbose=True)
    build_host(force=True, verbose=True)
