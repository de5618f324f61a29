"""Build the gfx950 shared library in-tree (allocnet_amd/lib/liballocnet_amd.so).

    python -m allocnet_amd.build [--force] [--verbose]

hipcc cross-compiles for gfx950 without a GPU present.  The built .so is git-ignored but ships
to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "liballocnet_amd.so")
SOURCES = ["allocnet_amd.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: the one MFMA of the library (the FP64 Schur update of k_qp_ipm) keeps its accumulator in VGPRs; in
# AGPRs it pushes the jerk instantiation past 256 registers in total, i.e. from two workgroups per CU to one.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ldl", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-I", os.path.join(ROOT, "include")]


def _deps():
    out = [os.path.join(ROOT, "include", "allocnet_amd.h")]
    for f in os.listdir(SRC_DIR):
        if f.endswith((".hip", ".h", ".hpp")):
            out.append(os.path.join(SRC_DIR, f))
    return out


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("ANET_BUILD_FLAGS", "").split()      # e.g. -DANET_PERSIST_PROF (tools/ only)
    cmd = [HIPCC] + FLAGS + extra + [os.path.join(SRC_DIR, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building liballocnet_amd.so")
    if verbose and res.stderr:
        sys.stderr.write(res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
