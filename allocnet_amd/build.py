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
# translation units with flags of their own: (source, extra flags)
UNITS = [("piece_grad_unit.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         ("qp_ipm_fuse_unit.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: the one MFMA of the library (the FP64 Schur update of k_qp_ipm) keeps its accumulator in VGPRs; in
# AGPRs it pushes the jerk instantiation past 256 registers in total, i.e. from two workgroups per CU to one.
MFMA_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ldl", "-I", os.path.join(ROOT, "include")]


def probe_flags(flags):
    """`flags` if this hipcc accepts them (an LLVM without the option rejects an unknown -mllvm argument and the whole
    library would fail to build), else [] with a warning: a trivial device compile decides."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.hip")
        with open(src, "w") as f:
            f.write("#include <hip/hip_runtime.h>\n__global__ void k(double *p) { p[0] = 1.0; }\n")
        res = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-c", src, "-o", os.path.join(td, "probe.o")] + flags,
                             capture_output=True, text=True)
    if res.returncode != 0 and subprocess.run([a for a in res.args if a not in flags], capture_output=True).returncode != 0:
        raise RuntimeError("hipcc cannot compile a trivial kernel for gfx950:\n" + res.stderr)
    if res.returncode != 0:
        sys.stderr.write("allocnet_amd.build: hipcc rejects " + " ".join(flags) + " -- building without it (the FP64 MFMA "
                         "accumulator of k_qp_ipm goes to AGPRs: one workgroup per CU for the jerk instantiation)\n")
        return []
    return flags


def _deps():
    out = [os.path.join(ROOT, "include", "allocnet_amd.h")]
    for f in os.listdir(SRC_DIR):
        if f.endswith((".hip", ".h", ".hpp", ".inc")):
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
    import tempfile
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("ANET_BUILD_FLAGS", "").split()      # e.g. -DANET_PERSIST_PROF (tools/ only)
    # the units compile while the main source does (-mllvm flags are per invocation), then everything is linked
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl")] + probe_flags(MFMA_VGPR_FORM)
    jobs, objs = [], []
    tmp = tempfile.TemporaryDirectory(prefix="anet_build_")     # the objects do not stay in the tree
    OBJ_DIR = tmp.name
    for src, uflags in UNITS:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        ucmd = [HIPCC] + cflags + uflags + extra + ["-c", os.path.join(SRC_DIR, src), "-o", obj]
        if verbose:
            print(" ".join(ucmd), flush=True)
        jobs.append((ucmd, subprocess.Popen(ucmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    main_objs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        main_objs.append(obj)
        mcmd = [HIPCC] + cflags + extra + ["-c", os.path.join(SRC_DIR, src), "-o", obj]
        if verbose:
            print(" ".join(mcmd), flush=True)
        jobs.append((mcmd, subprocess.Popen(mcmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for jcmd, pr in jobs:
        out, err = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out + err)
            raise RuntimeError("hipcc failed: " + " ".join(jcmd))
        if verbose and err:
            sys.stderr.write(err)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + main_objs + objs + ["-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking liballocnet_amd.so")
    return LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
