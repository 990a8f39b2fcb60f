"""Build libstar_nugget.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["error.cpp", "select.hip", "gather.hip", "cif.hip", "cifagg.hip", "typeembed.hip", "gatherln.hip", "gemm.hip", "xattn.hip"]
LIB = os.path.join(HERE, "libstar_nugget.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file extras (STAR_XATTN_FLAGS etc. override for experiments)
EXTRA = {"xattn.hip": os.environ.get("STAR_XATTN_FLAGS", "-mllvm -amdgpu-mfma-vgpr-form=1").split(),
         "gemm.hip": os.environ.get("STAR_GEMM_FLAGS", "-fno-slp-vectorize").split()}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "..", "..", "include", "star_nugget.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        op = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(d, op) for d in deps):
            cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
