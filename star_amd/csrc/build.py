"""Build libstar_nugget.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build().

The build also audits the compiler's assembly of gather.hip: the row-gather backward issues its row loads as inline
asm (untracked by the compiler) and waits for them by hand, which is only sound while no instruction touches a load's
destination registers between the load and that wait.  A compiler, flag or register-pressure change that breaks this
fails the build instead of producing stale row gradients."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["error.cpp", "select.hip", "gather.hip", "harness.hip", "cif.hip", "cifagg.hip", "typeembed.hip", "gatherln.hip", "gemm.hip", "xattn.hip", "xattn_bwd4.hip"]
LIB = os.path.join(HERE, "libstar_nugget.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# per-file extras (STAR_XATTN_FLAGS etc. override for experiments)
EXTRA = {"xattn.hip": os.environ.get("STAR_XATTN_FLAGS", "-mllvm -amdgpu-mfma-vgpr-form=1").split(),
         "gemm.hip": os.environ.get("STAR_GEMM_FLAGS", "-fno-slp-vectorize").split()}


# sources whose assembly is kept (-save-temps=obj) and audited: file -> regex of the kernels that hold untracked loads
ASM_AUDIT = {"gather.hip": r"^(_ZN4star22gather_rows_bwd_kernel\w+):"}


def audit_untracked_loads(asm_path, kernel_re):
    """Every inline-asm global_load (between #ASMSTART / #ASMEND) of the matching kernels: no instruction between the
    load and the next `s_waitcnt vmcnt(0)` may name one of its destination VGPRs.  Returns (loads, violations)."""
    lines = open(asm_path).read().split("\n")
    start, loads_total, bad = None, 0, []
    for i, l in enumerate(lines):
        if re.match(kernel_re, l):
            start = i
        if start is not None and "s_endpgm" in l:
            body = lines[start:i]
            inapp, loads = False, []
            for k, b in enumerate(body):
                if "ASMSTART" in b:
                    inapp = True
                if "ASMEND" in b:
                    inapp = False
                m = re.search(r"global_load_dwordx4 (v\[\d+:\d+\])", b)
                if m and inapp:
                    loads.append((k, m.group(1)))
            for li, r in loads:
                wait = next((k for k in range(li, len(body)) if "s_waitcnt vmcnt(0)" in body[k]), None)
                if wait is None:
                    bad.append((r, "no s_waitcnt vmcnt(0) after the load"))
                    continue
                a, z = map(int, re.findall(r"\d+", r))
                regs = set(range(a, z + 1))
                for k in range(li + 1, wait):
                    if body[k].strip().startswith(";"):
                        continue
                    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", body[k]):
                        rr = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
                        if rr & regs:
                            bad.append((r, body[k].strip()))
            loads_total += len(loads)
            start = None
    return loads_total, bad


def _asm_path(objdir, src):
    return os.path.join(objdir, src.rsplit(".", 1)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "xattn_drop.h"), os.path.join(HERE, "gemm_nt4.h"),
            os.path.join(HERE, "..", "..", "include", "star_nugget.h"), os.path.join(HERE, "..", "..", "include", "star_harness.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        op = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        audit = src in ASM_AUDIT
        if force or _newer(sp, op) or any(_newer(d, op) for d in deps) or (audit and not os.path.exists(_asm_path(objdir, src))):
            cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + (["-save-temps=obj"] if audit else []) + \
                (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    for src, kre in ASM_AUDIT.items():
        n, bad = audit_untracked_loads(_asm_path(objdir, src), kre)
        if verbose:
            print(f"asm audit {src}: {n} untracked loads, {len(bad)} touches of a destination before its wait", flush=True)
        if n == 0 or bad:
            if os.path.exists(LIB):
                os.remove(LIB)          # never leave a library built from unaudited code behind
            raise RuntimeError(f"{src}: untracked-load audit failed ({n} loads found): {bad[:4]}")
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
