"""Dev tool: the untracked-load audit of star_amd/csrc/build.py on an assembly file of your choice.
usage: hipcc ... -save-temps -c star_amd/csrc/gather.hip ; python tools/asm_load_check.py <gather-hip-amdgcn-amd-amdhsa-gfx950.s>"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from star_amd.csrc.build import ASM_AUDIT, audit_untracked_loads  # noqa: E402

n, bad = audit_untracked_loads(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ASM_AUDIT["gather.hip"])
print(f"{n} asm loads, {len(bad)} touches of a destination before its wait")
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad or not n else 0)
