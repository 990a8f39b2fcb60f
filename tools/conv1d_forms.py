"""Dev tool: the q_attn / c_proj projection (HF Conv1D: y = x @ W + b, W [in, out]) at the configs[1] decoder shape
(M = B*L = 4096, d = 1024, bf16) through the call forms torch offers and through this library's GEMM -- which hipBLASLt
kernel each form lands on decides 17 against 31 us.  Cache-cold (a 512 MB buffer is rewritten between launches) and
warm (back-to-back replays).
usage: python tools/conv1d_forms.py [--M 4096]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from star_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=4096)
ap.add_argument("--d", type=int, default=1024)
ap.add_argument("--tunable", action="store_true", help="let torch's TunableOp pick the GEMM solutions (first calls tune)")
a = ap.parse_args()
dev = "cuda:0"
if a.tunable:
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.set_filename("/tmp/star_tunableop.csv")
    tunable.set_max_tuning_duration(30)
    tunable.set_max_tuning_iterations(20)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
torch.manual_seed(0)
M, d = a.M, a.d
x = torch.randn(M, d, device=dev).bfloat16()
W = (torch.randn(d, d, device=dev) * 0.02).bfloat16()
Wt = W.t().contiguous()
b = torch.randn(d, device=dev).bfloat16()
bf = b.float()
g = torch.randn(M, d, device=dev).bfloat16()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(name, fn, cold):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    iters = 10
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in evs:
        if cold: flush.fill_(1)
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    return ts[len(ts) // 2]


def splitk_dw(S, out_dtype=torch.float32):
    xv, gv = x.view(S, M // S, d), g.view(S, M // S, d)
    if out_dtype is None:
        return torch.bmm(xv.transpose(1, 2), gv).sum(0)
    return torch.bmm(xv.transpose(1, 2), gv, out_dtype=out_dtype).sum(0).to(torch.bfloat16)


forms = [
    ("addmm(b, x, W)                 [Conv1D as in HF]", lambda: torch.addmm(b, x, W)),
    ("mm(x, W)", lambda: torch.mm(x, W)),
    ("mm(x, W) + b", lambda: torch.mm(x, W) + b),
    ("mm(x, W).add_(b)", lambda: torch.mm(x, W).add_(b)),
    ("F.linear(x, W^T copy, b)", lambda: F.linear(x, Wt, b)),
    ("F.linear(x, W^T copy)", lambda: F.linear(x, Wt)),
    ("F.linear(x, W.t() view, b)", lambda: F.linear(x, W.t(), b)),
    ("star gemm_nt_bf16(x, W^T copy, b)", lambda: ops.gemm_nt_bf16(x, Wt, bf)),
    ("backward dx: mm(g, W.t())", lambda: torch.mm(g, W.t())),
    ("backward dx: mm(g, W^T copy)", lambda: torch.mm(g, Wt)),
    ("backward dW: mm(x.t(), g)", lambda: torch.mm(x.t(), g)),
    ("W.t().contiguous()", lambda: W.t().contiguous()),
    ("backward dW: mm(g.t(), x).t()", lambda: torch.mm(g.t(), x).t()),
    ("backward dW: 2-way split-K bmm, fp32 partials", lambda: splitk_dw(2)),
    ("backward dW: 4-way split-K bmm, fp32 partials", lambda: splitk_dw(4)),
    ("backward dW: 4-way split-K bmm, bf16 partials", lambda: splitk_dw(4, None)),
]
print(f"M = {M}, K = N = {d}, bf16; us per call (events around each call incl. launch; median of 10)")
for name, fn in forms:
    fn(); torch.cuda.synchronize()       # (TunableOp tunes here)
    print("..", name, flush=True)
    torch.arange(5, device=dev)          # marker kernel between forms for tools/prof_conv1d.sh
    w = timeit(name, fn, False)
    c = timeit(name, fn, True)
    print(f"{name:52s} warm {w:7.1f}   cold {c:7.1f}", flush=True)
if a.tunable:
    for r in tunable.get_results():
        print(r)
