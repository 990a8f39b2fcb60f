"""Dev tool: the simultaneous NuggetStep (CIF-regularised scorer + fire scan + ILK attention) at a configs[3]-like
shape, replayed from a HIP graph; prints ms/step.  Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from star_amd.nugget import NuggetScorer
from star_amd.step import make_inputs, nugget_step_simul

B, T, d, h, nl = 32, 1500, 1024, 8, 4
ilk = "--ilk" in sys.argv
rng = np.random.default_rng(0)
lengths = [int(x) for x in rng.integers(T // 2, T + 1, size=B)]
lengths[0] = T
tlen = [int(0.075 * n) + 2 for n in lengths]
L = max(tlen)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
scorer = NuggetScorer(d).to(dev)
with torch.no_grad():
    scorer[2].bias.fill_(-2.5)
inp = make_inputs(lengths, T, d, h, L, nl, dev, seed=7, target_lengths=tlen)
targets = list(scorer.parameters()) + [inp.s] + inp.qs


def step():
    for t in targets:
        t.grad = None
    return nugget_step_simul(scorer, inp, use_ilk=ilk)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        out = step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("fires max", int(out["max_count"].item()), "L", L, "cap", L + 2)
g = torch.cuda.CUDAGraph()
for t in targets:
    t.grad = None
with torch.cuda.graph(g):
    nugget_step_simul(scorer, inp, use_ilk=ilk)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"simul step (ilk={ilk}): {ms:.3f} ms/step, {sum(lengths) / ms / 1e3:.1f} M frames/s")
