"""Dev tool: the RCCL calls of the N > 1 bench path on a ONE-rank process group (the pool gives this build a single
GPU): init_process_group("nccl"), all_reduce of the flat gradient buffers on the main and on a side stream between HIP
graph replays, float64 MAX / SUM reductions, barrier.  usage: python tools/rccl_1rank_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
from star_amd import dist as sd
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
flat = torch.arange(1 << 20, dtype=torch.float32, device=dev)
big = torch.ones(16_801_792, dtype=torch.float32, device=dev)
side = sd.SideStreamAllReduce(big)
buck = sd.BucketedSideAllReduce(torch.ones(4 * 4_200_448, dtype=torch.float32, device=dev), 4)
g = torch.cuda.CUDAGraph()
x = torch.zeros(1 << 20, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    x.add_(1.0)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    x.add_(1.0)
for it in range(3):
    g.replay()
    side.start(world=2)          # world = 2 on purpose: exercises the collective + the scale on the side stream
    for l in reversed(range(4)):   # bench.py's N > 1 step: a bucket behind every layer graph
        g.replay()
        buck.start(l, world=2)
    sd._all_reduce(flat, dist.ReduceOp.SUM)
    side.join()
    buck.join()
torch.cuda.synchronize()
assert abs(big[0].item() - 0.5 ** 3) < 1e-7, big[0].item()
assert all(abs(b[-1].item() - 0.5 ** 3) < 1e-7 for b in buck.buckets)
assert flat[5].item() == 5.0 and x[0].item() == 1 + 3 * 5, (flat[5].item(), x[0].item())
assert sd.max_over_ranks(1.25, dev) == 1.25 and sd.sum_over_ranks(3.0, dev) == 3.0
sd.barrier()
dist.destroy_process_group()
print("rccl 1-rank check ok")
