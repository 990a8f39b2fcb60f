"""Per-chunk latency of the streaming segmenter: StreamingNuggetSegmenter.push =
star_scorer_fwd (GEMM + finish kernel) + star_cif_stream_push, for chunk sizes Tc in {16, 64, 160} frames
(0.32 / 1.28 / 3.2 s of audio at 20 ms per frame) and B in {1, 8} concurrent utterances, d = 1024.
  eager: host wall time per push over 200 pushes (launches issued from Python, one sync at the end)
  graph: the same push captured in a HIP graph, device time per replay (HIP events around 200 replays)
usage: python tools/stream_latency.py > profiles/r03_stream_latency.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from star_amd.nugget import NuggetScorer  # noqa: E402
from star_amd.streaming import StreamingNuggetSegmenter  # noqa: E402

DEV = "cuda:0"
d, N = 1024, 200
torch.manual_seed(0)
scorer = NuggetScorer(d).to(DEV)
with torch.no_grad():
    scorer[2].bias.fill_(-2.5)
print(f"# StreamingNuggetSegmenter.push, d = {d}, bf16 states, {N} pushes per cell; launches per push: 3 "
      "(gemm_nt_kernel<EPI_SCORE>, scorer_finish_kernel, cif_stream_push_kernel)")
print(f"{'B':>3s} {'Tc':>4s} {'eager us/push (host wall)':>26s} {'graph us/push (device)':>24s} {'nuggets/utt after %d pushes' % N:>30s}")
for B in (1, 8):
    for Tc in (16, 64, 160):
        kmax = int(N * Tc * 0.12) + 16
        seg = StreamingNuggetSegmenter(scorer, B, d, max_nuggets=kmax, device=DEV)
        x = torch.randn(B, Tc, d, device=DEV).to(torch.bfloat16)
        s = torch.randn(B, Tc, d, device=DEV).to(torch.bfloat16)
        for _ in range(5):
            seg.push(x, s)
        seg.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            seg.push(x, s)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / N * 1e6
        counts = seg.state.counts.float().mean().item()
        assert int(seg.overflow().sum()) == 0
        seg.reset()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                seg.push(x, s)
            seg.reset()
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(N):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            graph = a.elapsed_time(b) / N * 1e3
        print(f"{B:3d} {Tc:4d} {eager:26.1f} {graph:24.1f} {counts:30.1f}")
