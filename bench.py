#!/usr/bin/env python3
"""bench.py -- valid encoder frames/s of one NuggetStep (scorer + select + gather + nl x score-biased
cross-attention, forward + backward) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling, per-GPU work fixed): BASELINE.json configs[1] -- wav2vec2-large-shaped
states B=32, T=1500, d=1024, compress rate 4 (K=375), 4 decoder layers x 8 heads, L=128, bf16, synthetic.
For N>1 every rank owns its own 32 utterances and the step ends with one flat RCCL all-reduce of the scorer
gradients.  The step is captured once in a HIP graph (no host sync inside it) and replayed.
Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rate", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--tgt", type=int, default=128)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--breakdown", action="store_true", help="also print per-stage HIP-event timings to stderr")
    return ap.parse_args()


def cpu_baseline(args, seconds):
    """The CPU oracle (port of the reference path, fp32, torch CPU threads) on a bounded sample of the SAME
    workload: 2 utterances of T frames per step, same d / rate / layers / L."""
    from oracle import nugget_oracle as orc
    torch.manual_seed(0)
    B, T, d, h, L, nl = 2, args.frames, args.dim, args.heads, args.tgt, args.layers
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, T, d, generator=g).requires_grad_()
    s = torch.randn(B, T, d, generator=g).requires_grad_()
    lim = 1 / d ** 0.5
    w1 = ((torch.rand(d, d, generator=g) * 2 - 1) * lim).requires_grad_()
    b1 = ((torch.rand(d, generator=g) * 2 - 1) * lim).requires_grad_()
    w2 = ((torch.rand(1, d, generator=g) * 2 - 1) * lim).requires_grad_()
    b2 = ((torch.rand(1, generator=g) * 2 - 1) * lim).requires_grad_()
    qs = [torch.randn(B, h, L, d // h, generator=g).requires_grad_() for _ in range(nl)]
    gs = [torch.randn(B, h, L, d // h, generator=g) for _ in range(nl)]
    kv = [(1 + 0.25 * torch.randn(d, generator=g), 1 + 0.25 * torch.randn(d, generator=g)) for _ in range(nl)]
    lengths = [T] * B

    def step():
        for t in (x, s, w1, b1, w2, b2, *qs):
            t.grad = None
        out = orc.nugget_step_asr(x, s, lengths, args.rate, w1, b1, w2, b2, qs, kv, gs)
        out["loss"].backward()

    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": B * T * n / el, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of B={B} x T={T} (d={d}, rate={args.rate}, {nl} layers, L={L}), fp32 torch-CPU oracle, "
                      f"{el:.1f} s"}


def pmc_traffic(N, d):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled
    as the gfx950 note prescribes + WRITE_SIZE; profiles/r01_v42_pmc_scorer_fwd.json).  PMC passes cannot run
    inside this process, so the figure is only reported for the exact workload it was collected on; else None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_v42_pmc_scorer_fwd.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("algorithmic_bytes_per_launch") == N * d * 2 + d * d * 2 + N * 4:
            return round(float(doc["traffic_bytes_per_launch"]))
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    args = parse()
    if os.environ.get("STAR_BENCH_WATCHDOG"):       # debugging aid: dump the Python stacks if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["STAR_BENCH_WATCHDOG"]), exit=True)
    from star_amd import dist as sd
    rank, local, world = sd.init()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("STAR_DIST_BACKEND") == "gloo":       # rehearsal: several ranks may share one card
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from star_amd import _lib, ops
    from star_amd.nugget import NuggetScorer
    from star_amd.step import make_inputs, nugget_step_asr

    B, T, d, h, L, nl, rate = args.batch, args.frames, args.dim, args.heads, args.tgt, args.layers, args.rate
    lengths = [T] * B
    torch.manual_seed(1234)                      # identical scorer replicas on every rank
    scorer = NuggetScorer(d).to(dev)
    inp = make_inputs(lengths, T, d, h, L, nl, dev, seed=1234 + rank)
    params = list(scorer.parameters())
    grad_targets = params + [inp.x, inp.s] + inp.qs
    allreduce = sd.FlatGradAllReduce(params)
    if world > 1:
        # the scorer gradients live in the flat all-reduce buffer: no copies around the collective
        allreduce.attach()
        grad_targets = [inp.x, inp.s] + inp.qs

    def reset_grads():
        for t in grad_targets:
            t.grad = None
        if world > 1:
            allreduce.zero()

    def step():
        reset_grads()
        nugget_step_asr(scorer, inp, rate)

    # ---- eager warm-up (also fills the per-rate caches and the allocator), then graph capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        reset_grads()
        with torch.cuda.graph(graph):
            if world > 1:
                allreduce.zero()               # part of the replayed step: backward accumulates into the buffer
            nugget_step_asr(scorer, inp, rate)

    def run_step():
        if graph is not None:
            graph.replay()
        else:
            step()
        if world > 1:
            allreduce(world)

    for _ in range(args.warmup):
        run_step()
    sd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    sd.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sd.max_over_ranks(elapsed, dev)
    frames_per_step = float(sum(lengths)) * world
    ms_per_step = elapsed / args.steps * 1e3

    # for N > 1 also report the step without its gradient all-reduce (same K steps, same
    # barrier / max-over-ranks bracket); `value` and `ms_per_step` above always include the all-reduce.
    compute_ms = None
    if world > 1:
        sd.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            if graph is not None:
                graph.replay()
            else:
                step()
        torch.cuda.synchronize()
        sd.barrier()
        compute_ms = sd.max_over_ranks(time.perf_counter() - t1, dev) / args.steps * 1e3

    # ---- roofline of the dominant kernel: the fused scorer GEMM (gemm_nt_kernel<EPI_SCORE>), timed with
    # HIP events on the stream it is launched on.  Algorithmic FLOP per launch = N_f * (2 d^2 + 2 d).
    roofline = None
    if rank == 0:
        N = B * T
        xb = inp.x.detach().view(N, d)
        w1b = scorer[0].weight.detach().to(torch.bfloat16).contiguous()
        b1f, w2f, b2f = (scorer[0].bias.detach().float().contiguous(), scorer[2].weight.detach().float().view(-1).contiguous(),
                         scorer[2].bias.detach().float().contiguous())
        wsb = _lib.lib().star_scorer_fwd_workspace(N, d)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        sc = torch.empty(N, dtype=torch.float32, device=dev)
        # the training forward: it also emits the ReLU sign bits the backward consumes
        bits = torch.empty(_lib.lib().star_scorer_relu_bits_bytes(N, d), dtype=torch.uint8, device=dev)

        def launch():
            _lib.check(_lib.lib().star_scorer_fwd(xb.data_ptr(), w1b.data_ptr(), b1f.data_ptr(), w2f.data_ptr(),
                                                  b2f.data_ptr(), N, d, sc.data_ptr(), bits.data_ptr(), ws.data_ptr(),
                                                  wsb, _lib.stream_ptr()), "star_scorer_fwd")
        for _ in range(3):
            launch()
        iters = max(args.steps, 10)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            launch()
            b.record()
        torch.cuda.synchronize()
        avg_ms = sum(a.elapsed_time(b) for a, b in evs) / iters
        flops = float(N) * (2.0 * d * d + 2.0 * d)
        ach = flops / (avg_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm_nt_kernel<EPI_SCORE> (+ scorer_finish_kernel)", "achieved": round(ach, 2),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(N, d), "avg_us": round(avg_ms * 1e3, 2), "flop_per_launch": flops}

    if args.breakdown and rank == 0:
        breakdown(scorer, inp, rate, dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, args.cpu_seconds)

    if rank == 0:
        rec = {
            "metric": "encoder frames/sec, nugget scorer+select+cross-attn fwd+bwd",
            "value": frames_per_step / (elapsed / args.steps),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("configs[1]: non-streaming nugget compression, wav2vec2-large states"
                                    if (B, T, d, h, L, nl, rate) == (32, 1500, 1024, 8, 128, 4, 4)
                                    else "custom shape (parity-case configuration, not the headline workload)"),
                       "batch_per_gpu": B, "global_batch": B * world, "frames": T, "d": d, "rate": rate,
                       "K": int(inp.key_len_dev(rate)[1]), "layers": nl, "heads": h, "tgt_len": L,
                       "parallelism": f"dp{world}", "hip_graph": graph is not None,
                       "grad_allreduce": "flat fp32 scorer grads, 1 per step" if world > 1 else None,
                       "compute_only_ms_per_step": compute_ms,
                       "allreduce_ms_per_step": None if compute_ms is None else max(ms_per_step - compute_ms, 0.0)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(rec), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def breakdown(scorer, inp, rate, dev):
    """Per-stage HIP-event timings (eager launches), printed to stderr."""
    from star_amd import ops
    B, T, d = inp.x.shape

    def timed(name, fn, n=10):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"[breakdown] {name:28s} {a.elapsed_time(b) / n * 1e3:9.1f} us", file=sys.stderr, flush=True)

    print("[breakdown] eager launches: stage times include host launch overhead; per-kernel GPU times are in "
          "profiles/*_kernel_stats.csv", file=sys.stderr, flush=True)
    key_len, K = inp.key_len_dev(rate)
    with torch.no_grad():
        scores = scorer(inp.x).squeeze(-1).float()
    timed("scorer fwd", lambda: scorer(inp.x.detach()))
    timed("topk select", lambda: ops.topk_select(scores, lengths=inp.lengths_dev, rate=rate, K=K))
    nmask, idx, sg, table, pos = ops.topk_select(scores, lengths=inp.lengths_dev, rate=rate, K=K)
    timed("gather rows fwd", lambda: ops.gather_rows(inp.s.detach(), idx))
    mem = ops.gather_rows(inp.s.detach(), idx)
    h = inp.qs[0].shape[1]
    k = mem.view(B, K, h, d // h).permute(0, 2, 1, 3)
    q = inp.qs[0].detach()
    timed("xattn fwd (1 layer)", lambda: ops.xattn(q, k, k, bias=sg, key_len=key_len))
    qg, kg, vg = q.clone().requires_grad_(), k.clone().requires_grad_(), k.clone().requires_grad_()
    sgg = sg.detach().clone().requires_grad_()
    o, _ = ops.xattn(qg, kg, vg, bias=sgg, key_len=key_len)
    go = torch.randn_like(o)
    timed("xattn bwd (1 layer)", lambda: torch.autograd.grad(o, (qg, kg, vg, sgg), go, retain_graph=True))
    xs = inp.x.detach().clone().requires_grad_()
    s2 = scorer(xs).squeeze(-1)
    rows = (idx + torch.arange(B, device=dev).unsqueeze(1) * T).to(torch.int32).reshape(-1)
    gsc = torch.zeros_like(s2)
    gsc.view(-1)[rows.long()] = 1.0

    def scorer_bwd():
        scorer.set_active_rows(rows)              # single-use hint: set again for every backward
        return torch.autograd.grad(s2, (xs, *scorer.parameters()), gsc, retain_graph=True)
    timed("scorer bwd (B*K rows, +dX)", scorer_bwd)
    gm = torch.randn_like(mem)
    memg = ops.gather_rows(inp.s, idx)
    timed("gather rows bwd", lambda: torch.autograd.grad(memg, inp.s, gm, retain_graph=True))


if __name__ == "__main__":
    main()
