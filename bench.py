#!/usr/bin/env python3
"""bench.py -- valid encoder frames/s of one NuggetStep (scorer + select / fire scan + gather + nl x score-biased
cross-attention, forward + backward) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5                 # configs[1], the headline workload
    python bench.py --config 2 | --config 3 | --rate 2|8|12        # the other BASELINE.json configurations
    python bench.py --config 4 [--gpus N]                          # configs[4]: global B = 512, strong scaling
    python bench.py --config 0                                     # configs[0]: the reference's CPU-runnable case
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (weak scaling: per-GPU work is fixed, every rank owns its own utterances), all d = 1024, 4 decoder layers x
8 heads, bf16, synthetic wav2vec2-shaped states:
  --config 1  BASELINE.json configs[1]: non-streaming step, B = 32, T = 1500, compress rate 4 (K = 375), L = 128
  --config 2  configs[2]: simultaneous step (CIF-regularised scorer, resize, fire scan), B = 64, MuST-C-shaped
              log-normal lengths up to 1500 frames, L_b = 0.075 n_b + 2, ILK off
  --config 3  configs[3]: simultaneous step with the infinite-lookback mask, B = 16, T = 3000, L = 227
  --config 4  configs[4]: the batch-sharded training step, GLOBAL B = 512 MuST-C-shaped utterances (the lengths of
              configs[2], one global draw), 512 / N per rank (star_amd.dist.shard), strong scaling
  --config 0  configs[0]: B = 4, T = 250 (lengths 250/230/200/170), d = 768, rate 4, 6 decoder layers, L = 24 -- the
              case the reference runs on a CPU; its cpu_baseline leg is the oracle's NuggetStep AND the oracle's
              whole-decoder step on all four utterances
For N > 1 the step carries the gradient all-reduces of SURVEY.md section 8e over RCCL: the scorer's flat fp32
gradient buffer (1.05 M values) and the decoder's cross-attention parameter gradients (nl x (4 d^2 + 6 d) = 16.8 M
values; the decoder weights themselves are outside the hot path, so that buffer is payload of the right size).  The
second buffer is reduced on a side stream as soon as the attention backward kernels are done, beside the gather and
scorer backward.  The step is captured in HIP graphs (no host sync inside; one graph at N = 1, two around the
start of the side-stream all-reduce for N > 1) and replayed.
Prints ONE JSON line on rank 0 (fields: DESIGN.md section 5).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0            # HBM3E (same table; a streaming copy reaches about 6.3 TB/s of it)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=1, choices=(0, 1, 2, 3, 4), help="BASELINE.json configs[i]")
    ap.add_argument("--rate", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--tgt", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--staged", action="store_true",
                    help="diagnostic: run the N > 1 step structure (one graph per decoder layer's attention backward) at "
                         "N = 1, to price its replay gaps; the line is marked as a custom configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ramp", action="store_true", help="skip the untimed clock-ramp steps in front of the warm-up")
    ap.add_argument("--no-roofline-all", action="store_true", help="skip the per-kernel-class roofline legs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--decoder-grad-dtype", choices=("fp32", "bf16"), default="fp32",
                    help="N > 1: dtype of the decoder-side gradient payload on the wire (the reference's DDP reduces "
                         "fp32; bf16 halves the bytes).  The other dtype is timed as well and reported beside it")
    ap.add_argument("--breakdown", action="store_true", help="also print per-stage HIP-event timings to stderr")
    return ap.parse_args()


def mustc_lengths(B, T, seed):
    """MuST-C-shaped durations: n_b = clip(round(50 * LogNormal(ln 6.0, 0.6)), 50, T) frames and
    target lengths L_b = clip(round(0.075 n_b) + 2, 4, 512)."""
    rng = np.random.default_rng(seed)
    n = np.clip(np.round(50.0 * rng.lognormal(np.log(6.0), 0.6, size=B)), 50, T).astype(int)
    n[0] = T                                   # the padded batch is T frames long
    tl = np.clip(np.round(0.075 * n) + 2, 4, 512).astype(int)
    return [int(v) for v in n], [int(v) for v in tl]


def workload(args, rank, world=1):
    """Lengths / target lengths / shapes of the chosen BASELINE.json configuration (seeded per rank)."""
    if args.dim is None:
        args.dim = 768 if args.config == 0 else 1024
    if args.layers is None:
        args.layers = 6 if args.config == 0 else 4
    if args.config == 0:
        lengths = [250, 230, 200, 170]
        if args.batch:
            lengths = (lengths * ((args.batch + 3) // 4))[:args.batch]
        T, L = args.frames or 250, args.tgt or 24
        return dict(kind="asr", B=len(lengths), T=T, L=L, lengths=[min(v, T) for v in lengths], targets=None, ilk=False,
                    scaling="weak",
                    name="configs[0]: non_streaming.sh-style nugget compression, wav2vec2-base states (the reference's "
                         "CPU-runnable case)")
    if args.config == 1:
        B, T, L = args.batch or 32, args.frames or 1500, args.tgt or 128
        return dict(kind="asr", B=B, T=T, L=L, lengths=[T] * B, targets=None, ilk=False, scaling="weak",
                    name="configs[1]: non-streaming nugget compression, wav2vec2-large states"
                    + ("" if args.rate == 4 else f", compress rate {args.rate}"))
    if args.config == 2:
        B, T = args.batch or 64, args.frames or 1500
        n, tl = mustc_lengths(B, T, 4321 + rank)
        return dict(kind="simul", B=B, T=T, L=max(tl), lengths=n, targets=tl, ilk=False, scaling="weak",
                    name="configs[2]: simul_s2t path, CIF-regularised scorer + fire scan, MuST-C-shaped lengths")
    if args.config == 4:
        # ONE global batch (the same draw on every rank), sharded by utterance: strong scaling.  The reference shards
        # a global batch with Lightning's distributed sampler under DDPStrategy (train_w2v.py:195-196).
        from star_amd import dist as sd
        GB, T = args.batch or 512, args.frames or 1500
        n, tl = mustc_lengths(GB, T, 4321)
        n, tl = sd.shard(n, rank, world), sd.shard(tl, rank, world)
        return dict(kind="simul", B=len(n), T=T, L=max(tl), lengths=n, targets=tl, ilk=False, scaling="strong",
                    global_batch=GB,
                    name="configs[4]: batch-sharded training step, global B = %d MuST-C-shaped utterances, %d per rank"
                         % (GB, len(n)))
    B, T, L = args.batch or 16, args.frames or 3000, args.tgt or 227
    return dict(kind="simul", B=B, T=T, L=L, lengths=[T] * B, targets=[L] * B, ilk=True, scaling="weak",
                name="configs[3]: simultaneous S2T with --use_ilk, long-form utterances")


def _timed_loop(step, seconds):
    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            return n, el


def cpu_baseline(args, wl, seconds):
    """The CPU oracle (port of the reference path, fp32, torch CPU threads) on a bounded sample of the SAME
    workload: the first 2 utterances of the batch (all 4 for configs[0]) per step, same T / d / rate / layers / L and
    the same kind of step: the non-streaming NuggetStep for configs[0..1], the simultaneous one (quantity loss, resize,
    serial fire scan, ILK as configured) for configs[2..4]."""
    from oracle import nugget_oracle as orc
    torch.manual_seed(0)
    simul = wl["kind"] == "simul"
    B = min(4 if args.config == 0 else 2, wl["B"])
    T, d, h, L, nl = wl["T"], args.dim, args.heads, wl["L"], args.layers
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, T, d, generator=g).requires_grad_(not simul)
    s = torch.randn(B, T, d, generator=g).requires_grad_()
    lim = 1 / d ** 0.5
    w1 = ((torch.rand(d, d, generator=g) * 2 - 1) * lim).requires_grad_()
    b1 = ((torch.rand(d, generator=g) * 2 - 1) * lim).requires_grad_()
    w2 = ((torch.rand(1, d, generator=g) * 2 - 1) * lim).requires_grad_()
    b2 = ((torch.rand(1, generator=g) * 2 - 1) * lim - (2.5 if simul else 0.0)).requires_grad_()
    qs = [torch.randn(B, h, L, d // h, generator=g).requires_grad_() for _ in range(nl)]
    gs = [torch.randn(B, h, L, d // h, generator=g) for _ in range(nl)]
    kv = [(1 + 0.25 * torch.randn(d, generator=g), 1 + 0.25 * torch.randn(d, generator=g)) for _ in range(nl)]
    lengths = list(wl["lengths"][:B])
    targets = None if not simul else list(wl["targets"][:B])

    def step():
        for t in (x, s, w1, b1, w2, b2, *qs):
            t.grad = None
        if simul:
            out = orc.nugget_step_simul(x, s, lengths, targets, wl["ilk"], w1, b1, w2, b2, qs, kv, gs)
        else:
            out = orc.nugget_step_asr(x, s, lengths, args.rate, w1, b1, w2, b2, qs, kv, gs)
        out["loss"].backward()

    n, el = _timed_loop(step, seconds)
    frames = float(sum(lengths))
    what = ("simultaneous NuggetStep (quantity loss, resize, serial fire scan, ILK %s)" % ("on" if wl["ilk"] else "off")) \
        if simul else f"non-streaming NuggetStep (rate={args.rate})"
    rec = {"value": frames * n / el, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{n} steps of the batch's first {B} utterances ({int(frames)} valid of {B} x {T} frames; d={d}, "
                     f"{nl} layers, L={L}), {what}, fp32 torch-CPU oracle, {el:.1f} s"}
    if args.config == 0:
        # for configs[0] also the whole-decoder step, to stand beside the survey's figure for the
        # reference itself on this container's 8 threads (124.3 ms = 6.84 k valid frames/s)
        n2, el2 = _timed_loop(orc.DecoderStepConfig0(lengths, T, d, h, L, nl, args.rate, seed=99), max(seconds / 2, 3.0))
        rec["whole_decoder_step"] = {
            "value": frames * n2 / el2, "unit": "frames/s", "ms_per_step": el2 / n2 * 1e3, "cores": cores,
            "sample": f"{n2} steps: the NuggetStep above inside a 6-layer GPT-2-shaped decoder (self-attention, MLP, LM "
                      f"head over V = 10000, cross-entropy; oracle/nugget_oracle.py: DecoderStepConfig0), {el2:.1f} s",
            "survey_reference_ms_per_step": 124.3, "survey_reference_cores": 8}
    return rec


def pmc_traffic(N, d):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled
    as the gfx950 note prescribes + WRITE_SIZE; profiles/r0*_pmc_scorer_fwd.json).  PMC passes cannot run
    inside this process, so the figure is only reported for the exact workload it was collected on; else None."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("r03_pmc_scorer_fwd.json", "r02_pmc_scorer_fwd.json", "r01_v42_pmc_scorer_fwd.json"):
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                doc = json.load(f)
            if doc.get("algorithmic_bytes_per_launch") == N * d * 2 + d * d * 2 + N * 4:
                return round(float(doc["traffic_bytes_per_launch"]))
        except (OSError, ValueError, KeyError):
            continue
    return None


class CacheEvictor:
    """Streams 2 GiB through the chip (a 1-GiB device copy) so that nothing of the timed operands is left in L2 or
    the 256-MiB Infinity Cache: inside the step ~1.7 GB of other traffic lies between two uses of any operand, so a leg
    replayed back to back on warm caches reads 15-45 % faster than the same kernel in the step.  The copy also takes
    ~0.5 ms of device time, which lets the host queue every launch of the timed leg behind it (no launch gaps inside
    the event bracket)."""

    def __init__(self, dev, mib=1024):
        self.src = torch.empty(mib << 20, dtype=torch.uint8, device=dev).fill_(1)
        self.dst = torch.empty_like(self.src)

    def __call__(self):
        self.dst.copy_(self.src, non_blocking=True)


def event_time_us(fn, iters=10, warm=3, graph=True, evict=None):
    """Average device time of fn() in microseconds, HIP events on the stream the kernels are launched on.
    evict given: caches are flushed before every repeat (CacheEvictor) and one pair of events brackets every eager
    launch -- the cold-cache time, the one comparable with the kernel inside the step.
    graph=True: `iters` launches captured in one HIP graph and replayed between one pair of events (no host launch
    gaps: right for legs of a few microseconds; operands stay cache-warm between repeats); graph=False: one pair of
    events around every eager launch."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if evict is not None or not graph:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            if evict is not None:
                evict()
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3
    g = torch.cuda.CUDAGraph()
    # capture on the CURRENT stream: autograd runs a backward node on the stream its forward ran on, so a leg whose
    # forward was built outside the capture must be captured on that same stream
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        for _ in range(iters):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


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
    from star_amd.step import (attention_backward, attention_backward_layer, backward_rest, backward_rest_layers,
                               make_inputs, nugget_step_asr, nugget_step_simul)

    wl = workload(args, rank, world)
    B, T, L, d, h, nl, rate = wl["B"], wl["T"], wl["L"], args.dim, args.heads, args.layers, args.rate
    lengths = wl["lengths"]
    simul = wl["kind"] == "simul"
    torch.manual_seed(1234)                      # identical scorer replicas on every rank
    scorer = NuggetScorer(d).to(dev)
    if simul:
        with torch.no_grad():
            scorer[2].bias.fill_(-2.5)           # sigmoid(score) ~ 0.08: sum of alphas in the range of the target lengths
    inp = make_inputs(lengths, T, d, h, L, nl, dev, seed=1234 + rank, target_lengths=wl["targets"])
    params = list(scorer.parameters())
    act_targets = ([] if simul else [inp.x]) + [inp.s] + inp.qs     # the simultaneous scorer sees X detached
    grad_targets = params + act_targets
    allreduce = sd.FlatGradAllReduce(params)
    decoder_side = None
    if world > 1:
        # the scorer gradients live in the flat all-reduce buffer: no copies around the collective
        allreduce.attach()
        grad_targets = act_targets
        # decoder cross-attention parameter gradients per layer: c_attn d x 2d, q_attn d x d, c_proj d x d, their
        # biases (4d) and ln_cross_attn (2d).  The weights are outside the hot path; their
        # all-reduce is part of the step, so the buffer has the real size.
        # One bucket per decoder layer, started as soon as that layer's attention backward is queued (DDP's bucketed
        # overlap): the collective runs beside the remaining layers' backward and the gather / scorer backward.
        n_dec_vals = nl * (4 * d * d + 6 * d)
        wire = {"fp32": torch.float32, "bf16": torch.bfloat16}
        decoder_sides = {k: sd.BucketedSideAllReduce(torch.zeros(n_dec_vals, dtype=v, device=dev), nl) for k, v in wire.items()}
        decoder_side = decoder_sides[args.decoder_grad_dtype]

    def reset_grads():
        for t in grad_targets:
            t.grad = None
        if world > 1:
            allreduce.zero()

    def forward():
        if simul:
            return nugget_step_simul(scorer, inp, use_ilk=wl["ilk"], backward=False)
        return nugget_step_asr(scorer, inp, rate, backward=False)

    staged = world > 1 or args.staged

    def extra(out):
        if simul:
            ql = out["quantity_loss"]
            return [ql], [torch.full_like(ql, 0.01)]
        return None, None

    def step():                                  # eager, N = 1 structure
        reset_grads()
        out = forward()
        dmem, dsg = attention_backward(out, inp)
        backward_rest(out, dmem, dsg, *extra(out))

    def step_layers(comm, decoder_side=decoder_side):      # eager, N > 1 structure: layer by layer, last layer first
        reset_grads()
        out = forward()
        lg = [None] * nl
        for l in reversed(range(nl)):
            lg[l] = attention_backward_layer(out, inp, l)
            if comm and world > 1:
                decoder_side.start(l, world)
        backward_rest_layers(out, lg, *extra(out))

    # ---- eager warm-up (also fills the per-rate caches and the allocator), then graph capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            if staged:
                step_layers(False)
            else:
                step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph_a = None
    graphs_l, graph_r = [], None
    if not args.no_graph:
        # N = 1: one graph.  N > 1: forward | attention backward of layer nl-1 | ... | layer 0 | gather / scorer backward,
        # with the start of a decoder-side all-reduce bucket after every layer graph (RCCL launches stay outside the
        # graphs).
        reset_grads()
        graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_a):
            if world > 1:
                allreduce.zero()               # part of the replayed step: backward accumulates into the buffer
            out_c = forward()
            if not staged:                     # nothing happens between the halves: one graph, no replay gap
                dmem_c, dsg_c = attention_backward(out_c, inp)
                backward_rest(out_c, dmem_c, dsg_c, *extra(out_c))
        if staged:
            lg_c = [None] * nl
            for l in reversed(range(nl)):
                g_l = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_l, pool=graph_a.pool()):
                    lg_c[l] = attention_backward_layer(out_c, inp, l)
                graphs_l.append((l, g_l))
            graph_r = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_r, pool=graph_a.pool()):
                backward_rest_layers(out_c, lg_c, *extra(out_c))

    def run_step(with_allreduce=True, decoder_side=decoder_side):
        if graph_a is None:
            if staged:
                step_layers(with_allreduce, decoder_side)
            else:
                step()
        else:
            graph_a.replay()
            for l, g_l in graphs_l:
                g_l.replay()
                if with_allreduce and world > 1:
                    decoder_side.start(l, world)   # side stream: runs beside everything queued after this point
            if graph_r is not None:
                graph_r.replay()
        if world > 1 and with_allreduce:
            allreduce(world)
            decoder_side.join()

    # Clock ramp: a fresh process starts at idle clocks, and the chip takes a few hundred steps (~0.2 s) of this load to
    # settle -- a 20-step run measured right behind 5 warm-up steps reads 2-3 % slow.  The ramp steps are the same step,
    # untimed, BEFORE the W warm-up steps the contract asks for; their number is reported in config.clock_ramp_steps.
    ramp_steps = max(0, 300 - args.warmup) if not args.no_ramp else 0
    for _ in range(ramp_steps):
        run_step()
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
    frames_local = float(sum(lengths))
    frames_per_step = sd.sum_over_ranks(frames_local, dev)
    ms_per_step = elapsed / args.steps * 1e3

    # for N > 1 also report the step without its gradient all-reduces (same K steps, same
    # barrier / max-over-ranks bracket); `value` and `ms_per_step` above always include them.
    compute_ms = other_ms = None
    if world > 1:
        sd.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            run_step(with_allreduce=False)
        torch.cuda.synchronize()
        sd.barrier()
        compute_ms = sd.max_over_ranks(time.perf_counter() - t1, dev) / args.steps * 1e3
        # the same K steps with the decoder-side payload in the OTHER wire dtype (bf16 halves the 67 MB; the reference's
        # DDP reduces fp32, so `value` is the fp32 run unless --decoder-grad-dtype bf16 is given)
        other = "bf16" if args.decoder_grad_dtype == "fp32" else "fp32"
        for _ in range(min(args.warmup, 5)):
            run_step(True, decoder_sides[other])
        sd.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            run_step(True, decoder_sides[other])
        torch.cuda.synchronize()
        sd.barrier()
        other_ms = sd.max_over_ranks(time.perf_counter() - t1, dev) / args.steps * 1e3

    # ---- roofline of the dominant kernel: the fused scorer GEMM (gemm_nt_kernel<EPI_SCORE>), timed with
    # HIP events on the stream it is launched on.  Algorithmic FLOP per launch = N_f * (2 d^2 + 2 d).
    roofline = roofline_all = None
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

        valid = inp.valid                          # ragged batch: the launch the step makes computes the valid frames only

        def launch():
            if valid is None:
                _lib.check(_lib.lib().star_scorer_fwd(xb.data_ptr(), w1b.data_ptr(), b1f.data_ptr(), w2f.data_ptr(),
                                                      b2f.data_ptr(), N, d, sc.data_ptr(), bits.data_ptr(), ws.data_ptr(),
                                                      wsb, _lib.stream_ptr()), "star_scorer_fwd")
            elif valid.compact:
                _lib.check(_lib.lib().star_scorer_fwd_compact(xb.data_ptr(), w1b.data_ptr(), b1f.data_ptr(), w2f.data_ptr(),
                                                              b2f.data_ptr(), valid.lengths.data_ptr(),
                                                              valid.offsets.data_ptr(), valid._rows_ptr.data_ptr(),
                                                              valid.n_valid, B, T, d, sc.data_ptr(), bits.data_ptr(),
                                                              ws.data_ptr(), wsb, _lib.stream_ptr()),
                           "star_scorer_fwd_compact")
            else:
                _lib.check(_lib.lib().star_scorer_fwd_ragged(xb.data_ptr(), w1b.data_ptr(), b1f.data_ptr(), w2f.data_ptr(),
                                                             b2f.data_ptr(), valid.lengths.data_ptr(), B, T, d,
                                                             valid.live_tiles.data_ptr(), valid.n_live, sc.data_ptr(),
                                                             bits.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()),
                           "star_scorer_fwd_ragged")
        legs_stream = torch.cuda.Stream()         # graph capture needs a non-default stream
        legs_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(legs_stream)
        evict = CacheEvictor(dev)
        warm_us = event_time_us(launch, iters=max(min(args.steps, 50), 10), graph=False)
        avg_us = event_time_us(launch, iters=max(min(args.steps, 50), 10), evict=evict)
        # algorithmic work: the valid frames only
        flops = float(N if valid is None else valid.n_valid) * (2.0 * d * d + 2.0 * d)
        ach = flops / (avg_us * 1e-6) / 1e12
        roofline = {"bound": "mfma", "kernel": "gemm_nt_kernel<EPI_SCORE> (+ scorer_finish_kernel)", "achieved": round(ach, 2),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(N, d) if valid is None else None, "avg_us": round(avg_us, 2),
                    "warm_us": round(warm_us, 2),
                    "timing": "avg_us / achieved: HIP events around each launch with L2 and the Infinity Cache flushed "
                              "before it (as inside the step, where X is cold); warm_us: launches back to back",
                    "flop_per_launch": flops,
                    "launch": "star_scorer_fwd" if valid is None else
                    ("star_scorer_fwd_compact: the %d valid of %d frames, gathered" % (valid.n_valid, N) if valid.compact else
                     "star_scorer_fwd_ragged: %d of %d row tiles computed, %d valid frames" % (valid.n_live, (N + 255) // 256,
                                                                                           valid.n_valid))}
        if not args.no_roofline_all:
            roofline_all = roofline_legs(ops, scorer, inp, wl, rate, dev, roofline, evict)
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream())

    if args.breakdown and rank == 0 and not simul:
        breakdown(scorer, inp, rate, dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, wl, args.cpu_seconds)

    if rank == 0:
        headline = args.config != 1 or (B, T, d, h, L, nl) == (32, 1500, 1024, 8, 128, 4)
        if args.config == 0:
            headline = (T, d, h, L, nl, rate) == (250, 768, 8, 24, 6, 4)
        elif args.config != 1:
            headline = (d, h, nl) == (1024, 8, 4)
        headline = headline and args.batch is None and args.frames is None and args.tgt is None and not args.staged
        n_dec = decoder_side.flat.numel() if decoder_side is not None else 0
        rec = {
            "metric": "encoder frames/sec, nugget scorer+select+cross-attn fwd+bwd",
            "value": frames_per_step / (elapsed / args.steps),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": wl["scaling"],
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": wl["name"] if headline else
                       "custom shape (parity-case configuration, not a BASELINE.json workload)",
                       "step": ("simultaneous (quantity loss, resize, fire scan, ILK %s)" % ("on" if wl["ilk"] else "off"))
                       if simul else "non-streaming (top-k select)",
                       "batch_per_gpu": B, "global_batch": wl.get("global_batch", B * world), "frames": T,
                       "valid_frames_per_gpu": int(frames_local), "d": d, "rate": None if simul else rate,
                       "K": int(L + 2) if simul else int(inp.key_len_dev(rate)[1]), "layers": nl, "heads": h, "tgt_len": L,
                       "parallelism": f"dp{world}", "hip_graph": graph_a is not None, "clock_ramp_steps": ramp_steps,
                       "loss_scalar": "not formed in the timed step: the cotangents G_l of loss = sum_l <O_l, G_l> are fed "
                                      "to backward directly (one reduction kernel per layer left out; tests form the scalar)",
                       "grad_allreduce": ("2 flat buffers per step: scorer (%d fp32 values) after the step; decoder "
                                          "cross-attention parameters (%d %s values) in one bucket per layer on a side "
                                          "stream, each started behind its layer's attention backward"
                                          % (allreduce.flat.numel(), n_dec, args.decoder_grad_dtype)) if world > 1 else None,
                       "allreduce_bytes": (4 * allreduce.flat.numel() + decoder_side.flat.element_size() * n_dec)
                       if world > 1 else None,
                       "decoder_grad_dtype": args.decoder_grad_dtype if world > 1 else None,
                       ("with_allreduce_%s_decoder_grads_ms_per_step" % ("bf16" if args.decoder_grad_dtype == "fp32" else "fp32")):
                           other_ms,
                       "compute_only_ms_per_step": compute_ms,
                       "with_allreduce_ms_per_step": ms_per_step if world > 1 else None,
                       "allreduce_ms_per_step": None if compute_ms is None else max(ms_per_step - compute_ms, 0.0)},
            "roofline": roofline,
            "roofline_all": roofline_all,
            "cpu_baseline": cpu,
        }
        print(json.dumps(rec), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def roofline_legs(ops, scorer, inp, wl, rate, dev, scorer_leg, evict):
    """One entry per kernel class of the path (north_star: MFMA utilisation for the scorer and the attention core,
    HBM GB/s for select / gather / scan).  Every leg is a stand-alone op timed with HIP events on the launch stream,
    twice: avg_us / achieved / frac with the caches flushed before every repeat (comparable with the kernel inside
    the step: profiles/*_kernel_stats.csv), warm_us replayed back to back from a HIP graph (operands L2 / Infinity-
    Cache warm).  achieved = ALGORITHMIC flops or bytes / avg_us."""
    B, T, d = inp.x.shape
    h, L = inp.qs[0].shape[1], inp.qs[0].shape[2]
    legs = [dict(scorer_leg, name="scorer forward GEMM")]
    simul = wl["kind"] == "simul"

    def leg(name, kernel, fn, flops=None, bytes_=None):
        warm = event_time_us(fn)
        us = event_time_us(fn, evict=evict)
        e = {"name": name, "kernel": kernel, "avg_us": round(us, 2), "warm_us": round(warm, 2)}
        if flops is not None:
            # the binding roof is the one whose time at peak is longer: flops / MFMA peak against bytes / HBM peak (the
            # attention core at these shapes has ~95 FLOP per byte, below the ~310 of the ridge: HBM binds it)
            tf = flops / (us * 1e-6) / 1e12
            gb = bytes_ / (us * 1e-6) / 1e9
            if bytes_ / (HBM_PEAK_GBS * 1e9) > flops / (MFMA_BF16_PEAK_TFLOPS * 1e12):
                e.update(bound="hbm", achieved=round(gb, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=round(gb / HBM_PEAK_GBS, 4))
            else:
                e.update(bound="mfma", achieved=round(tf, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                         frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4))
            e.update(mfma_tflops=round(tf, 2), mfma_frac=round(tf / MFMA_BF16_PEAK_TFLOPS, 4), flop_per_launch=flops,
                     hbm_gbs=round(gb, 1), hbm_frac=round(gb / HBM_PEAK_GBS, 4), bytes_per_launch=bytes_)
        else:
            gb = bytes_ / (us * 1e-6) / 1e9
            e.update(bound="hbm", achieved=round(gb, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gb / HBM_PEAK_GBS, 4),
                     bytes_per_launch=bytes_)
        legs.append(e)

    with torch.no_grad():
        scores = scorer(inp.x).squeeze(-1).float()
    if simul:
        alphas, _ = ops.cif_quantity(scores, inp.target_lengths_dev, lengths=inp.lengths_dev)
        resized = ops.cif_resize(alphas, inp.target_lengths_dev)
        cap = int(max(inp.target_lengths)) + 2
        fn = lambda: ops.cif_quantity(scores, inp.target_lengths_dev, lengths=inp.lengths_dev)
        leg("quantity (sigmoid, masked sum, L1 loss)", "cif_quantity_kernel", fn, bytes_=8.0 * B * T)
        fn = lambda: ops.cif_resize(alphas, inp.target_lengths_dev)
        leg("resize", "cif_resize_kernel", fn, bytes_=8.0 * B * T)
        fn = lambda: ops.cif_fire_scan(resized, lengths=inp.lengths_dev, cap=cap)
        leg("fire scan", "cif_fire_scan_kernel", fn, bytes_=4.0 * B * T + 9.0 * B * cap)
        fn = lambda: ops.cif_segment(scores, inp.target_lengths_dev, inp.lengths_dev, cap=cap)
        leg("segmenter as the step runs it: quantity + resize + fire scan + score gather + type table, one launch",
            "cif_segment_kernel (+ cif_quantity_loss_kernel)", fn, bytes_=4.0 * B * T * 3 + 8.0 * B * T + 21.0 * B * cap)
        idx, nmask, counts, _, _ = ops.cif_fire_scan(resized, lengths=inp.lengths_dev, cap=cap)
        key_len, K = counts, cap
        ilk = 2 if wl["ilk"] else -1
        kb = counts.cpu().numpy().astype(np.float64)
        lq = np.minimum(np.asarray(inp.target_lengths, dtype=np.float64), L)
    else:
        key_len, K = inp.key_len_dev(rate)
        fn = lambda: ops.topk_select(scores, lengths=inp.lengths_dev, rate=rate, K=K)
        leg("top-k select + type table", "topk_select_kernel", fn, bytes_=4.0 * B * T + 8.0 * B * T + 21.0 * B * K)
        _, idx, _, _, _ = ops.topk_select(scores, lengths=inp.lengths_dev, rate=rate, K=K)
        ilk = -1
        kb = np.asarray(ops.nugget_counts_host(inp.lengths, rate), dtype=np.float64)
        lq = np.full(B, float(L))
    # scorer backward (G' pass, dW1 GEMM + reductions, dX GEMM) on the rows the step's backward reaches
    import copy
    scorer_leg_mod = copy.deepcopy(scorer)     # fresh leaves: the step's parameters carry AccumulateGrad nodes of its stream
    xs = inp.x.detach().clone().requires_grad_(not simul)      # the simultaneous model scores detached encoder states
    sfull = scorer_leg_mod(xs, inp.valid).squeeze(-1)
    if simul:
        rows = inp.valid.rows if inp.valid is not None else torch.arange(B * T, dtype=torch.int32, device=dev)
    else:
        rows = ops.topk_select(scores, lengths=inp.lengths_dev, rate=rate, K=K, want_rows=True)[-1]
    gsc = torch.zeros(B * T, device=dev)
    gsc[rows.long()] = torch.randn(rows.numel(), device=dev)
    gsc = gsc.view(B, T)
    params = list(scorer_leg_mod.parameters())

    def scorer_bwd():
        scorer_leg_mod.set_active_rows(rows)         # single-use hint, as in the step
        return torch.autograd.grad(sfull, ([] if simul else [xs]) + params, gsc, retain_graph=True)
    fn = scorer_bwd
    R = float(rows.numel())
    if simul:
        leg("scorer backward over the valid frames (G' pass, dW1 GEMM + reductions)",
            "scorer_bwd_g_kernel, gemm_tn_kernel | gemm_nt_kernel<EPI_SLAB>, scorer_bwd_finish_kernel",
            fn, flops=R * (2.0 * d * d + 4.0 * d), bytes_=2.0 * d * R + 3 * 4.0 * d * d)
    else:
        leg("scorer backward over the selected frames (G' pass, dW1 GEMM + reductions, dX GEMM)",
            "scorer_bwd_g_kernel, gemm_tn_kernel | gemm_nt_kernel<EPI_SLAB>, gemm_nt_kernel<EPI_ROWS>, scorer_bwd_finish_kernel",
            fn, flops=R * (4.0 * d * d + 4.0 * d), bytes_=2.0 * d * R * 2 + 2.0 * d * B * T + 3 * 4.0 * d * d)
    sg = ops.gather_rows(scores, idx)
    s_det = inp.s.detach()
    fn = lambda: ops.gather_rows(s_det, idx)
    leg("row gather forward", "gather_rows_fwd_kernel", fn, bytes_=2.0 * 2 * d * B * K)
    sreq = s_det.clone().requires_grad_()
    mem = ops.gather_rows(sreq, idx)
    gm = torch.randn_like(mem)
    fn = lambda: torch.autograd.grad(mem, sreq, gm, retain_graph=True)
    leg("row gather backward (every dS row written once)", "gather_rows_bwd_kernel", fn,
        bytes_=2.0 * d * B * K + 2.0 * d * B * T)
    # attention core, one layer: all B x L x K pairs are computed for the ILK-off core, the lower band for ILK
    memd = mem.detach()
    k = memd.view(B, K, h, d // h).permute(0, 2, 1, 3)
    q = inp.qs[0].detach()
    if ilk < 0:
        pairs = float((lq * kb).sum())
    else:
        pairs = float(sum(min(i + ilk + 1, kk) for l_, kk in zip(lq, kb) for i in range(int(l_))))
    fwd_flops = 4.0 * pairs * d
    fwd_bytes = 2.0 * d * (2 * B * L + 2 * B * K)
    fn = lambda: ops.xattn(q, k, k, bias=sg, key_len=key_len, ilk_diag=ilk)
    leg("attention core forward (1 layer)", "xattn_fwd_ring_kernel", fn, flops=fwd_flops, bytes_=fwd_bytes)
    qg, kg, vg = q.clone().requires_grad_(), k.clone().requires_grad_(), k.clone().requires_grad_()
    sgg = sg.detach().clone().requires_grad_()
    o, _ = ops.xattn(qg, kg, vg, bias=sgg, key_len=key_len, ilk_diag=ilk)
    go = torch.randn(B, L, h, d // h, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
    fn = lambda: torch.autograd.grad(o, (qg, kg, vg, sgg), go, retain_graph=True)
    leg("attention core backward (1 layer, with the bias-gradient reduction)",
        "xattn_bwd_fused_small_kernel | xattn_bwd_dq_ring_kernel + xattn_bwd_dkv_*", fn,
        flops=2.5 * fwd_flops, bytes_=2.0 * d * (4 * B * L + 4 * B * K))
    return legs


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
