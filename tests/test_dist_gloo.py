"""world_size-2 gloo tests of the N>1 path: sharded utterances + the flat gradient all-reduces give the
single-process gradients (utterances are independent, so this is exact up to summation order).  The second test
covers bench.py's two-buffer step: a two-stage backward with the decoder-side buffer reduced between the stages."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scorer_grads(x, lengths, rate, params):
    """CPU stand-in for a rank's NuggetStep: oracle scorer + oracle select; loss = sum of selected scores^2."""
    from oracle import nugget_oracle as orc
    w1, b1, w2, b2 = params
    t = x.shape[1]
    mask = np.arange(t)[None, :] < np.asarray(lengths)[:, None]
    s = orc.scorer(x, w1, b1, w2, b2)
    nmask, idx, _ = orc.topk_select(s.detach().numpy(), mask, rate)
    sg = s.gather(1, torch.from_numpy(idx)) * torch.from_numpy(nmask)
    return (sg ** 2).sum()


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from star_amd import dist as sd
    r, _, w = sd.init(backend="gloo")
    assert (r, w) == (rank, world)
    data = torch.load(os.path.join(tmp, "data.pt"))
    mine = sd.shard(list(range(len(data["lengths"]))), rank, world)
    params = [torch.nn.Parameter(p.clone()) for p in data["params"]]
    ar = sd.FlatGradAllReduce(params)
    if data.get("attach"):
        ar.attach()            # gradients accumulate straight into the flat buffer
        ar.zero()
    loss = _scorer_grads(data["x"][mine], [data["lengths"][i] for i in mine], 4, params)
    loss.backward()
    if data.get("attach"):
        assert all(p.grad.data_ptr() >= ar.flat.data_ptr() for p in params) and ar._still_attached()
    ar(world)
    sd.barrier()
    if rank == 0:
        torch.save([p.grad for p in params], os.path.join(tmp, "grads.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("attach", [False, True])
def test_sharded_allreduce_matches_single_process(tmp_path, attach):
    torch.manual_seed(0)
    d, t, b = 16, 40, 6
    lengths = [40, 33, 21, 40, 8, 17]
    x = torch.randn(b, t, d)
    params = [torch.randn(d, d) * 0.2, torch.randn(d) * 0.1, torch.randn(1, d) * 0.2, torch.randn(1) * 0.1]
    torch.save({"x": x, "lengths": lengths, "params": params, "attach": attach}, tmp_path / "data.pt")
    port = 29611 + os.getpid() % 200 + (300 if attach else 0)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "grads.pt")
    full = [torch.nn.Parameter(p.clone()) for p in params]
    _scorer_grads(x, lengths, 4, full).backward()
    for g, p in zip(got, full):
        # DDP semantics: mean over ranks of per-rank sums == full-batch sum / world
        np.testing.assert_allclose(g.numpy(), (p.grad / 2).numpy(), rtol=1e-5, atol=1e-6)


def _two_stage_loss(x, lengths, rate, params, dec):
    """Stand-in for the step with a decoder side: selected scores -> (scores * dec) 'attention' -> loss.  Returns
    (loss, sg): sg is the boundary between the two backward stages, like scores_gather in star_amd.step."""
    from oracle import nugget_oracle as orc
    w1, b1, w2, b2 = params
    t = x.shape[1]
    mask = np.arange(t)[None, :] < np.asarray(lengths)[:, None]
    s = orc.scorer(x, w1, b1, w2, b2)
    nmask, idx, _ = orc.topk_select(s.detach().numpy(), mask, rate)
    sg = s.gather(1, torch.from_numpy(idx)) * torch.from_numpy(nmask)
    out = torch.tanh(sg.unsqueeze(-1) * dec.unsqueeze(0).unsqueeze(0))        # [b, k, m]
    return (out ** 2).sum(), sg


def _worker2(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from star_amd import dist as sd
    sd.init(backend="gloo")
    data = torch.load(os.path.join(tmp, "data.pt"))
    mine = sd.shard(list(range(len(data["lengths"]))), rank, world)
    params = [torch.nn.Parameter(p.clone()) for p in data["params"]]
    dec = torch.nn.Parameter(data["dec"].clone())
    ar = sd.FlatGradAllReduce(params).attach()
    ar.zero()
    dec_flat = torch.zeros(dec.numel())
    side = sd.SideStreamAllReduce(dec_flat)
    loss, sg = _two_stage_loss(data["x"][mine], [data["lengths"][i] for i in mine], 4, params, dec)
    # stage 1: "attention backward": decoder-side gradients + the cotangent of the boundary
    dsg, ddec = torch.autograd.grad(loss, [sg, dec])
    dec_flat.copy_(ddec.reshape(-1))
    side.start(world)                              # reduced while stage 2 runs (in order on the host under gloo)
    torch.autograd.backward([sg], [dsg])           # stage 2: scorer backward into the attached flat buffer
    ar(world)
    side.join()
    frames = sd.sum_over_ranks(float(sum(data["lengths"][i] for i in mine)), "cpu")
    assert frames == float(sum(data["lengths"]))
    sd.barrier()
    if rank == 0:
        torch.save({"params": [p.grad for p in params], "dec": dec_flat}, os.path.join(tmp, "grads.pt"))
    torch.distributed.destroy_process_group()


def test_two_buffer_two_stage_backward_matches_single_process(tmp_path):
    torch.manual_seed(1)
    d, t, b, m = 16, 40, 5, 7
    lengths = [40, 12, 29, 40, 9]
    x = torch.randn(b, t, d)
    params = [torch.randn(d, d) * 0.2, torch.randn(d) * 0.1, torch.randn(1, d) * 0.2, torch.randn(1) * 0.1]
    dec = torch.randn(m) * 0.5
    torch.save({"x": x, "lengths": lengths, "params": params, "dec": dec}, tmp_path / "data.pt")
    port = 30311 + os.getpid() % 200
    mp.spawn(_worker2, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "grads.pt")
    full = [torch.nn.Parameter(p.clone()) for p in params]
    fdec = torch.nn.Parameter(dec.clone())
    _two_stage_loss(x, lengths, 4, full, fdec)[0].backward()
    for g, p in zip(got["params"], full):
        np.testing.assert_allclose(g.numpy(), (p.grad / 2).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["dec"].numpy(), (fdec.grad / 2).numpy(), rtol=1e-5, atol=1e-6)


def _layered_loss(x, lengths, rate, params, decs):
    """Stand-in with nl decoder layers fed from one boundary: returns (per-layer losses, per-layer boundary copies, sg)."""
    from oracle import nugget_oracle as orc
    w1, b1, w2, b2 = params
    t = x.shape[1]
    mask = np.arange(t)[None, :] < np.asarray(lengths)[:, None]
    s = orc.scorer(x, w1, b1, w2, b2)
    nmask, idx, _ = orc.topk_select(s.detach().numpy(), mask, rate)
    sg = s.gather(1, torch.from_numpy(idx)) * torch.from_numpy(nmask)
    feeds = [sg * 1.0 for _ in decs]             # the per-layer inputs this side of the shared node (like K_l / V_l)
    losses = [(torch.tanh(f.unsqueeze(-1) * dd.unsqueeze(0).unsqueeze(0)) ** 2).sum() for f, dd in zip(feeds, decs)]
    return losses, feeds, sg


def _worker3(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from star_amd import dist as sd
    sd.init(backend="gloo")
    data = torch.load(os.path.join(tmp, "data.pt"))
    mine = sd.shard(list(range(len(data["lengths"]))), rank, world)
    params = [torch.nn.Parameter(p.clone()) for p in data["params"]]
    decs = [torch.nn.Parameter(dd.clone()) for dd in data["decs"]]
    nl, m = len(decs), decs[0].numel()
    ar = sd.FlatGradAllReduce(params).attach()
    ar.zero()
    side = sd.BucketedSideAllReduce(torch.zeros(nl * m), nl)
    losses, feeds, sg = _layered_loss(data["x"][mine], [data["lengths"][i] for i in mine], 4, params, decs)
    # bench.py's N > 1 step: layer by layer (last first), a bucket started behind every layer, then the rest
    dfeeds = [None] * nl
    for l in reversed(range(nl)):
        dfeeds[l], ddec = torch.autograd.grad(losses[l], [feeds[l], decs[l]])
        side.buckets[l].copy_(ddec.reshape(-1))
        side.start(l, world)
    torch.autograd.backward(feeds, dfeeds)
    ar(world)
    side.join()
    sd.barrier()
    if rank == 0:
        torch.save({"params": [p.grad for p in params], "dec": side.flat}, os.path.join(tmp, "grads.pt"))
    torch.distributed.destroy_process_group()


def test_bucket_per_layer_backward_matches_single_process(tmp_path):
    torch.manual_seed(2)
    d, t, b, m, nl = 16, 40, 5, 6, 3
    lengths = [40, 12, 29, 40, 9]
    x = torch.randn(b, t, d)
    params = [torch.randn(d, d) * 0.2, torch.randn(d) * 0.1, torch.randn(1, d) * 0.2, torch.randn(1) * 0.1]
    decs = [torch.randn(m) * 0.5 for _ in range(nl)]
    torch.save({"x": x, "lengths": lengths, "params": params, "decs": decs}, tmp_path / "data.pt")
    port = 30711 + os.getpid() % 200
    mp.spawn(_worker3, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "grads.pt")
    full = [torch.nn.Parameter(p.clone()) for p in params]
    fdecs = [torch.nn.Parameter(dd.clone()) for dd in decs]
    sum(_layered_loss(x, lengths, 4, full, fdecs)[0]).backward()
    for g, p in zip(got["params"], full):
        np.testing.assert_allclose(g.numpy(), (p.grad / 2).numpy(), rtol=1e-5, atol=1e-6)
    ref = torch.cat([dd.grad.reshape(-1) for dd in fdecs]) / 2
    np.testing.assert_allclose(got["dec"].numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        from star_amd import dist as sd
        sd.BucketedSideAllReduce(torch.zeros(7), 2)


def test_shard_partitions_everything():
    from star_amd import dist as sd
    items = list(range(13))
    for world in (1, 2, 4, 8):
        parts = [sd.shard(items, r, world) for r in range(world)]
        assert sum(parts, []) == items
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
