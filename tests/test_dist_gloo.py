"""N > 1 data-parallel path on CPU: world_size 2 over gloo, kernels emulated (tests/emu_kernels.py).
Checks the reference's DDP semantics (main.py:366-367): parameters broadcast from rank 0, gradients averaged over
ranks, identical parameters on every rank after the optimizer step, per-rank seeds -> different sampled archs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Patch:
    def setattr(self, obj, name, value, raising=True):
        setattr(obj, name, value)


def _worker(rank, world, port, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vit-search_amd"), HERE,
              os.path.join(HERE, "golden"), os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_kernels
    import recipe
    import vitres
    from vitres import engine
    emu_kernels.install(_Patch())
    torch.manual_seed(100 + rank)                      # reference: seed + rank (main.py:261-267)
    model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=recipe.MICRO_IMG,
                                num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[0], drop_path_rate=0.0,
                                num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
    model.set_compute_dtype(torch.float32)
    model._ensure_arena(torch.device("cpu"))
    sync = engine.GradSync(model)
    sync.broadcast_parameters()
    p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    model.train()
    model.set_epoch(31)
    x, t, pt, _ = recipe.inputs(500 + rank, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    crit = lambda a, b: torch.sum(-b * torch.log_softmax(a, -1), -1).mean()   # noqa: E731
    # local gradients first (no exchange), then the real step from the same CPU RNG state (same masks)
    rng = torch.random.get_rng_state()
    cls, pat = model(x, patch_output_type="seq")
    (crit(cls, t) + crit(pat, pt)).backward()
    keeps = torch.stack(model.last_keeps).clone()
    g_local = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    # the split exchange of engine.GraphedTrainStep.step_with_sync: tail range while part 2 of the backward runs, then
    # the rest -- same result as the single all-reduce
    torch.random.set_rng_state(rng)
    cut, start = model.split_plan()
    cls, pat = model(x, patch_output_type="seq")
    model._bwd_split = cut
    (crit(cls, t) + crit(pat, pt)).backward()
    model._bwd_split = None
    works = [sync.all_reduce_range(start, model._arena["gcur"].numel())]
    model.resume_backward()
    works.append(sync.all_reduce_range(0, start))
    sync.finish(works)
    g_split = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    # three parts: one exchange per completed arena range
    torch.random.set_rng_state(rng)
    cuts = model.split_plan(parts=3)
    cls, pat = model(x, patch_output_type="seq")
    model._bwd_split = [c for c, _ in cuts]
    (crit(cls, t) + crit(pat, pt)).backward()
    model._bwd_split = None
    works, end = [], model._arena["gcur"].numel()
    for _, st in cuts:
        works.append(sync.all_reduce_range(st, end))
        end = st
        model.resume_backward()
    works.append(sync.all_reduce_range(0, end))
    sync.finish(works)
    g_split3 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    # the same three-part exchange with bf16 on the wire (GradSync(wire_dtype=torch.bfloat16)): half the bytes per link
    sync16 = engine.GradSync(model, wire_dtype=torch.bfloat16)
    torch.random.set_rng_state(rng)
    cls, pat = model(x, patch_output_type="seq")
    model._bwd_split = [c for c, _ in cuts]
    (crit(cls, t) + crit(pat, pt)).backward()
    model._bwd_split = None
    works, end = [], model._arena["gcur"].numel()
    for _, st in cuts:
        works.append(sync16.all_reduce_range(st, end))
        end = st
        model.resume_backward()
    works.append(sync16.all_reduce_range(0, end))
    sync16.finish(works)
    g_bf16 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    torch.random.set_rng_state(rng)
    cls, pat = model(x, patch_output_type="seq")
    (crit(cls, t) + crit(pat, pt)).backward()
    sync16.all_reduce_grads()                                       # ... and as one blocking call
    g_bf16_one = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    torch.random.set_rng_state(rng)
    engine.train_step(model, crit, opt, x, t, pt, "seq", epoch=31, train_iter=0, arch_sample="multi", grad_sync=sync)
    g_sync = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()
    p1 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    # the reference signature always passes a NativeScaler (engine.py:175-177): with several ranks the exchange must still happen
    class Scaler:                                    # timm.utils.NativeScaler's shape: a torch GradScaler under `_scaler`
        def __init__(self):
            self._scaler = torch.amp.GradScaler("cpu", enabled=False)
    torch.random.set_rng_state(rng)
    engine.train_step(model, crit, opt, x, t, pt, "seq", epoch=31, train_iter=0, arch_sample="multi", grad_sync=sync,
                      loss_scaler=Scaler())
    p2 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    torch.save({"p0": p0, "p1": p1, "p2": p2, "g_local": g_local, "g_sync": g_sync, "keeps": keeps, "g_split": g_split,
                "g_split3": g_split3, "g_bf16": g_bf16, "g_bf16_one": g_bf16_one}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2))
    assert torch.equal(r0["p0"], r1["p0"])                                # rank-0 parameters everywhere
    assert not torch.equal(r0["keeps"], r1["keeps"])                      # different archs per rank (seed + rank)
    assert torch.allclose(r0["g_sync"], r1["g_sync"], rtol=0, atol=0)     # identical after the all-reduce
    assert torch.equal(r0["p1"], r1["p1"])                                # and identical parameters after the step
    # the exchanged gradient is the mean of what each rank computes alone for ITS masks; the second forward of a rank
    # re-samples from the same RNG state (train_step restores the CPU RNG), so local grads are comparable
    mean = 0.5 * (r0["g_local"] + r1["g_local"])
    err = float((r0["g_sync"] - mean).abs().max() / mean.abs().max())
    assert err < 1e-5, err
    assert torch.equal(r0["g_split"], r1["g_split"])
    assert float((r0["g_split"] - r0["g_sync"]).abs().max() / r0["g_sync"].abs().max()) < 1e-6
    assert torch.equal(r0["g_split3"], r1["g_split3"]) and torch.equal(r0["g_split3"], r0["g_split"])
    assert torch.equal(r0["p2"], r1["p2"]) and not torch.equal(r0["p2"], r0["p1"])      # loss_scaler + grad_sync: replicas stay equal
    # bf16 on the wire: both ranks hold the same values; against the fp32 exchange every element is within three bf16 roundings
    # (unit roundoff u = 2^-8: each rank's value, then their sum -- at most u (|a| + |b| + |a + b|) / 2 <= 2 u mean(|a|, |b|) on the
    # averaged gradient), and within 2^-8 in the L2 norm
    assert torch.equal(r0["g_bf16"], r1["g_bf16"]) and torch.equal(r0["g_bf16"], r0["g_bf16_one"])
    ref16 = r0["g_split3"]
    scale = 0.5 * (r0["g_local"].abs() + r1["g_local"].abs())
    assert float(((r0["g_bf16"] - ref16).abs() - 2 * 2.0 ** -8 * scale).max()) <= 1e-9
    assert float((r0["g_bf16"] - ref16).norm() / ref16.norm()) < 2.0 ** -8
    assert not torch.equal(r0["g_bf16"], ref16)


def _buffer_worker(rank, world, port, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vit-search_amd"), HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import recipe
    import vitres
    from vitres import engine
    torch.manual_seed(7 + rank)
    model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output", img_size=recipe.MICRO_IMG,
                                num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[4])
    sync = engine.GradSync(model)
    sync.broadcast_parameters()
    with torch.no_grad():
        for b in model.buffers():                                  # ranks drift apart (each normalises its own batch) ...
            if b.is_floating_point():
                b.add_(float(rank + 1))
            else:
                b.add_(rank + 3)
    before = [b.clone() for b in model.buffers()]
    sync.broadcast_buffers()                                       # ... until the next forward's DDP buffer broadcast (X4)
    torch.save({"before": before, "after": [b.clone() for b in model.buffers()]}, os.path.join(out_dir, "b%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_buffer_broadcast_per_forward(tmp_path):
    """DDP broadcast_buffers=True (main.py:367): rank 0's BatchNorm running statistics replace the other ranks'."""
    mp.spawn(_buffer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "b%d.pt" % r)) for r in range(2))
    assert len(r0["after"]) == 9                                   # 3 x (running_mean, running_var, num_batches_tracked)
    for a0, a1, b0, b1 in zip(r0["after"], r1["after"], r0["before"], r1["before"]):
        assert torch.equal(a0, b0) and torch.equal(a1, b0) and not torch.equal(b1, b0)


def _search_worker(rank, world, port, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vit-search_amd"), HERE,
              os.path.join(HERE, "golden"), os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_kernels
    import recipe
    import vitres
    from vitres import evo_search
    from vitres.network_utils.compute_flop_mac import ComputationEstimator
    emu_kernels.install(_Patch())
    nd, keep = recipe.MICRO_DEFS[0], recipe.micro_keep_config()
    sup = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=recipe.MICRO_IMG,
                              num_classes=recipe.MICRO_CLASSES, network_def=nd, num_channels_to_keep=keep, example_per_arch=2,
                              num_warmup_epochs=30)
    sup.load_state_dict(recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in sup.state_dict().items()], 100))
    sup.set_compute_dtype(torch.float32).eval()
    batches = []
    for s in (9, 10):
        x, _, _, labels = recipe.inputs(s, 6, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        batches.append((x, labels))
    est = ComputationEstimator(distill=False, input_resolution=recipe.MICRO_IMG, patch_size=14)
    best = evo_search.search(sup, batches, nd, keep, 0.8 * est(nd), search_iter=2, init_popu_size=5, parent_size=3, mutate_size=2,
                             mutate_prob=0.3, input_size=recipe.MICRO_IMG, output_dir=os.path.join(out_dir, "w%d" % world), seed=0)
    torch.save([(b.network_def, b.score) for b in best], os.path.join(out_dir, "best_w%d_r%d.pt" % (world, rank)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_search_shards_candidates_and_agrees_with_one_rank(tmp_path):
    """evo_search.search on 2 ranks (candidates dealt round robin, scores summed over ranks -- evo_eval.score_population)
    selects the same winners with the same scores as one rank; only rank 0 writes the result files."""
    mp.spawn(_search_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_search_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = torch.load(os.path.join(str(tmp_path), "best_w1_r0.pt"))
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "best_w2_r%d.pt" % r)) for r in range(2))
    assert one == r0 == r1
    assert open(os.path.join(str(tmp_path), "w1", "summary.txt")).read() == open(os.path.join(str(tmp_path), "w2", "summary.txt")).read()


def _rng_worker(rank, world, port, out_dir):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "vit-search_amd"), HERE,
              os.path.join(HERE, "golden"), os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import recipe
    import vitres
    from vitres import checkpoint

    def make():
        return vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=recipe.MICRO_IMG,
                                   num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[0], drop_path_rate=0.1,
                                   num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
    torch.manual_seed(7)                               # the SAME global seed on both ranks: the generator adds the rank itself
    m = make()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    gen = m.drop_path_generator()
    torch.rand(3 + rank, generator=gen)                # every rank somewhere else in its own stream
    mine = m.drop_path_rng_state().clone()
    states = checkpoint.collect_rng_states(m)          # collective
    full = checkpoint.checkpoint_dict(m, opt, None, 3, rng_states=states)
    solo = checkpoint.checkpoint_dict(m, opt, None, 3)                  # what rank 0 alone would have written
    box = [solo if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    solo0 = box[0]
    res = {}
    # (a) per-rank list: every rank continues its own stream exactly
    m2 = make()
    checkpoint.resume(full, m2, torch.optim.AdamW(m2.parameters(), lr=1e-3))
    res["own"] = bool(torch.equal(m2.drop_path_rng_state(), mine))
    # (b) rank 0's state alone: rank 0 continues exactly, rank 1 gets a derived stream -- not rank 0's
    m3 = make()
    checkpoint.resume(solo0, m3, torch.optim.AdamW(m3.parameters(), lr=1e-3))
    st3 = m3.drop_path_rng_state()
    res["solo_same_as_rank0"] = bool(torch.equal(st3, solo0["vitres_rng"]["drop_path"]))
    res["draw"] = torch.rand(4, generator=m3.drop_path_generator()).tolist()
    torch.save(res, os.path.join(out_dir, "rng%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_resume_keeps_per_rank_drop_path_streams(tmp_path):
    """a checkpoint's DropPath generator state must not make every rank draw the same noise after a resume
    (reference: per-rank noise from seed + rank, main.py:261-267)."""
    mp.spawn(_rng_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "rng%d.pt" % r)) for r in (0, 1))
    assert r0["own"] and r1["own"]
    assert r0["solo_same_as_rank0"] and not r1["solo_same_as_rank0"]
    assert r0["draw"] != r1["draw"]
