"""CPU checks: the C-ABI library loads and exports every symbol include/vitres_hip.h declares (no compute calls
without a GPU); host-side data (search spaces, MAC estimator, sub-net slicing) against the reference's golden vectors."""
import ast
import os
import re

import numpy as np
import torch

import recipe
import vitres
from vitres import _lib, supernet_config
from vitres.nets import net_utils
from vitres.network_utils import ComputationEstimator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vitres_hip.h")).read()
    declared = set(re.findall(r"^int\s+(vr_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.vr_version() >= 1000


def test_gemm_argument_marshalling_runs_without_a_gpu(monkeypatch):
    """vitres.kernels._gemm_args fills the ctypes struct of include/vitres_hip.h (the emulation used by the CPU host tests
    replaces K.gemm wholesale, so this is the only CPU coverage of the marshalling code: pointers are faked, nothing is launched)."""
    from vitres import kernels as K
    monkeypatch.setattr(K, "_p", lambda t: None if t is None else 0x1000)
    a, b, out = torch.zeros(8, 16, dtype=torch.bfloat16), torch.zeros(4, 16, dtype=torch.bfloat16), torch.zeros(8, 4)
    keep = torch.tensor([16, 8], dtype=torch.int32)
    K.M_GROUPS[0] = 2
    try:
        args = K._gemm_args(a, b, out, M=8, N=4, K=16, lda=16, ldb=16, ldc=4, keep_k=keep, rows_in=4, sched=16, atomic=2, ws=None)
    finally:
        K.M_GROUPS[0] = 1
    assert (args.M, args.N, args.K, args.atomic, args.m_groups, args.sched, args.in_dtype, args.out_dtype) == (8, 4, 16, 2, 2, 16, 1, 0)
    assert args.ws is None and args.ws_bytes == 0
    assert ctypes_sizeof_gemm_args() % 8 == 0


def ctypes_sizeof_gemm_args():
    import ctypes
    return ctypes.sizeof(_lib.GemmArgs)


def test_flop_mode_of_the_estimator_matches_reference():
    """ComputationEstimator(return_mac=False) (compute_flop_mac.py:53-194, 227-307) -- fixture F19: exact integers."""
    g = np.load(os.path.join(G, "f19_flops.npz"))
    nets = {"ref_tiny": recipe.REF_TINY_DEF, "sr_tiny": recipe.SR_TINY_DEF, "sr_small": recipe.SR_SMALL_DEF,
            "sr_tiny_mh": recipe.SR_TINY_MH_DEF, "sr_small_mh": recipe.SR_SMALL_MH_DEF}
    for name, nd in nets.items():
        for distill in (False, True):
            est = ComputationEstimator(distill=distill, input_resolution=224, patch_size=14, return_mac=False)
            assert est(nd) == int(g["%s.flops%s" % (name, "_distill" if distill else "")]), (name, distill)
    for i, nd in enumerate(recipe.MICRO_CANDIDATES):
        assert ComputationEstimator(False, 56, 14, return_mac=False)(nd) == int(g["micro_cand%d.flops" % i])
    vit_t = ((0, 192),) + ((1, (192, 3, 64), (192, 768), 1),) * 12 + ((2, 192, 1000),)
    assert ComputationEstimator(True, 224, 16)(vit_t) == int(g["vit_t_p16.macs"]) == 1261003776       # the reference's own print
    assert ComputationEstimator(True, 224, 16, return_mac=False)(vit_t) == int(g["vit_t_p16.flops"])
    assert "return_mac=False" in repr(ComputationEstimator(True, 224, 16, return_mac=False))


def test_search_spaces_and_macs_match_reference():
    g = np.load(os.path.join(G, "f6_schema_macs.npz"))
    est = ComputationEstimator(distill=False, input_resolution=224, patch_size=14)
    for name, nd in (("sr_tiny", recipe.SR_TINY_DEF), ("sr_small", recipe.SR_SMALL_DEF), ("sr_tiny_mh", recipe.SR_TINY_MH_DEF),
                     ("sr_small_mh", recipe.SR_SMALL_MH_DEF)):
        sp = getattr(supernet_config, name)
        assert sp.network_def == nd
        flat = []
        for ent in sp.num_channels_to_keep:
            if ent is None:
                flat.append("None")
            elif isinstance(ent, dict):
                flat.append(str({k: (None if v is None else [int(a) for a in v]) for k, v in ent.items()}))
            else:
                flat.append(str([int(a) for a in ent]))
        assert flat == list(g[name + ".choices"])
        assert est(nd) == int(g[name + ".macs"])
    assert est(recipe.REF_TINY_DEF) == int(g["ref_tiny.macs"]) == 1794400000 + (int(g["ref_tiny.macs"]) - 1794400000)
    assert abs(est(recipe.REF_TINY_DEF) - 1.7944e9) < 1e5
    est56 = ComputationEstimator(distill=False, input_resolution=56, patch_size=14)
    for i, nd in enumerate(recipe.MICRO_CANDIDATES):
        assert est56(nd) == int(g["micro_cand%d.macs" % i])


def test_state_dict_schema_and_sub_state_dict():
    g = np.load(os.path.join(G, "f6_schema_macs.npz"))
    with torch.device("meta"):
        m = vitres.create_model("flexible_vit_sr_patch14_224_patch_output", num_classes=1000, network_def=recipe.REF_TINY_DEF)
    assert list(m.state_dict().keys()) == list(g["ref_tiny.keys"])
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == list(g["ref_tiny.shapes"])
    assert m.no_weight_decay() == {"tokens"}
    g5 = np.load(os.path.join(G, "f5_subnet.npz"))
    sup = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=56, num_classes=10,
                              network_def=recipe.MICRO_DEFS[0], num_channels_to_keep=recipe.micro_keep_config(),
                              example_per_arch=2, num_warmup_epochs=30)
    sd = recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in sup.state_dict().items()], 100)
    sup.load_state_dict(sd)
    for i, nd in enumerate(recipe.MICRO_CANDIDATES):
        sub = vitres.create_model("flexible_vit_sr_patch14_224_patch_output", img_size=56, num_classes=10, network_def=nd)
        ssd = net_utils.get_sub_state_dict(sup.state_dict(), sub.state_dict())
        assert recipe.checksum(ssd) == int(g5["cand%d.crc" % i])
        sub.load_state_dict(ssd)


def test_rewiring_matches_reference():
    g = np.load(os.path.join(G, "f7_rewiring.npz"))
    m = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=56, num_classes=10,
                            network_def=recipe.MICRO_DEFS[0], num_channels_to_keep=recipe.micro_keep_config(),
                            example_per_arch=2, num_warmup_epochs=30)
    m.load_state_dict(recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()], 100))
    m.set_epoch(0)
    after = m.state_dict()
    n = 0
    for k in g.files:
        if k.startswith("blocks."):
            assert np.array_equal(after[k].numpy(), g[k]), k
            n += 1
    assert n == 12


def test_channel_drop_tables_and_rng_protocol():
    from vitres.nets.channel_drop import ChannelDrop
    g = np.load(os.path.join(G, "f3_channel_drop.npz"))
    for ci, case in enumerate(g["cases"]):
        choices, B, epa, warm, single = ast.literal_eval(str(case))
        for e in (0, 8, 15, 30, 31):
            cd = ChannelDrop(np.array(choices), num_warmup_epochs=warm, example_per_arch=epa, single_arch=single)
            cd.train()
            cd.set_epoch(e)
            torch.manual_seed(1000 + ci * 10 + e)
            draws = np.stack([cd.sample_keep(B, max(choices)).numpy() for _ in range(3)])
            tag = "c%d.e%d." % (ci, e)
            assert np.array_equal(draws, g[tag + "draws"])
            assert list(cd.table.numpy()) == list(g[tag + "table"]) and cd.num_layer_config == int(g[tag + "nlc"])


def test_cosine_scheduler_known_answers():
    """main.py:388 create_scheduler (--sched cosine, --epochs 120, lr 5e-4, --warmup-lr 1e-6, --min-lr 1e-5, 5 warm-up epochs,
    10 cool-down epochs): five closed-form points, the two parameter groups of add_weight_decay, checkpoint round trip."""
    import argparse
    import math
    import torch
    from vitres.scheduler import create_scheduler
    w = [torch.zeros(2, requires_grad=True), torch.zeros(2, requires_grad=True)]
    opt = torch.optim.AdamW([{"params": [w[0]], "weight_decay": 0.0}, {"params": [w[1]], "weight_decay": 0.05}], lr=5e-4)
    args = argparse.Namespace(epochs=120, sched="cosine", min_lr=1e-5, warmup_lr=1e-6, warmup_epochs=5, cooldown_epochs=10,
                              decay_rate=0.1)
    sched, n_epochs = create_scheduler(args, opt)
    assert n_epochs == 130
    assert [g["lr"] for g in opt.param_groups] == [1e-6, 1e-6]                 # starts at the warm-up rate
    want = {0: 1e-6, 3: 1e-6 + 3 * (5e-4 - 1e-6) / 5, 5: 1e-5 + 0.5 * 4.9e-4 * (1 + math.cos(math.pi * 5 / 120)),
            60: 1e-5 + 0.5 * 4.9e-4, 119: 1e-5 + 0.5 * 4.9e-4 * (1 + math.cos(math.pi * 119 / 120)), 120: 1e-5, 129: 1e-5}
    for e, v in want.items():
        sched.step(e)
        assert all(abs(g["lr"] - v) < 1e-12 for g in opt.param_groups), (e, opt.param_groups[0]["lr"], v)
    sd = sched.state_dict()
    assert "optimizer" not in sd
    opt2 = torch.optim.AdamW([{"params": [w[0]]}, {"params": [w[1]]}], lr=1.0)
    s2, _ = create_scheduler(argparse.Namespace(epochs=7), opt2)
    s2.load_state_dict(sd)
    s2.step(60)
    assert abs(opt2.param_groups[1]["lr"] - want[60]) < 1e-12


def test_sync_ranges_tile_the_gradient_arena_for_every_shipped_network():
    """the arena ranges of `split_for_sync=3` (engine.GraphedTrainStep: range k is all-reduced after backward
    part k) tile the flat gradient arena exactly -- no gap, no overlap, every parameter in exactly one range -- for the five
    shipped networks (reference main.py:366-367: DDP reduces every gradient once)."""
    import importlib.util
    import vitres
    from vitres import supernet_config
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    nets = [("flexible_vit_sr_patch14_224_patch_output", dict(network_def=bench.REF_TINY_DEF))]
    for space in ("sr_tiny", "sr_tiny_mh", "sr_small", "sr_small_mh"):
        sp = getattr(supernet_config, space)
        nets.append(("flexible_vit_sr_patch14_224_patch_output_supernet",
                     dict(network_def=sp.network_def, num_channels_to_keep=sp.num_channels_to_keep, example_per_arch=8,
                          num_warmup_epochs=30)))
    for name, kw in nets:
        model = vitres.create_model(name, num_classes=1000, **kw)
        model._ensure_arena(torch.device("cpu"))
        a = model._arena
        n = a["flat"].numel()
        cuts = model.split_plan(parts=3)
        assert cuts and len(cuts) == 2, (name, cuts)
        starts = [s for _, s in cuts]                       # descending: range k = [starts[k], starts[k - 1])
        assert n > starts[0] > starts[1] > 0
        ranges = [(starts[0], n), (starts[1], starts[0]), (0, starts[1])]
        assert sorted(ranges)[0][0] == 0 and sorted(ranges)[-1][1] == n
        for (lo0, hi0), (lo1, hi1) in zip(sorted(ranges)[:-1], sorted(ranges)[1:]):
            assert hi0 == lo1                               # no gap, no overlap
        covered = 0
        for p, (off, cnt) in zip(a["params"], a["offsets"]):
            inside = [lo <= off and off + p.numel() <= hi for lo, hi in ranges]
            assert sum(inside) == 1, (name, off)            # no parameter straddles a cut
            covered += p.numel()
        assert covered == sum(p.numel() for p in model.parameters()) <= n
        # the blocks behind a cut are exactly the parameters of its range
        blocks = list(model.blocks)
        for (cut, start) in cuts:
            for m in blocks[cut:]:
                for p in m.parameters():
                    assert a["offsets"][a["index"][id(p)]][0] >= start
            for m in blocks[:cut]:
                for p in m.parameters():
                    assert a["offsets"][a["index"][id(p)]][0] < start


def test_plan_marks_for_write_skipping():
    """Round 5 host logic (no GPU): _Plan.skip_writes is set only when every kernel tile of the batch sees one architecture (G
    contiguous groups, or one architecture for the whole batch), the network's masked widths are multiples of the 64-wide slice and the
    bf16 kernels run; the rows the KERNELS read mark a DropPath-dropped sample of a live layer as -(k + 2) (k = its group's width),
    keep a dropped layer's 0, and leave the sampled keeps reported to the caller alone."""
    sp = supernet_config.sr_tiny

    def make(epa, nd=sp.network_def, cfg=sp.num_channels_to_keep, img=224, classes=1000):
        with torch.device("meta"):
            m = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=img, num_classes=classes, network_def=nd,
                                    drop_path_rate=0.4, num_channels_to_keep=cfg, example_per_arch=epa, num_warmup_epochs=30)
        m.train()
        m.set_epoch(31)
        return m
    m = make(8)
    torch.manual_seed(5)
    plan = m.sample_plan(32)
    assert plan.groups == 4 and plan.skip_writes
    assert make(32).sample_plan(32).skip_writes                       # one architecture for the whole batch
    assert not make(1).sample_plan(32).skip_writes                    # an architecture per sample: tiles mix them
    m32 = make(8)
    m32.compute_dtype = torch.float32
    assert not m32.sample_plan(32).skip_writes                        # exact-fp32 kernels: no skipping forms
    micro = make(2, nd=recipe.MICRO_DEFS[0], cfg=recipe.micro_keep_config(), img=recipe.MICRO_IMG, classes=recipe.MICRO_CLASSES)
    assert not micro.sample_plan(8).skip_writes                       # widths of 32 / 48: not the lean kernels' forms
    m.drop_path_generator(seed=3)
    seen_mark = seen_dead = False
    for seed in range(6):
        torch.manual_seed(40 + seed)
        plan = m.sample_plan(32)
        reported = [k.clone() for k in m.last_keeps]
        flat, nk = m.plan_host_buffer(plan)
        kh = flat[:nk].reshape(plan.keeps_host.shape)
        for L in plan.layers:
            if L is None or L.get("attn") is None or L.get("dp") is None:
                continue
            for row, srow in ((L["attn"], L["dp"]), (L["mlp"], L["dp"] + 1)):
                sampled, sent = plan.keeps_host[row], kh[row]
                dropped = plan.scales_host[srow] == 0
                live = sampled > 0
                assert (sent[~dropped] == sampled[~dropped]).all()
                assert (sent[dropped & live] == -sampled[dropped & live] - 2).all()
                assert (sent[~live] == 0).all()
                seen_mark |= bool((dropped & live).any())
                seen_dead |= bool((~live).any())
        assert all(torch.equal(a, b) for a, b in zip(reported, m.last_keeps))
    assert seen_mark and seen_dead
