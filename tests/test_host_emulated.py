"""Host-side orchestration of the product model (plan sampling, tape, gradient routing) checked on CPU by
swapping the HIP kernels for tests/emu_kernels.py.  The real kernels are checked on the GPU (tests/test_gpu_*.py)."""
import os

import numpy as np
import pytest
import torch

import emu_kernels
import recipe
import vitres
import vitres_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_pair(et, mode, seed):
    nd = recipe.MICRO_DEFS[et]
    sup = mode != "plain"
    kw = {}
    if sup:
        kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30,
                  single_arch=(mode == "single"), hybrid_arch=(mode == "hybrid"))
    name = "flexible_vit_sr_patch14_224_patch_output" + ("_supernet" if sup else "")
    prod = vitres.create_model(name, img_size=recipe.MICRO_IMG, num_classes=recipe.MICRO_CLASSES, network_def=nd,
                               drop_path_rate=0.0, drop_block_rate=None, **kw)
    orc = O.OracleViTSR(nd, img_size=recipe.MICRO_IMG, num_classes=recipe.MICRO_CLASSES, supernet=sup,
                        patch_output=True, **kw)
    shapes = [(k, tuple(v.shape)) for k, v in orc.state_dict().items()]
    assert shapes == [(k, tuple(v.shape)) for k, v in prod.state_dict().items()]
    sd = recipe.fill_state_dict(shapes, seed)
    prod.load_state_dict(sd)
    orc.load_state_dict(sd)
    return prod, orc, sd


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


@pytest.mark.parametrize("et,mode", [(0, "plain"), (0, "multi"), (0, "single"), (0, "hybrid"), (4, "plain"), (4, "multi"),
                                     (5, "plain"), (5, "multi")])
def test_emulated_model_matches_oracle_and_golden(monkeypatch, et, mode):
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f1_micro_t%d_%s.npz" % (et, mode)))
    prod, orc, sd = build_pair(et, mode, 100 + et)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    for e in ([31] if mode == "plain" else [0, 15, 31]):
        prod.train(); orc.train()
        if mode != "plain":
            prod.set_epoch(e); orc.set_epoch(e)
            prod.load_state_dict(sd); orc.load_state_dict(sd)
        prod.zero_grad(); orc.zero_grad()
        torch.manual_seed(555 + e)
        rng = torch.random.get_rng_state()
        if mode in ("single", "hybrid"):
            torch.manual_seed(e * 10000 + 3)
        cls, pat = prod(x, patch_output_type="seq")
        torch.random.set_rng_state(rng)
        tag = "e%d." % e
        if mode != "plain":
            keeps = torch.stack(prod.last_keeps).numpy()
            assert np.array_equal(keeps, g[tag + "keeps"])          # bit-exact masks vs the reference
        assert rel(cls.detach(), torch.from_numpy(g[tag + "cls"])) < 5e-5
        assert rel(pat.detach(), torch.from_numpy(g[tag + "pat"])) < 5e-5
        loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)
        loss.backward()
        ocls, opat = orc(x, keeps=prod.last_keeps if mode != "plain" else None, patch_output_type="seq")
        (O.soft_target_ce(ocls, t) + O.soft_target_ce(opat, pt)).backward()
        op = dict(orc.named_parameters())
        for n, p in prod.named_parameters():
            assert p.grad is not None, n
            assert rel(p.grad, op[n].grad) < 2e-4, (n, rel(p.grad, op[n].grad))
        if et != 0:                                                # BatchNorm running statistics (buffers)
            ob = dict(orc.named_buffers())
            for n, bf in prod.named_buffers():
                assert rel(bf.float(), ob[n].float()) < 1e-5, n
    prod.eval()
    prod.load_state_dict(sd)
    with torch.no_grad():
        out = prod(x)
    assert rel(out, torch.from_numpy(g["eval.cls"])) < 5e-5


def test_emulated_second_backward_accumulates(monkeypatch):
    emu_kernels.install(monkeypatch)
    prod, orc, sd = build_pair(0, "plain", 100)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    for _ in range(2):
        cls, pat = prod(x)
        (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)).backward()
    g2 = prod.cls_head.weight.grad.clone()
    prod.zero_grad()
    cls, pat = prod(x)
    (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)).backward()
    assert rel(g2, 2 * prod.cls_head.weight.grad) < 1e-5


def test_emulated_split_backward_equals_monolithic(monkeypatch):
    """The backward cut after the last stage (engine.GraphedTrainStep(split_for_sync=True): second hipGraph overlapped with
    the all-reduce of the finished arena tail) writes the same gradients as the single pass, and part 1 alone already
    completes every gradient of the arena tail."""
    emu_kernels.install(monkeypatch)
    prod, orc, sd = build_pair(0, "multi", 100)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    prod.set_epoch(31)
    prod._ensure_arena(torch.device("cpu"))
    cut, start = prod.split_plan()
    assert 0 < start < prod._arena["flat"].numel()

    def run(split):
        torch.manual_seed(5)
        prod.zero_grad(set_to_none=True)
        cls, pat = prod(x, patch_output_type="seq")
        prod._bwd_split = cut if split else None
        (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)).backward()
        prod._bwd_split = None
        part1 = prod._arena["gcur"].clone()
        if split:
            assert prod._bwd_state is not None
            prod.resume_backward()
        assert prod._bwd_state is None
        return part1, prod._arena["gcur"].clone()
    _, mono = run(False)
    part1, full = run(True)
    assert torch.equal(full, mono)
    assert torch.equal(part1[start:], mono[start:])          # the tail is final after part 1 ...
    assert not torch.equal(part1[:start], mono[:start])      # ... the rest is not
    # three parts (a cut in front of every spatial reduction): range k of the arena is final after part k
    cuts = prod.split_plan(parts=3)
    assert len(cuts) == 2 and cuts[0] == (cut, start) and cuts[1][0] < cut and 0 < cuts[1][1] < start
    torch.manual_seed(5)
    prod.zero_grad(set_to_none=True)
    cls, pat = prod(x, patch_output_type="seq")
    prod._bwd_split = [c for c, _ in cuts]
    (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)).backward()
    prod._bwd_split = None
    p1 = prod._arena["gcur"].clone()
    assert prod.resume_backward() is True
    p2 = prod._arena["gcur"].clone()
    assert prod.resume_backward() is False and prod._bwd_state is None
    s2 = cuts[1][1]
    assert torch.equal(prod._arena["gcur"], mono)
    assert torch.equal(p1[start:], mono[start:]) and torch.equal(p2[s2:], mono[s2:]) and not torch.equal(p2[:s2], mono[:s2])


@pytest.mark.parametrize("mode", ["plain", "multi"])
def test_emulated_patch_output_avg(monkeypatch, mode):
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f9_patch_avg.npz"))
    prod, orc, sd = build_pair(0, mode, 100)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    if mode != "plain":
        prod.set_epoch(31)
        prod.load_state_dict(sd)
    torch.manual_seed(586)
    cls, pat = prod(x, patch_output_type="avg")
    (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, t)).backward()
    assert rel(cls.detach(), torch.from_numpy(g[mode + ".cls"])) < 5e-5
    assert rel(pat.detach(), torch.from_numpy(g[mode + ".pat"])) < 5e-5
    p = dict(prod.named_parameters())
    for k in g.files:
        if k.startswith(mode + ".grad."):
            assert rel(p[k[len(mode) + 6:]].grad, torch.from_numpy(g[k])) < 2e-4, k


@pytest.mark.parametrize("et", [4, 5])
def test_emulated_eval_stem_with_folded_batchnorm(monkeypatch, et):
    """Host logic of the evaluation stem (vitres/stem.py): BatchNorm folded into the three convolutions' weights / biases, conv1
    straight from the image, residual in conv3's epilogue -- same logits as the reference's eval forward (bf16 rounding of the
    operands apart) and as the un-folded kernel sequence; the folded weights follow a changed running statistic."""
    import vitres.stem as stem
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f1_micro_t%d_plain.npz" % et))
    prod, orc, sd = build_pair(et, "plain", 100 + et)
    prod.set_compute_dtype(torch.bfloat16)
    x, _, _, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.eval()
    with torch.no_grad():
        monkeypatch.setattr(stem, "FOLD_BN", True)
        folded = prod(x).clone()
        monkeypatch.setattr(stem, "FOLD_BN", False)
        plain = prod(x).clone()
        assert rel(folded, torch.from_numpy(g["eval.cls"])) < 3e-2
        assert rel(folded, plain) < 2e-2
        prod.patch_embed.conv2.bn.running_var.mul_(4.0)
        monkeypatch.setattr(stem, "FOLD_BN", True)
        assert rel(prod(x), folded) > 1e-3


def test_cpu_tensor_is_refused():
    prod, _, _ = build_pair(0, "plain", 100)
    with pytest.raises(RuntimeError):
        prod(torch.zeros(2, 3, 56, 56))


def test_emulated_fused_layernorm_path_equals_separate_kernels(monkeypatch):
    """functional.FUSE_LN (vr_gemm_ln: the LayerNorm computed in the epilogue of the Linear before / after it, opt-in) only
    re-routes kernel calls: the same logits and gradients as the separate vr_gemm + vr_ln_fwd / vr_ln_bwd sequence."""
    import vitres.functional as Fn
    emu_kernels.install(monkeypatch)
    prod, orc, sd = build_pair(0, "multi", 100)
    prod.set_compute_dtype(torch.bfloat16)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    prod.set_epoch(31)
    calls = []
    real_fwd, real_bwd = emu_kernels.gemm_ln_fwd, emu_kernels.gemm_ln_bwd
    import vitres.kernels as K
    monkeypatch.setattr(K, "gemm_ln_fwd", lambda *a, **k: (calls.append("f"), real_fwd(*a, **k))[1])
    monkeypatch.setattr(K, "gemm_ln_bwd", lambda *a, **k: (calls.append("b"), real_bwd(*a, **k))[1])

    def run(fuse):
        monkeypatch.setattr(Fn, "FUSE_LN", fuse)
        torch.manual_seed(5)
        prod.zero_grad(set_to_none=True)
        cls, pat = prod(x, patch_output_type="seq")
        (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)).backward()
        return cls.detach().clone(), prod._arena["gcur"].clone()
    c0, g0 = run(0)
    assert not calls
    c3, g3 = run(3)
    assert "f" in calls and "b" in calls
    assert rel(c3, c0) < 2e-2 and rel(g3, g0) < 5e-2          # bf16 mode: the fused backward keeps dy in fp32 instead of bf16


def build_distill_pair(et, mode, seed):
    nd = recipe.MICRO_DEFS[et]
    sup = mode != "plain"
    kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30) if sup else {}
    prod = vitres.create_model("flexible_vit_sr_distill_patch14_224" + ("_supernet" if sup else ""), img_size=recipe.MICRO_IMG,
                               num_classes=recipe.MICRO_CLASSES, network_def=nd, drop_path_rate=0.0, **kw)
    orc = O.OracleViTSR(nd, img_size=recipe.MICRO_IMG, num_classes=recipe.MICRO_CLASSES, supernet=sup, distill_token=True,
                        patch_output=False, **kw)
    shapes = [(k, tuple(v.shape)) for k, v in orc.state_dict().items()]
    assert shapes == [(k, tuple(v.shape)) for k, v in prod.state_dict().items()]
    sd = recipe.fill_state_dict(shapes, seed)
    prod.load_state_dict(sd)
    orc.load_state_dict(sd)
    return prod, orc, sd


@pytest.mark.parametrize("et,mode", [(0, "plain"), (0, "multi"), (4, "plain"), (4, "multi")])
def test_emulated_two_token_variant_matches_reference_golden(monkeypatch, et, mode):
    """Class + distillation token (flexible_vit_sr_distill_patch14_224[_supernet]): schema, masks, (cls, dst) logits in train
    and eval mode and every gradient against fixture F14 (imported reference) and the oracle."""
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f14_distill_t%d_%s.npz" % (et, mode)))
    prod, orc, sd = build_distill_pair(et, mode, 140 + et)
    assert list(prod.state_dict().keys()) == list(g["keys"]) and recipe.checksum(sd) == int(g["state_crc"])
    assert sorted(prod.no_weight_decay()) == list(g["no_weight_decay"])
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    t2 = pt[:, 0, :].contiguous()
    prod.train(); orc.train()
    if mode != "plain":
        prod.set_epoch(31); orc.set_epoch(31)
        prod.load_state_dict(sd); orc.load_state_dict(sd)
    torch.manual_seed(555 + 31)
    cls, dst = prod(x)
    if mode != "plain":
        assert np.array_equal(torch.stack(prod.last_keeps).numpy(), g["keeps"])
    assert rel(cls.detach(), torch.from_numpy(g["cls"])) < 5e-5 and rel(dst.detach(), torch.from_numpy(g["dst"])) < 5e-5
    (O.soft_target_ce(cls, t) + O.soft_target_ce(dst, t2)).backward()
    params = dict(prod.named_parameters())
    for k in g.files:
        if k.startswith("grad."):
            assert rel(params[k[5:]].grad, torch.from_numpy(g[k])) < 2e-4, k
    ocls, odst = orc(x, keeps=prod.last_keeps if mode != "plain" else None)
    (O.soft_target_ce(ocls, t) + O.soft_target_ce(odst, t2)).backward()
    op = dict(orc.named_parameters())
    for n, p in prod.named_parameters():
        assert p.grad is not None and rel(p.grad, op[n].grad) < 2e-4, n
    prod.eval()
    prod.load_state_dict(sd)
    with torch.no_grad():
        ec, ed = prod(x)
    assert rel(ec, torch.from_numpy(g["eval.cls"])) < 5e-5 and rel(ed, torch.from_numpy(g["eval.dst"])) < 5e-5


def _kd_epoch(prod, dev, mode, hard):
    from vitres import engine
    opt = torch.optim.AdamW(engine.param_groups_weight_decay(prod, 0.05), lr=1e-3)
    loader = []
    for it in range(3):
        x, t, _, _ = recipe.inputs(300 + it, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        loader.append((x, t))

    def scaler(loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
        loss.backward()
        optimizer.step()

    class Crit(torch.nn.Module):
        def forward(self, x, t):
            return O.soft_target_ce(x, t)
    if mode != "plain":
        prod.set_epoch(31)
    torch.manual_seed(4321)
    keeps = []
    orig = type(prod).forward

    def fwd(self, *a, **k):
        out = orig(self, *a, **k)
        if self.last_keeps:
            keeps.extend(self.last_keeps)
        return out
    type(prod).forward = fwd
    try:
        stats = engine.train_one_epoch(prod, Crit(), loader, opt, dev, 31, scaler, max_norm=None, print_freq=0,
                                       teacher_model=recipe.toy_teacher(recipe.MICRO_CLASSES).to(dev), hard_distill=hard, alpha=0.5,
                                       arch_sample=("multi" if mode != "plain" else None), logger=type("L", (), {"info": staticmethod(lambda *_: None)}))
    finally:
        type(prod).forward = orig
    return stats, keeps


def test_knowledge_distillation_loss_matches_reference():
    from vitres.engine import KnowledgeDistillationLoss
    g = np.load(os.path.join(G, "f15_distillation_engine.npz"))
    gen = torch.Generator().manual_seed(5)
    xs, ts = torch.randn(8, 10, generator=gen), torch.randn(8, 10, generator=gen) * 2
    for hard, tag in ((True, "hard"), (False, "soft")):
        xv = xs.clone().requires_grad_(True)
        loss = KnowledgeDistillationLoss(hard_distill=hard)(xv, ts)
        loss.backward()
        assert abs(loss.item() - float(g["kd.%s.loss" % tag])) < 1e-6
        assert rel(xv.grad, torch.from_numpy(g["kd.%s.grad" % tag])) < 1e-5


@pytest.mark.parametrize("mode", ["plain", "multi"])
@pytest.mark.parametrize("hard", [True, False])
def test_emulated_distillation_epoch_matches_reference_engine(monkeypatch, mode, hard):
    """engine.train_one_epoch with a teacher (hard / soft distillation through the distillation token, alpha = 0.5) on the
    two-token micro net: masks, mean loss and parameters after three AdamW steps against the imported reference (F15)."""
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f15_distillation_engine.npz"))
    torch.manual_seed(2024)
    prod, orc, sd = build_distill_pair(0, mode, 140)
    prod.set_compute_dtype(torch.float32)
    stats, keeps = _kd_epoch(prod, torch.device("cpu"), mode, hard)
    tag = "%s.%s." % (mode, "hard" if hard else "soft")
    assert abs(stats["loss"] - float(g[tag + "avg_loss"])) < 1e-5 * abs(float(g[tag + "avg_loss"]))
    if mode != "plain":
        assert np.array_equal(torch.stack(keeps).numpy(), g[tag + "keeps"])
    after = prod.state_dict()
    for k in g.files:
        if k.startswith(tag + "after."):
            assert rel(after[k[len(tag) + 6:]], torch.from_numpy(g[k])) < 2e-5, k


def build_vit16(mode):
    sup = mode != "plain"
    kw = dict(num_channels_to_keep=recipe.vit16_keep_config(), example_per_arch=2, num_warmup_epochs=30) if sup else {}
    return vitres.create_model("flexible_vit_patch16_224" + ("_supernet" if sup else ""), img_size=recipe.VIT16_IMG,
                               num_classes=recipe.MICRO_CLASSES, network_def=recipe.VIT16_DEF, **kw)


@pytest.mark.parametrize("mode", ["plain", "multi"])
def test_emulated_single_stage_patch16_sibling_matches_reference(monkeypatch, mode):
    """flexible_vit_patch16_224[_supernet] (nets/vision_transformer_supernet.py): same-seed initialisation, schema,
    no_weight_decay, masks, (cls, dst) logits, loss, gradients and eval outputs against the imported reference (F16)."""
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f16_vit16_%s.npz" % mode))
    torch.manual_seed(77)
    prod = build_vit16(mode)
    assert recipe.checksum(prod.state_dict()) == int(g["init_crc"])          # the constructor draws like the reference's
    assert sorted(prod.no_weight_decay()) == list(g["no_weight_decay"])
    shapes = [(k, tuple(v.shape)) for k, v in prod.state_dict().items()]
    assert [k for k, _ in shapes] == list(g["keys"]) and [str(s) for _, s in shapes] == list(g["shapes"])
    sd = recipe.fill_state_dict(shapes, 160)
    assert recipe.checksum(sd) == int(g["state_crc"])
    prod.load_state_dict(sd)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.VIT16_IMG, recipe.MICRO_CLASSES, 1)
    t2 = pt[:, 0, :].contiguous()
    prod.train()
    if mode != "plain":
        prod.set_epoch(31)
        prod.load_state_dict(sd)
    torch.manual_seed(555 + 31)
    cls, dst = prod(x)
    if mode != "plain":
        assert np.array_equal(torch.stack(prod.last_keeps).numpy(), g["keeps"])
    assert rel(cls.detach(), torch.from_numpy(g["cls"])) < 5e-5 and rel(dst.detach(), torch.from_numpy(g["dst"])) < 5e-5
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(dst, t2)
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    loss.backward()
    params = dict(prod.named_parameters())
    n = 0
    for k in g.files:
        if k.startswith("grad."):
            assert rel(params[k[5:]].grad, torch.from_numpy(g[k])) < 2e-4, k
            n += 1
    assert n > 20
    prod.eval()
    prod.load_state_dict(sd)
    with torch.no_grad():
        ec, ed = prod(x)
    assert rel(ec, torch.from_numpy(g["eval.cls"])) < 5e-5 and rel(ed, torch.from_numpy(g["eval.dst"])) < 5e-5
    with pytest.raises(AssertionError):                                        # SR entries are not part of this grammar
        vitres.create_model("flexible_vit_patch16_224", img_size=64, num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[0])


@pytest.mark.parametrize("et,mode,pot", [(0, "multi", "seq"), (0, "multi", "avg"), (4, "plain", "seq"), (0, "single", "seq")])
def test_emulated_loss_and_grad_equals_autograd_path(monkeypatch, et, mode, pot):
    """model.loss_and_grad (forward + vr_softce_train + backward without autograd, logits kept in the internal sample order)
    gives the loss and the gradients of criterion(forward()).backward()."""
    emu_kernels.install(monkeypatch)
    prod, orc, sd = build_pair(et, mode, 100 + et)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    if mode != "plain":
        prod.set_epoch(31)
        prod.load_state_dict(sd)
    torch.manual_seed(11)
    prod.zero_grad(set_to_none=True)
    cls, pat = prod(x, patch_output_type=pot)
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt if pot == "seq" else t)
    loss.backward()
    want = {n: p.grad.clone() for n, p in prod.named_parameters()}
    keeps = [k.clone() for k in prod.last_keeps] if prod.last_keeps else None
    torch.manual_seed(11)
    prod.zero_grad(set_to_none=True)
    got = prod.loss_and_grad(x, t, pt if pot == "seq" else None, pot)
    if keeps is not None:
        assert all(torch.equal(a, b) for a, b in zip(keeps, prod.last_keeps))
    assert abs(got.item() - loss.item()) < 1e-5 * abs(loss.item())
    for n, p in prod.named_parameters():
        assert p.grad is not None and rel(p.grad, want[n]) < 1e-5, n
    with pytest.raises(RuntimeError):
        prod.loss_and_grad(x, t, pt, pot)                      # gradients not cleared


@pytest.mark.parametrize("et", [0, 4])
def test_emulated_drop_path_wiring_matches_reference(monkeypatch, et):
    """DropPath (nets/drop.py:11-26) at model level: the per-sample scale vectors are drawn in the caller's sample order while the
    rows run arch-grouped (plan.order) -- with injected draws (fixture F18, incl. dropped samples) logits, loss and EVERY
    gradient must equal the reference's."""
    emu_kernels.install(monkeypatch)
    g = np.load(os.path.join(G, "f18_micro_t%d_multi_dp.npz" % et))
    nd = recipe.MICRO_DEFS[et]
    kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
    prod = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=recipe.MICRO_IMG,
                               num_classes=recipe.MICRO_CLASSES, network_def=nd, drop_path_rate=0.2, **kw)
    sd = recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in prod.state_dict().items()], 100 + et)
    prod.load_state_dict(sd)
    prod.set_compute_dtype(torch.float32)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    prod.set_epoch(31)
    prod.load_state_dict(sd)
    for use_fused in (False, True):
        prod.zero_grad(set_to_none=True)
        torch.manual_seed(555 + 31)
        plan = prod.sample_plan(8)
        assert plan.order is not None and plan.n_dp == g["noise"].shape[0]
        plan.dp_noise = torch.from_numpy(g["noise"])
        if use_fused:
            loss = prod.loss_and_grad(x, t, pt, "seq", plan=plan)
        else:
            cls, pat = prod(x, patch_output_type="seq", plan=plan)
            assert np.array_equal(torch.stack(prod.last_keeps).numpy(), g["keeps"])
            assert rel(cls.detach(), torch.from_numpy(g["cls"])) < 1e-4 and rel(pat.detach(), torch.from_numpy(g["pat"])) < 1e-4
            loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)
            loss.backward()
        assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
        for n, p in prod.named_parameters():
            assert rel(p.grad, torch.from_numpy(g["grad." + n])) < 5e-4, (n, use_fused)


def test_second_model_forward_between_the_parts_of_a_split_backward(monkeypatch):
    """the pending LayerNorm folds / collected weight-gradient calls are module-global; a training-mode forward of a
    SECOND model between the parts of another model's split backward must leave them alone (it used to clear them: the first model
    then lost gradients without an error)."""
    emu_kernels.install(monkeypatch)
    from vitres import functional as Fn
    prod, orc, sd = build_pair(0, "multi", 100)
    other, _, sd2 = build_pair(0, "multi", 101)
    for m in (prod, other):
        m.set_compute_dtype(torch.float32)
        m.train()
        m.set_epoch(31)
        m._ensure_arena(torch.device("cpu"))
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    cut, start = prod.split_plan()

    def run(intrude):
        torch.manual_seed(5)
        prod.zero_grad(set_to_none=True)
        cls, pat = prod(x, patch_output_type="seq")
        prod._bwd_split = cut
        (O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)).backward()
        prod._bwd_split = None
        assert prod._bwd_state is not None
        if intrude:
            sentinel = ("sentinel",)                               # (on the CPU emulation the lists are empty between the parts: stand-in entry)
            Fn._ln_pending.append(sentinel)
            torch.manual_seed(6)
            other(x, patch_output_type="seq")                      # training-mode forward of another model
            assert Fn._ln_pending and Fn._ln_pending[-1] is sentinel
            Fn._ln_pending.remove(sentinel)
            # ... while the owner's own next forward, or anybody's once the owner is through, does clear leftovers
            Fn._ln_pending.append(sentinel)
            assert Fn.reset_ln_grads(prod) is True and not Fn._ln_pending
        prod.resume_backward()
        assert prod._bwd_state is None
        return prod._arena["gcur"].clone()
    assert torch.equal(run(True), run(False))
