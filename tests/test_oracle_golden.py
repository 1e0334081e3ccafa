"""Pins the CPU oracle (oracle/vitres_oracle.py) against golden vectors produced by the reference
itself (tests/golden/make_golden.py).  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

import recipe
import vitres_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def build(nd, mode="plain", img=recipe.MICRO_IMG, classes=recipe.MICRO_CLASSES, cfg=None, epa=2, **kw):
    sup = mode != "plain"
    if sup:
        kw.update(num_channels_to_keep=cfg or recipe.micro_keep_config(), example_per_arch=epa, num_warmup_epochs=30,
                  single_arch=(mode == "single"), hybrid_arch=(mode == "hybrid"))
    return O.OracleViTSR(nd, img_size=img, num_classes=classes, supernet=sup, patch_output=True, **kw)


def load_recipe(m, seed):
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = recipe.fill_state_dict(shapes, seed)
    m.load_state_dict(sd)
    return sd, shapes


def close(a, b, tol=2e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    denom = max(np.abs(b).max(), 1e-6)
    err = np.abs(a - b).max() / denom
    assert err < tol, "rel err %.3e" % err


MICRO_CASES = [(0, "plain"), (0, "multi"), (0, "single"), (0, "hybrid"), (4, "plain"), (4, "multi"), (5, "plain"), (5, "multi")]


@pytest.mark.parametrize("et,mode", MICRO_CASES)
def test_f1_micro(et, mode):
    g = load("f1_micro_t%d_%s" % (et, mode))
    m = build(recipe.MICRO_DEFS[et], mode)
    sd, shapes = load_recipe(m, 100 + et)
    assert recipe.checksum(sd) == int(g["state_crc"])
    assert [k for k, _ in shapes] == list(g["keys"])                  # identical state_dict schema
    assert [str(s) for _, s in shapes] == list(g["shapes"])
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    epochs = [31] if mode == "plain" else [0, 8, 15, 30, 31]
    for e in epochs:
        m.train()
        if mode != "plain":
            m.set_epoch(e)
            m.load_state_dict(sd)
        m.zero_grad()
        tag = "e%d." % e
        # (a) RNG protocol reproduces the reference's keep tables bit-exactly
        torch.manual_seed(555 + e)
        rng = torch.random.get_rng_state()
        if mode in ("single", "hybrid"):
            torch.manual_seed(e * 10000 + 3)
        (cls, pat), used = m(x, patch_output_type="seq", return_keeps=True)
        torch.random.set_rng_state(rng)
        if mode != "plain":
            assert np.array_equal(torch.stack(used).numpy(), g[tag + "keeps"])
            mod_order = [m.embed_drop]            # reference stores nlc in nn.Module registration order
            for blk in m.blocks:
                if isinstance(blk, O.OracleSR):
                    mod_order.append(blk.drop)
                elif isinstance(blk, O.OracleBlock):
                    mod_order += [d for d in (blk.layer_drop, blk.attn.drop, blk.mlp.drop) if d is not None]
            nlc = [O.num_layer_config(len(d.choices), e, d.warmup) for d in mod_order]
            assert nlc == list(g[tag + "nlc"])
        loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)
        close(cls.detach(), g[tag + "cls"])
        close(pat.detach(), g[tag + "pat"])
        assert abs(loss.item() - float(g[tag + "loss"])) < 2e-5 * abs(float(g[tag + "loss"]))
        loss.backward()
        params = dict(m.named_parameters())
        n_checked = 0
        for k in g.files:
            if k.startswith(tag + "grad."):
                close(params[k[len(tag) + 5:]].grad, g[k], tol=5e-5)
                n_checked += 1
        if e == 31:
            assert n_checked > 15
        if et != 0 and (tag + "bn.conv1.running_mean") in g.files:
            bsd = m.state_dict()
            for bn in ("conv1", "conv3"):
                close(bsd["patch_embed.%s.bn.running_mean" % bn], g[tag + "bn.%s.running_mean" % bn])
                close(bsd["patch_embed.%s.bn.running_var" % bn], g[tag + "bn.%s.running_var" % bn])
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        close(m(x), g["eval.cls"])


def test_f1_explicit_keeps_equal_sampled():
    """Feeding the recorded keep vectors explicitly gives the same result as the RNG protocol."""
    g = load("f1_micro_t0_multi")
    m = build(recipe.MICRO_DEFS[0], "multi")
    load_recipe(m, 100)
    x, _, _, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    m.train()
    m.set_epoch(31)
    load_recipe(m, 100)
    cls, pat = m(x, keeps=[torch.from_numpy(k) for k in g["e31.keeps"]], patch_output_type="seq")
    close(cls.detach(), g["e31.cls"])


def test_f2_masked_layer_norm():
    g = load("f2_masked_ln")
    x, w, b, gr = (torch.from_numpy(g[k]) for k in ("x", "w", "b", "g"))
    keep = torch.from_numpy(g["keep"])
    y, z, inv_std = O.masked_layer_norm_fwd(x, w, b, 1e-6, keep)
    close(y, g["y"])
    gx, gw, gb = O.masked_layer_norm_bwd(gr, z, inv_std, w, keep)
    close(gx, g["gx"]); close(gw, g["gw"]); close(gb, g["gb"])
    y, z, inv_std = O.masked_layer_norm_fwd(x, w, b, 1e-6, None)
    close(y, g["y_plain"])
    gx, gw, gb = O.masked_layer_norm_bwd(gr, z, inv_std, w, None)
    close(gx, g["gx_plain"]); close(gw, g["gw_plain"]); close(gb, g["gb_plain"])
    # autograd of the forward formula == the reference's hand-written backward
    xr = x.clone().requires_grad_(True)
    O.masked_layer_norm_fwd(xr, w, b, 1e-6, keep)[0].backward(gr)
    close(xr.grad, g["gx"])


def test_f3_channel_drop_tables_and_rng():
    g = load("f3_channel_drop")
    for ci, case in enumerate(g["cases"]):
        choices, B, epa, warm, single = ast.literal_eval(str(case))
        for e in (0, 8, 15, 30, 31):
            tag = "c%d.e%d." % (ci, e)
            table = O.keep_table(choices, B, epa, e, warm, single)
            assert table == list(g[tag + "table"])
            assert O.num_layer_config(len(choices), e, warm) == int(g[tag + "nlc"])
            cd = O.OracleChannelDrop(choices, warm, epa, single)
            cd.set_epoch(e)
            torch.manual_seed(1000 + ci * 10 + e)
            draws = np.stack([cd.sample(B, max(choices), True).numpy() for _ in range(3)])
            assert np.array_equal(draws, g[tag + "draws"])


def test_f5_subnet_slicing_and_equivalence():
    g = load("f5_subnet")
    sup = build(recipe.MICRO_DEFS[0], "multi")
    sd, _ = load_recipe(sup, 100)
    assert recipe.checksum(sd) == int(g["state_crc"])
    x, _, _, _ = recipe.inputs(9, 6, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    for i, nd in enumerate(recipe.MICRO_CANDIDATES):
        sub = build(nd)
        ssd = O.sub_state_dict(sup.state_dict(), sub.state_dict())
        assert recipe.checksum(ssd) == int(g["cand%d.crc" % i])
        assert np.array_equal(ssd["blocks.0.attn.qkv.weight"].numpy(), g["cand%d.qkv0" % i])
        sub.load_state_dict(ssd)
        sub.eval()
        with torch.no_grad():
            logits = sub(x)
        close(logits, g["cand%d.logits" % i])


def test_f6_schema_params_macs():
    g = load("f6_schema_macs")
    nets = {"ref_tiny": recipe.REF_TINY_DEF, "sr_tiny": recipe.SR_TINY_DEF, "sr_small": recipe.SR_SMALL_DEF,
            "sr_tiny_mh": recipe.SR_TINY_MH_DEF, "sr_small_mh": recipe.SR_SMALL_MH_DEF}
    expect_params = {"ref_tiny": 42781736, "sr_tiny": 69673168, "sr_small": 144108208,
                     "sr_tiny_mh": 85839464, "sr_small_mh": 156599528}
    expect_macs = {"ref_tiny": 1.7944e9, "sr_tiny": 3.4735e9, "sr_small": 6.0043e9, "sr_tiny_mh": 3.4976e9,
                   "sr_small_mh": 6.2603e9}
    for name, nd in nets.items():
        assert int(g[name + ".n_params"]) == expect_params[name]
        assert O.macs(nd) == int(g[name + ".macs"])
        assert O.macs(nd, distill=True) == int(g[name + ".macs_distill"])
        assert abs(O.macs(nd) - expect_macs[name]) / expect_macs[name] < 1e-4
        with torch.device("meta"):
            m = O.OracleViTSR(nd, img_size=224, num_classes=1000, patch_output=True)
        sd = m.state_dict()
        assert list(sd.keys()) == list(g[name + ".keys"])
        assert [str(tuple(v.shape)) for v in sd.values()] == list(g[name + ".shapes"])
        assert sum(p.numel() for p in m.parameters()) == expect_params[name]
        assert list(g[name + ".no_weight_decay"]) == ["tokens"]
    for i, nd in enumerate(recipe.MICRO_CANDIDATES):
        assert O.macs(nd, resolution=56) == int(g["micro_cand%d.macs" % i])


def test_f7_rewiring():
    g = load("f7_rewiring")
    m = build(recipe.MICRO_DEFS[0], "multi")
    load_recipe(m, 100)
    m.set_epoch(0)
    after = m.state_dict()
    n = 0
    for k in g.files:
        if k.startswith("blocks."):
            assert np.array_equal(after[k].numpy(), g[k]), k
            n += 1
    assert n == 12


@pytest.mark.parametrize("mode", ["multi", "single", "plain"])
def test_f8_engine_protocol(mode):
    g = load("f8_engine_%s" % mode)
    torch.manual_seed(2024)
    m = build(recipe.MICRO_DEFS[0], mode)
    sd, _ = load_recipe(m, 100)
    opt = torch.optim.AdamW(O.param_groups_weight_decay(m, 0.05), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    if mode != "plain":
        m.set_epoch(31)
    torch.manual_seed(4321)
    losses, keeps = [], []
    rng0 = torch.random.get_rng_state()
    for it in range(3):
        x, t, pt, _ = recipe.inputs(300 + it, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        loss, used = O.train_step(m, opt, x, t, pt, 31, it, arch_sample=(None if mode == "plain" else mode))
        losses.append(loss)
        keeps += used
    assert torch.equal(rng0, torch.random.get_rng_state()) and bool(g["rng_unchanged"])
    assert abs(np.mean(losses) - float(g["avg_loss"])) < 1e-5 * abs(float(g["avg_loss"]))
    if mode != "plain":
        assert np.array_equal(torch.stack(keeps).numpy(), g["keeps"])
    after = m.state_dict()
    for k in g.files:
        if k.startswith("after."):
            close(after[k[6:]], g[k], tol=1e-5)


def test_f8_evaluate():
    g = load("f8_engine_eval")
    m = build(recipe.MICRO_DEFS[0])
    load_recipe(m, 100)
    batches = []
    for it in range(2):
        x, _, _, labels = recipe.inputs(400 + it, 8 if it == 0 else 4, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        batches.append((x, labels))
    stats = O.evaluate(m, batches)
    for k in ("loss", "acc1", "acc5"):
        assert abs(stats[k] - float(g[k])) < 1e-5 * max(abs(float(g[k])), 1.0)


def test_f4_fullsize_ref_tiny_c1():
    g = load("f4_ref_tiny_c1")
    m = O.OracleViTSR(recipe.REF_TINY_DEF, img_size=224, num_classes=1000, patch_output=True)
    sd, shapes = load_recipe(m, 4242)
    assert recipe.checksum(sd) == int(g["state_crc"])
    assert int(g["n_params"]) == 42781736
    x, t, pt, _ = recipe.inputs(11, 2, 224, 1000, 16)
    m.train()
    with torch.no_grad():
        cls, pat = m(x, patch_output_type="seq")
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)
    close(cls, g["cls"], tol=1e-4); close(pat, g["pat"], tol=1e-4)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        close(m(x), g["eval.cls"], tol=1e-4)


@pytest.mark.parametrize("mode", ["plain", "multi"])
def test_f9_patch_output_avg(mode):
    """patch_output_type='avg' (vit_sr_supernet.py:447-449): patch head on the mean patch token, trained against `targets`."""
    g = load("f9_patch_avg")
    m = build(recipe.MICRO_DEFS[0], mode)
    sd, _ = load_recipe(m, 100)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    m.train()
    if mode != "plain":
        m.set_epoch(31)
        m.load_state_dict(sd)
    torch.manual_seed(586)
    (cls, pat), keeps = m(x, patch_output_type="avg", return_keeps=True)
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, t)
    loss.backward()
    assert pat.shape == (8, recipe.MICRO_CLASSES)
    close(cls.detach(), g[mode + ".cls"])
    close(pat.detach(), g[mode + ".pat"])
    close(loss.item(), g[mode + ".loss"])
    if mode != "plain":
        assert np.array_equal(torch.stack(keeps).numpy(), g[mode + ".keeps"])
    p = dict(m.named_parameters())
    for k in g.files:
        if k.startswith(mode + ".grad."):
            close(p[k[len(mode) + 6:]].grad, g[k], tol=1e-4)


def test_f10_switch_token_mix():
    """SwitchTokenMix restatement == the reference, bit for bit, including how far both RNG streams advance."""
    g = load("f10_token_mix")
    for case in range(4):
        tag = "c%d." % case
        B, H, pl, nc, seed = (int(v) for v in g[tag + "cfg"])
        rs = np.random.RandomState(100 + case)
        x = torch.from_numpy(rs.standard_normal((B, 3, H, H)).astype(np.float32))
        y = torch.from_numpy(rs.randint(0, nc, size=(B,)).astype(np.int64))
        torch.manual_seed(40 + seed)
        np.random.seed(50 + seed)
        xs, t, pt, pot = O.switch_token_mix(x, y, pl, nc, 0.1)
        assert pot == "seq"
        assert np.array_equal(xs.numpy(), g[tag + "samples"])
        assert np.array_equal(t.numpy(), g[tag + "targets"])
        assert np.array_equal(pt.numpy(), g[tag + "patch_targets"])
        assert np.random.randint(0, 1 << 30) == int(g[tag + "np_after"])
        assert int(torch.randint(0, 1 << 30, (1,))) == int(g[tag + "torch_after"][0])


def test_f11_pos_embed_interpolation():
    """vitres.network_utils.finetune_state_dict == the reference's resampling of every positional embedding (56 -> 140 px)."""
    import vitres
    from vitres.network_utils.finetune_state_dict import state_dict_interpolate_pos_embed
    g = load("f11_pos_embed_interp")
    kw = dict(num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[0], drop_path_rate=0.0)
    lo = vitres.create_model("flexible_vit_sr_patch14_224_patch_output", img_size=56, **kw)
    hi = vitres.create_model("flexible_vit_sr_patch14_224_patch_output", img_size=140, **kw)
    sd = recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in lo.state_dict().items()], 321)
    out = state_dict_interpolate_pos_embed(hi.state_dict(), {k: v.clone() for k, v in sd.items()})
    keys = [k for k in out if "pos_embed" in k]
    assert sorted(keys) == sorted(g.files) and len(keys) == 3
    for k in keys:
        assert out[k].shape == hi.state_dict()[k].shape
        assert np.array_equal(out[k].numpy(), g[k]), k
    hi.load_state_dict(out)                                # and the resized checkpoint loads


@pytest.mark.parametrize("et,sup", [(0, False), (0, True), (4, True), (5, False)])
def test_f12_same_seed_same_initialisation(et, sup):
    """vitres.create_model consumes the random stream exactly like the reference's constructor: the same torch seed gives the
    same initial parameters (trunc_normal Linears / tokens / pos_embeds, default conv init), bit for bit."""
    import vitres
    g = load("f12_init")
    kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30) if sup else {}
    torch.manual_seed(77)
    name = "flexible_vit_sr_patch14_224_patch_output" + ("_supernet" if sup else "")
    m = vitres.create_model(name, img_size=recipe.MICRO_IMG, num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[et], **kw)
    sd = m.state_dict()
    tag = "t%d_%d." % (et, int(sup))
    assert np.array_equal(sd["tokens"].numpy(), g[tag + "tokens"])
    assert np.array_equal(sd["cls_head.weight"].numpy(), g[tag + "head"])
    assert recipe.checksum(sd) == int(g[tag + "crc"])


DISTILL_CASES = [(0, "plain"), (0, "multi"), (4, "plain"), (4, "multi")]


def build_distill(et, mode):
    kw = {}
    if mode != "plain":
        kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
    return O.OracleViTSR(recipe.MICRO_DEFS[et], img_size=recipe.MICRO_IMG, num_classes=recipe.MICRO_CLASSES,
                         supernet=(mode != "plain"), distill_token=True, patch_output=False, **kw)


@pytest.mark.parametrize("et,mode", DISTILL_CASES)
def test_f14_two_token_variant(et, mode):
    """Class + distillation token (factories flexible_vit_sr_distill_patch14_224[_supernet]): schema, keep tables, both logits,
    loss and gradients of the oracle against the imported reference."""
    g = load("f14_distill_t%d_%s" % (et, mode))
    m = build_distill(et, mode)
    sd, shapes = load_recipe(m, 140 + et)
    assert recipe.checksum(sd) == int(g["state_crc"])
    assert [k for k, _ in shapes] == list(g["keys"]) and [str(s) for _, s in shapes] == list(g["shapes"])
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    t2 = pt[:, 0, :].contiguous()
    m.train()
    if mode != "plain":
        m.set_epoch(31)
        m.load_state_dict(sd)
    torch.manual_seed(555 + 31)
    (cls, dst), used = m(x, return_keeps=True)
    if mode != "plain":
        assert np.array_equal(torch.stack(used).numpy(), g["keeps"])
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(dst, t2)
    close(cls.detach(), g["cls"])
    close(dst.detach(), g["dst"])
    assert abs(loss.item() - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    loss.backward()
    params = dict(m.named_parameters())
    n = 0
    for k in g.files:
        if k.startswith("grad."):
            close(params[k[5:]].grad, g[k], tol=5e-5)
            n += 1
    assert n > 15
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        ec, ed = m(x)
    close(ec, g["eval.cls"])
    close(ed, g["eval.dst"])


# ---- F18: DropPath with injected draws (nets/drop.py:11-26), full-size gradients, config C4 ---------------------------
def oracle_noise(m, noise):
    """The oracle draws one noise row per branch of EVERY block (rate 0 included): pad the fixture's rows (blocks with rate > 0
    only) accordingly."""
    rows, out = list(noise), []
    for blk in m.blocks:
        if isinstance(blk, O.OracleBlock):
            for _ in range(2):
                out.append(torch.from_numpy(rows.pop(0)) if blk.dp > 0 else None)
        elif isinstance(blk, O._Bypass):
            out += [None, None]
    assert not rows
    return out


def check_grad_samples(named_grads, g, tol):
    worst = 0.0
    for n, gr in named_grads:
        gr = gr.detach().reshape(-1).cpu()
        idx = torch.from_numpy(recipe.grad_sample_index(n, gr.numel()))
        ref = np.asarray(g["gs." + n], dtype=np.float64)
        scale = max(float(g["gn." + n]) / np.sqrt(gr.numel()), np.abs(ref).max(), 1e-12)   # rms of the whole tensor
        err = np.abs(gr[idx].double().numpy() - ref).max() / scale
        nerr = abs(float(gr.double().norm()) - float(g["gn." + n])) / max(float(g["gn." + n]), 1e-12)
        worst = max(worst, err, nerr)
        assert err < tol and nerr < tol, (n, err, nerr)
    return worst


@pytest.mark.parametrize("et", [0, 4])
def test_f18_micro_drop_path(et):
    g = load("f18_micro_t%d_multi_dp" % et)
    m = build(recipe.MICRO_DEFS[et], "multi", drop_path_rate=0.2)
    sd, _ = load_recipe(m, 100 + et)
    assert recipe.checksum(sd) == int(g["state_crc"])
    rates = [b.dp for b in m.blocks if isinstance(b, O.OracleBlock) and b.dp > 0]
    assert np.allclose(rates, g["rates"])
    noise = recipe.drop_path_noise(1800 + et, rates, 8)
    assert np.array_equal(np.stack(noise), g["noise"])
    assert (np.floor(1 - np.repeat(g["rates"], 2)[:, None] + g["noise"]) == 0).any()      # dropped samples do occur
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    m.train()
    m.set_epoch(31)
    m.load_state_dict(sd)
    torch.manual_seed(555 + 31)
    (cls, pat), keeps = m(x, dp_noise=oracle_noise(m, noise), patch_output_type="seq", return_keeps=True)
    assert np.array_equal(torch.stack(keeps).numpy(), g["keeps"])
    close(cls.detach(), g["cls"]); close(pat.detach(), g["pat"])
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    for n, p in m.named_parameters():
        close(p.grad, g["grad." + n], 1e-4)


def test_f18_full_size_sr_small_c4_gradients():
    """C4 geometry on the oracle: sr_small supernet, B = 8, drop_path 0.3, explicit draws: logits, loss, every gradient (sampled)."""
    from vitres import supernet_config
    g = load("f18_sr_small_c4")
    m = build(recipe.SR_SMALL_DEF, "multi", img=224, classes=1000, cfg=supernet_config.sr_small.num_channels_to_keep,
              drop_path_rate=0.3)
    sd, shapes = load_recipe(m, 4444)
    assert recipe.checksum(sd) == int(g["state_crc"]) and [k for k, _ in shapes] == list(g["keys"])
    rates = [b.dp for b in m.blocks if isinstance(b, O.OracleBlock) and b.dp > 0]
    noise = recipe.drop_path_noise(1900 + 13, rates, 8)
    assert np.array_equal(np.stack(noise), g["noise"])
    x, t, pt, _ = recipe.inputs(13, 8, 224, 1000, 16)
    m.train()
    m.set_epoch(31)
    m.load_state_dict(sd)
    torch.manual_seed(77)
    (cls, pat), keeps = m(x, dp_noise=oracle_noise(m, noise), patch_output_type="seq", return_keeps=True)
    assert np.array_equal(torch.stack(keeps).numpy(), g["keeps"])
    close(cls.detach(), g["cls"], 1e-4); close(pat.detach()[:, :, :8], g["pat_head8"], 1e-4)
    loss = O.soft_target_ce(cls, t) + O.soft_target_ce(pat, pt)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss.backward()
    check_grad_samples([(n, p.grad) for n, p in m.named_parameters()], g, 2e-4)
