#!/usr/bin/env python
"""Generate the committed golden vectors by running the REFERENCE itself (dev container only).

    cd /root/repo && python tests/golden/make_golden.py

Imports /root/reference under tests/golden/ref_shim.py, feeds it tensors from
tests/golden/recipe.py and stores inputs' recipe ids + the reference's outputs as small .npz
files next to this script.  Nothing of the reference's source is stored -- vectors only.
Fixture ids follow SURVEY.md section 8c (F1..F8).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

import recipe  # noqa: E402
import ref_shim  # noqa: E402

R = ref_shim.ref_modules()
CD = R.channel_drop.ChannelDrop


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


class KeepRecorder:
    """Records the keep-count vector returned by every ChannelDrop.forward, in call order."""

    def __init__(self):
        self.log = []
        self._orig = CD.forward

    def __enter__(self):
        rec = self

        def fwd(self_cd, x):
            out, mask = rec._orig(self_cd, x)
            rec.log.append(mask.sum(dim=2).reshape(-1).to(torch.int64).clone())
            return out, mask
        CD.forward = fwd
        return self

    def __exit__(self, *a):
        CD.forward = self._orig


def build_ref(network_def, supernet=False, img=recipe.MICRO_IMG, classes=recipe.MICRO_CLASSES, **kw):
    m = R.vit_sr.FlexibleDistillVisionTransformerSR(
        img_size=img, patch_size=14, num_classes=classes, distill_token=False, network_def=network_def,
        patch_output=True, supernet=supernet, **kw)
    return m


def load_recipe(model, seed):
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = recipe.fill_state_dict(shapes, seed)
    model.load_state_dict(sd)
    return sd, shapes


def grads_of(model, names):
    p = dict(model.named_parameters())
    return {("grad." + n): p[n].grad.detach().numpy().copy() for n in names}


GRAD_NAMES_COMMON = ["tokens", "pos_embed", "blocks.0.norm1.weight", "blocks.0.norm1.bias",
                     "blocks.0.attn.qkv.weight", "blocks.0.attn.qkv.bias", "blocks.0.attn.proj.weight",
                     "blocks.0.mlp.fc1.weight", "blocks.0.mlp.fc2.bias",
                     "blocks.2.pos_embed", "blocks.2.norm.weight", "blocks.2.patch_reduce.weight",
                     "blocks.2.patch_reduce.bias", "blocks.2.token_transform.weight",
                     "blocks.4.attn.qkv.weight", "blocks.4.mlp.fc1.bias",
                     "blocks.6.attn.proj.bias", "blocks.6.mlp.fc2.weight",
                     "norm.weight", "norm.bias", "cls_head.weight", "patch_head.weight", "patch_head.bias"]


def soft_ce(x, t):
    return torch.sum(-t * torch.nn.functional.log_softmax(x, dim=-1), dim=-1).mean()


# ------------------------------------------------------------------------------------------------
# F1: micro nets, plain and supernet, embed types 0/4/5
# ------------------------------------------------------------------------------------------------
def f1_micro():
    B = 8
    for et in (0, 4, 5):
        nd = recipe.MICRO_DEFS[et]
        for mode in ("plain", "multi", "single", "hybrid"):
            if mode in ("single", "hybrid") and et != 0:
                continue
            torch.manual_seed(1234)
            kw = {}
            if mode != "plain":
                kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2,
                          num_warmup_epochs=30, single_arch=(mode == "single"), hybrid_arch=(mode == "hybrid"))
            m = build_ref(nd, supernet=(mode != "plain"), **kw)
            sd, shapes = load_recipe(m, seed=100 + et)
            x, t, pt, labels = recipe.inputs(7, B, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
            out = {"state_crc": recipe.checksum(sd), "keys": np.array([k for k, _ in shapes]),
                   "shapes": np.array([str(s) for _, s in shapes])}
            epochs = [31] if mode == "plain" else [0, 8, 15, 30, 31]
            for e in epochs:
                m.train()
                m.set_epoch(e) if mode != "plain" else None
                # set_epoch rewires weights while warm-up >= epoch; reload so every epoch sees the recipe weights
                m.load_state_dict(sd)
                m.zero_grad()
                torch.manual_seed(555 + e)
                rng = torch.random.get_rng_state()
                if mode in ("single", "hybrid"):
                    torch.manual_seed(e * 10000 + 3)              # engine.py:121-122 with train_iter=3
                with KeepRecorder() as rec:
                    cls, pat = m(x.clone(), patch_output_type="seq")
                torch.random.set_rng_state(rng)
                loss = soft_ce(cls, t) + soft_ce(pat, pt)
                loss.backward()
                tag = "e%d." % e
                out[tag + "cls"] = cls.detach().numpy()
                out[tag + "pat"] = pat.detach().numpy()
                out[tag + "loss"] = loss.item()
                if rec.log:
                    out[tag + "keeps"] = torch.stack(rec.log).numpy()
                    out[tag + "nlc"] = np.array([d.num_layer_config for d in m.modules() if isinstance(d, CD)])
                if e not in (31,) and not (e == 0 and mode == "multi" and et == 0):
                    continue          # gradients only at the fully-widened epoch (+ dense epoch 0 once)
                names = [n for n in GRAD_NAMES_COMMON if n in dict(m.named_parameters())]
                if et == 0:
                    names += ["patch_embed.proj.weight", "patch_embed.proj.bias"]
                else:
                    names += ["patch_embed.conv1.conv.weight", "patch_embed.conv1.bn.weight", "patch_embed.conv1.bn.bias",
                              "patch_embed.conv2.conv.weight", "patch_embed.conv3.bn.weight",
                              "patch_embed.conv_proj.weight", "patch_embed.conv_proj.bias"]
                for k, v in grads_of(m, names).items():
                    out[tag + k] = v
                if et != 0:
                    bsd = m.state_dict()
                    for bn in ("conv1", "conv3"):
                        out[tag + "bn.%s.running_mean" % bn] = bsd["patch_embed.%s.bn.running_mean" % bn].numpy().copy()
                        out[tag + "bn.%s.running_var" % bn] = bsd["patch_embed.%s.bn.running_var" % bn].numpy().copy()
            # eval-mode forward (all-true masks for supernets)
            m.load_state_dict(sd)
            m.eval()
            with torch.no_grad():
                out["eval.cls"] = m(x.clone()).numpy()
            save("f1_micro_t%d_%s" % (et, mode), **out)


# ------------------------------------------------------------------------------------------------
# F2: MaskedLayerNormFunc forward / backward
# ------------------------------------------------------------------------------------------------
def f2_masked_ln():
    rs = np.random.RandomState(21)
    B, N, C = 4, 5, 24
    keep = torch.tensor([24, 16, 8, 20])
    mask = (torch.arange(C)[None, :] < keep[:, None]).unsqueeze(1)
    x = torch.from_numpy(rs.standard_normal((B, N, C)).astype(np.float32)) * mask
    w = torch.from_numpy((1 + 0.2 * rs.standard_normal(C)).astype(np.float32))
    b = torch.from_numpy((0.1 * rs.standard_normal(C)).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((B, N, C)).astype(np.float32))
    ln = R.mln.MaskedLayerNorm(C)
    ln.weight.data.copy_(w)
    ln.bias.data.copy_(b)
    xr = x.clone().requires_grad_(True)
    y = ln(xr, mask)
    y.backward(g)
    xp = x.clone().requires_grad_(True)
    out = dict(x=x.numpy(), w=w.numpy(), b=b.numpy(), g=g.numpy(), keep=keep.numpy(),
               y=y.detach().numpy(), gx=xr.grad.numpy(), gw=ln.weight.grad.numpy().copy(), gb=ln.bias.grad.numpy().copy())
    ln.zero_grad()
    y2 = ln(xp, None)
    y2.backward(g)
    out.update(y_plain=y2.detach().numpy(), gx_plain=xp.grad.numpy(), gw_plain=ln.weight.grad.numpy().copy(),
               gb_plain=ln.bias.grad.numpy().copy())
    save("f2_masked_ln", **out)


# ------------------------------------------------------------------------------------------------
# F3: ChannelDrop tables + RNG protocol
# ------------------------------------------------------------------------------------------------
def f3_channel_drop():
    out = {}
    cases = []
    grid = [
        ([256, 224, 192, 176, 160], 128, 64, 30, False),
        ([256, 224, 192, 176, 160], 16, 2, 30, False),
        ([768, 640, 512, 384], 12, 3, 30, False),
        ([256, 256, 256, 0], 8, 2, 30, False),
        ([320, 320, 0, 0], 64, 32, 30, False),
        ([960, 880, 800, 720, 640, 560, 480], 64, 32, 30, False),
        ([96, 48], 8, 2, 0, False),
        ([32, 24, 16], 8, 2, 30, True),
        ([192, 160, 128, 96], 6, 1, 15, False),
    ]
    for ci, (choices, B, epa, warm, single) in enumerate(grid):
        for e in (0, 8, 15, 30, 31):
            cd = CD(num_channels_to_keep=np.array(choices), num_warmup_epochs=warm, example_per_arch=epa,
                    single_arch=single)
            cd.train()
            cd.set_epoch(e)
            C = max(choices)
            torch.manual_seed(1000 + ci * 10 + e)
            x = torch.ones(B, 3, C)
            draws = []
            for _ in range(3):
                _, mask = cd(x)
                draws.append(mask.sum(dim=2).reshape(-1).numpy())
            table = cd.mask.sum(dim=2).reshape(-1).numpy()
            tag = "c%d.e%d." % (ci, e)
            out[tag + "table"] = table
            out[tag + "draws"] = np.stack(draws)
            out[tag + "nlc"] = cd.num_layer_config
        cases.append(str((choices, B, epa, warm, single)))
    out["cases"] = np.array(cases)
    save("f3_channel_drop", **out)


# ------------------------------------------------------------------------------------------------
# F4: full-size nets (C1 ref-tiny; sr_tiny supernet) -- logits only, weights by recipe
# ------------------------------------------------------------------------------------------------
def f4_fullsize():
    torch.manual_seed(0)
    m = build_ref(recipe.REF_TINY_DEF, img=224, classes=1000, drop_path_rate=0.0)
    sd, shapes = load_recipe(m, seed=4242)
    x, t, pt, labels = recipe.inputs(11, 2, 224, 1000, 16)
    m.train()
    cls, pat = m(x.clone(), patch_output_type="seq")
    loss = soft_ce(cls, t) + soft_ce(pat, pt)
    out = dict(state_crc=recipe.checksum(sd), n_params=sum(p.numel() for p in m.parameters()),
               cls=cls.detach().numpy(), pat=pat.detach().numpy(), loss=loss.item(),
               keys=np.array([k for k, _ in shapes]), shapes=np.array([str(s) for _, s in shapes]))
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        out["eval.cls"] = m(x.clone()).numpy()
    save("f4_ref_tiny_c1", **out)
    del m

    cfg = R.cfg["sr_tiny"].num_channels_to_keep
    m = build_ref(recipe.SR_TINY_DEF, supernet=True, img=224, classes=1000, drop_path_rate=0.0,
                  num_channels_to_keep=cfg, example_per_arch=2, num_warmup_epochs=30)
    sd, shapes = load_recipe(m, seed=4343)
    x, t, pt, labels = recipe.inputs(12, 8, 224, 1000, 16)
    m.train()
    m.set_epoch(31)
    torch.manual_seed(77)
    with KeepRecorder() as rec:
        cls, pat = m(x.clone(), patch_output_type="seq")
    loss = soft_ce(cls, t) + soft_ce(pat, pt)
    save("f4_sr_tiny_c3", state_crc=recipe.checksum(sd), n_params=sum(p.numel() for p in m.parameters()),
         cls=cls.detach().numpy(), pat_head8=pat.detach().numpy()[:, :, :8], loss=loss.item(),
         keeps=torch.stack(rec.log).numpy(), keys=np.array([k for k, _ in shapes]),
         shapes=np.array([str(s) for _, s in shapes]))


# ------------------------------------------------------------------------------------------------
# F1b: state_dict schemas + param totals of the five shipped network_defs; F6: MACs
# ------------------------------------------------------------------------------------------------
def f6_schema_and_macs():
    out = {}
    nets = {"ref_tiny": (recipe.REF_TINY_DEF, None), "sr_tiny": (recipe.SR_TINY_DEF, "sr_tiny"),
            "sr_small": (recipe.SR_SMALL_DEF, "sr_small"), "sr_tiny_mh": (recipe.SR_TINY_MH_DEF, "sr_tiny_mh"),
            "sr_small_mh": (recipe.SR_SMALL_MH_DEF, "sr_small_mh")}
    import io
    import contextlib
    for name, (nd, cfg) in nets.items():
        kw = {}
        if cfg:
            kw = dict(num_channels_to_keep=R.cfg[cfg].num_channels_to_keep, example_per_arch=64, num_warmup_epochs=30)
        m = build_ref(nd, supernet=bool(cfg), img=224, classes=1000, **kw)
        out[name + ".keys"] = np.array(list(m.state_dict().keys()))
        out[name + ".shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
        out[name + ".n_params"] = sum(p.numel() for p in m.parameters())
        out[name + ".no_weight_decay"] = np.array(sorted(m.no_weight_decay()))
        out[name + ".n_channel_drop"] = sum(1 for d in m.modules() if isinstance(d, CD))
        with contextlib.redirect_stdout(io.StringIO()):
            est = R.flop.ComputationEstimator(distill=False, input_resolution=224, patch_size=14)
            out[name + ".macs"] = est(nd)
            est2 = R.flop.ComputationEstimator(distill=True, input_resolution=224, patch_size=14)
            out[name + ".macs_distill"] = est2(nd)
        if cfg:
            tbl = R.cfg[cfg].num_channels_to_keep
            flat = []
            for ent in tbl:
                if ent is None:
                    flat.append("None")
                elif isinstance(ent, dict):
                    flat.append(str({k: (None if v is None else [int(a) for a in v]) for k, v in ent.items()}))
                else:
                    flat.append(str([int(a) for a in ent]))
            out[name + ".choices"] = np.array(flat)
        del m
    with contextlib.redirect_stdout(io.StringIO()):
        for i, nd in enumerate(recipe.MICRO_CANDIDATES):
            est = R.flop.ComputationEstimator(distill=False, input_resolution=56, patch_size=14)
            out["micro_cand%d.macs" % i] = est(nd)
    save("f6_schema_macs", **out)


# ------------------------------------------------------------------------------------------------
# F5: get_sub_state_dict + masked-supernet == sliced sub-net
# ------------------------------------------------------------------------------------------------
def f5_subnet():
    sup = build_ref(recipe.MICRO_DEFS[0], supernet=True, num_channels_to_keep=recipe.micro_keep_config(),
                    example_per_arch=2, num_warmup_epochs=30)
    sd, _ = load_recipe(sup, seed=100)
    x, t, pt, labels = recipe.inputs(9, 6, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    out = {"state_crc": recipe.checksum(sd)}
    for i, nd in enumerate(recipe.MICRO_CANDIDATES):
        sub = build_ref(nd)
        ssd = R.net_utils.get_sub_state_dict(sup.state_dict(), sub.state_dict())
        sub.load_state_dict(ssd)
        sub.eval()
        with torch.no_grad():
            logits = sub(x.clone())
        out["cand%d.logits" % i] = logits.numpy()
        out["cand%d.crc" % i] = recipe.checksum(ssd)
        out["cand%d.qkv0" % i] = ssd["blocks.0.attn.qkv.weight"].numpy()
        out["cand%d.sr_w_sum" % i] = float(ssd["blocks.2.patch_reduce.weight"].double().sum())
    save("f5_subnet", **out)


# ------------------------------------------------------------------------------------------------
# F7: rewiring
# ------------------------------------------------------------------------------------------------
def f7_rewiring():
    m = build_ref(recipe.MICRO_DEFS[0], supernet=True, num_channels_to_keep=recipe.micro_keep_config(),
                  example_per_arch=2, num_warmup_epochs=30)
    sd, _ = load_recipe(m, seed=100)
    m.set_epoch(0)      # warm-up >= epoch -> every Block.rewiring()
    after = m.state_dict()
    out = {"state_crc": recipe.checksum(sd)}
    for i in (0, 3):
        for k in ("attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight"):
            key = "blocks.%d.%s" % (i, k)
            out[key] = after[key].numpy().copy()
    save("f7_rewiring", **out)


# ------------------------------------------------------------------------------------------------
# F8: engine.train_one_epoch / evaluate protocol (two iterations, AdamW)
# ------------------------------------------------------------------------------------------------
def f8_engine():
    import importlib
    engine = importlib.import_module("engine")
    for mode in ("multi", "single", None):
        torch.manual_seed(2024)
        np.random.seed(2024)
        sup = mode is not None
        kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30,
                  single_arch=(mode == "single")) if sup else {}
        m = build_ref(recipe.MICRO_DEFS[0], supernet=sup, **kw)
        sd, _ = load_recipe(m, seed=100)
        skip = m.no_weight_decay()
        decay, no_decay = [], []
        for n, p in m.named_parameters():
            (no_decay if (p.ndim == 1 or n.endswith(".bias") or n in skip) else decay).append(p)
        opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}],
                                lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
        batches = []
        for it in range(3):
            x, t, pt, labels = recipe.inputs(300 + it, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
            batches.append((x, (t, pt)))

        def patch_mixup(samples, targets):
            return samples, targets[0], targets[1], "seq"

        class Loader(list):
            pass

        def scaler(loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
            loss.backward()
            optimizer.step()

        losses = []

        class Crit(torch.nn.Module):
            def forward(self, x, t):
                return soft_ce(x, t)

        crit = Crit()
        if sup:
            m.set_epoch(31)
        torch.manual_seed(4321)          # RNG state at loop entry is part of the fixture
        rng_before = torch.random.get_rng_state().numpy().copy()
        with KeepRecorder() as rec:
            # targets are tuples: keep them on CPU; engine calls .to(device) on both -> wrap
            class T(tuple):
                def to(self, *a, **k):
                    return self
            loader = Loader([(x, T(tt)) for x, tt in batches])
            import io
            import contextlib
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                stats = engine.train_one_epoch(m, crit, loader, opt, torch.device("cpu"), 31, scaler,
                                               max_norm=None, model_ema=None, mixup_fn=None, print_freq=1,
                                               arch_sample=mode, patch_mixup_fn=patch_mixup)
        rng_after = torch.random.get_rng_state().numpy().copy()
        import re
        per_it = [float(v) for v in re.findall(r"loss: ([0-9.]+) \(", buf.getvalue())]
        out = dict(avg_loss=stats["loss"], lr=stats["lr"], state_crc=recipe.checksum(sd),
                   rng_unchanged=bool((rng_before == rng_after).all()), printed_losses=np.array(per_it))
        if rec.log:
            out["keeps"] = torch.stack(rec.log).numpy()
        after = m.state_dict()
        for k in ("tokens", "blocks.0.attn.qkv.weight", "blocks.6.mlp.fc2.bias", "cls_head.weight", "norm.weight"):
            out["after." + k] = after[k].numpy().copy()
        # exact per-iteration losses: replay with a fresh copy
        save("f8_engine_%s" % (mode or "plain"), **out)

    # evaluate protocol
    m = build_ref(recipe.MICRO_DEFS[0])
    sd, _ = load_recipe(m, seed=100)
    batches = []
    for it in range(2):
        x, t, pt, labels = recipe.inputs(400 + it, 8 if it == 0 else 4, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        batches.append((x, labels))
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        stats = engine.evaluate(batches, m, torch.device("cpu"), print_freq=1)
    save("f8_engine_eval", loss=stats["loss"], acc1=stats["acc1"], acc5=stats["acc5"], state_crc=recipe.checksum(sd))


# ------------------------------------------------------------------------------------------------
# F9: patch_output_type='avg' (vit_sr_supernet.py:447-449): patch head on the mean of the patch tokens
# ------------------------------------------------------------------------------------------------
def f9_patch_avg():
    B = 8
    out = {}
    for mode in ("plain", "multi"):
        torch.manual_seed(1234)
        kw = {}
        if mode != "plain":
            kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
        m = build_ref(recipe.MICRO_DEFS[0], supernet=(mode != "plain"), **kw)
        sd, shapes = load_recipe(m, seed=100)
        x, t, pt, labels = recipe.inputs(7, B, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        m.train()
        if mode != "plain":
            m.set_epoch(31)
            m.load_state_dict(sd)
        m.zero_grad()
        torch.manual_seed(586)
        with KeepRecorder() as rec:
            cls, pat = m(x.clone(), patch_output_type="avg")
        loss = soft_ce(cls, t) + soft_ce(pat, t)                   # engine.py:158-159: 'avg' is trained against `targets`
        loss.backward()
        out[mode + ".cls"], out[mode + ".pat"], out[mode + ".loss"] = cls.detach().numpy(), pat.detach().numpy(), loss.item()
        if rec.log:
            out[mode + ".keeps"] = torch.stack(rec.log).numpy()
        names = ["norm.weight", "norm.bias", "cls_head.weight", "patch_head.weight", "patch_head.bias",
                 "blocks.6.mlp.fc2.weight", "blocks.0.attn.qkv.weight", "pos_embed"]
        for k, v in grads_of(m, names).items():
            out[mode + "." + k] = v
    save("f9_patch_avg", **out)


# ------------------------------------------------------------------------------------------------
# F10: SwitchTokenMix (token_mixup.py:39-162): first half patch-level mix, second half image-level mixup
# ------------------------------------------------------------------------------------------------
def f10_token_mix():
    import importlib.util
    import numpy.random as npr
    spec = importlib.util.spec_from_file_location("ref_token_mixup", os.path.join(ref_shim.REF, "token_mixup.py"))
    tm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tm)
    # the reference hard-codes device='cuda' in torch.full / torch.zeros: drop the kwarg while it runs on CPU
    full, zeros = torch.full, torch.zeros
    strip = lambda f: (lambda *a, **k: f(*a, **{kk: vv for kk, vv in k.items() if kk != "device"}))   # noqa: E731
    out = {}
    try:
        torch.full, torch.zeros = strip(full), strip(zeros)
        for case, (B, H, pl, nc, seed) in enumerate([(8, 56, 4, 10, 0), (8, 56, 4, 10, 1), (6, 32, 4, 7, 2), (16, 28, 2, 5, 3)]):
            rs = np.random.RandomState(100 + case)
            x = torch.from_numpy(rs.standard_normal((B, 3, H, H)).astype(np.float32))
            y = torch.from_numpy(rs.randint(0, nc, size=(B,)).astype(np.int64))
            torch.manual_seed(40 + seed)
            npr.seed(50 + seed)
            mix = tm.SwitchTokenMix(pl, switch_prob=0.5, num_classes=nc, smoothing=0.1)
            xs, t, pt, pot = mix(x.clone(), y.clone())
            tag = "c%d." % case
            out[tag + "cfg"] = np.array([B, H, pl, nc, seed])
            out[tag + "samples"], out[tag + "targets"], out[tag + "patch_targets"] = xs.numpy(), t.numpy(), pt.numpy()
            assert pot == "seq"
            out[tag + "np_after"] = npr.randint(0, 1 << 30)             # RNG streams advanced exactly as far
            out[tag + "torch_after"] = torch.randint(0, 1 << 30, (1,)).numpy()
    finally:
        torch.full, torch.zeros = full, zeros
    save("f10_token_mix", **out)


# ------------------------------------------------------------------------------------------------
# F11: positional-embedding interpolation for higher-resolution fine-tuning (network_utils/finetune_state_dict.py:24-65)
# ------------------------------------------------------------------------------------------------
def f11_pos_embed_interp():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ft", os.path.join(ref_shim.REF, "network_utils", "finetune_state_dict.py"))
    ft = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ft)
    lo = build_ref(recipe.MICRO_DEFS[0], img=56)
    hi = build_ref(recipe.MICRO_DEFS[0], img=140)          # grid 4 -> 10 (SR grids 2 -> 5, 1 -> 2)
    sd, _ = load_recipe(lo, seed=321)
    out = ft.state_dict_interpolate_pos_embed(hi.state_dict(), {k: v.clone() for k, v in sd.items()})
    save("f11_pos_embed_interp", **{k: v.numpy() for k, v in out.items() if "pos_embed" in k})


# ------------------------------------------------------------------------------------------------
# F12: parameter initialisation from a seed (vit_sr_supernet.py:352-376: trunc_normal(.02) Linears / tokens / pos_embeds, default
#      Conv2d init, LN ones / zeros) -- the order random numbers are consumed in is part of the drop-in surface
# ------------------------------------------------------------------------------------------------
def f12_init():
    out = {}
    for et, sup in ((0, False), (0, True), (4, True), (5, False)):
        kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30) if sup else {}
        torch.manual_seed(77)
        m = build_ref(recipe.MICRO_DEFS[et], supernet=sup, **kw)
        sd = m.state_dict()
        out["t%d_%d.crc" % (et, int(sup))] = recipe.checksum(sd)
        out["t%d_%d.tokens" % (et, int(sup))] = sd["tokens"].numpy()
        out["t%d_%d.head" % (et, int(sup))] = sd["cls_head.weight"].numpy()
    save("f12_init", **out)


# ------------------------------------------------------------------------------------------------
# F14: two-token (class + distillation token) variants -- factories flexible_vit_sr_distill_patch14_224[_supernet]
#      (vit_sr_supernet.py:204-359 with distill_token=True; forward :455-460 returns (cls_pred, dst_pred) in train AND eval)
# ------------------------------------------------------------------------------------------------
def f14_distill_token():
    B = 8
    for et in (0, 4):
        nd = recipe.MICRO_DEFS[et]
        for mode in ("plain", "multi"):
            torch.manual_seed(1234)
            kw = {}
            if mode != "plain":
                kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
            m = R.vit_sr.FlexibleDistillVisionTransformerSR(
                img_size=recipe.MICRO_IMG, patch_size=14, num_classes=recipe.MICRO_CLASSES, distill_token=True, network_def=nd,
                patch_output=False, supernet=(mode != "plain"), **kw)
            sd, shapes = load_recipe(m, seed=140 + et)
            x, t, pt, labels = recipe.inputs(7, B, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
            t2 = pt[:, 0, :].contiguous()                        # soft target of the distillation head
            out = {"state_crc": recipe.checksum(sd), "keys": np.array([k for k, _ in shapes]),
                   "shapes": np.array([str(s) for _, s in shapes]), "no_weight_decay": np.array(sorted(m.no_weight_decay()))}
            m.train()
            if mode != "plain":
                m.set_epoch(31)
                m.load_state_dict(sd)
            m.zero_grad()
            torch.manual_seed(555 + 31)
            with KeepRecorder() as rec:
                cls, dst = m(x.clone())
            loss = soft_ce(cls, t) + soft_ce(dst, t2)
            loss.backward()
            out["cls"], out["dst"], out["loss"] = cls.detach().numpy(), dst.detach().numpy(), loss.item()
            if rec.log:
                out["keeps"] = torch.stack(rec.log).numpy()
            names = [n for n in GRAD_NAMES_COMMON if n in dict(m.named_parameters())] + ["dst_head.weight", "dst_head.bias", "cls_head.bias"]
            names += ["patch_embed.proj.weight"] if et == 0 else ["patch_embed.conv_proj.weight", "patch_embed.conv1.conv.weight"]
            for k, v in grads_of(m, names).items():
                out[k] = v
            m.load_state_dict(sd)
            m.eval()
            with torch.no_grad():
                ec, ed = m(x.clone())
            out["eval.cls"], out["eval.dst"] = ec.numpy(), ed.numpy()
            save("f14_distill_t%d_%s" % (et, mode), **out)


# ------------------------------------------------------------------------------------------------
# F15: knowledge distillation through the engine (engine.py:25-46 KnowledgeDistillationLoss; :112-148 teacher forward,
#      loss = (1 - alpha) * criterion(cls, targets) + alpha * kd(dst, teacher_output)) on the two-token micro net
# ------------------------------------------------------------------------------------------------
def f15_distillation_engine():
    import contextlib
    import importlib
    import io
    engine = importlib.import_module("engine")
    out = {}
    g = torch.Generator().manual_seed(5)
    xs, ts = torch.randn(8, 10, generator=g), torch.randn(8, 10, generator=g) * 2
    for hard in (True, False):
        xv = xs.clone().requires_grad_(True)
        loss = engine.KnowledgeDistillationLoss(hard_distill=hard)(xv, ts)
        loss.backward()
        out["kd.%s.loss" % ("hard" if hard else "soft")] = loss.item()
        out["kd.%s.grad" % ("hard" if hard else "soft")] = xv.grad.numpy().copy()
    for mode in ("plain", "multi"):
        for hard in (True, False):
            torch.manual_seed(2024)
            sup = mode != "plain"
            kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30) if sup else {}
            m = R.vit_sr.FlexibleDistillVisionTransformerSR(
                img_size=recipe.MICRO_IMG, patch_size=14, num_classes=recipe.MICRO_CLASSES, distill_token=True,
                network_def=recipe.MICRO_DEFS[0], patch_output=False, supernet=sup, **kw)
            sd, _ = load_recipe(m, seed=140)
            skip = m.no_weight_decay()
            decay, no_decay = [], []
            for n, p in m.named_parameters():
                (no_decay if (p.ndim == 1 or n.endswith(".bias") or n in skip) else decay).append(p)
            opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}], lr=1e-3)
            loader = []
            for it in range(3):
                x, t, pt, labels = recipe.inputs(300 + it, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
                loader.append((x, t))

            def scaler(loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
                loss.backward()
                optimizer.step()

            class Crit(torch.nn.Module):
                def forward(self, x, t):
                    return soft_ce(x, t)
            if sup:
                m.set_epoch(31)
            torch.manual_seed(4321)
            with KeepRecorder() as rec, contextlib.redirect_stdout(io.StringIO()):
                stats = engine.train_one_epoch(m, Crit(), loader, opt, torch.device("cpu"), 31, scaler, max_norm=None,
                                               model_ema=None, mixup_fn=None, print_freq=1,
                                               teacher_model=recipe.toy_teacher(recipe.MICRO_CLASSES), hard_distill=hard, alpha=0.5,
                                               arch_sample=("multi" if sup else None), patch_mixup_fn=None)
            tag = "%s.%s." % (mode, "hard" if hard else "soft")
            out[tag + "avg_loss"] = stats["loss"]
            if rec.log:
                out[tag + "keeps"] = torch.stack(rec.log).numpy()
            after = m.state_dict()
            for k in ("tokens", "blocks.0.attn.qkv.weight", "blocks.6.mlp.fc2.bias", "cls_head.weight", "dst_head.weight", "norm.weight"):
                out[tag + "after." + k] = after[k].numpy().copy()
    save("f15_distillation_engine", **out)


# ------------------------------------------------------------------------------------------------
# F16: the single-stage patch-16 sibling (nets/vision_transformer_supernet.py:45-283): schema, init from a seed, (cls, dst)
#      logits, loss, gradients, masks, eval outputs
# ------------------------------------------------------------------------------------------------
def f16_vit16():
    import importlib
    vt = importlib.import_module("nets.vision_transformer_supernet")
    B = 8
    for mode in ("plain", "multi"):
        sup = mode != "plain"
        kw = dict(num_channels_to_keep=recipe.vit16_keep_config(), example_per_arch=2, num_warmup_epochs=30) if sup else {}
        torch.manual_seed(77)
        m = vt.FlexibleDistillVisionTransformer(img_size=recipe.VIT16_IMG, patch_size=16, num_classes=recipe.MICRO_CLASSES,
                                                distill_token=True, network_def=recipe.VIT16_DEF, supernet=sup, **kw)
        out = {"init_crc": recipe.checksum(m.state_dict()), "no_weight_decay": np.array(sorted(m.no_weight_decay()))}
        sd, shapes = load_recipe(m, seed=160)
        out.update(state_crc=recipe.checksum(sd), keys=np.array([k for k, _ in shapes]), shapes=np.array([str(s_) for _, s_ in shapes]))
        x, t, pt, labels = recipe.inputs(7, B, recipe.VIT16_IMG, recipe.MICRO_CLASSES, 1)
        t2 = pt[:, 0, :].contiguous()
        m.train()
        if sup:
            m.set_epoch(31)
            m.load_state_dict(sd)
        m.zero_grad()
        torch.manual_seed(555 + 31)
        with KeepRecorder() as rec:
            cls, dst = m(x.clone())
        loss = soft_ce(cls, t) + soft_ce(dst, t2)
        loss.backward()
        out["cls"], out["dst"], out["loss"] = cls.detach().numpy(), dst.detach().numpy(), loss.item()
        if rec.log:
            out["keeps"] = torch.stack(rec.log).numpy()
        for n, p in m.named_parameters():
            if n.startswith(("tokens", "pos_embed", "patch_embed", "blocks.0.", "blocks.3.mlp.fc2", "norm.", "cls_head", "dst_head")):
                out["grad." + n] = p.grad.detach().numpy().copy()
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            ec, ed = m(x.clone())
        out["eval.cls"], out["eval.dst"] = ec.numpy(), ed.numpy()
        save("f16_vit16_%s" % mode, **out)
    # choice tables of the patch-16 spaces (supernet_config/tiny.py, tiny_deep.py, small_deep.py)
    tables = {}
    for name in ("tiny", "tiny_deep", "small_deep"):
        tbl = importlib.import_module("supernet_config." + name).num_channels_to_keep
        flat = []
        for ent in tbl:
            if ent is None:
                flat.append("None")
            elif isinstance(ent, dict):
                flat.append(str({k: (None if v is None else [int(a) for a in v]) for k, v in ent.items()}))
            else:
                flat.append(str([int(a) for a in ent]))
        tables[name] = np.array(flat)
    save("f16_patch16_spaces", **tables)


# ------------------------------------------------------------------------------------------------
# F17: repeated-augmentation sampler (samplers.py:12-64)
# ------------------------------------------------------------------------------------------------
def f17_ra_sampler():
    import importlib
    smp = importlib.import_module("samplers")
    out = {}
    for n, world in ((1000, 4), (777, 3), (5000, 8), (300, 2)):
        ds = list(range(n))
        for rank in (0, world - 1):
            for shuffle in (True, False):
                sp = smp.RASampler(ds, num_replicas=world, rank=rank, shuffle=shuffle)
                sp.set_epoch(5)
                out["n%d_w%d_r%d_s%d" % (n, world, rank, int(shuffle))] = np.array(list(iter(sp)), dtype=np.int64)
                out["n%d_w%d_r%d_s%d.len" % (n, world, rank, int(shuffle))] = len(sp)
    save("f17_ra_sampler", **out)


# ------------------------------------------------------------------------------------------------
# F13: evolutionary-search population bookkeeping + candidate generation (search_utils/evolver.py, gen_utils.py) under a seed:
#      the reference's own classes, numpy global RNG, sr_tiny and sr_small spaces, toy scores from recipe.toy_candidate_score
# ------------------------------------------------------------------------------------------------
def f13_evolver():
    import contextlib
    import importlib
    import io
    import json
    evo = importlib.import_module("search_utils.evolver")
    gen = importlib.import_module("search_utils.gen_utils")
    flop = importlib.import_module("search_utils.compute_flop_mac")
    out = {}
    runs = {"sr_tiny": (recipe.SR_TINY_DEF, 0.45, 0, 12, 3, 6, 6), "sr_small": (recipe.SR_SMALL_DEF, None, 3, 10, 2, 5, 4)}
    with contextlib.redirect_stdout(io.StringIO()):
        for name, (nd, frac, seed, n_init, n_gen, parents, msize) in runs.items():
            est = flop.ComputationEstimator(distill=False, input_resolution=224, patch_size=14)
            constraint = est(nd) * frac if frac else 2.9e9          # 2.9e9: evolutionary_search/no_distill/small_flexible-conv-patch.sh:19
            keep = R.cfg[name].num_channels_to_keep
            pe = evo.PopulationEvolver(largest_network_def=nd, num_channels_to_keep=keep, constraint=constraint, compute_resource=est)
            np.random.seed(seed)
            gens = []
            for it in range(1 + n_gen):
                if it == 0:
                    pe.random_sample(n_init)
                else:
                    pe.evolve_sample(parent_size=parents, mutate_prob=0.3, mutate_size=msize)
                for ind in pe.popu:
                    ind.score = recipe.toy_candidate_score(ind.network_def)
                gens.append([[ind.network_def, ind.score] for ind in pe.popu])
                pe.update_history()
                pe.sort_history()
            out[name + ".generations"] = json.dumps(gens)
            out[name + ".history"] = json.dumps([[ind.network_def, ind.score] for ind in pe.history_popu])
            out[name + ".constraint"] = float(constraint)
            out[name + ".rng_after"] = float(np.random.uniform())
            # the pieces one by one, from a fresh seed
            np.random.seed(seed + 100)
            a = gen.gen_random_network_def(nd, keep, constraint, est)
            b = gen.gen_random_network_def(nd, keep, constraint, est)
            m = gen.mutate_network_def(a, keep, 0.3, constraint, est)
            c = gen.crossover_network_def(a, b, keep, constraint, est)
            p = gen.tupleit(gen.reduce_constraint(nd, keep, constraint * 0.8, est))
            out[name + ".pieces"] = json.dumps([a, b, m, c, p])
    save("f13_evolver", **out)

# ------------------------------------------------------------------------------------------------
# F18: DropPath (nets/drop.py:11-26) with INJECTED uniform draws + full-size gradients + config C4 (sr_small)
# ------------------------------------------------------------------------------------------------
class RandInjector:
    """nets/drop.py:23 is the only torch.rand call on the path: while active, every call returns the next row of a prepared
    noise table (caller order, shape (B,)) reshaped to the requested shape; rows are consumed in call order."""

    def __init__(self, noise):
        self.noise, self.used = list(noise), 0
        self._orig = torch.rand

    def __enter__(self):
        inj = self

        def rand(shape, *a, **k):
            row = inj.noise[inj.used]
            inj.used += 1
            return row.reshape(shape).to(k.get("dtype") or torch.float32)
        torch.rand = rand
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


def grad_samples(model):
    """Per parameter: L2 norm (float64) and recipe.GRAD_SAMPLES elements at recipe.grad_sample_index positions."""
    out = {}
    for n, p_ in model.named_parameters():
        g = p_.grad.detach().reshape(-1)
        idx = recipe.grad_sample_index(n, g.numel())
        out["gn." + n] = float(g.double().norm())
        out["gs." + n] = g[torch.from_numpy(idx)].numpy().copy()
    return out


def f18_droppath_fullsize_c4():
    # (a) micro supernets, embed types 0 and 4, 'multi', drop_path 0.2, explicit draws incl. dropped samples
    B = 8
    for et in (0, 4):
        nd = recipe.MICRO_DEFS[et]
        kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
        m = build_ref(nd, supernet=True, drop_path_rate=0.2, **kw)
        sd, shapes = load_recipe(m, seed=100 + et)
        x, t, pt, labels = recipe.inputs(7, B, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
        m.train()
        m.set_epoch(31)
        m.load_state_dict(sd)
        rates = [b.drop_path.drop_prob for b in m.blocks if hasattr(b, "drop_path") and not isinstance(b.drop_path, torch.nn.Identity)]
        noise = recipe.drop_path_noise(1800 + et, rates, B)
        torch.manual_seed(555 + 31)
        with KeepRecorder() as rec, RandInjector([torch.from_numpy(r) for r in noise]) as inj:
            cls, pat = m(x.clone(), patch_output_type="seq")
        assert inj.used == len(noise)
        loss = soft_ce(cls, t) + soft_ce(pat, pt)
        loss.backward()
        out = dict(state_crc=recipe.checksum(sd), rates=np.array(rates), noise=np.stack(noise), cls=cls.detach().numpy(),
                   pat=pat.detach().numpy(), loss=loss.item(), keeps=torch.stack(rec.log).numpy())
        out.update({"grad." + n: p_.grad.detach().numpy().copy() for n, p_ in m.named_parameters()})
        save("f18_micro_t%d_multi_dp" % et, **out)

    # (b) full size: sr_tiny supernet (C3 geometry) B=8 with drop_path 0.2, ref-tiny (C1/C2 geometry) B=2, sr_small supernet
    # (C4: super_net/no_distill/small_flexible-conv-patch.sh:19, drop_path 0.3) B=8 -- logits, loss, keeps, gradient samples
    cases = [("sr_tiny_c3", recipe.SR_TINY_DEF, "sr_tiny", 0.2, 8, 4343, 12),
             ("ref_tiny_c2", recipe.REF_TINY_DEF, None, 0.2, 2, 4242, 11),
             ("sr_small_c4", recipe.SR_SMALL_DEF, "sr_small", 0.3, 8, 4444, 13)]
    for name, nd, space, dp, B, wseed, iseed in cases:
        kw = {}
        if space:
            kw = dict(num_channels_to_keep=R.cfg[space].num_channels_to_keep, example_per_arch=2, num_warmup_epochs=30)
        torch.manual_seed(0)
        m = build_ref(nd, supernet=bool(space), img=224, classes=1000, drop_path_rate=dp, **kw)
        sd, shapes = load_recipe(m, seed=wseed)
        x, t, pt, labels = recipe.inputs(iseed, B, 224, 1000, 16)
        m.train()
        if space:
            m.set_epoch(31)
            m.load_state_dict(sd)
        rates = [b.drop_path.drop_prob for b in m.blocks if hasattr(b, "drop_path") and not isinstance(b.drop_path, torch.nn.Identity)]
        noise = recipe.drop_path_noise(1900 + iseed, rates, B)
        torch.manual_seed(77)
        with KeepRecorder() as rec, RandInjector([torch.from_numpy(r) for r in noise]) as inj:
            cls, pat = m(x.clone(), patch_output_type="seq")
        assert inj.used == len(noise)
        loss = soft_ce(cls, t) + soft_ce(pat, pt)
        loss.backward()
        out = dict(state_crc=recipe.checksum(sd), n_params=sum(p_.numel() for p_ in m.parameters()), rates=np.array(rates),
                   noise=np.stack(noise), cls=cls.detach().numpy(), pat_head8=pat.detach().numpy()[:, :, :8], loss=loss.item(),
                   keys=np.array([k for k, _ in shapes]))
        if rec.log:
            out["keeps"] = torch.stack(rec.log).numpy()
        if not space:                                   # conv stem: BatchNorm running statistics after the forward
            bsd = m.state_dict()
            for bn in ("conv1", "conv2", "conv3"):
                out["bn.%s.running_mean" % bn] = bsd["patch_embed.%s.bn.running_mean" % bn].numpy().copy()
                out["bn.%s.running_var" % bn] = bsd["patch_embed.%s.bn.running_var" % bn].numpy().copy()
        out.update(grad_samples(m))
        save("f18_" + name, **out)
        del m


def f19_flops():
    """ComputationEstimator(return_mac=False): FLOP counts (multiply-adds x 2, biases, softmax / LayerNorm / GELU / residual
    terms, compute_flop_mac.py:53-194) of the five shipped network_defs with and without the distillation token, of the
    micro candidates at 56 px, and of the README's patch-16 ViT-T example."""
    import io
    import contextlib
    out = {}
    nets = {"ref_tiny": recipe.REF_TINY_DEF, "sr_tiny": recipe.SR_TINY_DEF, "sr_small": recipe.SR_SMALL_DEF,
            "sr_tiny_mh": recipe.SR_TINY_MH_DEF, "sr_small_mh": recipe.SR_SMALL_MH_DEF}
    with contextlib.redirect_stdout(io.StringIO()):
        for name, nd in nets.items():
            for distill in (False, True):
                est = R.flop.ComputationEstimator(distill=distill, input_resolution=224, patch_size=14, return_mac=False)
                out["%s.flops%s" % (name, "_distill" if distill else "")] = est(nd)
        for i, nd in enumerate(recipe.MICRO_CANDIDATES):
            est = R.flop.ComputationEstimator(distill=False, input_resolution=56, patch_size=14, return_mac=False)
            out["micro_cand%d.flops" % i] = est(nd)
        vit_t = ((0, 192),) + ((1, (192, 3, 64), (192, 768), 1),) * 12 + ((2, 192, 1000),)
        for mac in (True, False):
            est = R.flop.ComputationEstimator(distill=True, input_resolution=224, patch_size=16, return_mac=mac)
            out["vit_t_p16." + ("macs" if mac else "flops")] = est(vit_t)
    save("f19_flops", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2", "f3", "f4", "f5", "f6", "f7", "f8", "f9", "f10", "f11", "f12", "f13", "f14", "f15", "f16", "f17", "f18", "f19"]
    table = dict(f1=f1_micro, f2=f2_masked_ln, f3=f3_channel_drop, f4=f4_fullsize, f5=f5_subnet,
                 f6=f6_schema_and_macs, f7=f7_rewiring, f8=f8_engine, f9=f9_patch_avg, f10=f10_token_mix, f11=f11_pos_embed_interp, f12=f12_init, f13=f13_evolver, f14=f14_distill_token, f15=f15_distillation_engine, f16=f16_vit16, f17=f17_ra_sampler, f18=f18_droppath_fullsize_c4, f19=f19_flops)
    for w in which:
        table[w]()
