"""Checkpoint compatibility with the reference layout (main.py:400-424,496-515): the dict keys, the
torch.optim.AdamW layout of 'optimizer' (FlatAdamW <-> per-parameter moments), resume / eval-with-EMA / supernet inheritance."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import recipe  # noqa: E402

import vitres  # noqa: E402
from vitres import checkpoint, engine  # noqa: E402
from vitres.optim import FlatAdamW  # noqa: E402


def micro(supernet=True, nd=None):
    kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30) if supernet else {}
    name = "flexible_vit_sr_patch14_224_patch_output" + ("_supernet" if supernet else "")
    return vitres.create_model(name, img_size=recipe.MICRO_IMG, num_classes=recipe.MICRO_CLASSES,
                               network_def=nd or recipe.MICRO_DEFS[0], **kw)


def torch_adamw_with_state(model, steps=3, seed=0):
    """torch.optim.AdamW over timm's two groups, stepped on CPU with seeded random gradients: a 'reference' optimizer state."""
    opt = torch.optim.AdamW(engine.param_groups_weight_decay(model, 0.05), lr=1e-3)
    g = torch.Generator().manual_seed(seed)
    for _ in range(steps):
        for p in model.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        opt.step()
    for p in model.parameters():
        p.grad = None
    return opt


def test_flat_adamw_imports_and_exports_the_torch_adamw_layout():
    torch.manual_seed(1)
    m = micro()
    ref = torch_adamw_with_state(m)
    sd = ref.state_dict()
    flat = FlatAdamW(m, engine.param_groups_weight_decay(m, 0.05), lr=5e-4)
    flat.load_state_dict(sd)                                    # 'state' key -> torch layout
    assert flat._step == 3 and flat.param_groups[0]["lr"] == 1e-3 and flat.param_groups[1]["weight_decay"] == 0.05
    back = flat.torch_state_dict()
    assert [g["params"] for g in back["param_groups"]] == [g["params"] for g in sd["param_groups"]]
    assert set(back["state"]) == set(sd["state"])
    for i, st in sd["state"].items():
        assert torch.equal(back["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(back["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
        assert float(back["state"][i]["step"]) == float(st["step"])
    torch.optim.AdamW(engine.param_groups_weight_decay(m, 0.05), lr=1e-3).load_state_dict(back)      # torch accepts it
    own = flat.state_dict()                                     # the flat layout still round-trips
    flat2 = FlatAdamW(m, engine.param_groups_weight_decay(m, 0.05), lr=5e-4)
    flat2.load_state_dict(own)
    assert torch.equal(flat2._flat_state["m"], flat._flat_state["m"]) and flat2._step == 3
    with pytest.raises(ValueError):
        FlatAdamW(m, [p for p in m.parameters()], lr=1e-3).load_state_dict(sd)      # one group vs two


def test_checkpoint_dict_layout_resume_and_eval_with_ema(tmp_path):
    torch.manual_seed(2)
    m = micro()
    opt = FlatAdamW(m, engine.param_groups_weight_decay(m, 0.05), lr=1e-3)
    opt.load_state_dict(torch_adamw_with_state(m).state_dict())
    sched = torch.optim.lr_scheduler.LambdaLR(torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=1.0), lambda e: 0.5 ** e)
    ema = {k: v + 1.0 if v.dtype.is_floating_point else v for k, v in m.state_dict().items()}
    path = checkpoint.save_checkpoint(str(tmp_path), m, opt, sched, epoch=9, args={"lr": 1e-3}, model_ema=ema)
    assert os.path.basename(path) == "checkpoint.pth.tar" and os.path.exists(os.path.join(tmp_path, "epoch@9_checkpoint.pth.tar"))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    # the reference's six keys (main.py:506-512) + this stack's DropPath generator state (the reference's loaders index by name)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch", "args", "model_ema", "vitres_rng"} and ck["epoch"] == 9
    assert list(ck["model"].keys()) == list(m.state_dict().keys())
    assert set(ck["optimizer"]) == {"state", "param_groups"}
    torch.manual_seed(3)
    m2 = micro()
    opt2 = FlatAdamW(m2, engine.param_groups_weight_decay(m2, 0.05), lr=7e-4)
    got = {}
    start = checkpoint.resume(path, m2, opt2, None, load_ema=got.update)
    assert start == 10 and opt2._step == 3 and recipe.checksum(m2.state_dict()) == recipe.checksum(m.state_dict())
    assert torch.equal(m2.drop_path_rng_state(), m.drop_path_rng_state())        # a resumed run continues the same noise stream
    m.drop_path_generator(seed=123)
    assert not torch.equal(m2.drop_path_rng_state(), m.drop_path_rng_state())
    assert torch.equal(opt2._flat_state["v"], opt._flat_state["v"]) and set(got) == set(ema)
    m3 = micro()
    assert checkpoint.resume(path, m3, eval_mode=True) is None                   # --eval: EMA weights, no optimizer
    assert torch.equal(m3.state_dict()["cls_head.weight"], ema["cls_head.weight"])
    del ck["lr_scheduler"]
    assert checkpoint.resume(ck, micro(), FlatAdamW(m3, m3.parameters())) is None   # weights-only checkpoint: no epoch to resume


def test_subnet_inherits_prefix_slices_of_a_supernet_checkpoint(tmp_path):
    sup = micro()
    sup.load_state_dict(recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in sup.state_dict().items()], 100))
    ck = {"model": sup.state_dict()}
    nd = recipe.MICRO_CANDIDATES[1]
    sub = checkpoint.inherit_supernet_weights(micro(False, nd), ck)
    w_sup, w_sub = sup.state_dict()["blocks.0.mlp.fc1.weight"], sub.state_dict()["blocks.0.mlp.fc1.weight"]
    assert torch.equal(w_sub, w_sup[:w_sub.shape[0], :w_sub.shape[1]])
    import numpy as np
    g = np.load(os.path.join(HERE, "golden", "f5_subnet.npz"))                   # the reference's own sliced state (F5)
    assert recipe.checksum(sub.state_dict()) == int(g["cand1.crc"])


def test_flat_adamw_ema_survives_resume(tmp_path):
    """a FlatAdamW EMA is written as 'model_ema' and must come back through resume (reference
    utils._load_checkpoint_for_ema), not restart from the raw weights."""
    torch.manual_seed(5)
    m = micro()
    opt = FlatAdamW(m, engine.param_groups_weight_decay(m, 0.05), lr=1e-3, ema_decay=0.99)
    opt._bind()
    with torch.no_grad():
        opt._flat_state["ema"].add_(0.25)                          # an EMA that differs from the weights
    ema_sd = opt.ema_state_dict()
    path = checkpoint.save_checkpoint(str(tmp_path), m, opt, None, epoch=3, model_ema=ema_sd)
    torch.manual_seed(6)
    m2 = micro()
    opt2 = FlatAdamW(m2, engine.param_groups_weight_decay(m2, 0.05), lr=1e-3, ema_decay=0.99)
    assert checkpoint.resume(path, m2, opt2, None) == 4
    assert not torch.equal(opt2._flat_state["ema"], m2._arena["flat"])      # not restarted from the raw weights
    back = opt2.ema_state_dict()
    assert all(torch.equal(back[k], ema_sd[k]) for k in ema_sd)
    bad = dict(ema_sd)
    del bad["cls_head.weight"]
    with pytest.raises(KeyError):
        opt2.load_ema_state_dict(bad)


def test_finetune_loader_reads_reference_layout_with_namespace_args(tmp_path):
    """reference checkpoints store args as argparse.Namespace; torch >= 2.6 defaults to weights_only=True."""
    import argparse
    from vitres.network_utils.finetune_state_dict import load_interpolated_state_dict
    m = micro(False)
    path = checkpoint.save_checkpoint(str(tmp_path), m, torch.optim.SGD(m.parameters(), lr=0.1), None, epoch=0,
                                      args=argparse.Namespace(lr=1e-3, model="x"))
    sd = load_interpolated_state_dict(m.state_dict(), path)
    assert recipe.checksum(sd) == recipe.checksum(m.state_dict())
