#!/usr/bin/env python
"""Sanity beyond parity: the whole stack (hipGraph step, masks, FlatAdamW, bf16 shadows) actually trains -- overfit one
synthetic batch of the sr_tiny supernet with hard one-hot targets and watch the loss fall (dev tool; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
import bench
from vitres import engine
from vitres.losses import SoftTargetCrossEntropy
from vitres.optim import FlatAdamW

dev = torch.device("cuda:0")
torch.manual_seed(0)
model, nd = bench.build_model("sr_tiny_supernet", torch.bfloat16, dev)
model.train(); model.set_epoch(31); model._ensure_arena(dev)
B = 128
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
y = torch.randint(0, 1000, (B,), generator=g)
t = torch.nn.functional.one_hot(y, 1000).float().to(dev)
pt = t[:, None, :].repeat(1, 16, 1).contiguous()
opt = FlatAdamW(model, engine.param_groups_weight_decay(model, 0.05), lr=3e-4)
opt.own_shadow()
step = engine.GraphedTrainStep(model, SoftTargetCrossEntropy(), x, t, pt, "seq")
sync = engine.GradSync(model)
hist = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    loss = step.step_with_sync(sync, x, t, pt, average=False, epoch=31, train_iter=i, arch_sample="multi")
    opt.step()
    if i % 25 == 0 or i == 199:
        hist.append((i, round(float(loss), 4)))
print("loss trajectory (cls CE + patch CE, different sub-networks every step):", hist)
assert hist[-1][1] < 0.25 * hist[0][1], "loss did not fall"
print("OK")
