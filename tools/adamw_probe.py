#!/usr/bin/env python
"""vr_adamw_flat_dev over ranges of the sr_tiny arena alone (dev probe): GB/s by range size and workgroup cap."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
import bench
from vitres import engine
from vitres.optim import FlatAdamW
model, nd = bench.build_model("sr_tiny_supernet", torch.bfloat16, "cuda")
model.train(); model.set_epoch(31)
x, t, pt = bench.synthetic_batch(128, "cuda", 0)
opt = FlatAdamW(model, engine.param_groups_weight_decay(model, 0.05), lr=1e-3)
opt.own_shadow()
loss = model.loss_and_grad(x, t, pt, "seq")
opt.prepare_step()
n = model._arena["flat"].numel()
cuts = model.split_plan(parts=3)
head = cuts[0][1]
print("arena %d parameters; head range [0, %d)" % (n, head))
for lo, hi, cap in [(0, head, 0), (0, head, 2048), (0, head, 1024), (0, head, 512), (head, n, 0), (head, n, 256), (0, n, 0)]:
    hi8 = hi // 8 * 8
    for _ in range(3):
        opt.step_device(lo, hi8, max_blocks=cap)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        opt.step_device(lo, hi8, max_blocks=cap)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("[%9d, %9d) cap %5d: %7.1f us  %5.2f TB/s" % (lo, hi8, cap, us, (hi8 - lo) * 30 / us / 1e6))
