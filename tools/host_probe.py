#!/usr/bin/env python
"""Where the host time of one graphed training step goes (dev tool; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from vitres import engine
from vitres.optim import FlatAdamW
from vitres.losses import SoftTargetCrossEntropy
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model, nd = bench.build_model("sr_tiny_supernet", torch.bfloat16, dev)
x, t, pt = bench.synthetic_batch(128, dev, 1000)
model.train(); model.set_epoch(31); model._ensure_arena(dev)
opt = FlatAdamW(model, engine.param_groups_weight_decay(model, 0.05), lr=1e-4); opt.own_shadow()
g = engine.GraphedTrainStep(model, SoftTargetCrossEntropy(), x, t, pt, "seq")
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
n = 40
for i in range(n + 5):
    if i == 5:
        torch.cuda.synchronize(); T.clear()
    t0 = time.perf_counter(); torch.randperm(64); torch.randperm(64); tick("mixup draws", t0)
    t0 = time.perf_counter(); rng = torch.random.get_rng_state(); plan = model.sample_plan(128); torch.random.set_rng_state(rng); tick("sample_plan (+rng save/restore)", t0)
    t0 = time.perf_counter(); flat, _ = model.plan_host_buffer(plan); tick("plan_host_buffer", t0)
    t0 = time.perf_counter()
    slot = g._stage[g._stage_i % len(g._stage)]; g._stage_i += 1
    if slot[1] is not None: slot[1].synchronize()
    slot[0].numpy()[:] = flat; g.keep_static.copy_(slot[0], non_blocking=True); slot[1] = torch.cuda.Event(); slot[1].record()
    tick("staging copy", t0)
    t0 = time.perf_counter(); g._gather(x, plan); tick("gather launch", t0)
    t0 = time.perf_counter(); g.graph.replay(); tick("graph.replay", t0)
    t0 = time.perf_counter(); opt.step(); tick("opt.step", t0)
torch.cuda.synchronize()
for k, v in T.items():
    print("%-34s %7.3f ms" % (k, v / n * 1e3))
print("%-34s %7.3f ms" % ("total", sum(T.values()) / n * 1e3))
