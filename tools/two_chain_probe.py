"""Feasibility probe for the two-half-batch-chains idea: how long does the graph of one
B = 128 step take against TWO independent B = 64 step graphs (two model instances, one architecture group each) replayed
concurrently on two streams?  No optimizer in either (the update does not change with the split).  Prints ms per 128 images."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from vitres import engine
from vitres.losses import SoftTargetCrossEntropy

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "sr_tiny_supernet"
w = bench.WORKLOADS[wl]
B = w["batch"]
crit = SoftTargetCrossEntropy()


def make(batch, seed):
    torch.manual_seed(seed)
    model, _ = bench.build_model(wl, torch.bfloat16, dev)
    x, t, pt = bench.synthetic_batch(batch, dev, 1000 + seed)
    model.train()
    if w["space"]:
        model.set_epoch(31)
    model._ensure_arena(dev)
    g = engine.GraphedTrainStep(model, crit, x, t, pt, "seq")
    return model, g, (x, t, pt)


def timed(fn, n=60, warm=20):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(warm + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


arch = "multi" if w["space"] else None
m0, g0, d0 = make(B, 0)
one = timed(lambda i: g0(*d0, epoch=31, train_iter=i, arch_sample=arch))
print("one graph, B=%d: %.3f ms" % (B, one))
half = timed(lambda i: g0.graph.replay())
print("one graph, B=%d, replay only (no host plan): %.3f ms" % (B, half))
del g0, m0
m1, g1, d1 = make(B // 2, 1)
m2, g2, d2 = make(B // 2, 2)
h = timed(lambda i: g1.graph.replay())
print("one graph, B=%d, replay only: %.3f ms (x2 serial = %.3f)" % (B // 2, h, 2 * h))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(i):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        g1.graph.replay()
    with torch.cuda.stream(s2):
        g2.graph.replay()
    cur.wait_stream(s1)
    cur.wait_stream(s2)


two = timed(both)
print("two graphs, B=%d each, two streams: %.3f ms per %d images" % (B // 2, two, B))
