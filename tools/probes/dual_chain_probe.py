"""How much of a step's time is latency that a second, independent chain of kernels could hide?
Two sr_tiny models with B = 64 each, forward + backward captured as one hipGraph each, replayed (a) one after the other on one
stream, (b) concurrently on two streams; against one model with B = 128.  (Measurement only: the two models do not share weights.)
usage: python tools/probes/dual_chain_probe.py [steps]"""
import os
import sys
import time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
import bench
from vitres import engine
from vitres.losses import SoftTargetCrossEntropy

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
crit = SoftTargetCrossEntropy()


def make(B, epa, seed):
    bench.WORKLOADS["sr_tiny_supernet"]["epa"] = epa
    model, _ = bench.build_model("sr_tiny_supernet", torch.bfloat16, dev)
    model.train()
    model.set_epoch(31)
    x, t, pt = bench.synthetic_batch(B, dev, seed)
    st = engine.GraphedTrainStep(model, crit, x, t, pt, "seq")
    return model, st, (x, t, pt)


def run(pairs, streams, n):
    for it in range(n):
        for (m, st, (x, t, pt)), s in zip(pairs, streams):
            with torch.cuda.stream(s):
                st(x, t, pt, epoch=31, train_iter=it, arch_sample="multi")


def timed(pairs, streams, label):
    run(pairs, streams, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(pairs, streams, steps)
    host = (time.perf_counter() - t0) * 1e3 / steps
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    print("%-44s %.3f ms per round (host done enqueueing after %.3f)" % (label, ms, host), flush=True)
    return ms


full = make(128, 64, 1)
s0 = torch.cuda.current_stream()
timed([full], [s0], "one chain, B = 128 (fwd + bwd, no optimizer)")
del full
torch.cuda.empty_cache()
a, b = make(64, 32, 2), make(64, 32, 3)
timed([a], [s0], "one chain, B = 64")
timed([a, b], [s0, s0], "two chains of B = 64, one stream")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
timed([a, b], [s1, s2], "two chains of B = 64, two streams")
