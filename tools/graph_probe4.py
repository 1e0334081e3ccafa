"""Probe: forward of one B=128 supernet batch in one stream vs. its two 64-sample architecture groups in two streams (hipGraph replays)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch, bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
mA, _ = bench.build_model("sr_tiny_supernet", torch.bfloat16, dev)
mB, _ = bench.build_model("sr_tiny_supernet", torch.bfloat16, dev)
mC, _ = bench.build_model("sr_tiny_supernet", torch.bfloat16, dev)
for m in (mA, mB, mC):
    m.train(); m.set_epoch(31)
x, t, pt = bench.synthetic_batch(128, dev, 1)
xa, xb = x[:64].contiguous(), x[64:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
pA, pB, pC = mA.sample_plan(128), mB.sample_plan(64), mC.sample_plan(64)
mA._upload_plan(pA, dev); mB._upload_plan(pB, dev); mC._upload_plan(pC, dev)
torch.cuda.synchronize()

def one():
    with torch.no_grad():
        return mA(x, "seq", plan=pA)

def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            oa = mB(xa, "seq", plan=pB)
        with torch.cuda.stream(s2):
            ob = mC(xb, "seq", plan=pC)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return oa, ob

def graphed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g

def timeit(g, n=30):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

g1 = graphed(one); print("one stream  B=128: %.3f ms" % timeit(g1), flush=True)
g2 = graphed(two); print("two streams 2x64 : %.3f ms" % timeit(g2), flush=True)
