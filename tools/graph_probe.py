import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch
from vitres import kernels as K

which = sys.argv[1]
dev = "cuda"
x = torch.randn(4, 17, 64, device=dev)
w = torch.ones(64, device=dev); b = torch.zeros(64, device=dev)
qkv32 = torch.randn(2, 257, 3 * 2 * 64, device=dev)
qkv16 = qkv32.bfloat16()
a = torch.randn(256, 64, device=dev).bfloat16(); bb = torch.randn(128, 64, device=dev).bfloat16()
out = torch.empty(256, 128, device=dev, dtype=torch.bfloat16)


def body():
    if which == "ln":
        return K.ln_fwd(x, w, b, None, 17, 1e-6, torch.float32)
    if which == "gemm":
        return K.gemm(a, bb, out, M=256, N=128, K=64, lda=64, ldb=64, ldc=128)
    if which == "attn32":
        return K.attn_fwd(qkv32, None, 2, 257, 2, 64, 0.125)
    if which == "attn16":
        return K.attn_fwd(qkv16, None, 2, 257, 2, 64, 0.125)
    if which == "zero":
        return out.zero_()
    if which == "softce":
        return K.softce(torch.randn(8, 10, device=dev), torch.rand(8, 10, device=dev), 0.125)


s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = body()
g.replay(); torch.cuda.synchronize()
print(which, "OK")
