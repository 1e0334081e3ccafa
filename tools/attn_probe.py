#!/usr/bin/env python
"""A few launches of the attention kernels at one shape, single stream: workload for rocprofv3 --pmc passes (dev tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import torch  # noqa: E402
from vitres import kernels as K  # noqa: E402

B, N, H, D = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 257, 4, 64)))
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, N, 3 * H * D, generator=g).cuda().to(torch.bfloat16)
d_o = torch.randn(B, N, H * D, generator=g).cuda().to(torch.bfloat16)
keep = torch.tensor([H * D] * (B // 2) + [(H - 1) * D] * (B - B // 2), dtype=torch.int32, device="cuda")
for _ in range(4):
    o, lse = K.attn_fwd(qkv, keep, B, N, H, D, D ** -0.5)
    K.attn_bwd(qkv, o, d_o, lse, keep, B, N, H, D, D ** -0.5)
torch.cuda.synchronize()
