#!/usr/bin/env python
"""BASELINE config C5 on one GPU: forward-only scoring of evolutionary-search candidates on the RESIDENT sr_small supernet
(vitres.evo_eval: a candidate is a keep descriptor, no sub-network is built or copied).  Random candidates are drawn from the
sr_small space under the 2.9e9 MAC constraint of the reference's search script
(evolutionary_search/no_distill/small_flexible-conv-patch.sh:19), each scored on synthetic validation batches (val-bs 256).
Prints one JSON line: candidates/s and images/s.  dev / measurement tool; run on the GPU box."""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-search_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vitres  # noqa: E402
from vitres import evo_eval, supernet_config  # noqa: E402
from vitres.network_utils.compute_flop_mac import ComputationEstimator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--candidates", type=int, default=16)
ap.add_argument("--batches", type=int, default=8, help="validation batches of 256 images per candidate")
ap.add_argument("--space", default="sr_small")
ap.add_argument("--mac", type=float, default=2.9e9)
args = ap.parse_args()
dev = torch.device("cuda:0")
sp = getattr(supernet_config, args.space)
model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", num_classes=1000, network_def=sp.network_def,
                            num_channels_to_keep=sp.num_channels_to_keep, example_per_arch=64, num_warmup_epochs=30)
model = model.to(dev).set_compute_dtype(torch.bfloat16).eval()
est = ComputationEstimator(distill=False, input_resolution=224, patch_size=14)
rng = np.random.RandomState(0)
cands, macs = [], []
while len(cands) < args.candidates:
    c = evo_eval.random_candidate(sp.network_def, sp.num_channels_to_keep, rng)
    m = est(c)
    if 0.8 * args.mac <= m <= args.mac:
        cands.append(c)
        macs.append(m)
g = torch.Generator().manual_seed(0)
batches = [(torch.randn(256, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (256,), generator=g).to(dev))
           for _ in range(args.batches)]
evo_eval.score_population(model, cands[:2], batches[:2])                  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
scores = evo_eval.score_population(model, cands, batches)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
imgs = len(cands) * len(batches) * 256
print(json.dumps({"metric": "evo-eval candidates/s (C5, 1 GPU, resident sr_small supernet, bf16)", "candidates_per_s": round(len(cands) / dt, 3),
                  "images_per_s": round(imgs / dt, 1), "candidates": len(cands), "images_per_candidate": len(batches) * 256,
                  "mean_candidate_gmac": round(float(np.mean(macs)) / 1e9, 3),
                  "effective_tflops": round(2 * float(np.mean(macs)) * imgs / dt / 1e12, 1), "scores_head": scores[:3]}))
