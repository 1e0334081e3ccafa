"""Run-to-run determinism of the training forward / backward at full widths: the same plan twice, outputs compared bit by bit.
usage: python tools/probes/determinism_probe.py [sr_tiny|sr_small] [B] [epa]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vit-search_amd"))
import torch
import recipe
import vitres
from vitres import supernet_config

space = sys.argv[1] if len(sys.argv) > 1 else "sr_small"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
epa = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nd = {"sr_tiny": recipe.SR_TINY_DEF, "sr_small": recipe.SR_SMALL_DEF}[space]
prod = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=224, num_classes=1000, network_def=nd,
                           drop_path_rate=float(os.environ.get("DP", "0.4")), num_channels_to_keep=getattr(supernet_config, space).num_channels_to_keep,
                           example_per_arch=epa, num_warmup_epochs=30)
sd = recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in prod.state_dict().items()], 4343)
prod.load_state_dict(sd)
prod = prod.to("cuda").set_compute_dtype(torch.bfloat16)
prod.train()
prod.set_epoch(31)
prod.load_state_dict(sd)
g = torch.Generator().manual_seed(78)
x = torch.randn(B, 3, 224, 224, generator=g).cuda()
t = torch.nn.functional.one_hot(torch.randint(0, 1000, (B,), generator=g), 1000).float().cuda()
pt = t[:, None, :].repeat(1, 16, 1).contiguous()
for seed in range(3):
    outs = []
    for rep in range(3):
        torch.manual_seed(800 + seed)
        prod.drop_path_generator(seed=seed)
        prod.zero_grad(set_to_none=True)
        plan = prod.sample_plan(B)
        cls, pat = prod(x, plan=plan) if os.environ.get("FWD_ONLY") else (None, None)
        if cls is None:
            loss = prod.loss_and_grad(x, t, pt, "seq", plan=plan)
            torch.cuda.synchronize()
            gr = {n: p.grad.detach().float().cpu().clone() for n, p in prod.named_parameters()}
            outs.append((float(loss), gr))
        else:
            torch.cuda.synchronize()
            outs.append((cls.detach().float().cpu(), pat.detach().float().cpu()))
    if os.environ.get("FWD_ONLY"):
        print("seed", seed, "cls equal", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[0][0], outs[2][0]),
              "max diff", float((outs[0][0] - outs[1][0]).abs().max()), "pat equal", torch.equal(outs[0][1], outs[1][1]))
    else:
        worst = sorted(((float((outs[0][1][n] - outs[1][1][n]).abs().max() / max(float(outs[0][1][n].abs().max()), 1e-12)), n)
                        for n in outs[0][1]), reverse=True)[:6]
        print("seed", seed, "loss", outs[0][0], outs[1][0], outs[2][0], "worst grads", [(("%.1e" % v), n) for v, n in worst])
