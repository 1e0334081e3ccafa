"""Worker of test_gpu_model.test_opt_in_forms_at_model_level: one bf16 training step (forward + loss + backward) of the micro
supernet on fixed inputs with whatever VITRES_* knobs the parent put in the environment (they are read at import time, so every
setting needs its own process); saves keeps, logits, loss and every parameter gradient."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "vit-search_amd"), os.path.join(ROOT, "oracle"), os.path.join(HERE, "golden")):
    sys.path.insert(0, p)
import torch

import recipe
import vitres
import vitres_oracle as O


def main(out):
    nd = recipe.MICRO_DEFS[0]
    kw = dict(num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
    prod = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=recipe.MICRO_IMG,
                               num_classes=recipe.MICRO_CLASSES, network_def=nd, drop_path_rate=0.0, drop_block_rate=None, **kw)
    sd = recipe.fill_state_dict([(k, tuple(v.shape)) for k, v in prod.state_dict().items()], 100)
    prod.load_state_dict(sd)
    prod = prod.to("cuda").set_compute_dtype(torch.bfloat16)
    x, t, pt, _ = recipe.inputs(7, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1)
    prod.train()
    prod.set_epoch(31)
    torch.manual_seed(0)
    cls, pat = prod(x.cuda(), patch_output_type="seq")
    loss = O.soft_target_ce(cls, t.cuda()) + O.soft_target_ce(pat, pt.cuda())
    loss.backward()
    torch.cuda.synchronize()
    torch.save({"keeps": [k.cpu() if torch.is_tensor(k) else k for k in prod.last_keeps], "cls": cls.detach().float().cpu(),
                "pat": pat.detach().float().cpu(), "loss": float(loss),
                "grads": {n: p.grad.detach().float().cpu() for n, p in prod.named_parameters()}}, out)


if __name__ == "__main__":
    main(sys.argv[1])
