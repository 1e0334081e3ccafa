import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("vit-search_amd", "tests/golden", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, recipe, vitres
from vitres import engine
from vitres.losses import SoftTargetCrossEntropy
m = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=56, num_classes=10, network_def=recipe.MICRO_DEFS[0],
                        num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30).cuda()
m.set_compute_dtype(torch.float32 if sys.argv[1] == "f32" else torch.bfloat16)
m.train(); m.set_epoch(31)
x, t, pt, _ = (v.cuda() for v in recipe.inputs(7, 8, 56, 10, 1))
crit = SoftTargetCrossEntropy()
print("init", flush=True)
g = engine.GraphedTrainStep(m, crit, x, t, pt, "seq")
print("captured", flush=True)
for i in range(3):
    l = g(x, t, pt, epoch=31, train_iter=i, arch_sample="multi")
    print(l.item(), flush=True)
