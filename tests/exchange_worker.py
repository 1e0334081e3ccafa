"""Worker of tests/test_gpu_dist.py::test_async_range_exchange_equals_blocking_exchange_*: one rank of a 2-rank job
(torch.distributed.run) on the GPU.  The same training step is replayed through engine.GraphedTrainStep three ways --

  one graph  + ONE blocking all-reduce of the flat gradient arena after it,
  three graphs + an ASYNCHRONOUS all-reduce of every finished arena range while the next graph runs (what bench.py --gpus N
  uses: the ordering of those all-reduces against the following backward graph is what differs between gloo's host threads
  and RCCL's stream),
  three graphs + the same exchange with bf16 on the wire

-- from the same weights, batch, masks and DropPath draws.  Rank 0 writes what the test asserts: the first two gradient arenas
are bit-identical (and identical on both ranks); the third is within bf16 rounding."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "vit-search_amd"), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
import recipe  # noqa: E402
import vitres  # noqa: E402
from vitres import engine  # noqa: E402
from vitres.losses import SoftTargetCrossEntropy  # noqa: E402


def main():
    backend = os.environ.get("VITRES_DIST_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", local if torch.cuda.device_count() > local else 0)     # gloo run: both ranks share GPU 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "device_count": torch.cuda.device_count()}
    grads = {}
    for mode in ("one_graph_blocking", "three_graphs_async", "three_graphs_async_bf16_wire"):
        torch.manual_seed(100)                                   # same initial weights in every mode (and on every rank)
        model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", img_size=recipe.MICRO_IMG,
                                    num_classes=recipe.MICRO_CLASSES, network_def=recipe.MICRO_DEFS[0], drop_path_rate=0.1,
                                    num_channels_to_keep=recipe.micro_keep_config(), example_per_arch=2, num_warmup_epochs=30)
        model = model.to(dev).set_compute_dtype(torch.bfloat16)
        model.train()
        model.set_epoch(31)
        model.drop_path_generator(seed=7)                        # (+ rank inside): same noise stream in every mode
        sync = engine.GradSync(model, wire_dtype=torch.bfloat16 if mode.endswith("bf16_wire") else torch.float32)
        sync.broadcast_parameters()
        x, t, pt, _ = (v.to(dev) for v in recipe.inputs(500 + rank, 8, recipe.MICRO_IMG, recipe.MICRO_CLASSES, 1))
        graphed = engine.GraphedTrainStep(model, SoftTargetCrossEntropy(), x, t, pt, "seq",
                                          split_for_sync=False if mode == "one_graph_blocking" else 3)
        assert len(graphed.more_graphs) == (0 if mode == "one_graph_blocking" else 2), len(graphed.more_graphs)
        model.drop_path_generator(seed=7)                        # (the constructor's warm-up steps drew from it)
        snaps = []
        for it in range(3):                                      # several replays: an exchange must not race the NEXT step either
            torch.manual_seed(900 + it + 17 * rank)              # different sub-networks per rank and step, same in every mode
            graphed.step_with_sync(sync, x, t, pt, average=True, epoch=31, train_iter=it, arch_sample=None)
            torch.cuda.synchronize()
            snaps.append(model._arena["gcur"].detach().clone())
        grads[mode] = snaps
        del graphed, model
    a, b, c = (grads[m] for m in ("one_graph_blocking", "three_graphs_async", "three_graphs_async_bf16_wire"))
    out["bit_identical"] = all(torch.equal(u, v) for u, v in zip(a, b))
    out["steps_differ"] = not torch.equal(a[0], a[1])
    out["bf16_wire_rel_l2"] = max(float((u - v).norm() / u.norm()) for u, v in zip(a, c))
    # the other rank holds the same averaged gradients
    mine = torch.stack([u.double().sum() for u in b] + [u.double().abs().sum() for u in b])
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    out["ranks_agree"] = all(torch.equal(both[0], o) for o in both[1:])
    out["finite"] = all(bool(torch.isfinite(u).all()) for u in a + b + c)
    if rank == 0:
        print("EXCHANGE " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
