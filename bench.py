#!/usr/bin/env python
"""Headline benchmark: images/sec of the ViT-ResNAS-Tiny supernet training step (BASELINE.json `metric`),
one process per GPU, gradients all-reduced over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One step = forward + loss + backward + gradient exchange + AdamW on one synthetic batch (B=128/GPU, resident in
HBM before the timed region).  Prints ONE JSON line (rank 0) with the driver's contract fields plus
`roofline` (dominant kernel, measured with HIP events around its launches) and `cpu_baseline`
(the CPU oracle timed on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vit-search_amd"),):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[2] / `metric`: sr_tiny supernet (supernet_config/sr_tiny.py), multi-arch sampling
    "sr_tiny_supernet": dict(space="sr_tiny", batch=128, epa=64, drop_path=0.2),
    # configs[2'] the README's ViT-ResNAS-Tiny recipe (conv patch embedding)
    "sr_tiny_mh_supernet": dict(space="sr_tiny_mh", batch=128, epa=64, drop_path=0.2),
    # configs[3]
    "sr_small_supernet": dict(space="sr_small", batch=64, epa=32, drop_path=0.3),
    # configs[1]: ViT-Res-Tiny reference net
    "ref_tiny": dict(space=None, batch=128, epa=None, drop_path=0.2),
    # configs[4] (C5): forward-only scoring of evolutionary-search candidates on the resident sr_small supernet, val-bs 256
    "evo_eval_sr_small": dict(space="sr_small", batch=256, epa=64, drop_path=0.0, evo=True),
}
REF_TINY_DEF = ((4, 192),) + ((1, (192, 3, 64), (192, 768), 1),) * 4 + ((3, 192, 384),) + \
    ((1, (384, 6, 64), (384, 1536), 1),) * 4 + ((3, 384, 768),) + \
    ((1, (768, 12, 64), (768, 3072), 1),) * 4 + ((2, 768, 1000),)
MFMA_PEAK = {"bf16": 2500.0, "f32": 157.3}      # dense TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = "r06"                            # profiles/<round>_traffic_<workload>.json, <round>_graph_kernels_<workload>.json


def load_traffic(workload, batch, dtype):
    """PMC traffic of this workload (tools/traffic_run.sh -> profiles/r06_traffic_<workload>.json), used only when the file was
    measured on the same workload, per-GPU batch and dtype as this run (its "key"); otherwise `traffic` stays null."""
    name = "%s_traffic_%s.json" % (PROFILE_ROUND, workload)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
    try:
        tj = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if tj.get("key") != {"workload": workload, "batch": int(batch), "dtype": dtype}:
        return None, None
    return tj, name


def build_model(name, dtype, device):
    import vitres
    from vitres import supernet_config
    w = WORKLOADS[name]
    if w["space"]:
        sp = getattr(supernet_config, w["space"])
        model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", num_classes=1000,
                                    network_def=sp.network_def, drop_path_rate=w["drop_path"],
                                    num_channels_to_keep=sp.num_channels_to_keep, example_per_arch=w["epa"],
                                    num_warmup_epochs=30, single_arch=False)
        nd = sp.network_def
    else:
        model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output", num_classes=1000,
                                    network_def=REF_TINY_DEF, drop_path_rate=w["drop_path"])
        nd = REF_TINY_DEF
    model = model.to(device).set_compute_dtype(dtype)
    return model, nd


def synthetic_batch(batch, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, 3, 224, 224, generator=g)
    t = torch.softmax(torch.randn(batch, 1000, generator=g), -1)
    pt = torch.softmax(torch.randn(batch, 16, 1000, generator=g), -1)
    return x.to(device), t.to(device), pt.to(device)


def cpu_baseline(name, seconds_budget=25.0):
    """The CPU oracle (oracle/vitres_oracle.py, pinned to the reference by golden vectors) on the host cores:
    full training step (fwd + bwd + AdamW), same network, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vitres_oracle as O
    from vitres import supernet_config
    w = WORKLOADS[name]
    cores = min(os.cpu_count() or 1, 32)       # more threads than this only adds fork/join overhead at batch 16
    torch.set_num_threads(cores)
    B = 16                                     # CPU baseline at B = 16 (train)
    if w["space"]:
        sp = getattr(supernet_config, w["space"])
        m = O.OracleViTSR(sp.network_def, num_classes=1000, drop_path_rate=w["drop_path"], supernet=True,
                          num_channels_to_keep=sp.num_channels_to_keep, example_per_arch=B // 2,
                          num_warmup_epochs=30, patch_output=True)
        m.set_epoch(31)
    else:
        m = O.OracleViTSR(REF_TINY_DEF, num_classes=1000, drop_path_rate=w["drop_path"], patch_output=True)
    opt = torch.optim.AdamW(O.param_groups_weight_decay(m, 0.05), lr=1e-4)
    x, t, pt = synthetic_batch(B, "cpu", 0)
    n, t0 = 0, time.time()
    while True:                                   # no separate warm-up: bounded to ~seconds_budget of CPU work
        O.train_step(m, opt, x, t, pt, 31, n + 1, arch_sample=("multi" if w["space"] else None))
        n += 1
        dt = time.time() - t0
        if dt + dt / n > seconds_budget or n >= 6:
            break
    return {"value": round(B * n / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d train steps (fwd+bwd+AdamW) of the %s CPU oracle, batch %d, fp32, torch %d threads"
                      % (n, name, B, cores)}


def cpu_baseline_evo(cands, seconds_budget=20.0):
    """CPU baseline of C5: the CPU oracle scores candidates the reference's way (evo_search.py:253-285: build the prefix-sliced
    sub-network of a candidate, eval forward) -- batch 32, fp32, on this box's host cores; bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vitres_oracle as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    B = 32
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    n, t0 = 0, time.time()
    while True:
        m = O.OracleViTSR(cands[n % len(cands)], num_classes=1000, patch_output=True).eval()      # (construction is part of the
        with torch.no_grad():                                                                   # reference's per-candidate cost)
            m(x)
        n += 1
        dt = time.time() - t0
        if dt + dt / n > seconds_budget or n >= 8:
            break
    return {"value": round(B * n / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d candidates x one eval forward of batch %d through the CPU oracle (sub-network built per candidate), fp32, "
                      "torch %d threads" % (n, B, cores)}


def write_launch_table(path, profile, descs, steps):
    rows = {}
    for (kind, fl, dense, by, e0, e1), desc in zip(profile, descs):
        r = rows.setdefault(desc, [0, 0.0, 0.0, 0.0, 0.0])
        r[0] += 1; r[1] += e0.elapsed_time(e1) * 1e-3; r[2] += fl; r[3] += dense; r[4] += by
    with open(path, "w") as f:
        f.write("%-86s %5s %8s %8s %8s %8s %9s\n" % ("launch", "n", "us", "TF kept", "TF dense", "GB/s", "us/step"))
        for desc, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            f.write("%-86s %5d %8.1f %8.1f %8.1f %8.1f %9.1f\n" % (desc[:86], r[0], r[1] / r[0] * 1e6, r[2] / r[1] / 1e12,
                                                                 r[3] / r[1] / 1e12, r[4] / r[1] / 1e9, r[1] / steps * 1e6))


def run_evo_eval(args, rank, world, device):
    """C5: 512 candidates drawn by the restated gen_random_network_def under the 2.9e9-MAC constraint of
    evolutionary_search/no_distill/small_flexible-conv-patch.sh:19, dealt over the ranks (candidate-sharded), each
    scored forward-only on synthetic validation batches of 256 images against the RESIDENT supernet (a candidate is a keep
    descriptor: vitres.evo_eval).  One step = one candidate on one batch of 256 images; value = images/s over all ranks."""
    import numpy as np
    from vitres import evo_eval, kernels as K, supernet_config
    from vitres.network_utils.compute_flop_mac import ComputationEstimator
    from vitres.search_utils import gen_utils
    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    sp = getattr(supernet_config, w["space"])
    torch.manual_seed(0)
    import vitres
    model = vitres.create_model("flexible_vit_sr_patch14_224_patch_output_supernet", num_classes=1000, network_def=sp.network_def,
                                num_channels_to_keep=sp.num_channels_to_keep, example_per_arch=w["epa"], num_warmup_epochs=30)
    model = model.to(device).set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32).eval()
    est = ComputationEstimator(distill=False, input_resolution=224, patch_size=14)
    np.random.seed(0)
    n_cand = 512
    cands = [gen_utils.gen_random_network_def(sp.network_def, sp.num_channels_to_keep, 2.9e9, est) for _ in range(n_cand)]
    macs = [est(c) for c in cands]
    mine = list(range(rank, n_cand, world))
    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    x = torch.randn(B, 3, 224, 224, generator=g).to(device)
    y = torch.randint(0, 1000, (B,), generator=g).to(device)
    plans = {}

    def step(i):
        ci = mine[i % len(mine)]
        if ci not in plans:
            plans[ci] = evo_eval.plan_for_subnet(model, cands[ci], B)
        with torch.no_grad():
            out = model(x, plan=plans[ci])
        out = out[0] if isinstance(out, tuple) else out
        return (out.argmax(dim=1) == y).sum()
    step_probe = [] if os.environ.get("VITRES_DBG_STEPS") else None   # dev aid: GPU time of every step from the first warm-up step on
    if step_probe is not None:
        _step = step

        def step(i):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = _step(i)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            step_probe.append((e0, e1))
            return r
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hits = [step(args.warmup + i) for i in range(args.steps)]
    host_enqueue = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    assert all(0 <= int(h) <= B for h in hits)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    used = [mine[(args.warmup + i) % len(mine)] for i in range(args.steps)]
    roof = None
    if args.profile_steps > 0:
        K.PROFILE = []
        K.PROFILE_ATTN = []
        K.PROFILE_DESC = [] if args.launch_table else None
        for i in range(args.profile_steps):
            torch.cuda.synchronize()
            torch.cuda._sleep(int(0.04 * 2.0e9))         # (see the training workloads: the step is queued before the GPU starts it)
            step(args.warmup + i)
        torch.cuda.synchronize()
        if args.launch_table:
            write_launch_table(args.launch_table, K.PROFILE, K.PROFILE_DESC, args.profile_steps)
            K.PROFILE_DESC = None
        sec = fl = by = dense = 0.0
        n = 0
        for kind, f_, d_, b_, e0, e1 in K.PROFILE:
            if kind[0] == "bf16" and not kind[1] and not kind[2]:
                sec += e0.elapsed_time(e1) * 1e-3
                fl, dense, by, n = fl + f_, dense + d_, by + b_, n + 1
        K.PROFILE = None
        if n:
            gbps, ach = by / sec / 1e9, fl / sec / 1e12
            traffic, traffic_note = None, None
            tj, tj_name = load_traffic(args.workload, B, args.dtype)
            if tj is not None:
                for fam, tv in tj["kernels"].items():
                    if fam.startswith("vr_gemm_nt::nt_kernel"):
                        traffic = round(tv["read_bytes_per_launch"] + tv["write_bytes_per_launch"], 1)
                        traffic_note = "HBM-side bytes per launch (read + write) from profiles/%s: %s" % (tj_name, tj["source"])
            roof = {"bound": "hbm", "kernel": "vr_gemm_nt::nt_kernel (forward)", "achieved": round(gbps, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(gbps / 8000.0, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "traffic_over_algorithmic": (round(traffic / (by / n), 3) if traffic else None),
                    "launches_per_step": n // args.profile_steps, "avg_launch_us": round(sec / n * 1e6, 2),
                    "flops_per_launch": fl / n, "algorithmic_bytes_per_launch": by / n,
                    "mfma_check": {"achieved_TFLOPs_kept": round(ach, 2), "peak_TFLOPs": MFMA_PEAK["bf16"],
                                   "frac_kept": round(ach / MFMA_PEAK["bf16"], 4)},
                    "note": "kept FLOPs / kept-width algorithmic bytes of the candidates' sub-networks; HIP events around every "
                            "vr_gemm launch of %d extra steps" % args.profile_steps}
    img_s = B * world * args.steps / elapsed
    mean_mac = float(np.mean([macs[c] for c in used]))
    print(json.dumps({
        "metric": "images/sec/node evo-search candidate scoring (BASELINE configs[4]: forward-only sub-network evaluation on the resident sr_small supernet)",
        "value": round(img_s, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": args.workload, "val_batch": B, "candidates": n_cand, "candidates_per_rank": len(mine),
                   "mac_constraint": 2.9e9, "mean_candidate_gmac": round(mean_mac / 1e9, 3), "parallelism": "candidates dealt over %d rank(s)" % world,
                   "candidates_per_sec_at_25000_images": round(img_s / 25000.0, 3), "host_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
                   "effective_tflops": round(2 * mean_mac * img_s / 1e12, 1)},
        "roofline": roof,
        "cpu_baseline": (None if (args.no_cpu_baseline or world > 1) else cpu_baseline_evo(cands[:8]))}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps (the driver's protocol: --steps 20 --warmup 5; steady state = --steps 100 --warmup 30, ~0.5 %% lower "
                         "since the graphed step bounds its run-ahead)")
    ap.add_argument("--workload", default="sr_tiny_supernet", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-table", default="", help="dev aid: write a per-shape table of the profiled GEMM launches to this file")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--optimizer", default="flat", choices=["flat", "torch"], help="flat: vitres.optim.FlatAdamW (one "
                    "fused HIP pass over the arena); torch: torch.optim.AdamW(fused=True)")
    ap.add_argument("--split-sync", type=int, default=-1, help="backward captured as this many hipGraphs (cut in front of the "
                    "spatial reductions), each followed by the all-reduce of the gradient range it completed, overlapped with the "
                    "next one; 0/1: one graph, one all-reduce; -1: 3 when --gpus > 1 (only ~24 MB of the 279 MB are exchanged after the "
                    "backward has finished, against ~116 MB with 2 graphs; +0.2 ms of graph boundaries measured on one GPU; not yet timed "
                    "over RCCL -- no multi-GPU box was available)")
    ap.add_argument("--wire", default="f32", choices=["f32", "bf16"], help="dtype of the gradients on the wire (N > 1): bf16 halves "
                    "the bytes of the exchange (engine.GradSync(wire_dtype=torch.bfloat16)); the default is the reference's fp32")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel from Python instead of replaying a hipGraph")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (the form the driver uses at N = 1): become the launcher -- one rank per GPU through
        # torch.distributed.run, rendezvous on 127.0.0.1; rank 0 of the children prints the JSON line
        import socket
        import subprocess
        if torch.cuda.device_count() < args.gpus and os.environ.get("VITRES_DIST_BACKEND", "nccl") == "nccl":
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # RCCL's ring kernels take one workgroup (one CU) per channel and run beside the backward, whose side stream already fills
        # the CUs the main chain leaves idle: cap them at 8 channels (8 of 256 CUs; a ring over 7 xGMI links needs no more to
        # saturate them at 92 - 228 MB per range) unless the caller chose otherwise.  DESIGN.md section 6.
        env.setdefault("NCCL_MIN_NCHANNELS", "4")
        env.setdefault("NCCL_MAX_NCHANNELS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # VITRES_DIST_BACKEND=gloo + ranks sharing one GPU: functional check of the N > 1 path on a 1-GPU box (not a measurement)
    backend = os.environ.get("VITRES_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)

    if WORKLOADS[args.workload].get("evo"):
        return run_evo_eval(args, rank, world, device)
    from vitres import engine, kernels as K
    from vitres.losses import SoftTargetCrossEntropy
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    torch.manual_seed(0 + rank)                                    # reference: seed + rank (main.py:261-267)
    model, nd = build_model(args.workload, dtype, device)
    sync = engine.GradSync(model, wire_dtype=torch.bfloat16 if args.wire == "bf16" else torch.float32)
    x, t, pt = synthetic_batch(B, device, 1000 + rank)
    model.train()
    if w["space"]:
        model.set_epoch(31)                                        # past warm-up: every width choice active
    model._ensure_arena(device)
    sync.broadcast_parameters()
    lr = 5e-4 * B * world / 512.0
    if args.optimizer == "flat":
        # AdamW on the flat arena in one HIP pass, fused with the bf16 weight shadow and the 1/world gradient averaging
        from vitres.optim import FlatAdamW
        opt = FlatAdamW(model, engine.param_groups_weight_decay(model, 0.05), lr=lr)
        opt.grad_scale = 1.0 / world
        if dtype == torch.bfloat16:
            opt.own_shadow()
    else:
        opt = torch.optim.AdamW(engine.param_groups_weight_decay(model, 0.05), lr=lr, fused=True)
    average = args.optimizer != "flat"
    crit = SoftTargetCrossEntropy()
    arch = "multi" if w["space"] else None

    def eager_step(i, exchange=True):
        return engine.train_step(model, crit, opt, x, t, pt, "seq", epoch=31, train_iter=i, arch_sample=arch,
                                 grad_sync=sync if exchange else None, average_grads=average)

    graphed = None
    split = (3 if world > 1 else 0) if args.split_sync < 0 else (args.split_sync if args.split_sync >= 2 else 0)
    if not args.no_graph:
        # N > 1: the backward is captured as two graphs so that the all-reduce of the last stage's gradients (the tail of
        # the flat arena, most of the parameters) runs on RCCL's stream while the rest of the backward is still computing
        # one rank, FlatAdamW: the update is captured into the graph as well, the arena's tail (last stage + heads: most parameters)
        # updated on the weight gradients' stream beside the rest of the backward by a capped launch (engine.GraphedTrainStep,
        # VITRES_OPT_OVERLAP; round 4: 7.47 -> 7.37 ms).  VITRES_OPT_IN_GRAPH=0: replay, then optimizer.step().  With more than one
        # rank the exchange sits between the backward and the update: step_with_sync + optimizer.step()
        opt_in_graph = world == 1 and args.optimizer == "flat" and os.environ.get("VITRES_OPT_IN_GRAPH", "1") != "0"
        graphed = engine.GraphedTrainStep(model, crit, x, t, pt, "seq", split_for_sync=split,
                                          optimizer=opt if opt_in_graph else None)

    def step(i):
        # the reference draws SwitchTokenMix's two sample permutations from the CPU generator every iteration
        # (token_mixup.py:104-106,126-127) BEFORE the forward saves / restores its state (engine.py:119-165): without them every
        # step of a synthetic-data run would re-draw the same architectures
        torch.randperm(B // 2)
        torch.randperm(B - B // 2)
        if graphed is None:
            return eager_step(i)
        if graphed.optimizer is not None:
            if graphed.defer is None:
                opt.prepare_step()                               # this step's lr / bias corrections -> device; the graph does the rest
            return graphed(x, t, pt, epoch=31, train_iter=i, arch_sample=arch)
        # fwd + loss + bwd (hipGraph replay) + gradient exchange (averaged), then the optimizer
        loss = graphed.step_with_sync(sync, x, t, pt, average=average, epoch=31, train_iter=i, arch_sample=arch)
        opt.step()
        return loss

    if graphed is not None and world > 1:
        graphed.exposed = []                                     # (event after the last backward graph, event after the exchange)

    step_probe = [] if os.environ.get("VITRES_DBG_STEPS") else None   # dev aid: GPU time of every step from the first warm-up step on
    if step_probe is not None:
        _step = step

        def step(i):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = _step(i)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            step_probe.append((e0, e1))
            return r
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    if graphed is not None and world > 1:
        graphed.exposed = []
    gap_probe = [] if os.environ.get("VITRES_DBG_GAP") else None   # dev aid: GPU time inside a step / between two steps
    for i in range(args.steps):
        if gap_probe is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        losses.append(step(args.warmup + i))
        if gap_probe is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            gap_probe.append((e0, e1))
    host_enqueue = time.perf_counter() - t0              # host time to issue the K steps (the GPU may still be running)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    lossv = torch.stack(losses).tolist()
    assert all(v == v and abs(v) != float("inf") for v in lossv), "non-finite loss"
    if step_probe:
        print("step probe (ms, from the first warm-up step): " + " ".join("%.2f" % a.elapsed_time(b) for a, b in step_probe),
              file=sys.stderr)
    if graphed is not None and getattr(graphed, "_gap_probe", None):
        print("gap probe 2 (GPU ms: plan copy | gather + target copies | graph replay | between calls): %s" % graphed.gap_report(),
              file=sys.stderr)
    if gap_probe:
        inside = sum(a.elapsed_time(b) for a, b in gap_probe) / len(gap_probe)
        between = sum(gap_probe[i][1].elapsed_time(gap_probe[i + 1][0]) for i in range(len(gap_probe) - 1)) / max(len(gap_probe) - 1, 1)
        print("gap probe: %.3f ms inside a step, %.3f ms between the end of one step and the start of the next" % (inside, between),
              file=sys.stderr)
    exchange = None
    if world > 1:
        n_arena = model._arena["flat"].numel()
        exposed = None
        if graphed is not None and getattr(graphed, "exposed", None):
            exposed = round(sum(a.elapsed_time(b) for a, b in graphed.exposed) / len(graphed.exposed), 3)
        exchange = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "allreduce_bytes_per_step": (2 if args.wire == "bf16" else 4) * n_arena,
                    "dtype": args.wire, "ranges": (len(graphed.ranges) if graphed is not None and graphed.ranges else 1),
                    "exposed_ms_per_step": exposed,
                    "nccl_channels": {k: os.environ.get(k) for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS")},
                    "note": "exposed = GPU time between the end of the last backward graph and the end of the last all-reduce "
                            "(rank 0, HIP events on the compute stream); compare ms_per_step with the N = 1 line of the same box: "
                            "ms_per_step(N) - ms_per_step(1) = exposed exchange + what RCCL's kernels cost the backward they run beside"}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- dominant-kernel roofline: HIP events around every vr_gemm launch (same stream), extra steps ------
    roof = None
    if args.profile_steps > 0:
        K.PROFILE = []
        K.PROFILE_ATTN = []
        K.PROFILE_DESC = [] if args.launch_table else None
        for i in range(args.profile_steps):
            # eager launches are host-bound (~10 us of Python per kernel): with an idle GPU every event pair would also time the
            # host's gap between recording the event and launching the kernel.  The GPU is parked behind a 40 ms spin first, so that
            # the whole step is queued when it starts and the pairs bracket GPU time only (kernel + its launch boundary).
            torch.cuda.synchronize()
            torch.cuda._sleep(int(0.04 * 2.0e9))
            eager_step(10_000 + i, exchange=False)      # rank 0 only, after the timed region: no collectives here
        torch.cuda.synchronize()
        if args.launch_table:                            # dev aid: per-shape table of the GEMM launches (time, TF/s, GB/s)
            write_launch_table(args.launch_table, K.PROFILE, K.PROFILE_DESC, args.profile_steps)
            K.PROFILE_DESC = None
        agg = {}
        for kind, fl, dense, by, e0, e1 in K.PROFILE:
            a = agg.setdefault(kind, [0.0, 0.0, 0.0, 0, 0.0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += fl
            a[2] += by
            a[3] += 1
            a[4] += dense
        K.PROFILE = None
        # SURVEY 8d "MFMA utilisation on the transformer blocks": kept FLOPs of every block Linear (forward, data gradient, weight
        # gradient, LayerNorm-fused forms: the launches without row maps) and of the attention kernels over the time of exactly those
        # launches, against the dense bf16 MFMA peak
        blk_fl = sum(v[1] for k, v in agg.items() if k[3] != 1) + sum(f for f, _, _ in K.PROFILE_ATTN)
        blk_s = sum(v[0] for k, v in agg.items() if k[3] != 1) + sum(a.elapsed_time(b) * 1e-3 for _, a, b in K.PROFILE_ATTN)
        blocks_util = {"value": round(blk_fl / blk_s / (MFMA_PEAK[args.dtype] * 1e12), 4) if blk_s > 0 else None,
                       "kept_gflop_per_step": round(blk_fl / args.profile_steps / 1e9, 1),
                       "kernel_ms_per_step": round(blk_s / args.profile_steps * 1e3, 3), "target": 0.40,
                       "note": "kept FLOPs of the transformer blocks' GEMMs (forward, data and weight gradients) and attention kernels / "
                               "(sum of their eager launch times x dense MFMA peak); launches on two streams overlap in the step, so the "
                               "step-level figure is higher: kept_gflop_per_step / ms_per_step"}
        K.PROFILE_ATTN = None
        def kname(k):
            dt_, at, bt, mapped = k
            if mapped == 2:
                return "vr_gemm_ntln::ntln_kernel (Linear + LayerNorm epilogue)"
            if dt_ == "bf16" and not at:         # (b_trans data gradients run on the same kernel since round 2: BKM)
                return "vr_gemm_nt::nt_kernel (forward+dgrad: gemm_ntk.hip lean-loop kernels + gemm_nt.hip forms)"
            if dt_ == "bf16" and at and bt:
                return "vr_gemm_tn::tn_group_kernel / tn_kernel (wgrad)"
            if at:
                return "gemm_kernel<%s,true,true,float,EPI_ATOMIC> (row-mapped wgrad)" % dt_
            return "gemm_kernel<%s,false,%s> (%s)" % (dt_, "true" if bt else "false", "dgrad, contraction-major W" if bt
                                                      else "forward")
        byname = {}
        for k, v in agg.items():
            a = byname.setdefault(kname(k), [0.0, 0.0, 0.0, 0, 0.0, k[0]])
            for i in range(5):
                a[i] += v[i]
        HBM_PEAK = 8000.0                                                   # GB/s, /opt/skills/guides/MI355X_MICROARCH.md
        tj, tj_name = load_traffic(args.workload, B, args.dtype)

        def family_roof(name, sec, fl, by, n, dense, dt_):
            """Which roof binds a kernel family: its arithmetic intensity -- KEPT FLOPs per KEPT algorithmic byte of a launch --
            against the machine balance peak_flops / peak_bandwidth; the other roof is reported beside it (mfma_check)."""
            peak = MFMA_PEAK[dt_]
            ach, gbps = fl / sec / 1e12, by / sec / 1e9
            intensity, ridge = fl / by, peak * 1e12 / (HBM_PEAK * 1e9)
            if intensity < ridge:
                r = {"bound": "hbm", "kernel": name, "achieved": round(gbps, 1), "peak": HBM_PEAK, "unit": "GB/s",
                     "frac": round(gbps / HBM_PEAK, 4)}
            else:
                r = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(ach / peak, 4)}
            traffic, traffic_note = None, None
            if tj is not None:
                for fam, tv in tj["kernels"].items():
                    if name.startswith(fam):
                        traffic = round(tv["read_bytes_per_launch"] + tv["write_bytes_per_launch"], 1)
                        traffic_note = "HBM-side bytes per launch (read + write) of %s from profiles/%s: %s" % (fam, tj_name, tj["source"])
            r.update({"traffic": traffic, "traffic_note": traffic_note,
                      "traffic_over_algorithmic": (round(traffic / (by / n), 3) if traffic else None),
                      "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                      "launches_per_step": n // args.profile_steps, "avg_launch_us": round(sec / n * 1e6, 2),
                      "flops_per_launch": fl / n, "dense_flops_per_launch": dense / n, "algorithmic_bytes_per_launch": by / n,
                      "mfma_check": {"achieved_TFLOPs_kept": round(ach, 2), "achieved_TFLOPs_dense_equiv": round(dense / sec / 1e12, 2),
                                     "peak_TFLOPs": peak, "frac_kept": round(ach / peak, 4)},
                      "hbm_bound_check": {"algorithmic_GBps": round(gbps, 1), "peak_GBps": HBM_PEAK}})
            return r

        ranked = sorted(byname.items(), key=lambda kv: -kv[1][0])
        dom, (sec, fl, by, n, dense, dom_dt) = ranked[0]
        gemm_sec = sum(v[0] for v in agg.values()) / args.profile_steps
        roof = family_roof(dom, sec, fl, by, n, dense, dom_dt)
        # the weight-gradient family is the largest single rocprof row of the step: reported beside the dominant family
        wg = [kv for kv in ranked[1:] if "wgrad" in kv[0]]
        if wg:
            roof["wgrad"] = family_roof(wg[0][0], *wg[0][1])
        peak, ach, gbps = MFMA_PEAK[dom_dt], fl / sec / 1e12, by / sec / 1e9
        # the same family inside the replayed graph (rocprofv3 trace of this workload, tools/prof_step.sh): eager launches run ~10 %
        # slower than the graph's, so `frac` above is the pessimistic figure
        try:
            gj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "%s_graph_kernels_%s.json" % (PROFILE_ROUND, args.workload))))
        except (OSError, ValueError):
            gj = None
        if gj and gj.get("key") == {"workload": args.workload, "batch": int(B), "dtype": args.dtype}:
            fam_ = [k for k in gj["kernels"] if roof["kernel"].startswith(k)]
            if fam_:
                us = gj["kernels"][fam_[0]]["avg_us"]
                gb_ = roof["algorithmic_bytes_per_launch"] / us / 1e3
                roof["graph"] = {"avg_launch_us": round(us, 2), "achieved_GBps": round(gb_, 1), "frac_hbm": round(gb_ / HBM_PEAK, 4),
                                 "achieved_TFLOPs_kept": round(roof["flops_per_launch"] / us / 1e6, 2),
                                 "source": "profiles/%s_graph_kernels_%s.json: %s" % (PROFILE_ROUND, args.workload, gj["source"])}
        roof.update({
                "note": "FLOPs and bytes = kept (un-masked) sub-problems only; HIP events (recorded on the stream each kernel is launched "
                        "on) around every vr_gemm launch of %d extra eager steps after the timed region, each queued behind a 40 ms GPU spin so that "
                        "the pairs time the GPU, not the host's launch gaps (roofline.graph has the replayed graph's "
                        "averages)" % args.profile_steps,
                "blocks_mfma_util": blocks_util,
                "all_gemm_ms_per_step": round(gemm_sec * 1e3, 3),
                "all_gemm_kinds": {k: {
                    "tflops_kept": round(v[1] / v[0] / 1e12, 2), "tflops_dense_equiv": round(v[4] / v[0] / 1e12, 2),
                    "algorithmic_GBps": round(v[2] / v[0] / 1e9, 1), "ms_per_step": round(v[0] / args.profile_steps * 1e3, 3)}
                    for k, v in byname.items()}})
    here = os.path.dirname(os.path.abspath(__file__))
    n1_file = os.path.join(here, "gpurun_out", ".bench_n1_%s.json" % args.workload)
    import platform
    import subprocess
    try:
        commit = subprocess.run(["git", "-C", here, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        commit = ""
    if not commit:                                                 # (the GPU box gets a snapshot without .git: the library identifies the tree)
        import hashlib
        from vitres import _lib as _L
        commit = "lib:" + hashlib.sha256(open(_L.LIB_PATH, "rb").read()).hexdigest()[:16]
    run_key = {"host": platform.node(), "commit": commit, "workload": args.workload, "batch": B, "dtype": args.dtype}
    if world > 1:                                                   # timed on rank 0 at N = 1 only
        cpu = {"value": None, "unit": "images/sec", "cores": None, "kind": "port",
               "sample": "not timed at N > 1: the CPU oracle runs beside the N = 1 line only (same workload, see that line)"}
        # ... but the N = 1 line this box wrote for the same workload, batch, dtype and commit carries it over (anything else --
        # another box, another tree, a tracked line of an earlier round -- would misstate the speed-up: left null)
        try:
            sib = json.load(open(n1_file))
        except (OSError, ValueError):
            sib = None
        if sib and sib.get("n_gpus") == 1 and sib.get("run_key") == run_key and (sib.get("cpu_baseline") or {}).get("value"):
            cpu = dict(sib["cpu_baseline"], copied_from="the N = 1 line of this box, commit and configuration (%s); not re-timed at "
                                                        "N = %d" % (os.path.relpath(n1_file, here), world))
    else:
        cpu = None if args.no_cpu_baseline else cpu_baseline(args.workload)
    from vitres.network_utils.compute_flop_mac import train_flops_per_image
    dense_flops = train_flops_per_image(nd)
    img_s = B * world * args.steps / elapsed
    out = {
        "metric": "images/sec/node ViT-ResNAS-Tiny supernet train, bs128/GPU, 1/2/4/8 MI355X",
        "value": round(B * world * args.steps / elapsed, 2), "unit": "images/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": args.workload, "global_batch": B * world, "per_gpu_batch": B,
                   "example_per_arch": w["epa"], "epoch": 31, "drop_path": w["drop_path"], "parallelism": "dp%d" % world,
                   "optimizer": ("AdamW (vitres.optim.FlatAdamW: vr_adamw_flat%s)" % (", inside the graph, tail range beside the backward" if (graphed is not None and graphed.optimizer is not None) else "")) if args.optimizer == "flat" else "AdamW(torch fused)", "hipgraph": graphed is not None, "host_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
                   "grad_exchange": ("all-reduce of the flat fp32 gradient arena in %d ranges, each overlapped with the next backward graph" % split
                                     if (graphed is not None and graphed.graph_b is not None) else
                                     "1 all-reduce of the flat fp32 arena" if world > 1 else "none (1 rank)"),
                   "exchange": exchange, "final_loss": round(lossv[-1], 4)},
        "roofline": roof, "cpu_baseline": cpu, "run_key": run_key,
        "dense_equiv": {"train_gflop_per_image": round(dense_flops / 1e9, 2),
                        "tflops_per_gpu": round(dense_flops * img_s / world / 1e12, 1),
                        "frac_of_bf16_mfma_peak": round(dense_flops * img_s / world / 2.5e15, 4),
                        "note": "6 x MAC of the LARGEST network_def (SURVEY 8d); supernet steps execute ~0.56x of it"},
    }
    print(json.dumps(out))
    if world == 1 and cpu is not None and cpu.get("value"):         # the N > 1 lines that follow on this box copy the CPU baseline
        try:
            os.makedirs(os.path.dirname(n1_file), exist_ok=True)
            with open(n1_file, "w") as f:
                json.dump(out, f)
        except OSError:
            pass
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
