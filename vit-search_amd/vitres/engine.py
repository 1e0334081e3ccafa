"""Train / eval loops: the build's counterpart of reference engine.py (train_one_epoch :57-190,
evaluate :194-261), same call signature and per-iteration protocol:

  (patch-)mixup -> save CPU RNG -> [manual_seed(epoch*10000+iter) if single/hybrid] -> forward ->
  loss = CE(cls) + CE(patch) -> restore CPU RNG -> finite check -> zero_grad -> backward (+ gradient
  exchange over RCCL) -> optimizer step -> meters.

Differences, all deliberate: bf16 compute needs no GradScaler (loss_scaler may be None); gradients of all
ranks are averaged by one flat all-reduce (GradSync) instead of DDP buckets; the non-finite-loss check is
done on the device and read back every `sync_every` iterations (reference: every iteration).
"""
import math
import os
import sys
import time
from collections import defaultdict

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


class Meter:
    """Running sum / count with cross-process reduction (reference utils.SmoothedValue :24-88)."""

    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.total += float(value) * n
        self.count += n

    def synchronize_between_processes(self):
        if not is_dist():
            return
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        t = t.tolist()
        self.count, self.total = int(t[0]), t[1]

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)


class _WireWork:
    """Handle of an all-reduce that ran on the wire buffer: the work + the arena range GradSync.finish() copies back."""
    __slots__ = ("work", "lo", "hi")

    def __init__(self, work, lo, hi):
        self.work, self.lo, self.hi = work, lo, hi

    def wait(self):
        return self.work.wait()


class GradSync:
    """Data-parallel gradient exchange for a vitres model: parameters are broadcast once from rank 0
    (reference DDP constructor, main.py:367) and, every step, the flat gradient arena is all-reduced
    (sum) and scaled by 1/world -- one RCCL call over xGMI instead of DDP's 25 MB buckets."""

    def __init__(self, model, wire_dtype=torch.float32):
        """wire_dtype=torch.bfloat16: the gradients cross xGMI as bf16 -- half the bytes of the exchange the ring pays per link
        (279 MB of fp32 per step for sr_tiny).  Every range is rounded to bf16 into a persistent wire buffer,
        all-reduced there (RCCL sums bf16 in bf16: each of the world - 1 additions rounds once more) and written back to the fp32
        arena by finish(); AdamW's moments and the master weights stay fp32.  Error of an averaged gradient element against the
        fp32 exchange: at most world bf16 roundings (unit roundoff 2^-8) of the operands' mean magnitude; tests/test_dist_gloo.py
        pins it at world 2: every element within 2 * 2^-8 * mean(|g_rank|), the whole arena within 2^-8 in the L2 norm.
        The default stays fp32 -- the reference's DDP reduces fp32 gradients (main.py:366-367)."""
        self.model = model
        self.world = world_size()
        self.wire_dtype = wire_dtype
        self._wire = None
        self._warned = False

    def broadcast_parameters(self):
        if self.world == 1:
            return
        arena = self.model._arena
        if arena is not None:
            dist.broadcast(arena["flat"], src=0)
        else:
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0)
        for b in self.model.buffers():
            dist.broadcast(b, src=0)
        if hasattr(self.model, "invalidate_shadow"):
            self.model.invalidate_shadow()          # ranks > 0 just received new fp32 weights: re-cast a bf16 shadow kept by FlatAdamW

    def broadcast_buffers(self):
        """DDP's `broadcast_buffers=True` default (reference main.py:367): before every forward rank 0's buffers --
        here the BatchNorm running statistics of the convolutional patch embedding, the only buffers of the model -- overwrite
        the other ranks'.  A no-op for buffer-free models (type-0 patch embedding) and on one rank."""
        if self.world == 1:
            return
        bufs = [b for b in self.model.buffers() if b.numel()]
        if not bufs:
            return
        fl = [b for b in bufs if b.is_floating_point()]
        if fl:
            flat = torch.cat([b.reshape(-1).float() for b in fl])
            dist.broadcast(flat, src=0)
            off = 0
            for b in fl:
                b.copy_(flat[off:off + b.numel()].view(b.shape))
                off += b.numel()
        for b in bufs:
            if not b.is_floating_point():
                dist.broadcast(b, src=0)

    def all_reduce_range(self, lo, hi):
        """Asynchronous all-reduce (sum) of elements [lo, hi) of the flat gradient arena; returns the work handle (None
        on one rank).  RCCL runs it on the process group's stream after everything queued so far on the current one."""
        if self.world == 1 or hi <= lo:
            return None
        g = self.model._arena["gcur"]
        if self.wire_dtype == torch.float32:
            return dist.all_reduce(g[lo:hi], async_op=True)
        wire = self._wire_buffer(g)
        self._to_wire(g[lo:hi], wire[lo:hi])
        # the range travels WITH its handle: finish() copies back exactly the ranges whose handles it was given (a handle dropped
        # by an exception, or waited on by the caller, leaves nothing behind that a later finish() could copy over valid gradients)
        return _WireWork(dist.all_reduce(wire[lo:hi], async_op=True), lo, hi)

    def _wire_buffer(self, g):
        if self._wire is None or self._wire.numel() != g.numel() or self._wire.device != g.device:
            self._wire = torch.empty(g.numel(), dtype=self.wire_dtype, device=g.device)
        return self._wire

    @staticmethod
    def _to_wire(src, dst):
        if src.is_cuda and dst.dtype == torch.bfloat16:
            from . import kernels as K
            K.cast_bf16(src, dst)              # vr_cast_f32_bf16: round-to-nearest-even, 16-byte accesses
        else:
            dst.copy_(src)

    def finish(self, works, average=True):
        """Wait for all_reduce_range handles and apply the 1/world averaging (average=False leaves the SUM: an optimizer
        that scales gradients itself -- vitres.optim.FlatAdamW.grad_scale -- saves the extra pass over the arena).  The ranges of
        `works` must cover every gradient the optimizer consumes (bf16 wire: only the exchanged ranges are copied back / averaged)."""
        if self.world == 1:
            return
        works = list(works)                    # (a generator would be exhausted by the waits below)
        for w in works:
            if w is not None:
                w.wait()
        g = self.model._arena["gcur"]
        wired = [w for w in works if isinstance(w, _WireWork)]
        if wired:                              # bf16 wire: summed ranges back into the fp32 arena (averaging folded in)
            if len(wired) != sum(1 for w in works if w is not None):
                raise RuntimeError("GradSync.finish: fp32 and wire-dtype handles mixed in one exchange")
            for w in wired:
                g[w.lo:w.hi].copy_(self._wire[w.lo:w.hi])
                if average:
                    g[w.lo:w.hi].mul_(1.0 / self.world)
            return
        if average:
            g.mul_(1.0 / self.world)

    def all_reduce_grads(self, average=True):
        if self.world == 1:
            return
        a = self.model._arena
        g = a.get("gcur") if a is not None else None
        p0 = a["params"][0] if a is not None else None
        if g is not None and p0.grad is not None and p0.grad.data_ptr() == g.data_ptr() + 4 * a["offsets"][0][0]:
            if self.wire_dtype != torch.float32:
                return self.finish([self.all_reduce_range(0, g.numel())], average=average)
            dist.all_reduce(g)                                    # .grad tensors are views of the flat arena
            if average:
                g.mul_(1.0 / self.world)
            return
        if self.wire_dtype != torch.float32 and not self._warned:
            import warnings
            warnings.warn("GradSync: gradients are not views of the flat arena (autograd cloned them): the per-tensor fallback "
                          "exchanges fp32, wire_dtype=%s is ignored" % self.wire_dtype)
            self._warned = True
        for p in self.model.parameters():                         # autograd cloned the views: per-tensor fallback
            if p.grad is not None:
                dist.all_reduce(p.grad)
                if average:
                    p.grad.mul_(1.0 / self.world)


class KnowledgeDistillationLoss(torch.nn.Module):
    """Distillation term on the distillation token's logits (reference engine.py:25-46): hard = cross entropy against the
    teacher's arg-max class; soft = T^2 * mean_b sum_k -softmax(teacher / T) * log_softmax(x / T) with T = soft_temperature."""

    def __init__(self, hard_distill=True, soft_temperature=3.0):
        super().__init__()
        self.hard_distill = hard_distill
        self.soft_temperature = None if hard_distill else float(soft_temperature)

    def forward(self, x, teacher_output):
        if self.hard_distill:
            return torch.nn.functional.cross_entropy(x, torch.argmax(teacher_output, dim=1))
        t = self.soft_temperature
        soft = torch.softmax(teacher_output / t, dim=1)
        return torch.mean(torch.sum(-soft * torch.log_softmax(x / t, dim=1), 1)) * (t * t)

    def extra_repr(self):
        return 'hard_distill={}'.format(self.hard_distill) + ('' if self.hard_distill else
                                                              ', soft_temperature={}'.format(self.soft_temperature))


def train_step(model, criterion, optimizer, samples, targets, patch_targets=None, patch_output_type=None, epoch=0,
               train_iter=0, arch_sample=None, grad_sync=None, loss_scaler=None, max_norm=None, average_grads=True,
               teacher_output=None, kd_criterion=None, alpha=0.5):
    """One optimisation step; returns the loss tensor (on device, not synchronised).  average_grads=False leaves the
    all-reduced SUM in the arena (optimizer applies 1/world: vitres.optim.FlatAdamW.grad_scale).  teacher_output +
    kd_criterion: knowledge distillation, loss = (1 - alpha) * criterion(cls) + alpha * kd(dst, teacher) (engine.py:135-148;
    a one-token model distils through its class logits, as the reference's `output_dst = outputs`)."""
    rng = None
    if arch_sample is not None:                                   # engine.py:119-131
        rng = torch.random.get_rng_state()
        if arch_sample in ('single', 'hybrid'):
            torch.manual_seed(epoch * 10000 + train_iter)
        elif arch_sample != 'multi':
            raise ValueError('arch_sample has invalid value {}.'.format(arch_sample))
    if patch_targets is None:
        outputs = model(samples)
        output_cls, output_dst = (outputs[0], outputs[1]) if isinstance(outputs, tuple) else (outputs, outputs)
        loss = criterion(output_cls, targets)
        if teacher_output is not None:
            loss = loss * (1 - alpha) + kd_criterion(output_dst, teacher_output) * alpha
    else:
        cls_pred, patch_pred = model(samples, patch_output_type=patch_output_type)
        loss = criterion(cls_pred, targets)
        if patch_output_type == 'seq':
            loss = loss + criterion(patch_pred, patch_targets)
        elif patch_output_type == 'avg':
            loss = loss + criterion(patch_pred, targets)
        else:
            raise ValueError()
    if rng is not None:
        torch.random.set_rng_state(rng)                           # engine.py:164-165
    optimizer.zero_grad(set_to_none=True)
    if loss_scaler is not None and (grad_sync is None or grad_sync.world == 1):
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False)
    elif loss_scaler is not None:
        # timm NativeScaler does scale -> backward -> unscale -> clip -> step in one call; with a gradient exchange the
        # all-reduce has to sit between backward and unscale (DDP does it inside backward, main.py:367), so the call is unrolled
        scaler = getattr(loss_scaler, '_scaler', None)
        if scaler is None:
            raise RuntimeError('loss_scaler with grad_sync on several ranks needs a NativeScaler-like object exposing `_scaler` '
                               '(torch.cuda.amp.GradScaler); bf16 / fp32 training needs no scaler: pass loss_scaler=None')
        scaler.scale(loss).backward()
        grad_sync.all_reduce_grads(average=average_grads)
        scaler.unscale_(optimizer)
        if max_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        scaler.step(optimizer)
        scaler.update()
    else:
        loss.backward()
        if grad_sync is not None:
            grad_sync.all_reduce_grads(average=average_grads)
        if max_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
    return loss.detach()


RUN_AHEAD = int(os.environ.get("VITRES_RUN_AHEAD", "2"))
PLAN_COPY_KERNEL = os.environ.get("VITRES_PLAN_COPY", "kernel") == "kernel"     # the per-step plan buffer: kernel reading pinned memory / memcpy


class GraphedTrainStep:
    """forward + loss + backward of one iteration captured ONCE into a hipGraph and replayed every step
    (the step issues ~600 kernel launches; replay removes their host cost).  What changes between iterations goes
    through static device buffers: the batch, the soft targets, and the int32 keep rows of every ChannelDrop, which
    are still sampled on the host with the reference's RNG protocol before each replay, and the DropPath scale vectors, drawn
    from the model's private CPU generator (model.drop_path_generator(): seeded from torch.initial_seed() + rank, saved and
    restored with the checkpoint's RNG bundle).  The gradient exchange and the optimizer stay outside the graph."""

    def __init__(self, model, criterion, samples, targets, patch_targets=None, patch_output_type=None, warmup=2,
                 split_for_sync=False, optimizer=None):
        """split_for_sync: capture the backward as TWO graphs cut after the last stage (model.split_plan()), so that
        step_with_sync() can all-reduce the finished tail of the gradient arena (most of the parameters) while the rest
        of the backward -- most of the time -- is still running."""
        self.model, self.criterion, self.pot = model, criterion, patch_output_type
        self.graph_b, self.split, self.more_graphs, self.ranges = None, None, [], []
        self._inflight = []
        # optimizer (a vitres.optim.FlatAdamW, single rank): the update becomes part of the graph -- the arena tail (last stage +
        # heads, most parameters) is updated on the side stream as soon as its gradients are final, beside the rest of the
        # backward; the remainder after it.  Call optimizer.prepare_step() before every replay instead of optimizer.step().
        self.optimizer = optimizer if (optimizer is not None and hasattr(optimizer, "step_device")) else None
        if self.optimizer is not None and split_for_sync:
            raise ValueError("optimizer-in-graph is for one rank; with a gradient exchange step the optimizer after step_with_sync")
        # deferred form (VITRES_OPT_DEFER=1, nets with a spatial reduction): the graph OPENS with the update of the previous replay's
        # gradients -- arena head on the main stream, the rest (stages 2.., heads: most parameters) on the side stream beside the
        # first stage's forward -- instead of closing with a 0.36 ms pass nothing overlaps.  Same sequence of updates; the LAST one
        # is applied by finish_update() (call it before evaluating, checkpointing or changing the learning rate: end of an
        # epoch); this object then calls optimizer.prepare_step() itself.  Measured round 3: 7.73 - 7.84 against 7.80 - 7.85 ms
        # (the 8192-workgroup update takes the CUs first -- the forward's first GEMM waits 350 us behind it -- and capped at
        # 256 - 2048 workgroups, VITRES_ADAMW_BLOCKS, the forward slows by what the update reads): within noise, so opt-in.
        self.defer, self._pending = None, False
        if self.optimizer is not None and os.environ.get("VITRES_OPT_DEFER", "0") != "0" and hasattr(model, "split_plan"):
            model._ensure_arena(samples.device)
            cuts3 = model.split_plan(parts=99)
            has_sr = any(type(b).__name__ == "SpatialReductionPatchEmbedding" for b in getattr(model, "blocks", []))
            if has_sr and isinstance(cuts3, list) and cuts3 and cuts3[-1][1] % 8 == 0:
                self.defer = cuts3[-1][1]                          # arena offset of the first spatial reduction
                # checkpoint_dict / evaluate / FlatAdamW.state_dict refuse to run while an update is pending (weights one step behind)
                self.optimizer._graph_pending = lambda: self._pending
                model._graph_pending = self.optimizer._graph_pending
        # soft-target CE is the training loss of every shipped recipe (main.py:390-398): the whole step then runs without
        # autograd and without torch glue between the heads and the backward (model.loss_and_grad / vr_softce_train)
        from .losses import SoftTargetCrossEntropy
        self.fused_loss = isinstance(criterion, SoftTargetCrossEntropy) and hasattr(model, "loss_and_grad") and \
            os.environ.get("VITRES_FUSED_LOSS", "1") != "0"
        self.x, self.t = samples.clone(), targets.clone()
        self.pt = patch_targets.clone() if patch_targets is not None else None
        B = samples.shape[0]
        rng = torch.random.get_rng_state()
        # the DropPath draws come from the model's private generator: the warm-up steps and the capture's plan must not advance the
        # stream a checkpoint restored (the first replay then continues exactly where the saved run stopped)
        dp_rng = model.drop_path_rng_state() if hasattr(model, "drop_path_rng_state") else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                              # eager warm-up (arena, LDS attributes, allocator)
            for _ in range(warmup):
                model.zero_grad(set_to_none=True)
                self._step_body(None)
        torch.cuda.current_stream().wait_stream(side)
        plan = model.sample_plan(B)
        # what changes per replay reaches the graph through ONE static int32 buffer: the keep rows of every ChannelDrop and the
        # DropPath scale vectors (bit-cast floats), both made on the host (model.plan_host_buffer) and uploaded from a ring of
        # pinned staging blocks; for the type-0 patch embedding the patch gather runs in front of the graph, straight from the
        # caller's batch (no copy of the images into a static buffer, no re-ordering pass)
        self.keep_static = None                                   # (name kept: tests / tools look at it)
        self._plan_nk, self._stage, self._stage_i = 0, [], 0
        flat, nk = model.plan_host_buffer(plan)
        if flat.size:
            self.keep_static = torch.from_numpy(flat).to(samples.device)
            self._plan_nk = nk
            self._stage = [[torch.empty(flat.size, dtype=torch.int32).pin_memory(), None] for _ in range(64)]
        self.col_static = None
        if getattr(model, "embed_type", None) == 0 and model.compute_dtype == torch.bfloat16:
            ldk = (model.in_chans * model.patch_size ** 2 + 7) // 8 * 8
            self.col_static = torch.empty((B * model.patch_embed.num_patches, ldk), dtype=torch.bfloat16, device=samples.device)
            self._gather(samples, plan)
        model.zero_grad(set_to_none=True)
        # split_for_sync = number of backward parts (True = 2): part k's graph is followed by the all-reduce of the arena range
        # it completed, overlapping part k+1 (a cut in front of every spatial reduction, counted from the end)
        parts = 2 if split_for_sync is True else int(split_for_sync or 0)
        cuts = None
        if parts >= 2:
            cuts = model.split_plan(parts=max(parts, 3))
            cuts = cuts[:parts - 1] if cuts else None
        if cuts:
            self.split = cuts[0]
        self.graph = torch.cuda.CUDAGraph()
        from . import kernels as K
        K.ensure_workspaces(samples.device, roles=(0, 1))             # stream-K workspaces exist before anything is captured
        model._bwd_split = [c for c, _ in cuts] if cuts else None
        try:
            self._loss_buf = torch.zeros(1, dtype=torch.float32, device=samples.device)
            opt_cut = None
            if self.optimizer is not None:
                self.optimizer.prepare_step()                      # allocates / fills the device hyper-parameters (not captured)
                self.optimizer._step -= 1
                # VITRES_OPT_OVERLAP = number of arena ranges updated EARLY (0 off, 1 (default): head + last stage, 2: + the stage
                # before; measured round 4: 7.47 -> 7.36 - 7.39 ms with 1 or 2, profiles/r04_optimizer_overlap.txt):
                # the backward is cut in front of the spatial reductions (model.split_plan) and the range a part completes is
                # updated on the weight gradients' side stream -- IN ORDER with the groups there: a third branch would land on
                # their hardware queue in front of them (round 4) -- by at most VITRES_OPT_OVERLAP_BLOCKS resident workgroups
                # (256: one per CU; the uncapped update took the chip and cost more than it hid in rounds 1 - 3), beside the rest
                # of the backward; what is left (the first stage + embedding) follows the backward at full width.
                n_early = int(os.environ.get("VITRES_OPT_OVERLAP", "1"))
                opt_cut = None
                if n_early > 0 and self.defer is None:             # (the deferred form updates beside the NEXT forward: no cut)
                    oc = model.split_plan(parts=max(n_early + 1, 3))
                    oc = oc[:n_early] if isinstance(oc, list) else None
                    if oc:
                        opt_cut = oc
                        model._bwd_split = [c for c, _ in oc]
                        model._bwd_join_parts = False              # the next part follows in the same capture
            if self.defer is not None:
                opt_cut = None
                model._deferred_update = (self.optimizer, self.defer)
            with torch.cuda.graph(self.graph):
                if self.keep_static is not None:
                    model.attach_plan_buffer(plan, self.keep_static, self._plan_nk)
                plan.embed_col = self.col_static
                self.loss = self._step_body(plan)
                if self.optimizer is not None and self.defer is None:
                    from . import functional as Fn
                    n_arena = model._arena["flat"].numel()
                    if opt_cut is not None and getattr(model, "_bwd_state", None) is not None:
                        cap = int(os.environ.get("VITRES_OPT_OVERLAP_BLOCKS", "256"))
                        hi = n_arena
                        for _, lo in opt_cut:                          # ranges complete from the arena's end backwards
                            if Fn.OVERLAP:
                                Fn.on_side(lambda lo=lo, hi=hi: self.optimizer.step_device(lo, hi, max_blocks=cap), after_all_sides=True)
                            else:
                                self.optimizer.step_device(lo, hi)
                            hi = lo
                            if getattr(model, "_bwd_state", None) is not None:
                                model.resume_backward()
                        self.optimizer.step_device(0, hi)
                    else:
                        self.optimizer.step_device(0, n_arena)
            while getattr(model, "_bwd_state", None) is not None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.graph.pool()):
                    model.resume_backward()
                self.more_graphs.append(g)
        finally:
            model._bwd_split = None
            model._bwd_join_parts = True
            model._deferred_update = None
        if self.more_graphs:
            self.graph_b = self.more_graphs[0]
            end = model._arena["gcur"].numel()
            for _, start in cuts:                                  # arena range completed by part 1, 2, ... (the rest: last part)
                self.ranges.append((start, end))
                end = start
            self.ranges.append((0, end))
        self.loss = self.loss.detach()
        torch.random.set_rng_state(rng)
        if dp_rng is not None:
            model.set_drop_path_rng_state(dp_rng)

    def _gather(self, samples, plan):
        """Patch gather of the caller's batch into the graph's static patchify operand (internal, arch-grouped sample order)."""
        from . import kernels as K
        smap = None
        if plan.order is not None:
            smap, _ = self.model._order_tensors(plan.order, samples.device)
        K.im2col_patch(samples.contiguous().float(), self.model.patch_size, self.col_static.shape[1], torch.bfloat16,
                       sample_map=smap, out=self.col_static)

    def _step_body(self, plan):
        """forward + loss + (first part of the) backward of one step; returns the loss tensor."""
        if self.fused_loss:
            return self.model.loss_and_grad(self.x, self.t, self.pt, self.pot, plan=plan,
                                            loss_out=getattr(self, "_loss_buf", None) if plan is not None else None)
        loss = self._loss(self.model(self.x, patch_output_type=self.pot, plan=plan))
        loss.backward()
        return loss

    def _loss(self, out):
        if self.pt is None:
            return self.criterion(out[0] if isinstance(out, tuple) else out, self.t)
        cls_pred, patch_pred = out
        return self.criterion(cls_pred, self.t) + self.criterion(patch_pred, self.pt if self.pot == 'seq' else self.t)

    def __call__(self, samples, targets, patch_targets=None, epoch=0, train_iter=0, arch_sample=None):
        rng = None
        if arch_sample is not None:                                # engine.py:119-131
            rng = torch.random.get_rng_state()
            if arch_sample in ('single', 'hybrid'):
                torch.manual_seed(epoch * 10000 + train_iter)
            elif arch_sample != 'multi':
                raise ValueError('arch_sample has invalid value {}.'.format(arch_sample))
        plan = self.model.sample_plan(samples.shape[0])
        if rng is not None:
            torch.random.set_rng_state(rng)
        probe = self._gap_probe
        if probe is not None:                                      # dev aid (VITRES_DBG_GAP2): GPU time of the phases of one call
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        if self.keep_static is not None:
            flat, _ = self.model.plan_host_buffer(plan)
            slot = self._stage[self._stage_i % len(self._stage)]
            self._stage_i += 1
            if slot[1] is not None:
                slot[1].synchronize()                             # the host runs ahead of the device: the block's last copy is done?
            slot[0].numpy()[:] = flat
            if PLAN_COPY_KERNEL:
                from . import kernels as K
                K.copy_i32_from_pinned(slot[0], self.keep_static.view(-1))
            else:
                self.keep_static.copy_(slot[0], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
        if probe is not None:
            ev[1].record()
        if self.col_static is not None:
            self._gather(samples, plan)                           # reads the caller's tensor directly
        elif samples.data_ptr() != self.x.data_ptr():
            self.x.copy_(samples, non_blocking=True)
        if targets.data_ptr() != self.t.data_ptr():
            self.t.copy_(targets, non_blocking=True)
            if self.pt is not None:
                self.pt.copy_(patch_targets, non_blocking=True)
        if self.defer is not None:
            self.optimizer.prepare_step(noop=not self._pending)   # the update this replay opens with: the previous replay's gradients
            self._pending = True
        # bounded run-ahead: the host needs ~3 ms for a 7.5 ms step, so left alone it queues replay after replay until the runtime's
        # queue limit stops it (~25 steps in); on the way the runtime grows its per-launch resources a few times, and each growth
        # stalls the device for ~1 ms (six 8.3 - 8.9 ms steps among the first 26 of a run, none after).  RUN_AHEAD replays in
        # flight keep the device fed with the host two steps ahead from the third step on (VITRES_RUN_AHEAD, 0 = unbounded).
        if RUN_AHEAD > 0:
            if len(self._inflight) >= RUN_AHEAD:
                self._inflight.pop(0).synchronize()
        if probe is not None:
            ev[2].record()
        self.graph.replay()
        self.model._stem_fold = None                               # (stem.drop_fold: the replay moved BatchNorm's running statistics)
        for k, g in enumerate(self.more_graphs):
            if self._sync is not None:                            # the arena range of the part just replayed is final: exchange it now
                self._works.append(self._sync.all_reduce_range(*self.ranges[k]))
            g.replay()
        if probe is not None:
            ev[3].record()
            probe.append(ev)
        if RUN_AHEAD > 0:
            e_done = torch.cuda.Event()
            e_done.record()
            self._inflight.append(e_done)
        return self.loss

    def finish_update(self):
        """Deferred optimizer-in-graph: apply the update of the last replay's gradients now (eagerly).  No-op otherwise."""
        if self.defer is not None and self._pending:
            self.optimizer.prepare_step()
            self.optimizer.step_device(0, self.model._arena["flat"].numel())
            self._pending = False

    _sync, _works = None, ()
    _inflight = None
    _gap_probe = [] if os.environ.get("VITRES_DBG_GAP2") else None

    def gap_report(self, skip=10):
        """VITRES_DBG_GAP2: average GPU ms of [plan copy | gather + target copies | graph replay | end of a call -> start of the next]."""
        pr = self._gap_probe[skip:] if self._gap_probe else []
        if len(pr) < 2:
            return None
        n = len(pr)
        seg = [sum(e[k].elapsed_time(e[k + 1]) for e in pr) / n for k in range(3)]
        seg.append(sum(pr[i][3].elapsed_time(pr[i + 1][0]) for i in range(n - 1)) / (n - 1))
        return [round(v, 4) for v in seg]

    def step_with_sync(self, grad_sync, samples, targets, patch_targets=None, average=True, **kw):
        """Replay + data-parallel gradient exchange: with split_for_sync the all-reduce of the last stage's gradients
        overlaps the second backward graph; the remainder follows it.  Gradients are averaged on return (average=False:
        summed -- for an optimizer that applies 1/world itself)."""
        self._sync, self._works = grad_sync, []
        try:
            loss = self(samples, targets, patch_targets, **kw)
        finally:
            self._sync = None
        ev = None
        if getattr(self, "exposed", None) is not None and grad_sync.world > 1:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                           # the backward is complete here (on the compute stream)
        if self.more_graphs:
            self._works.append(grad_sync.all_reduce_range(*self.ranges[-1]))
            grad_sync.finish(self._works, average=average)
        else:
            grad_sync.all_reduce_grads(average=average)
        if ev is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()                                          # work.wait() made the compute stream wait for the exchange
            self.exposed.append((ev, ev1))
        self._works = ()
        return loss


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler=None, max_norm=0,
                    model_ema=None, mixup_fn=None, print_freq=100, teacher_model=None, hard_distill=True, alpha=0.5,
                    logger=None, arch_sample=False, patch_mixup_fn=None, grad_sync=None, sync_every=1):
    kd_criterion = None
    if teacher_model is not None:                                 # engine.py:91-95: any module mapping images to logits
        kd_criterion = KnowledgeDistillationLoss(hard_distill=hard_distill)
        teacher_model.eval()
    model.train()
    criterion.train()
    print_out = logger.info if logger else print
    meters = defaultdict(Meter)
    arch_sample = arch_sample or None
    pending = []
    t0 = time.time()
    for train_iter, (samples, targets) in enumerate(data_loader):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        patch_targets, patch_output_type = None, None
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
            assert patch_mixup_fn is None
        if patch_mixup_fn is not None:
            samples, targets, patch_targets, patch_output_type = patch_mixup_fn(samples, targets)
        teacher_output = None
        if teacher_model is not None:
            with torch.no_grad():
                teacher_output = teacher_model(samples)
        if grad_sync is not None:
            grad_sync.broadcast_buffers()                         # DDP broadcast_buffers=True (main.py:367): per forward
        loss = train_step(model, criterion, optimizer, samples, targets, patch_targets, patch_output_type, epoch,
                          train_iter, arch_sample, grad_sync, loss_scaler, max_norm, teacher_output=teacher_output,
                          kd_criterion=kd_criterion, alpha=alpha)
        pending.append(loss)
        if model_ema is not None:
            model_ema.update(model)
        if len(pending) >= sync_every:
            for v in torch.stack(pending).tolist():               # device -> host sync (reference: every iteration)
                if not math.isfinite(v):
                    print_out('Loss is {}, stopping training'.format(v))
                    sys.exit(1)
                meters['loss'].update(v)
            pending = []
        meters['lr'].update(optimizer.param_groups[0]['lr'])
        if print_freq and train_iter % print_freq == 0:
            print_out('Epoch: [{}] [{}] loss: {:.4f} time: {:.1f}s'.format(epoch, train_iter, meters['loss'].global_avg,
                                                                          time.time() - t0))
    for v in (torch.stack(pending).tolist() if pending else []):
        if not math.isfinite(v):
            print_out('Loss is {}, stopping training'.format(v))
            sys.exit(1)
        meters['loss'].update(v)
    for m in meters.values():
        m.synchronize_between_processes()
    print_out('Averaged stats: ' + '  '.join('{}: {:.6f}'.format(k, m.global_avg) for k, m in meters.items()))
    return {k: m.global_avg for k, m in meters.items()}


def accuracy(output, target, topk=(1,)):
    """timm.utils.accuracy: top-k accuracy in percent."""
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * 100.0 / target.size(0) for k in topk]


@torch.no_grad()
def evaluate(data_loader, model, device, print_freq=100, logger=None):
    """engine.evaluate (:194-261): eval forward, CE, top-1/5 weighted by batch size, global averages."""
    criterion = torch.nn.CrossEntropyLoss()
    print_out = logger.info if logger else print
    meters = defaultdict(Meter)
    if getattr(model, "_graph_pending", None) is not None and model._graph_pending():
        raise RuntimeError("a deferred in-graph optimizer update is pending (weights one step behind): call "
                           "GraphedTrainStep.finish_update() before evaluating")
    model.eval()
    for images, target in data_loader:
        images = images.to(device, non_blocking=True)
        target = target.to(device, non_blocking=True)
        output = model(images)
        output_cls, output_dst = (output[0], output[1]) if isinstance(output, tuple) else (output, None)
        loss = criterion(output_cls, target)
        acc1, acc5 = accuracy(output_cls, target, topk=(1, 5))
        n = images.shape[0]
        meters['loss'].update(loss.item())
        meters['acc1'].update(acc1.item(), n=n)
        meters['acc5'].update(acc5.item(), n=n)
        if output_dst is not None:                       # two-token variants: distillation head and joint softmax (:230-238)
            d1, d5 = accuracy(output_dst, target, topk=(1, 5))
            meters['dst_acc1'].update(d1.item(), n=n)
            meters['dst_acc5'].update(d5.item(), n=n)
            joint = torch.softmax(output_cls, dim=1) + torch.softmax(output_dst, dim=1)
            j1, j5 = accuracy(joint, target, topk=(1, 5))
            meters['jnt_acc1'].update(j1.item(), n=n)
            meters['jnt_acc5'].update(j5.item(), n=n)
    for m in meters.values():
        m.synchronize_between_processes()
    info = 'Acc@1: {:.2f}, Acc@5: {:.2f}, loss: {:.2f}'.format(meters['acc1'].global_avg, meters['acc5'].global_avg,
                                                             meters['loss'].global_avg)
    if 'dst_acc1' in meters:
        info += ', Distill Acc@1: {:.2f}, Distill Acc@5: {:.2f}, Joint Acc@1: {:.2f}, Joint Acc@5: {:.2f}'.format(
            meters['dst_acc1'].global_avg, meters['dst_acc5'].global_avg, meters['jnt_acc1'].global_avg,
            meters['jnt_acc5'].global_avg)
    print_out(info + '\n')
    return {k: m.global_avg for k, m in meters.items()}


def param_groups_weight_decay(model, weight_decay=0.05):
    """timm 0.3.2 optim_factory.add_weight_decay as used by create_optimizer (main.py:385): 1-D parameters,
    `.bias` and names returned by model.no_weight_decay() get no weight decay."""
    skip = model.no_weight_decay() if hasattr(model, 'no_weight_decay') else set()
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim == 1 or name.endswith('.bias') or name in skip) else decay).append(p)
    return [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]
