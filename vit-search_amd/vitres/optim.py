"""Step tail on the flat parameter arena: AdamW (+ bf16 shadow, gradient averaging, EMA) in ONE HIP pass.

Reference: timm 0.3.2 `create_optimizer(args, model)` -> torch.optim.AdamW(lr, weight_decay=0.05) over the two groups of
`add_weight_decay` (main.py:385; engine.param_groups_weight_decay restates the grouping), `ModelEmaV2` (main.py:357-363).
`FlatAdamW` keeps torch.optim.Optimizer's interface (param_groups with mutable 'lr' for the cosine scheduler,
state_dict / load_state_dict, zero_grad) but its state is two flat fp32 buffers shaped like the model's arena, and step()
is one `vr_adamw_flat` launch that also refreshes the bf16 weight shadow the next forward reads.
"""
import ctypes
import math

import torch

from . import _lib
from .kernels import _p, _stream

MAX_GROUPS = 16


class _Group(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "bias_c1", "sqrt_bias_c2",
                                              "grad_scale")]


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, ema_decay=None):
        """params: iterable of parameters or of param-group dicts (as torch.optim.AdamW); every parameter must belong to
        `model`, whose arena they live in.  ema_decay: keep an exponential moving average of the parameters
        (`ema_state_dict()` returns it under the model's state_dict keys)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) > MAX_GROUPS:
            raise ValueError("at most %d parameter groups" % MAX_GROUPS)
        self.model = model
        self.ema_decay = ema_decay
        self._step = 0
        self._arena_id = None
        self.grad_scale = 1.0              # e.g. 1/world when the all-reduce leaves a SUM in the gradient arena

    # ---- arena-shaped state -------------------------------------------------------------------------------
    def _bind(self):
        dev = next(self.model.parameters()).device
        a = self.model._ensure_arena(dev)
        if self._arena_id == id(a):
            return a
        flat = a["flat"]
        if flat.numel() % 8:
            raise RuntimeError("arena length must be a multiple of 8")
        gid = torch.full((flat.numel() // 8,), 255, dtype=torch.uint8)
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if id(p) not in a["index"]:
                    raise ValueError("FlatAdamW: parameter does not belong to the model's arena")
                off, n = a["offsets"][a["index"][id(p)]]
                gid[off // 8:(off + n + 7) // 8] = gi
        old = getattr(self, "_flat_state", None)
        self._flat_state = {"m": torch.zeros_like(flat), "v": torch.zeros_like(flat), "gid": gid.to(flat.device),
                            "ema": flat.clone() if self.ema_decay is not None else None}
        if old is not None and old["m"].numel() == flat.numel():      # arena rebuilt (e.g. .to(device)): carry the state
            for k in ("m", "v", "ema"):
                if old[k] is not None and self._flat_state[k] is not None:
                    self._flat_state[k].copy_(old[k])
        self._arena_id = id(a)
        return a

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        a = self._bind()
        g = a.get("gcur")
        p0 = a["params"][0]
        if g is None or p0.grad is None or p0.grad.data_ptr() != g.data_ptr() + 4 * a["offsets"][0][0]:
            raise RuntimeError("FlatAdamW needs the gradients in the model's flat arena (one backward since zero_grad)")
        self._step += 1
        self.model._stem_fold = None           # parameters change through raw pointers: no Tensor._version moves (stem.drop_fold)
        arr = self._group_structs(self._step)
        st = self._flat_state
        shadow = a["shadow"] if self.model.compute_dtype == torch.bfloat16 else None
        _lib.check(_lib.lib().vr_adamw_flat(_p(a["flat"]), _p(g), _p(st["m"]), _p(st["v"]), _p(shadow), _p(st["ema"]),
                                            float(self.ema_decay or 0.0), _p(st["gid"]), ctypes.byref(arr),
                                            len(self.param_groups), a["flat"].numel(), _stream()), "vr_adamw_flat")
        if shadow is not None:
            a["shadow_ok"] = True              # forwards skip their own vr_cast_f32_bf16 from now on (model.invalidate_shadow)
        return loss

    # ---- the update as part of a captured hipGraph ---------------------------------------------------------------------
    def _group_structs(self, t):
        arr = (_Group * len(self.param_groups))()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            arr[gi] = _Group(group["lr"], b1, b2, group["eps"], group["weight_decay"], 1.0 - b1 ** t,
                             math.sqrt(1.0 - b2 ** t), self.grad_scale)
        return arr

    def prepare_step(self, noop=False):
        """Advance the step count and upload this step's per-group hyper-parameters (learning rates written by the scheduler,
        bias corrections) to the device buffer step_device() launches read -- call once before every replay of a graph that
        contains step_device() launches (engine.GraphedTrainStep(optimizer=...)).  noop: upload all-zero groups instead -- the
        captured launches then change nothing (vr_adamw_flat_dev skips a group whose bias correction is 0) and the step count
        stays: the first replay of a graph that applies the PREVIOUS replay's gradients."""
        a = self._bind()
        dev = a["flat"].device
        if getattr(self, "_hp_dev", None) is None or self._hp_dev.device != dev:
            self._hp_dev = torch.zeros(MAX_GROUPS * 8, dtype=torch.float32, device=dev)
        vals = []
        if not noop:
            self._step += 1
            self.model._stem_fold = None       # parameters change through raw pointers: no Tensor._version moves (stem.drop_fold)
            arr = self._group_structs(self._step)
            for gi in range(len(self.param_groups)):
                vals += [getattr(arr[gi], n) for n, _ in _Group._fields_]
        host = torch.zeros(MAX_GROUPS * 8, dtype=torch.float32)
        if vals:
            host[:len(vals)] = torch.tensor(vals, dtype=torch.float32)
        if dev.type == "cuda":
            host = host.pin_memory()       # a fresh pinned block per step: the host runs several replays ahead of the device, a
                                           # reused staging buffer would be overwritten before its copy has executed
        self._hp_dev.copy_(host, non_blocking=True)

    @torch.no_grad()
    def step_device(self, lo=0, hi=None, max_blocks=0):
        """AdamW over the arena range [lo, hi) (multiples of 8) with the hyper-parameters prepare_step() uploaded: capturable.
        max_blocks > 0: launched with at most that many workgroups (an update that runs beside other work)."""
        a = self._bind()
        g = a.get("gcur")
        if g is None or getattr(self, "_hp_dev", None) is None:
            raise RuntimeError("step_device needs gradients in the arena and a prepare_step() before it")
        n = a["flat"].numel()
        hi = n if hi is None else hi
        if lo % 8 or hi % 8 or not (0 <= lo < hi <= n):
            raise ValueError("range must be non-empty and aligned to 8 elements")
        st = self._flat_state
        shadow = a["shadow"] if self.model.compute_dtype == torch.bfloat16 else None
        self.model._stem_fold = None           # (stem.drop_fold)

        def at(t, esz):
            return None if t is None else t.data_ptr() + lo * esz
        _lib.check(_lib.lib().vr_adamw_flat_dev_capped(at(a["flat"], 4), at(g, 4), at(st["m"], 4), at(st["v"], 4), at(shadow, 2),
                                                       at(st["ema"], 4), float(self.ema_decay or 0.0), st["gid"].data_ptr() + lo // 8,
                                                       _p(self._hp_dev), len(self.param_groups), hi - lo, int(max_blocks), _stream()),
                   "vr_adamw_flat_dev_capped")
        if shadow is not None:
            a["shadow_ok"] = True

    def own_shadow(self):
        """Declare before capturing a hipGraph that this optimizer keeps the bf16 weight shadow up to date: casts it once
        now; forwards (and graphs captured from now on) contain no cast of their own."""
        a = self._bind()
        if self.model.compute_dtype == torch.bfloat16:
            from . import kernels as K
            K.cast_bf16(a["flat"], a["shadow"])
            a["shadow_ok"] = True

    # ---- checkpoint / EMA views -------------------------------------------------------------------------------
    def state_dict(self):
        if getattr(self, "_graph_pending", None) is not None and self._graph_pending():
            raise RuntimeError("a deferred in-graph update is pending: call GraphedTrainStep.finish_update() first")
        self._bind()
        st = self._flat_state
        return {"step": self._step, "exp_avg": st["m"].clone(), "exp_avg_sq": st["v"].clone(),
                "ema": None if st["ema"] is None else st["ema"].clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        """Accepts its own flat layout or torch.optim.AdamW's (the 'optimizer' entry of a reference checkpoint.pth.tar,
        main.py:506-512): per-parameter exp_avg / exp_avg_sq are scattered into the arena-shaped moments."""
        if "state" in sd:
            return self.load_torch_state_dict(sd)
        self._bind()
        st = self._flat_state
        self._step = int(sd["step"])
        st["m"].copy_(sd["exp_avg"])
        st["v"].copy_(sd["exp_avg_sq"])
        if st["ema"] is not None and sd.get("ema") is not None:
            st["ema"].copy_(sd["ema"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)

    def _ordered_params(self):
        return [p for g in self.param_groups for p in g["params"]]      # torch numbers parameters in this order

    def torch_state_dict(self):
        """The state in torch.optim.AdamW.state_dict() layout ({'state': {i: {step, exp_avg, exp_avg_sq}}, 'param_groups'}),
        i.e. what the reference writes under 'optimizer' -- a checkpoint saved here resumes in the reference and vice versa."""
        a = self._bind()
        st = self._flat_state
        state, groups, i = {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                off, n = a["offsets"][a["index"][id(p)]]
                if self._step > 0:
                    state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": st["m"][off:off + n].view(p.shape).clone(),
                                "exp_avg_sq": st["v"][off:off + n].view(p.shape).clone()}
                ids.append(i)
                i += 1
            pg = {k: v for k, v in g.items() if k != "params"}
            pg.setdefault("amsgrad", False)
            pg["params"] = ids
            groups.append(pg)
        return {"state": state, "param_groups": groups}

    def load_torch_state_dict(self, sd):
        a = self._bind()
        st = self._flat_state
        params = self._ordered_params()
        n_saved = sum(len(g["params"]) for g in sd["param_groups"])
        if n_saved != len(params) or len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        steps = set()
        st["m"].zero_()
        st["v"].zero_()
        saved_ids = [i for g in sd["param_groups"] for i in g["params"]]
        for p, i in zip(params, saved_ids):
            ps = sd["state"].get(i)
            if ps is None:
                continue
            if tuple(ps["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state of parameter %d has shape %s, expected %s" % (i, tuple(ps["exp_avg"].shape),
                                                                                                tuple(p.shape)))
            off, n = a["offsets"][a["index"][id(p)]]
            st["m"][off:off + n].copy_(ps["exp_avg"].reshape(-1))
            st["v"][off:off + n].copy_(ps["exp_avg_sq"].reshape(-1))
            steps.add(int(ps["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): not an AdamW state this optimizer can continue" % sorted(steps))
        self._step = steps.pop() if steps else 0
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k not in ("params", "amsgrad", "foreach", "maximize", "capturable",
                                                               "differentiable", "fused", "decoupled_weight_decay")})

    def ema_state_dict(self):
        """EMA parameters under the model's state_dict keys (buffers are taken from the live model, as ModelEmaV2's
        deepcopy shares nothing but is updated from them every step)."""
        a = self._bind()
        if self._flat_state["ema"] is None:
            raise RuntimeError("constructed without ema_decay")
        ema = self._flat_state["ema"]
        byid = {id(p): ema[off:off + n].view(p.shape) for p, (off, n) in zip(a["params"], a["offsets"])}
        out = {}
        for k, v in self.model.state_dict(keep_vars=True).items():
            out[k] = byid[id(v)].clone() if id(v) in byid else v.detach().clone()
        return out

    def load_ema_state_dict(self, sd):
        """Inverse of ema_state_dict(): scatter a model-keyed state dict (the 'model_ema' entry of a checkpoint, reference
        utils._load_checkpoint_for_ema / main.py:413-414) into the arena-shaped EMA.  Buffers in `sd` are ignored (the EMA tracks
        parameters only; buffers come from the live model)."""
        a = self._bind()
        if self._flat_state["ema"] is None:
            raise RuntimeError("constructed without ema_decay")
        ema = self._flat_state["ema"]
        where = {id(p): (off, n) for p, (off, n) in zip(a["params"], a["offsets"])}
        missing = []
        with torch.no_grad():
            for k, v in self.model.state_dict(keep_vars=True).items():
                if id(v) not in where:
                    continue
                if k not in sd:
                    missing.append(k)
                    continue
                off, n = where[id(v)]
                if tuple(sd[k].shape) != tuple(v.shape):
                    raise ValueError("model_ema[%s] has shape %s, expected %s" % (k, tuple(sd[k].shape), tuple(v.shape)))
                ema[off:off + n].copy_(sd[k].reshape(-1).to(ema.device, torch.float32))
        if missing:
            raise KeyError("model_ema lacks parameters: %s" % ", ".join(missing[:5]))

