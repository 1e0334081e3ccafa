"""Learning-rate schedule of the training loop: the build's counterpart of `create_scheduler(args, optimizer)` (reference
main.py:388, default --sched cosine :111; stepped once per epoch by `lr_scheduler.step(epoch)` :462 and check-pointed through
state_dict / load_state_dict :410,509).

timm 0.3.2 is not vendored in the reference ("parity unpinned" at this boundary): `CosineLRScheduler` restates its
published semantics -- linear warm-up from `warmup_lr_init` over `warmup_t` epochs (no warm-up prefix: the cosine position is the
absolute epoch), then lr_min + (lr - lr_min) / 2 * (1 + cos(pi * t / t_initial)), and `lr_min` once `cycle_limit` cycles are over.
It only writes `param_group['lr']`, so it drives torch optimizers and vitres.optim.FlatAdamW alike (the flat AdamW reads the
group's lr at every step; inside a captured hipGraph through FlatAdamW.prepare_step()).
"""
import math


class CosineLRScheduler:
    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0, decay_rate=1.0, cycle_limit=1):
        assert t_initial > 0 and lr_min >= 0
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = t_initial, lr_min, warmup_t, warmup_lr_init
        self.decay_rate, self.cycle_limit = decay_rate, cycle_limit
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_values = [g['initial_lr'] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._update(self.warmup_lr_init)                     # training starts at the warm-up learning rate
        else:
            self.warmup_steps = [1.0 for _ in self.base_values]

    def _update(self, values):
        if not isinstance(values, (list, tuple)):
            values = [values] * len(self.optimizer.param_groups)
        for g, v in zip(self.optimizer.param_groups, values):
            g['lr'] = v

    def get_epoch_values(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        i = t // self.t_initial
        t_curr = t - self.t_initial * i
        gamma = self.decay_rate ** i
        lr_min = self.lr_min * gamma
        if self.cycle_limit == 0 or i < self.cycle_limit:
            return [lr_min + 0.5 * (v * gamma - lr_min) * (1 + math.cos(math.pi * t_curr / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def get_cycle_length(self, cycles=0):
        return self.t_initial * max(1, cycles or self.cycle_limit)

    def step(self, epoch, metric=None):
        self._update(self.get_epoch_values(epoch))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)


def create_scheduler(args, optimizer):
    """(scheduler, num_epochs) from an argparse-like object with the reference's flags (main.py:110-135): epochs, sched, min_lr,
    warmup_lr, warmup_epochs, cooldown_epochs, decay_rate.  Only --sched cosine (every shipped recipe) is built."""
    if getattr(args, 'sched', 'cosine') != 'cosine':
        raise NotImplementedError("only --sched cosine (the reference default, used by every shipped script)")
    sched = CosineLRScheduler(optimizer, t_initial=args.epochs, lr_min=getattr(args, 'min_lr', 1e-5),
                              warmup_t=getattr(args, 'warmup_epochs', 5), warmup_lr_init=getattr(args, 'warmup_lr', 1e-6),
                              decay_rate=getattr(args, 'decay_rate', 0.1), cycle_limit=1)
    return sched, sched.get_cycle_length() + getattr(args, 'cooldown_epochs', 10)
