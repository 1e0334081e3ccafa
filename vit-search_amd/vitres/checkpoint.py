"""Checkpoints in the reference's own layout (main.py:400-424,496-515): one dict
    {'model', 'optimizer', 'lr_scheduler', 'epoch', 'args'[, 'model_ema'][, 'vitres_rng']}
written with torch.save as `checkpoint.pth.tar` (+ `epoch@E_checkpoint.pth.tar` every tenth epoch), so that runs move between
the reference and this stack in either direction.  'model' / 'model_ema' use the reference's state_dict keys;
'optimizer' is torch.optim.AdamW's layout -- vitres.optim.FlatAdamW converts to and from its arena-shaped moments.
"""
import os

import torch

from .nets.net_utils import get_sub_state_dict

CHECKPOINT_NAME = 'checkpoint.pth.tar'


def _optimizer_state(optimizer):
    return optimizer.torch_state_dict() if hasattr(optimizer, 'torch_state_dict') else optimizer.state_dict()


def _cpu(tree):
    if isinstance(tree, torch.Tensor):
        return tree.detach().cpu()
    if isinstance(tree, dict):
        return {k: _cpu(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_cpu(v) for v in tree)
    return tree


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collect_rng_states(model):
    """COLLECTIVE (every rank calls it, e.g. at the end of an epoch before rank 0 writes the checkpoint): the DropPath generator
    state of every rank, indexed by rank -- pass the result to checkpoint_dict / save_checkpoint(rng_states=...) so that a
    resumed multi-rank run continues every rank's own noise stream.  One rank: a one-element list, no communication."""
    import torch.distributed as dist
    state = model.drop_path_rng_state() if hasattr(model, 'drop_path_rng_state') else None
    rank, world = _rank_world()
    if world == 1:
        return [state]
    out = [None] * world
    dist.all_gather_object(out, state)
    return out


def _derived_seed(state, rank):
    """A seed that is a function of a saved generator state and a rank: ranks whose own state is not in the checkpoint continue
    with streams that differ from one another and from the saving rank's, reproducibly."""
    import hashlib
    h = hashlib.sha256(bytes(state.numpy().tobytes()) + int(rank).to_bytes(8, 'little')).digest()
    return int.from_bytes(h[:8], 'little') % (2 ** 63)


def checkpoint_dict(model, optimizer, lr_scheduler, epoch, args=None, model_ema=None, rng_states=None):
    """model_ema: a state_dict (e.g. FlatAdamW.ema_state_dict()) or a module holding the averaged weights.
    rng_states: collect_rng_states(model) (per-rank DropPath generator states); None: this rank's state alone is saved."""
    out = {'model': _cpu(model.state_dict()), 'optimizer': _cpu(_optimizer_state(optimizer)),
           'lr_scheduler': lr_scheduler.state_dict() if lr_scheduler is not None else {}, 'epoch': epoch, 'args': args}
    if model_ema is not None:
        out['model_ema'] = _cpu(model_ema if isinstance(model_ema, dict) else model_ema.state_dict())
    if hasattr(model, 'drop_path_rng_state'):
        # one key beyond the reference's layout (its loaders index by name and ignore it): the DropPath draws of this stack come
        # from a private CPU generator, so a resumed run continues the same noise stream
        # (the reference draws DropPath noise per rank from seed + rank, main.py:261-267: states are kept PER RANK)
        rank, world = _rank_world()
        out['vitres_rng'] = {'drop_path': model.drop_path_rng_state(), 'rank': rank, 'world': world}
        if rng_states is not None:
            if len(rng_states) != world:
                raise ValueError('rng_states has %d entries for %d ranks' % (len(rng_states), world))
            out['vitres_rng']['drop_path_by_rank'] = list(rng_states)
    if getattr(optimizer, '_graph_pending', None) is not None and optimizer._graph_pending():
        raise RuntimeError('a deferred in-graph optimizer update is pending: call GraphedTrainStep.finish_update() before '
                           'checkpointing (the weights are one step behind)')
    return out


def save_checkpoint(output_dir, model, optimizer, lr_scheduler, epoch, args=None, model_ema=None, rng_states=None):
    """Writes <output_dir>/checkpoint.pth.tar and, for epoch % 10 == 9, epoch@<epoch>_checkpoint.pth.tar (main.py:504-515)."""
    d = checkpoint_dict(model, optimizer, lr_scheduler, epoch, args=args, model_ema=model_ema, rng_states=rng_states)
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, CHECKPOINT_NAME)
    torch.save(d, path)
    if epoch % 10 == 9:
        torch.save(d, os.path.join(output_dir, 'epoch@{}_checkpoint.pth.tar'.format(epoch)))
    return path


def resume(path_or_dict, model, optimizer=None, lr_scheduler=None, eval_mode=False, load_ema=None):
    """--resume (main.py:401-417): loads 'model'; unless eval_mode, and when the checkpoint carries all of 'optimizer',
    'lr_scheduler' and 'epoch', restores them and returns the epoch to start from (saved epoch + 1; else None); load_ema(state)
    receives 'model_ema' on that path (default: a FlatAdamW built with ema_decay takes it through
    load_ema_state_dict, the counterpart of utils._load_checkpoint_for_ema).  With eval_mode, 'model_ema' (when present) replaces the model's weights."""
    ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location='cpu', weights_only=False)
    model.load_state_dict(ck['model'])
    start = None
    if not eval_mode and all(k in ck for k in ('optimizer', 'lr_scheduler', 'epoch')):
        if optimizer is not None:
            optimizer.load_state_dict(ck['optimizer'])
        if lr_scheduler is not None:
            lr_scheduler.load_state_dict(ck['lr_scheduler'])
        start = ck['epoch'] + 1
        if 'vitres_rng' in ck and hasattr(model, 'set_drop_path_rng_state'):
            # every rank continues ITS OWN DropPath stream: the per-rank list when the checkpoint has one for this world size; else
            # the saved state on the rank that saved it and, on the others, a stream derived from (saved state, rank) -- never the
            # same state on two ranks (they would draw identical noise for the rest of training).  Without the key: the fresh
            # (seed + rank) stream drop_path_generator() makes.
            bundle = ck['vitres_rng']
            rank, world = _rank_world()
            by_rank = bundle.get('drop_path_by_rank')
            if by_rank is not None and len(by_rank) == world and by_rank[rank] is not None:
                model.set_drop_path_rng_state(by_rank[rank])
            elif world == 1 or rank == bundle.get('rank', 0):
                model.set_drop_path_rng_state(bundle['drop_path'])
            else:
                model.drop_path_generator(seed=None).manual_seed(_derived_seed(bundle['drop_path'], rank))
        if 'model_ema' in ck:
            if load_ema is not None:
                load_ema(ck['model_ema'])
            elif getattr(optimizer, 'ema_decay', None) is not None and hasattr(optimizer, 'load_ema_state_dict'):
                optimizer.load_ema_state_dict(ck['model_ema'])     # FlatAdamW keeps the EMA: continue it, do not restart it
    if eval_mode and 'model_ema' in ck:
        model.load_state_dict(ck['model_ema'])
    return start


def inherit_supernet_weights(model, path_or_dict, use_ema=False):
    """--resume-supernet-weights (main.py:418-424): a searched sub-network starts from the prefix slices of a supernet
    checkpoint (nets/net_utils.py:get_sub_state_dict)."""
    ck = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location='cpu', weights_only=False)
    src = ck['model_ema'] if use_ema else ck['model']
    model.load_state_dict(get_sub_state_dict(source_dict=src, sub_dict=model.state_dict()))
    return model
