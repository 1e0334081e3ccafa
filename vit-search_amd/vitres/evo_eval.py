"""Evolutionary-search inner loop on resident supernet weights (BASELINE config 5).

The reference scores every candidate by building a fresh sub-network, deep-copying + prefix-slicing the supernet
state_dict into it, moving it to the GPU, DDP-wrapping it (a parameter broadcast per candidate) and running
engine.evaluate (evo_search.py:253-285, nets/net_utils.py:34-57).  A prefix-sliced sub-network is exactly the
supernet under fixed prefix masks (pinned by tests against the reference's own sliced logits),
so here a candidate is only a DESCRIPTOR: its `network_def` is turned into one keep vector per ChannelDrop and the
resident supernet is run in eval mode with those keeps -- masked K slices / output tiles / heads are skipped by the
kernels, nothing is copied or broadcast.  Multi-GPU: candidates (not data) are sharded over ranks; scores are
all-gathered once per generation.
"""
import torch
import torch.distributed as dist

from .nets.supernet_blocks import Block
from .nets.vit_sr_supernet import SpatialReductionPatchEmbedding, _Plan

_T_TRANS, _T_SR = 1, 3
_SKIP_REMOVED = __import__("os").environ.get("VITRES_SKIP_DROPPED_LAYERS", "1") != "0"


def plan_for_subnet(model, sub_network_def, batch):
    """_Plan that makes `model` (a supernet) compute the sub-network `sub_network_def` for every sample."""
    sup = model.network_def
    assert len(sub_network_def) == len(sup), 'candidate and supernet network_def must align entry by entry'
    plan = _Plan()
    plan.batch = batch

    def row(v):
        return plan.add(torch.full((batch,), int(v), dtype=torch.int64))
    sub_embed = sub_network_def[0][1]
    assert sub_embed <= sup[0][1]
    e_idx = row(sub_embed)
    plan.layers.append({"embed": e_idx})
    embed_keep = sub_embed
    layer_keep = None
    entries = [(s, c) for s, c in zip(sup, sub_network_def) if s[0] in (_T_TRANS, _T_SR)]
    assert len(entries) == len(model.blocks)
    for blk, (s, c) in zip(model.blocks, entries):
        if isinstance(blk, Block):
            assert c[0] == _T_TRANS and c[1][0] == embed_keep and c[1][2] == s[1][2], 'head_dim is not searchable'
            exists = bool(c[3])
            hd = c[1][1] * c[1][2] if exists else s[1][1] * s[1][2]
            hidden = c[2][1] if exists else s[2][1]
            assert hd <= s[1][1] * s[1][2] and hidden <= s[2][1]
            cur = embed_keep if exists else 0                    # removed block == layer keep 0 (BypassBlock)
            if layer_keep is not None and blk.layer_drop is not None:
                cur = min(cur, layer_keep)
            # (a removed block's branches are multiplied by layer keep 0: the widths the kernels read for it are 0 as well, so that its
            # qkv / fc1 GEMMs and attention cores are skipped instead of computed at the supernet's full width and discarded)
            skip = (not exists) and _SKIP_REMOVED
            plan.layers.append({"embed": e_idx, "attn": row(0 if skip else hd), "mlp": row(0 if skip else hidden), "out": row(cur), "dp": None})
            # reference semantics: BypassBlock / blocks without layer_drop reset the layer mask to the embed mask
            layer_keep = None
        elif isinstance(blk, SpatialReductionPatchEmbedding):
            assert c[0] == _T_SR and c[1] == embed_keep and c[2] <= s[2]
            n_idx = row(c[2])
            plan.layers.append({"embed": e_idx, "new": n_idx})
            embed_keep, e_idx, layer_keep = c[2], n_idx, None
        else:
            plan.layers.append(None)
            layer_keep = None
    plan.head = e_idx
    assert sub_network_def[-1][1] == embed_keep
    return plan


@torch.no_grad()
def score_candidate(model, sub_network_def, batches):
    """top-1 accuracy (percent, batch-size weighted -- engine.py:224-229) of one candidate on `batches`.  The keep
    descriptor is built once per batch size and the hit count stays on the device: one host sync per candidate (the
    reference reads `.item()` metrics every batch, engine.py:227-229)."""
    model.eval()
    correct, count, plans = None, 0, {}
    for images, target in batches:
        b = images.shape[0]
        if b not in plans:
            plans[b] = plan_for_subnet(model, sub_network_def, b)
        out = model(images, plan=plans[b])
        # two-token (distillation) models are ranked by the distillation head's accuracy (evo_search.py:280-284: dst_acc1)
        out = (out[1] if getattr(model, "num_tokens", 1) == 2 else out[0]) if isinstance(out, tuple) else out
        hits = (out.argmax(dim=1) == target).sum()
        correct = hits if correct is None else correct + hits
        count += b
    return 100.0 * float(correct) / max(count, 1) if count else 0.0


def random_candidate(network_def, num_channels_to_keep, rng):
    """One uniformly drawn sub-network of the search space (same grammar as network_def; the role of
    search_utils/gen_utils.py:gen_random_func: every stage picks an embedding width, every block its heads x head_dim,
    its MLP width and -- where the table allows 0 -- whether it exists)."""
    out, embed = [], None
    for entry, choice in zip(network_def, num_channels_to_keep):
        kind = entry[0]
        if kind in (0, 4, 5):                                    # patch embedding
            embed = int(rng.choice(choice))
            out.append((kind, embed) + tuple(entry[2:]))
        elif kind == _T_TRANS:
            d = entry[1][2]
            hd = int(rng.choice(choice['attn']))
            hidden = int(rng.choice(choice['mlp']))
            exists = 1 if choice.get('layer') is None else int(int(rng.choice(choice['layer'])) > 0)
            out.append((kind, (embed, hd // d, d), (embed, hidden), exists))
        elif kind == _T_SR:
            new = int(rng.choice(choice))
            out.append((kind, embed, new))
            embed = new
        else:                                                    # head
            out.append((kind, embed) + tuple(entry[2:]))
    return tuple(out)


@torch.no_grad()
def score_population(model, population, batches):
    """Scores for a list of candidate network_defs; candidates are sharded over ranks (round robin) and the
    scores gathered, so every rank returns the full list (the reference re-broadcasts weights per candidate)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    scores = torch.zeros(len(population), dtype=torch.float64, device=next(model.parameters()).device)
    for i in range(rank, len(population), world):
        scores[i] = score_candidate(model, population[i], batches)
    if world > 1:
        dist.all_reduce(scores)
    return scores.tolist()
