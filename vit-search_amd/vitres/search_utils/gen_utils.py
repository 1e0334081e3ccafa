"""Candidate generation for the evolutionary search: random sampling under a MAC budget, mutation, crossover.

Restates the behaviour of the reference's search_utils/gen_utils.py (file:line cited per function) on the same
`network_def` grammar and the same `num_channels_to_keep` choice tables.  The global numpy RNG is consumed in exactly the
reference's order (which draw, with which bounds, under which condition), so a seeded search visits the same candidates:
tests/golden/f13_evolver.npz holds populations produced by the imported reference and tests/test_search_utils.py replays them.

A network_def entry is addressed through the small accessors below instead of index constants; all edits happen in place on a
nested-list copy of the caller's definition.
"""
import copy

import numpy as np

T_EMBED, T_TRANS, T_HEAD, T_SR, T_CONV_EMBED, T_FLEX_CONV_EMBED = 0, 1, 2, 3, 4, 5
_EMBED_KINDS = (T_EMBED, T_CONV_EMBED, T_FLEX_CONV_EMBED)
RESOURCE_LOWER_BOUND = 0.975            # candidates must use at least this fraction of the budget (gen_utils.py:51)
_PRUNE_EVERYTHING_AFTER = 100           # prune steps after which widths and whole blocks become prunable too (:203)


def listit(t):
    """Nested tuples -> nested lists (gen_utils.py:54-55)."""
    return [listit(v) for v in t] if isinstance(t, (list, tuple)) else t


def tupleit(t):
    """Nested lists -> nested tuples (gen_utils.py:58-59)."""
    return tuple(tupleit(v) for v in t) if isinstance(t, (list, tuple)) else t


def _kind(entry):
    return entry[0]


def _as_lists(network_def):
    nd = copy.deepcopy(network_def)
    return listit(nd) if isinstance(nd, tuple) else nd


def update_embed_size(nd):
    """Propagate the embedding width through the definition: every transformer entry, the head and the input side of every
    spatial reduction take the width produced by the closest embedding / reduction before them (gen_utils.py:62-79)."""
    width = nd[0][1]
    for entry in nd[1:]:
        k = _kind(entry)
        if k == T_TRANS:
            entry[1][0] = width
            entry[2][0] = width
        elif k == T_HEAD:
            entry[1] = width
        elif k == T_SR:
            entry[1] = width
            width = entry[2]
        else:
            raise ValueError()
    return nd


def update_depth(nd, num_channels_to_keep):
    """A removable block directly after a removed removable block is removed as well; a block without a `layer` choice ends
    the run (gen_utils.py:82-108; other entry kinds leave the run untouched)."""
    in_removed_run = False
    for entry, choice in zip(nd, num_channels_to_keep):
        if _kind(entry) != T_TRANS:
            continue
        if choice['layer'] is None:
            in_removed_run = False
        elif in_removed_run:
            entry[3] = 0
        elif not entry[3]:
            in_removed_run = True
    return nd


def _first_smaller(choices, current, unit=1):
    """First listed choice (tables are sorted from large to small) whose value // unit is below `current`, else None."""
    for c in choices:
        v = int(c) // unit
        if v < current:
            return v
    return None


def prune_random_one(network_def, num_channels_to_keep, prune_embed=True, prune_block=True):
    """One random simplification step (gen_utils.py:111-183): pick an entry (never the head; embeddings and reductions only
    when prune_embed), then shrink it to the next smaller choice -- for a transformer entry pick among heads / hidden width /
    (when allowed and removable) dropping the block."""
    nd = copy.deepcopy(network_def)
    last = len(nd) - 1
    first = 0 if prune_embed else 1
    idx = np.random.randint(first, last)
    if not prune_embed:
        while _kind(nd[idx]) in _EMBED_KINDS + (T_SR,):
            idx = np.random.randint(first, last)
    entry, choice = nd[idx], num_channels_to_keep[idx]
    k = _kind(entry)
    if k in _EMBED_KINDS:
        smaller = _first_smaller(choice, entry[1])
        if smaller is not None:
            entry[1] = smaller
        update_embed_size(nd)
    elif k == T_TRANS:
        n_options = 3 if (choice['layer'] is not None and prune_block) else 2
        what = np.random.randint(n_options)
        if what == 0:
            smaller = _first_smaller(choice['attn'], entry[1][1], unit=entry[1][2])
            if smaller is not None:
                entry[1][1] = smaller
        elif what == 1:
            smaller = _first_smaller(choice['mlp'], entry[2][1])
            if smaller is not None:
                entry[2][1] = smaller
        else:
            if not int(np.random.choice(choice['layer'])):
                entry[3] = 0
                update_depth(nd, num_channels_to_keep)
    elif k == T_SR:
        smaller = _first_smaller(choice, entry[2])
        if smaller is not None:
            entry[2] = smaller
            update_embed_size(nd)
    else:
        raise ValueError()
    return nd


def reduce_constraint(network_def, num_channels_to_keep, constraint, compute_resource):
    """Prune random entries until compute_resource(network_def) <= constraint: heads and hidden widths only for the first
    100 steps, embedding widths and whole blocks as well after that (gen_utils.py:186-214).  Returns nested lists."""
    nd = listit(network_def) if isinstance(network_def, tuple) else network_def
    steps = 0
    while compute_resource(nd) > constraint:
        everything = steps >= _PRUNE_EVERYTHING_AFTER
        nd = prune_random_one(nd, num_channels_to_keep, prune_embed=everything, prune_block=everything)
        steps += 1
    return nd


def random_sample_embed_depth(largest_network_def, num_channels_to_keep):
    """Draw every embedding / reduction width and every removable block's existence; heads and hidden widths stay at their
    maximum (gen_utils.py:217-241)."""
    nd = _as_lists(largest_network_def)
    for entry, choice in zip(nd, num_channels_to_keep):
        k = _kind(entry)
        if k in _EMBED_KINDS:
            entry[1] = int(np.random.choice(choice))
            update_embed_size(nd)
        elif k == T_TRANS:
            if choice['layer'] is not None and not int(np.random.choice(choice['layer'])):
                entry[3] = 0
        elif k == T_SR:
            entry[2] = int(np.random.choice(choice))
            update_embed_size(nd)
    return update_depth(nd, num_channels_to_keep)


def gen_random_func(largest_network_def, num_channels_to_keep, constraint, compute_resource):
    """Width/depth skeletons are redrawn until one is large enough to reach the budget's lower bound, then pruned down into
    the budget (gen_utils.py:244-252)."""
    floor = RESOURCE_LOWER_BOUND * constraint
    nd = random_sample_embed_depth(largest_network_def, num_channels_to_keep)
    while compute_resource(nd) < floor:
        nd = random_sample_embed_depth(largest_network_def, num_channels_to_keep)
    nd = reduce_constraint(nd, num_channels_to_keep, constraint, compute_resource)
    return tupleit(nd) if isinstance(nd, list) else nd


def _in_budget(resource, constraint):
    return RESOURCE_LOWER_BOUND * constraint <= resource <= constraint


def gen_random_network_def(largest_network_def, num_channels_to_keep, constraint, compute_resource):
    """A random candidate with 0.975 * constraint <= resource <= constraint (gen_utils.py:255-262)."""
    while True:
        nd = gen_random_func(largest_network_def, num_channels_to_keep, constraint, compute_resource)
        if _in_budget(compute_resource(nd), constraint):
            return nd


def mutate_func(parent_network_def, num_channels_to_keep, m_prob):
    """Each searchable quantity is redrawn with probability m_prob; a removable block's existence is FLIPPED with
    probability m_prob (no draw at all for blocks that cannot be removed) (gen_utils.py:265-316)."""
    nd = _as_lists(parent_network_def)
    for entry, choice in zip(nd, num_channels_to_keep):
        k = _kind(entry)
        if k in _EMBED_KINDS:
            if np.random.uniform() <= m_prob:
                entry[1] = int(np.random.choice(choice))
                update_embed_size(nd)
        elif k == T_TRANS:
            if np.random.uniform() <= m_prob:
                entry[1][1] = int(np.random.choice(choice['attn'])) // entry[1][2]
            if np.random.uniform() <= m_prob:
                entry[2][1] = int(np.random.choice(choice['mlp']))
            if choice['layer'] is not None and np.random.uniform() <= m_prob:
                entry[3] = 0 if entry[3] else 1
                update_depth(nd, num_channels_to_keep)
        elif k == T_SR:
            if np.random.uniform() <= m_prob:
                entry[2] = int(np.random.choice(choice))
                update_embed_size(nd)
        elif k != T_HEAD:
            raise ValueError()
    return nd


def mutate_network_def(parent_network_def, num_channels_to_keep, m_prob, constraint, compute_resource):
    """Mutations of the parent are drawn until one lands inside the budget window (gen_utils.py:319-334)."""
    while True:
        nd = mutate_func(parent_network_def, num_channels_to_keep, m_prob)
        if _in_budget(compute_resource(nd), constraint):
            return tupleit(nd) if isinstance(nd, list) else nd


def crossover_func(m_network_def, f_network_def, num_channels_to_keep):
    """Child = first parent with every searchable quantity replaced by the second parent's with probability 1/2 (the
    existence draw is made for every transformer entry, removable or not) (gen_utils.py:337-377)."""
    nd = _as_lists(m_network_def)
    for i, entry in enumerate(nd):
        other = f_network_def[i]
        k = _kind(entry)
        if k in _EMBED_KINDS:
            if np.random.uniform() <= 0.5:
                entry[1] = other[1]
                update_embed_size(nd)
        elif k == T_TRANS:
            if np.random.uniform() <= 0.5:
                entry[1][1] = other[1][1]
            if np.random.uniform() <= 0.5:
                entry[2][1] = other[2][1]
            if np.random.uniform() <= 0.5:
                entry[3] = other[3]
                update_depth(nd, num_channels_to_keep)
        elif k == T_SR:
            if np.random.uniform() <= 0.5:
                entry[2] = other[2]
                update_embed_size(nd)
        elif k != T_HEAD:
            raise ValueError()
    return nd


def crossover_network_def(m_network_def, f_network_def, num_channels_to_keep, constraint, compute_resource):
    """Children of the two parents are drawn until one lands inside the budget window (gen_utils.py:380-395)."""
    while True:
        nd = crossover_func(m_network_def, f_network_def, num_channels_to_keep)
        if _in_budget(compute_resource(nd), constraint):
            return tupleit(nd) if isinstance(nd, list) else nd
