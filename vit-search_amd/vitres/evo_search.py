"""Evolutionary search over the sub-networks of a resident supernet (the loop of reference evo_search.py:220-314).

Per iteration: a generation is drawn (iteration 0: `init_popu_size` random candidates under the MAC budget; later:
`mutate_size` mutations + as many crossovers of the top `parent_size` candidates, search_utils.PopulationEvolver), every candidate
is scored, the generation is merged into the sorted history and the per-iteration result files are written with the reference's
names and text layout (iter@K/popu.txt, iter@K/history_popu_top.txt, summary.txt; pickles of the Individual lists beside them).

What differs from the reference is only HOW a candidate is scored: no sub-network is created, sliced, moved and DDP-wrapped
(evo_search.py:262-275); the candidate's network_def becomes a keep descriptor of the supernet that already sits in HBM
(evo_eval.plan_for_subnet) and, with several ranks, CANDIDATES are dealt over the ranks instead of validation data
(evo_eval.score_population; every rank draws the same generation because numpy is seeded identically, evo_search.py:172-178).
"""
import os
import pickle

import numpy as np
import torch

from . import evo_eval
from .network_utils.compute_flop_mac import ComputationEstimator
from .search_utils import PopulationEvolver


def write_results(individuals, path, item_name_list=None):
    """`Idx, Acc, Network_def` table of a list of Individuals (evo_search.py:143-157)."""
    names = item_name_list or ['Idx', 'Acc', 'Network_def']
    assert len(names) == 3
    with open(path, 'w') as f:
        f.write('{}, {}, {}\n'.format(*names))
        for i, ind in enumerate(individuals):
            f.write('{}, {}, {}\n'.format(i, ind.score, ind.network_def))


def pickle_save(obj, path):
    with open(path, 'wb') as f:
        pickle.dump(obj, f)


def _is_main():
    d = torch.distributed
    return not (d.is_available() and d.is_initialized()) or d.get_rank() == 0


def search(model, batches, network_def, num_channels_to_keep, constraint_value, search_iter=20, init_popu_size=500,
           parent_size=75, mutate_size=75, mutate_prob=0.3, input_size=224, output_dir=None, seed=None, score_fn=None,
           log=None, distill=None):
    """Run the search; returns the best Individual of every iteration (the reference's `_best_result_history`).

    model: a vitres supernet (eval()); batches: list of (images, labels) on its device, the validation subset every candidate
    is scored on; score_fn(network_defs) -> scores overrides the evaluator (tests).  Defaults are the reference's
    (evo_search.py:126-134)."""
    assert len(network_def) == len(num_channels_to_keep)
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    if distill is None:                                  # evo_search.py:201: distillation-token models count its MACs too
        distill = getattr(model, "num_tokens", 1) == 2
    compute_mac = ComputationEstimator(distill=bool(distill), input_resolution=input_size, patch_size=14)
    evolver = PopulationEvolver(largest_network_def=network_def, num_channels_to_keep=num_channels_to_keep,
                                constraint=constraint_value, compute_resource=compute_mac)
    if score_fn is None:
        def score_fn(defs):
            return evo_eval.score_population(model, defs, batches)
    best = []
    for it in range(search_iter):
        if it == 0:
            evolver.random_sample(init_popu_size)
        else:
            evolver.evolve_sample(parent_size=parent_size, mutate_prob=mutate_prob, mutate_size=mutate_size)
        scores = score_fn([ind.network_def for ind in evolver.popu])
        for ind, s in zip(evolver.popu, scores):
            ind.score = s
        write = _is_main() and output_dir
        if write:
            d = os.path.join(output_dir, 'iter@{}'.format(it))
            os.makedirs(d, exist_ok=True)
            pickle_save(evolver.popu, os.path.join(d, 'popu.pickle'))
            write_results(evolver.popu, os.path.join(d, 'popu.txt'))
        evolver.update_history()
        evolver.sort_history()
        best.append(evolver.history_popu[0])
        if log:
            log('Iter [{}]: {}'.format(it, evolver.history_popu[0]))
        if write:
            pickle_save(evolver.history_popu, os.path.join(d, 'history_popu.pickle'))
            write_results(evolver.history_popu[0:parent_size], os.path.join(d, 'history_popu_top.txt'))
            write_results(best, os.path.join(output_dir, 'summary.txt'), item_name_list=['Iter', 'Acc', 'Network_def'])
    return best
