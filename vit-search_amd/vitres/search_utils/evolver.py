"""Population bookkeeping of the evolutionary search (reference search_utils/evolver.py:13-118): the current generation, the
history of everything ever scored (kept sorted, best first; candidates are never evaluated twice), and the three ways to fill a
generation -- random sampling, mutation of the top parents, crossover between two of them.  numpy's global RNG is used exactly as
the reference does, so seeded runs coincide (tests/golden/f13_evolver.npz)."""
import warnings

import numpy as np

from .gen_utils import crossover_network_def, gen_random_network_def, mutate_network_def

CROSSOVER_GIVE_UP_AFTER = 100   # consecutive duplicate children after which a duplicate is accepted (evolver.py:10,103-112)


class Individual:
    """A candidate: its network_def and its score (-1 = not evaluated).  Ordered by score, equal by network_def."""

    def __init__(self, network_def, score=-1):
        self.network_def = network_def
        self.score = score

    def __lt__(self, other):
        return self.score < other.score

    def __eq__(self, other):
        return self.network_def == other.network_def

    __hash__ = None

    def __repr__(self):
        return '(network_def={}, score={})'.format(self.network_def, self.score)


class PopulationEvolver:
    def __init__(self, largest_network_def, num_channels_to_keep, constraint, compute_resource):
        self.largest_network_def = largest_network_def
        self.num_channels_to_keep = num_channels_to_keep
        self.constraint = constraint
        self.compute_resource = compute_resource
        self.popu = []
        self.history_popu = []

    def _budget(self):
        return dict(num_channels_to_keep=self.num_channels_to_keep, constraint=self.constraint,
                    compute_resource=self.compute_resource)

    def _is_new(self, ind):
        return ind not in self.popu and ind not in self.history_popu

    def random_sample(self, num_samples):
        """Fill the generation with num_samples distinct, never-seen random candidates (evolver.py:40-50)."""
        added = 0
        while added < num_samples:
            ind = Individual(gen_random_network_def(largest_network_def=self.largest_network_def, **self._budget()))
            if self._is_new(ind):
                self.popu.append(ind)
                added += 1

    def update_history(self):
        """Move the (scored) generation into the history, skipping definitions already there (evolver.py:53-59)."""
        for ind in self.popu:
            if ind not in self.history_popu:
                self.history_popu.append(ind)
        self.popu = []

    def sort_history(self):
        self.history_popu.sort(reverse=True)

    def evolve_sample(self, parent_size, mutate_prob, mutate_size, crossover_size=None):
        """Next generation = mutate_size mutations of random top-`parent_size` parents + crossover_size (default: the same
        number) children of two distinct random parents; duplicates are redrawn, except that crossover accepts one after 100
        consecutive failures (evolver.py:67-115)."""
        if self.popu:
            warnings.warn('[evolve_sample] popu is not empty.')
        if not self.history_popu:
            warnings.warn('[evolve_sample] history_popu is empty. Use update_history() before evolve_sample().')
            return
        if parent_size > len(self.history_popu):
            raise ValueError('Parent size is larger than history population size')
        self.sort_history()
        if crossover_size is None:
            crossover_size = mutate_size

        added = 0
        while added < mutate_size:
            parent = self.history_popu[np.random.randint(parent_size)].network_def
            ind = Individual(mutate_network_def(parent, m_prob=mutate_prob, **self._budget()))
            if self._is_new(ind):
                self.popu.append(ind)
                added += 1

        added, misses = 0, 0
        while added < crossover_size:
            pair = np.random.choice(range(parent_size), size=2, replace=False)
            ind = Individual(crossover_network_def(self.history_popu[pair[0]].network_def,
                                                   self.history_popu[pair[1]].network_def, **self._budget()))
            if self._is_new(ind) or misses >= CROSSOVER_GIVE_UP_AFTER:
                self.popu.append(ind)
                added += 1
                misses = 0
            else:
                misses += 1
