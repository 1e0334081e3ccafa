"""Host side of the evolutionary search around the resident-supernet evaluator."""
from .evolver import Individual, PopulationEvolver  # noqa: F401
from .gen_utils import (crossover_network_def, gen_random_network_def, listit, mutate_network_def,  # noqa: F401
                        reduce_constraint, tupleit)
