from .compute_flop_mac import ComputationEstimator  # noqa: F401
