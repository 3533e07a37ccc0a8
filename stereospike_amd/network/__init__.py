"""Drop-in for the reference's `network` package (module paths network.SNN_models / network.blocks /
network.ANN_models / network.loss / network.metrics; `stereospike_amd.install_dropin()` aliases them)."""
from .SNN_models import StereoSpike  # noqa: F401
from .ANN_models import StereoSpike_equivalentANN, SteroSpike_equivalentANN  # noqa: F401
