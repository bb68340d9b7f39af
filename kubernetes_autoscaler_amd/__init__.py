"""kubernetes_autoscaler_amd — MI355X-native scale-up simulation engine for the Cluster Autoscaler.

One hot path, drop-in behind the reference's own interfaces:
  estimator.Estimator.Estimate (BinpackingNodeEstimator)  -> estimator.BinpackingNodeEstimator
  ClusterSnapshot.CheckPredicates (batched)               -> scaleup.schedulable_pod_groups
  expander.Filter.BestOptions                             -> expander.*
All compute runs in hand-written HIP kernels (libcasim.so, gfx950); there is no CPU fallback.
"""
from . import _abi  # noqa: F401
from ._ffi import CasimError, NoDeviceError, LIB_PATH  # noqa: F401
from .objects import *  # noqa: F401,F403
from .encoder import Encoder  # noqa: F401
from .engine import BatchResult, Context, MultiContext, PrefetchCache, Problem, ResidentCluster, StreamedBatch, device_count, estimate_batch  # noqa: F401
