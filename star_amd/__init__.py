"""star_amd: MI355X-native implementation of STAR's nugget-compression hot path.

Only what the path needs lives here: `csrc/` (HIP kernels + the C ABI of libstar_nugget.so),
`ops.py` (tensor-level operators that launch those kernels), `nugget.py` (drop-in mirror of the
reference's operator signatures), `step.py` (the NuggetStep used by smoke/bench) and `dist.py`
(batch sharding + one RCCL gradient all-reduce).
"""
from . import ops  # noqa: F401
from .nugget import (NuggetCrossAttention, NuggetOpsMixin, NuggetScorer, gather_nuggets,  # noqa: F401
                     score_biased_attn, share_scorer_logits)

__all__ = ["ops", "NuggetScorer", "NuggetOpsMixin", "NuggetCrossAttention", "gather_nuggets",
           "score_biased_attn", "share_scorer_logits"]
