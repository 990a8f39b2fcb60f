"""star_amd: MI355X-native implementation of STAR's nugget-compression hot path.

Only what the path needs lives here: `csrc/` (HIP kernels + the C ABI of libstar_nugget.so),
`ops.py` (tensor-level operators that launch those kernels), `nugget.py` (drop-in mirror of the
reference's operator signatures), `step.py` (the NuggetStep used by smoke/bench), `streaming.py` (chunked
inference side of the path) and `dist.py` (batch sharding + flat RCCL gradient all-reduces).
"""
from . import ops  # noqa: F401
from .nugget import (NuggetCrossAttention, NuggetOpsMixin, NuggetScorer, gather_nuggets,  # noqa: F401
                     gather_nuggets_layernorm, score_biased_attn, share_encoder_kv, share_scorer_logits,
                     defer_projection_weight_grads)

__all__ = ["ops", "NuggetScorer", "NuggetOpsMixin", "NuggetCrossAttention", "gather_nuggets",
           "gather_nuggets_layernorm", "score_biased_attn", "share_encoder_kv", "share_scorer_logits",
           "defer_projection_weight_grads"]
