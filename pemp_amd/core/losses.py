"""Loss selection with the reference's surface (core/losses.py:8-43): ``get(cfg)`` returns a callable
``loss(logits, target) -> scalar tensor``.  On this path the cross-entropy itself is fused into the
evaluation tail / head-backward kernels, so the objects mainly carry the choice (``ce`` / ``cedt``) and
the device-side weight map; calling them evaluates the loss from full-resolution logits with the same
kernels (used by callers that hold logits, e.g. the stage-2 evaluator)."""
import torch

from .. import ops


class CELoss:
    """nn.CrossEntropyLoss(ignore_index=255) (core/losses.py:10), evaluated by pemp_eval_tail_f32."""
    kind = "ce"

    def weight_map(self, target):
        return None

    def __call__(self, logits, target):
        # the tail kernel consumes a low-res prediction; identity-size "upsampling" reuses it for given logits
        am, stats, _ = ops.eval_tail(logits.contiguous(), target.contiguous(), weight=self.weight_map(target))
        return (stats[:, 0].sum() / stats[:, 1].sum()).float()


class CELossDT(CELoss):
    """Cross-entropy weighted by exp(-EDT(boundary)/sigma^2) + 1 (core/losses.py:17-43); the distance transform
    runs on the GPU (pemp_cedt_weight_f32) instead of scipy on the host."""
    kind = "cedt"

    def __init__(self, sigma):
        self.sigma = sigma

    def weight_map(self, target):
        return ops.cedt_weight(target.contiguous(), self.sigma)


def get(cfg):
    loss = cfg["loss"] if isinstance(cfg, dict) else cfg.loss
    if loss == "ce":
        return CELoss()
    if loss == "cedt":
        return CELossDT(cfg["sigma"] if isinstance(cfg, dict) else cfg.sigma)
    raise ValueError(f"Unsupported loss type, got {loss}. Please choose from [ce, cedt]")
