"""Evaluation harness of PEMP stage 2 on MI355X (counterpart of the reference's
entry/pemp_stage2.py:53-65): the stage-1 model predicts the query at the input resolution, its argmax
becomes the 4th input channel ("prior") of the stage-2 encoder; loss/argmax/IoU counts come from the
fused tail.  Both models replay captured hipGraphs; nothing reaches the host per episode."""
import torch

from .. import ops
from ..core.metrics import Accumulator, FewShotMetric  # noqa: F401
from ..networks.pemp_stage2 import ModelClass, PriorNet, net_ingredient  # noqa: F401
from .pemp_stage1 import Evaluator as _Stage1Evaluator
from .pemp_stage1 import SyntheticEpisodes, allreduce_round, get_val_labels, shard_indices  # noqa: F401


class Evaluator(_Stage1Evaluator):
    """Inherits the sharded evaluation loop (``start_eval_loop``); a step is stage-1 prior -> stage 2."""

    def __init__(self, stage1, model, device=None, use_graph=True):
        super().__init__(model, device=device, use_graph=use_graph)
        self.stage1 = stage1

    def test_step_batch(self, episodes):
        return torch.cat([self.test_step_device(inputs, qry_msk)[1] for inputs, qry_msk in episodes])

    def prior(self, dev_in):
        """Stage-1 argmax at the input size as the float plane the stage-2 stem consumes."""
        H, W = dev_in[0].shape[-2:]
        pred, _ = (self.stage1.lowres_graphed(*dev_in) if self.use_graph else self.stage1.lowres(*dev_in))
        am, _, _ = ops.eval_tail(pred, None, out_hw=(H, W), ws_cache=self._ws)
        return am.unsqueeze(1).float()                                  # [BQ,1,H,W]

    def test_step_device(self, inputs, qry_msk):
        dev_in = [x.to(self.device, non_blocking=True) for x in inputs]
        tgt = qry_msk.view(-1, *qry_msk.shape[-2:]).to(self.device, non_blocking=True)
        with torch.no_grad():
            prior = self.prior(dev_in)
            if self.use_graph:
                pred, _ = self.model.lowres_graphed(*dev_in, prior)
            else:
                pred, _ = self.model.lowres(*dev_in, prior)
            am, stats, _ = ops.eval_tail(pred, tgt, ws_cache=self._ws)
        return am, stats

    def test_step(self, inputs, qry_msk, **kwargs):
        """Reference contract (entry/pemp_stage2.py:58-65): -> (qry_pred numpy [B,H,W], loss float)."""
        am, stats = self.test_step_device(inputs, qry_msk)
        st = stats.cpu().numpy()
        return am.cpu().numpy(), float(st[:, 0].sum() / max(st[:, 1].sum(), 1.0))


class Trainer:
    """``train_step(*inputs, qry_msk=...)`` of the reference's stage-2 Trainer (entry/pemp_stage2.py:67-83): the
    frozen stage-1 model gives the prior, stage 2 steps on the HIP training path (no clipping for ResNet-50)."""

    def __new__(cls, stage1, model, **kw):
        from ..train_stage2 import Stage2Trainer

        class _Trainer(Stage2Trainer):
            def train_step(self, *inputs, qry_msk=None):
                return super().train_step(*inputs, qry_msk=qry_msk.view(-1, *qry_msk.shape[-2:]))

        return _Trainer(stage1, model, **kw)
