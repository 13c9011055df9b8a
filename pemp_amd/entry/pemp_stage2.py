"""Evaluation harness of PEMP stage 2 on MI355X (counterpart of the reference's
entry/pemp_stage2.py:53-65): the stage-1 model predicts the query at the input resolution, its argmax
becomes the 4th input channel ("prior") of the stage-2 encoder; loss/argmax/IoU counts come from the
fused tail.  Both models replay captured hipGraphs; nothing reaches the host per episode."""
import torch

from .. import ops
from ..core.metrics import Accumulator, FewShotMetric  # noqa: F401
from ..networks.pemp_stage2 import ModelClass, PriorNet, net_ingredient  # noqa: F401
from ..config import Experiment
from .pemp_stage1 import Evaluator as _Stage1Evaluator
from .pemp_stage1 import INGREDIENTS, SyntheticEpisodes, eval_episodes, allreduce_round, get_val_labels, num_classes, shard_indices  # noqa: F401

NAME = "PEMP_Stage2"
ex = Experiment(name=NAME, ingredients=INGREDIENTS)


@ex.config
def ex_config():
    tag = "pemp_stage2"         # str, configuration tag
    shot = 1                    # int, support samples per episode
    query = 1                   # int, query samples per episode (must stay 1)
    split = -1                  # int, split number [0, 1, 2, 3], required
    seed = 1234                 # int, random seed
    ckpt = "bestckpt.pth"       # str, checkpoint file of stage 2
    exp_id = -1                 # experiment id to load checkpoint
    loss = "ce"                 # str, loss type [ce/cedt]
    sigma = 5.                  # float, sigma of the DT loss
    s1 = {"ckpt": "bestckpt.pth", "id": -1}         # checkpoint / experiment id of the stage-1 model (entry/pemp_stage2.py:39-42)
    p = {"cls": -1, "sup": "", "qry": ""}


class Evaluator(_Stage1Evaluator):
    """Inherits the sharded evaluation loop (``start_eval_loop``); a step is stage-1 prior -> stage 2."""

    def __init__(self, stage1, model, device=None, use_graph=True):
        super().__init__(model, device=device, use_graph=use_graph)
        self.stage1 = stage1

    def _lowres(self, dev_in):
        prior = self.prior(dev_in)
        return (self.model.lowres_graphed(*dev_in, prior) if self.use_graph else self.model.lowres(*dev_in, prior))[0]

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


def load_stage1(_config, s1, logger=None, device=None):
    """The frozen prior network of stage 2: ``PriorNet`` with the stage-1 checkpoint named by ``s1.id`` / ``s1.ckpt``
    (entry/pemp_stage2.py:121-124: find_snapshot(cfg, s1.id, s1.ckpt) + load_weights + maybe_fix_params(fix=True))."""
    from ..core.snapshots import load_for_eval
    stage1 = PriorNet(logger)
    load_for_eval(stage1, _config, s1["id"], s1["ckpt"], logger)
    stage1.maybe_fix_params(True)
    return (stage1.to(device) if device is not None else stage1.cuda()).eval()


@ex.command
def test(_config, split, shot, seed, exp_id, ckpt, s1):
    """``python -m pemp_amd.entry.pemp_stage2 test with split=0 shot=5``: stage-1 prior -> stage 2 on synthetic episodes."""
    import logging
    import numpy as np
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    logger = logging.getLogger(NAME)
    if split < 0:
        raise ValueError("Argument `split` is required! For example: `python -m pemp_amd.entry.pemp_stage2 test with split=0`")
    torch.manual_seed(seed)
    from ..core.snapshots import load_for_eval
    stage1 = load_stage1(_config, s1, logger)
    model = ModelClass(shot, _config["query"], logger)
    load_for_eval(model, _config, exp_id, ckpt, logger, wgen_seed=4321)      # entry/pemp_stage2.py:176-177
    model = model.cuda().eval()
    d = _config["data"]
    data = eval_episodes(d, shot, split)
    ev = Evaluator(stage1, model)
    loss, miou, biou = ev.start_eval_loop(data, num_classes(d["dataset"]), split, _config["te"]["epochs"], logger,
                                          batch=d["test_bs"], dataset_name=d["dataset"])
    return f"Loss: {loss:.4f}, mIoU: {np.mean(miou) * 100:.2f}, bIoU: {np.mean(biou) * 100:.2f}"


@ex.command
def train(_config, split, shot, seed, loss, sigma, exp_id, s1):
    """Stage-2 training procedure (entry/pemp_stage2.py:67-83,86-140): the stage-1 model is loaded from ``s1.id`` /
    ``s1.ckpt``, frozen (``maybe_fix_params(True)``, :121-124) and only provides the prior; clipping only for VGG (never here)."""
    from .pemp_stage1 import run_training
    holder = {}

    def make_trainer(logger, dev):
        stage1 = load_stage1(_config, s1, logger, dev)
        holder["s1"] = stage1
        return Trainer(stage1, ModelClass(shot, _config["query"], logger), lr=_config["tr"]["lr"], device=dev, loss=loss, sigma=sigma)

    return run_training(_config, NAME, make_trainer, lambda tr, dev: Evaluator(holder["s1"], tr.model, device=dev),
                        split, shot, seed, exp_id)


if __name__ == "__main__":
    print(ex.run_commandline())
