"""PANet harness on MI355X (counterpart of the reference's entry/panet.py): the Baseline's evaluation loop plus the
auxiliary prototype-alignment loss -- ``Evaluator.test_step`` -> (qry_pred, loss, aux_loss) (entry/panet.py:51-57),
``start_eval_loop`` -> ((loss, aux_loss), mIoU, bIoU) (:59-100), ``Trainer.train_step`` -> (loss, aux_loss) with
``loss + loss_coef * aux_loss`` back-propagated (:103-110).  Model: ``pemp_amd.networks.panet.PANet``."""
import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..config import Experiment
from ..networks.panet import ModelClass, align_forward, net_ingredient  # noqa: F401
from .pemp_stage1 import INGREDIENTS, SyntheticEpisodes, eval_episodes, get_val_labels, num_classes  # noqa: F401
from .pemp_stage1 import Evaluator as _Evaluator

NAME = "PEMP"
ex = Experiment(name=NAME, ingredients=[net_ingredient] + INGREDIENTS[1:])      # own ``net`` (backbone = vgg16) + data, tr, te, g, d


@ex.config
def ex_config():
    tag = "panet"               # str, configuration tag
    shot = 1                    # int, support samples per episode
    query = 1                   # int, query samples per episode
    split = -1                  # int, split number [0, 1, 2, 3], required
    seed = 1234                 # int, random seed
    ckpt = "bestckpt.pth"       # str, checkpoint file
    exp_id = -1                 # experiment id to load checkpoint
    loss = "ce"                 # str, loss type [ce/cedt]
    sigma = 5.                  # float, sigma of the DT loss
    loss_coef = 1.              # float, coefficient of the auxiliary loss
    p = {"cls": -1, "sup": "", "qry": ""}


class Evaluator(_Evaluator):
    """The stage-1 evaluator (batched, sharded, fused tail) + the alignment loss of every episode."""

    def __init__(self, model, device=None, use_graph=True):
        super().__init__(model, device, use_graph)
        self._aux, self._align_ws = [], {}

    def _lowres(self, dev_in):
        pred = super()._lowres(dev_in)
        sup_img, sup_mask, qry_img = dev_in
        B, S = sup_img.shape[:2]
        al = align_forward(self.model.__dict__["_last_feats"], pred, sup_mask.float(), B, S, qry_img.shape[1],
                           net_ingredient.cfg["dist_scalar"], self._align_ws)
        st = al["stats"].view(B, S, 8)
        self._aux.append(st[:, :, 0].sum(dim=1) / st[:, :, 1].sum(dim=1))       # per-episode mean CE of the branch
        return pred

    def test_step_device(self, inputs, qry_msk):
        dev_in = [x.to(self.device, non_blocking=True) for x in inputs]
        tgt = qry_msk.view(-1, *qry_msk.shape[-2:]).to(self.device, non_blocking=True)
        with torch.no_grad():
            pred = self._lowres(dev_in)
            am, stats, _ = ops.eval_tail(pred, tgt, ws_cache=self._ws)
        return am, stats

    def test_step(self, inputs, qry_msk, **kwargs):
        """-> (qry_pred numpy [B,H,W], loss float, aux_loss float)  (entry/panet.py:51-57)."""
        self._aux = []
        pred, loss = super().test_step(inputs, qry_msk)
        return pred, loss, float(torch.cat(self._aux).mean().item())

    def start_eval_loop(self, dataset, num_classes, split, te_epochs=5, logger=None, batch=1, dataset_name="PASCAL"):
        self._aux = []
        loss, miou, biou = super().start_eval_loop(dataset, num_classes, split, te_epochs, logger, batch, dataset_name)
        tot = torch.stack([torch.cat(self._aux).double().sum(), torch.tensor(float(sum(a.numel() for a in self._aux)),
                                                                            dtype=torch.float64, device=self.device)])
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(tot)
        self._aux = []
        return (loss, float((tot[0] / tot[1].clamp(min=1.0)).item())), miou, biou


class Trainer:
    """``train_step(*inputs, qry_msk=...)`` -> (loss, aux_loss) of the reference's PANet Trainer (entry/panet.py:103-110)."""

    def __new__(cls, model, **kw):
        from ..train_baseline import PANetTrainer

        class _Trainer(PANetTrainer):
            def train_step(self, *inputs, qry_msk=None):
                return super().train_step(*inputs, qry_msk=qry_msk.view(-1, *qry_msk.shape[-2:]))

        return _Trainer(model, **kw)


@ex.command
def test(_config, split, shot, exp_id, ckpt):
    import logging
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    logger = logging.getLogger(NAME)
    if split < 0:
        raise ValueError("Argument `split` is required! For example: `python -m pemp_amd.entry.panet test with split=0`")
    from ..core.snapshots import load_for_eval
    model = ModelClass(logger)
    load_for_eval(model, _config, exp_id, ckpt, logger)          # find_snapshot + load_weights, as the reference's test
    model = model.cuda().eval()
    ev = Evaluator(model)
    d = _config["data"]
    data = eval_episodes(d, shot, split)
    (loss, aux), miou, biou = ev.start_eval_loop(data, num_classes(d["dataset"]), split, _config["te"]["epochs"],
                                                 logger, batch=d["test_bs"], dataset_name=d["dataset"])
    return f"Loss: {loss:.4f}, Aux loss: {aux:.4f}, mIoU: {np.mean(miou) * 100:.2f}, bIoU: {np.mean(biou) * 100:.2f}"


@ex.command
def train(_config, split, shot, seed, loss, sigma, exp_id, loss_coef):
    """PANet training procedure (entry/panet.py:103-146): loss + loss_coef * align_loss, no gradient clipping."""
    from .pemp_stage1 import run_training

    def make_trainer(logger, dev):
        return Trainer(ModelClass(logger), lr=_config["tr"]["lr"], device=dev, loss=loss, sigma=sigma, loss_coef=loss_coef)

    return run_training(_config, NAME, make_trainer, lambda tr, dev: Evaluator(tr.model, device=dev), split, shot, seed, exp_id)


if __name__ == "__main__":
    print(ex.run_commandline())
