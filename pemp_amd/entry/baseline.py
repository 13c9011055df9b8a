"""Evaluation harness of the Baseline model on MI355X (counterpart of the reference's entry/baseline.py:46-51,
BASELINE.json configs[0]: baseline, VGG-16, PASCAL-5i 1-shot).  Same loop, sharding and fused tail as
stage 1; the model is ``pemp_amd.networks.baseline.Baseline`` (full-resolution masked average pooling
through the adjoint of the bilinear upsample)."""
from ..config import Experiment
from ..networks.baseline import ModelClass, net_ingredient  # noqa: F401
from .pemp_stage1 import INGREDIENTS, Evaluator, SyntheticEpisodes, eval_episodes, get_val_labels, num_classes  # noqa: F401

NAME = "Baseline"
ex = Experiment(name=NAME, ingredients=[net_ingredient] + INGREDIENTS[1:])      # own ``net`` (backbone = vgg16) + data, tr, te, g, d


@ex.config
def ex_config():
    tag = "baseline"            # str, configuration tag
    shot = 1                    # int, support samples per episode
    query = 1                   # int, query samples per episode
    split = -1                  # int, split number [0, 1, 2, 3], required
    seed = 1234                 # int, random seed
    ckpt = "bestckpt.pth"       # str, checkpoint file
    exp_id = -1                 # experiment id to load checkpoint
    loss = "ce"                 # str, loss type [ce/cedt]
    sigma = 5.                  # float, sigma of the DT loss


@ex.command
def test(_config, split, shot, exp_id, ckpt):
    import logging
    import numpy as np
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    logger = logging.getLogger(NAME)
    if split < 0:
        raise ValueError("Argument `split` is required! For example: `python -m pemp_amd.entry.baseline test with split=0`")
    from ..core.snapshots import load_for_eval
    model = ModelClass(logger)
    load_for_eval(model, _config, exp_id, ckpt, logger)          # find_snapshot + load_weights, as the reference's test
    model = model.cuda().eval()
    ev = Evaluator(model)
    d = _config["data"]
    data = eval_episodes(d, shot, split)
    loss, miou, biou = ev.start_eval_loop(data, num_classes(d["dataset"]), split, _config["te"]["epochs"], logger,
                                          batch=d["test_bs"], dataset_name=d["dataset"])
    return f"Loss: {loss:.4f}, mIoU: {np.mean(miou) * 100:.2f}, bIoU: {np.mean(biou) * 100:.2f}"


@ex.command
def train(_config, split, shot, seed, loss, sigma, exp_id):
    """Baseline training procedure (entry/baseline.py:54-62,65-110): no gradient clipping; VGG-16 or ResNet-50 per net.backbone."""
    from .pemp_stage1 import run_training

    def make_trainer(logger, dev):
        return Trainer(ModelClass(logger), lr=_config["tr"]["lr"], device=dev, loss=loss, sigma=sigma)

    return run_training(_config, NAME, make_trainer, lambda tr, dev: Evaluator(tr.model, device=dev), split, shot, seed, exp_id)




class Trainer:
    """``train_step(*inputs, qry_msk=...)`` of the reference's baseline Trainer (entry/baseline.py:54-62)."""

    def __new__(cls, model, **kw):
        from ..train_baseline import BaselineTrainer

        class _Trainer(BaselineTrainer):
            def train_step(self, *inputs, qry_msk=None):
                return super().train_step(*inputs, qry_msk=qry_msk.view(-1, *qry_msk.shape[-2:]))

        return _Trainer(model, **kw)


if __name__ == "__main__":
    print(ex.run_commandline())
