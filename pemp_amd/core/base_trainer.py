"""Training loop with the reference's structure (core/base_trainer.py:183-294): per epoch -- train
``data.train_n // data.bs`` steps, snapshot, evaluate (``te.epochs`` rounds), keep ``bestckpt.pth`` on the best
mIoU, log; LR policies step per iteration (cosine / poly) or per epoch (the others), as ``step_lr`` does there.

The step itself is the fused HIP trainer (``pemp_amd.train_engine.Stage1Trainer`` and friends): the optimizer
object returned by ``core.solver.get`` is attached to it as the holder of hyper-parameters and scheduler state.
Checkpoints are ``torch.save(model.state_dict())`` exactly like the reference's, under
``<g.model_dir>/<tag>/<run id>/{ckpt.pth, bestckpt.pth}`` -- loadable by the reference and vice versa.  Only rank 0
writes; every rank trains (gradient all-reduce inside the trainer) and evaluates its shard of the episodes.
"""
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from . import solver


class TrainingLoop:
    def __init__(self, cfg, trainer, evaluator, logger=None, run_id=None):
        self.cfg, self.trainer, self.evaluator, self.logger = cfg, trainer, evaluator, logger
        tr, data = cfg["tr"], cfg["data"]
        self.steps_per_epoch = max(data["train_n"] // data["bs"], 1)
        self.optimizer, self.scheduler = solver.get(trainer.model, tr, max_steps=tr["total_epochs"] * self.steps_per_epoch)
        trainer.attach_optimizer(self.optimizer)
        self.optimizer._opt_called = True      # the update runs in the fused kernel; tells torch's schedulers not to warn
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        rid = run_id if run_id is not None else time.strftime("%y%m%d-%H%M%S", time.localtime())
        self.model_dir = Path(cfg["g"]["model_dir"]) / str(cfg["tag"]) / str(rid)
        self.best_iou, self.best_epoch = -1.0, -1
        self._lr_counter = 0

    def _log(self, msg):
        if self.logger is not None and self.rank == 0:
            self.logger.info(msg)

    def step_lr(self):
        if self.scheduler is None:
            return
        self._lr_counter += 1
        if self.cfg["tr"]["lrp"] in ("cosine", "poly"):          # per iteration
            self.scheduler.step()
        elif self._lr_counter == self.steps_per_epoch:            # per epoch
            if self.cfg["tr"]["lrp"] != "plateau":               # plateau steps on the validation loss (evaluation())
                self.scheduler.step()
            self._lr_counter = 0

    def sync_buffers(self):
        """BatchNorm running statistics (and every other buffer) of the trained model: rank 0's copy to all ranks, as
        DDP's ``broadcast_buffers`` does.  Each rank updates them from its own batches; without this the sharded
        evaluation would mix slightly different models and the saved checkpoint (rank 0's) would not reproduce the
        logged validation mIoU."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            for b in self.trainer.model.buffers():
                dist.broadcast(b.data, 0)

    def _save(self, name):
        if self.rank != 0:
            return None
        self.model_dir.mkdir(parents=True, exist_ok=True)
        path = self.model_dir / name
        torch.save({k: v.detach().cpu().contiguous() for k, v in self.trainer.model.state_dict().items()}, str(path))
        return path

    def try_snapshot(self, epoch=-1, final=False):
        self.sync_buffers()
        if final:
            return self._save("ckpt.pth")
        ce = self.cfg["tr"]["ckpt_epoch"]
        if ce > 0 and epoch % ce == 0:
            return self._save("ckpt.pth")
        return None

    def evaluation(self, epoch, dataset, num_classes, split):
        self.sync_buffers()
        self.trainer.model.eval()
        d = self.cfg["data"]
        mloss, miou, biou = self.evaluator.start_eval_loop(dataset, num_classes, split, self.cfg["te"]["epochs"], None,
                                                           batch=d["test_bs"], dataset_name=d["dataset"])
        if isinstance(mloss, tuple):                 # PANet's evaluator: (loss, aux_loss) (entry/panet.py:100)
            mloss = mloss[0]
        miou_m, biou_m = float(np.mean(miou)), float(np.mean(biou))
        best = miou_m > self.best_iou
        if best:
            self.best_iou, self.best_epoch = miou_m, epoch
            self._save("bestckpt.pth")
        if self.cfg["tr"]["lrp"] == "plateau" and self.scheduler is not None:
            self.scheduler.step(mloss)
        self.trainer.model.train()
        return mloss, miou_m, biou_m, best

    def start_training_loop(self, batches_for_epoch, val_dataset, num_classes, split):
        """``batches_for_epoch(epoch)`` yields ``(inputs, qry_msk)`` for one epoch (``steps_per_epoch`` of them)."""
        history = []
        for epoch in range(1, self.cfg["tr"]["total_epochs"] + 1):
            self.trainer.model.train()
            total, calls, t0 = 0.0, 0, time.time()
            losses = []
            for inputs, qry_msk in batches_for_epoch(epoch):
                out = self.trainer.train_step(*inputs, qry_msk=qry_msk)
                losses.append(out[0] if isinstance(out, tuple) else out)      # PANet: (loss, align_loss)
                calls += 1
                self.step_lr()
            total = float(torch.stack([l.reshape(()) for l in losses]).sum().item()) if losses else 0.0
            speed = calls / max(time.time() - t0, 1e-9)
            self.try_snapshot(epoch)
            mloss, miou, biou, best = self.evaluation(epoch, val_dataset, num_classes, split)
            lr = self.optimizer.param_groups[0]["lr"]
            self._log(f"[ep {epoch}/{self.cfg['tr']['total_epochs']}] train_loss: {total / max(calls, 1):.5f} val_loss: {mloss:.5f} "
                      f"val_mIoU: {miou * 100:.2f} val_bIoU: {biou * 100:.2f} lr: {lr:g} speed: {speed:.1f} it/s" + (" (best)" if best else ""))
            history.append(dict(epoch=epoch, train_loss=total / max(calls, 1), val_loss=mloss, val_mIoU=miou, val_bIoU=biou, best=best))
        self.try_snapshot(final=True)
        return history
