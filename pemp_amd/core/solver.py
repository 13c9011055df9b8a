"""Optimizer / LR-schedule factory with the reference's config surface (core/solver.py:10-127).

``get(trainer_or_model, _config, max_steps)`` returns ``(optimizer, scheduler)`` exactly like the
reference.  The optimizer object is a real ``torch.optim.SGD`` over the model's parameters and is used
as the holder of hyper-parameters and scheduler state (``param_groups[0]["lr"]``, StepLR / MultiStepLR /
ReduceLROnPlateau / CosineAnnealingLR / PolyLR all work on it unchanged); the parameter update itself
is NOT ``optimizer.step()`` but the fused clip-norm + SGD kernel on the flat buffers
(``Stage1Trainer.optimizer_step``), which reads lr / momentum / weight decay / nesterov from it.
``opt=adam`` returns a real ``torch.optim.Adam`` whose hyper-parameters feed the fused clip + Adam kernel in the same way
(the reference's own Trainer code can also call its ``step()`` directly through ``pemp_amd.autograd``).
"""
import torch

from ..config import Ingredient

train_ingredient = Ingredient("tr", save_git_info=False)
test_ingredient = Ingredient("te", save_git_info=False)


@train_ingredient.config
def train_config():
    """ Training Arguments """
    epochs = 0                              # int, epochs already trained
    total_epochs = 3                        # int, total epochs

    lr = 1e-3                               # float, base learning rate
    lrp = "period_step"                     # str, LR policy [custom_step/period_step/plateau/cosine/poly]
    if lrp == "custom_step":
        lr_boundaries = []                  # list, [custom_step] milestones
    if lrp == "period_step":
        lr_step = 999999999                 # int, [period_step] decay period (default: never)
    if lrp in ["custom_step", "period_step", "plateau"]:
        lr_rate = 0.1                       # float, decay factor
    if lrp in ["plateau", "cosine", "poly"]:
        lr_end = 0.                         # float, minimal learning rate
    if lrp == "plateau":
        lr_patience = 30
        lr_min_delta = 1e-4
        cool_down = 0
        monitor = "val_loss"
    if lrp == "poly":
        power = 0.9

    opt = "sgd"                             # str, optimizer [sgd/adam]
    if opt == "adam":
        adam_beta1 = 0.9
        adam_beta2 = 0.999
        adam_epsilon = 1e-8
    if opt == "sgd":
        sgd_momentum = 0.9
        sgd_nesterov = False

    weight_decay = 0.0005                   # float, weight decay coefficient
    ckpt_epoch = 1                          # int, checkpoint interval, 0 disables checkpoints


@test_ingredient.config
def test_config():
    """ Testing Arguments """
    epochs = 5


class PolyLR:
    """lr = (lr0 - lr_end) * (1 - step/max_iter)^power + lr_end   (reference core/solver.py:53-72)."""

    def __init__(self, optimizer, max_iter, power=0.9, lr_end=0, last_step=0):
        self.optimizer, self.max_iter, self.power, self.lr_end = optimizer, max_iter, power, lr_end
        self.last_step = last_step
        self.init_lr = optimizer.param_groups[0]["lr"]
        self.step()

    def step(self, step=None):
        self.last_step += 1
        if step is None:
            step = self.last_step
        else:
            self.last_step = step
        # the reference's constructor already consumes one step, so its last call has step = max_iter + 1 and a negative
        # base (a complex number in Python 3); clamped here
        self.optimizer.param_groups[0]["lr"] = \
            (self.init_lr - self.lr_end) * max(1 - step / self.max_iter, 0.0) ** self.power + self.lr_end


def get(model, _config=None, max_steps=200001):
    cfg = dict(train_ingredient.cfg if _config is None else _config)
    if isinstance(model, list):
        params = model
    elif isinstance(model, torch.nn.Module):
        params = [p for p in model.parameters()]
    else:
        raise TypeError(f"`model` must be an nn.Model or a list, got {type(model)}")
    if cfg["opt"] == "sgd":
        optimizer = torch.optim.SGD(params, cfg["lr"], momentum=cfg["sgd_momentum"], weight_decay=cfg["weight_decay"],
                                    nesterov=cfg["sgd_nesterov"])
    elif cfg["opt"] == "adam":
        # reference core/solver.py:92-96; the trainers run the fused clip + Adam kernel (pemp_adam_clip_step_f32) with this
        # object's hyper-parameters, as they do for SGD
        optimizer = torch.optim.Adam(params, cfg["lr"], betas=(cfg["adam_beta1"], cfg["adam_beta2"]),
                                     eps=cfg["adam_epsilon"], weight_decay=cfg["weight_decay"])
    else:
        raise ValueError("Not supported optimizer: " + str(cfg["opt"]))
    lrp = cfg["lrp"]
    S = torch.optim.lr_scheduler
    if lrp == "period_step":
        scheduler = S.StepLR(optimizer, step_size=cfg["lr_step"], gamma=cfg["lr_rate"])
    elif lrp == "custom_step":
        scheduler = S.MultiStepLR(optimizer, milestones=cfg["lr_boundaries"], gamma=cfg["lr_rate"])
    elif lrp == "plateau":
        scheduler = S.ReduceLROnPlateau(optimizer, factor=cfg["lr_rate"], patience=cfg["lr_patience"],
                                        threshold=cfg["lr_min_delta"], cooldown=cfg["cool_down"], min_lr=cfg["lr_end"])
    elif lrp == "cosine":
        scheduler = S.CosineAnnealingLR(optimizer, T_max=max_steps, eta_min=cfg["lr_end"])
    elif lrp == "poly":
        scheduler = PolyLR(optimizer, max_iter=max_steps, power=cfg["power"], lr_end=cfg["lr_end"])
    else:
        raise ValueError
    return optimizer, scheduler
