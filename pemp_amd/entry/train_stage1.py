"""Training harness (stage 1 by default; ``model=baseline|baseline_rn50|stage2`` for the others) on MI355X
(counterpart of the reference's
entry/pemp_stage1.py:57-65,68-113 and core/base_trainer.py:183-210 on synthetic episodes).

One process per GPU (``torchrun --nproc-per-node N -m pemp_amd.entry.train_stage1 ...``): every rank
draws its own ``bs`` episodes per step (distinct sampler offsets), BatchNorm uses the rank-local batch
statistics exactly as the reference's single-GPU batch does, gradients are SUM-all-reduced in one
flat 47.8 MB bucket over RCCL and averaged inside the fused clip-norm + SGD kernel, so every rank
holds identical weights after every step.
"""
import os
import time

import torch
import torch.distributed as dist

from .. import synth
from ..networks.pemp_stage1 import ModelClass
from ..train_engine import Stage1Trainer


class Trainer(Stage1Trainer):
    """``train_step(*inputs, qry_msk=...)`` exactly as the reference's Trainer (entry/pemp_stage1.py:57-65)."""

    def train_step(self, *inputs, qry_msk=None):
        return super().train_step(*inputs, qry_msk=qry_msk.view(-1, *qry_msk.shape[-2:]))


def synthetic_batches(bs, shot, n_steps, seed, rank, height=401, width=401):
    for step in range(n_steps):
        seeds = [seed + (step * 1000003 + rank * 7919 + i) % 2 ** 30 for i in range(bs)]
        b = synth.make_batch(seeds, shot=shot, height=height, width=width, out_hw=(height, width))
        yield (torch.from_numpy(b["sup_img"]), torch.from_numpy(b["sup_mask"]), torch.from_numpy(b["qry_img"])), \
            torch.from_numpy(b["qry_mask"])


def decoded_batches(bs, shot, n_steps, seed, rank, height=401, width=401):
    """Training episodes as the loader holds them right after ``Image.open`` (uint8 images / label images at their
    own sizes) plus the reference's augmentation draws (random scale, flip, ColorJitter, crop_obj window;
    data_kits/pascal_voc.py:194-226) from a per-rank Python ``random`` stream.  Yields Sample lists for
    ``pemp_amd.data_kits.episode.EpisodeLoader``: every pixel operation then runs on the device."""
    import random
    from ..data_kits import synth_u8
    from ..data_kits.episode import train_samples
    sizes = ((375, 500), (333, 500), (500, 375), (366, 500), (457, 500))
    rng = random.Random(seed * 7919 + rank)
    for step in range(n_steps):
        samples = []
        for i in range(bs):
            s = seed + (step * 1000003 + rank * 7919 + i) % 2 ** 30
            hs, ws = sizes[s % len(sizes)]
            hs, ws = max(hs, height), max(ws, width)          # the reference's images are at least as large as its crops
            pairs = [(synth_u8.image(s * 8 + k, hs, ws), synth_u8.mask(s * 8 + k, hs, ws)) for k in range(shot + 1)]
            samples += train_samples(pairs[:shot], pairs[shot:], height, width, rng)
        yield samples


def device_batches(loader, bs, shot, height, width):
    """EpisodeLoader outputs -> ((sup_img, sup_mask, qry_img), qry_mask) in the reference's batch layout, on the device."""
    for img, planes, labels in loader:
        img = img.view(bs, shot + 1, 3, height, width)
        yield (img[:, :shot].contiguous(), planes.view(bs, shot, 2, height, width), img[:, shot:].contiguous()), \
            torch.stack(labels).view(bs, 1, height, width)


def broadcast_model(model, src=0):
    """Identical start on every rank (parameters and BN buffers)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src)


def build_trainer(model_name, shot, lr, dev):
    """model_name: stage1 | baseline | baseline_rn50 | stage2 (stage 2 trains against a frozen stage-1 model)."""
    if model_name == "stage1":
        return Trainer(ModelClass(None), lr=lr, device=dev)
    if model_name in ("baseline", "baseline_rn50"):
        from ..networks import baseline as mb
        from . import baseline as eb
        return eb.Trainer(mb.Baseline(None, backbone="vgg16" if model_name == "baseline" else "resnet50"), lr=lr, device=dev)
    if model_name == "stage2":
        from ..networks import pemp_stage2 as m2
        from . import pemp_stage2 as e2
        return e2.Trainer(ModelClass(None).to(dev).eval(), m2.ModelClass(shot, 1, None), lr=lr, device=dev)
    raise ValueError(f"unknown model {model_name!r}")


def main(steps=20, bs=4, shot=1, lr=1e-3, seed=1234, log_every=5, model="stage1", decoded=0, height=401, width=401):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(seed + rank)
    trainer = build_trainer(model, shot, lr, dev)
    model = trainer.model
    broadcast_model(model)
    if getattr(trainer, "stage1", None) is not None:
        broadcast_model(trainer.stage1)
    if decoded:      # uint8 "decoded" episodes -> device-side resize / jitter / flip / crop / normalise, prefetched
        from ..data_kits.episode import EpisodeLoader, EpisodeTransform
        loader = EpisodeLoader(decoded_batches(bs, shot, steps, seed, rank, height, width), EpisodeTransform(height, width, device=dev))
        batches = device_batches(loader, bs, shot, height, width)
    else:
        batches = synthetic_batches(bs, shot, steps, seed, rank, height, width)
    t0 = time.time()
    for i, (inputs, qry_msk) in enumerate(batches):
        loss = trainer.train_step(*inputs, qry_msk=qry_msk)
        if rank == 0 and (i + 1) % log_every == 0:
            print(f"step {i + 1}/{steps} loss {loss.item():.5f} |g| {trainer.last_grad_norm.item():.4f} "
                  f"{(i + 1) * bs * world / (time.time() - t0):.1f} episodes/s")
    return model


if __name__ == "__main__":
    import sys
    kw = dict(a.split("=") for a in sys.argv[1:])
    main(**{k: (v if k == "model" else float(v) if "." in v or "e" in v else int(v)) for k, v in kw.items()})
