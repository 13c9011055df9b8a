"""Evaluation harness of PEMP stage 1 on MI355X (counterpart of the reference's
entry/pemp_stage1.py:47-53,116-167 and core/base_trainer.py:59-102).

``Evaluator.test_step`` keeps the reference contract -- forward at the ground truth's size,
cross-entropy, argmax -- but the whole tail (upsample + argmax + CE + tp/fp/fn) is one fused
launch and nothing has to reach the host per episode: ``test_step_device`` leaves the
prediction and an 8-double statistics row on the GPU, and ``start_eval_loop`` fetches the rows
once per round.  Episodes shard over ranks as ``tasks[rank::world]``; the integer metric table
and the loss sum are all-reduced (SUM) once per round over RCCL, so mIoU does not depend on the
number of GPUs.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import ops, synth
from ..config import Experiment, device_ingredient, global_ingredient
from ..core.metrics import Accumulator, FewShotMetric
from ..core.solver import test_ingredient, train_ingredient
from ..data_kits.datasets import PASCAL_CLASSES, data_ingredient, get_class_name, get_val_labels, num_classes  # noqa: F401
from ..networks.pemp_stage1 import ModelClass, net_ingredient

NAME = "PEMP"
#: the reference's ingredient set (entry/pemp_stage1.py:20-24): net, data, tr, te + global g / device d
INGREDIENTS = [net_ingredient, data_ingredient, train_ingredient, test_ingredient, global_ingredient, device_ingredient]
ex = Experiment(name=NAME, ingredients=INGREDIENTS)


@ex.config
def ex_config():
    tag = "pemp_stage1"         # str, configuration tag
    shot = 1                    # int, support samples per episode
    query = 1                   # int, query samples per episode (must stay 1)
    split = -1                  # int, split number [0, 1, 2, 3], required
    seed = 1234                 # int, random seed
    ckpt = "bestckpt.pth"       # str, checkpoint file
    exp_id = -1                 # experiment id to load checkpoint
    loss = "ce"                 # str, loss type [ce/cedt]
    sigma = 5.                  # float, sigma of the DT loss
    p = {"cls": -1, "sup": "", "qry": ""}       # visualize one chosen episode (class id, support / query sample names)


class SyntheticEpisodes:
    """Stand-in for the PASCAL-5i test loader (no dataset on either box): episode ``i`` of a round is
    ``E(test_seed + round * test_n + i)`` (pemp_amd.synth), yielded in the reference's batch layout
    ``((sup_img, sup_mask, qry_img), qry_mask, cls)`` with a leading batch dim of 1."""

    def __init__(self, test_n, test_seed, shot, split=0, height=401, width=401, dataset="PASCAL"):
        self.test_n, self.test_seed, self.shot, self.split = test_n, test_seed, shot, split
        self.height, self.width, self.dataset = height, width, dataset
        synth.val_labels(max(split, 0), dataset)          # unknown dataset names fail here, like datasets.py:83-104
        self.round = -1

    def reset_sampler(self):
        self.round = -1

    def sample_tasks(self):
        self.round += 1

    def __len__(self):
        return self.test_n

    def task(self, i):
        seed = self.test_seed + self.round * self.test_n + i
        ep = synth.make_episode(seed, self.shot, self.height, self.width, index=i, split=self.split, dataset=self.dataset)
        t = lambda a: torch.from_numpy(a)[None]
        return (t(ep["sup_img"]), t(ep["sup_mask"]), t(ep["qry_img"])), t(ep["qry_mask"]), torch.tensor([ep["cls"]])


class SyntheticDecodedEpisodes(SyntheticEpisodes):
    """Same protocol, but an episode is what the reference's loader holds right after ``Image.open``: uint8
    HWC images and uint8 {0,255} label images at their own sizes (pemp_amd.data_kits.synth_u8).  The
    evaluator then runs resize / normalise / mask planes on the device (pemp_amd.data_kits.episode)."""
    SIZES = ((375, 500), (333, 500), (500, 375), (366, 500), (457, 500))
    SIZES_COCO = ((480, 640), (640, 480), (427, 640), (640, 640), (375, 500), (640, 427), (512, 640))

    def decoded_task(self, i):
        from ..data_kits import synth_u8
        seed = self.test_seed + self.round * self.test_n + i
        sizes = self.SIZES if self.dataset == "PASCAL" else self.SIZES_COCO
        hs, ws = sizes[seed % len(sizes)]
        pairs = [(synth_u8.image(seed * 8 + k, hs, ws), synth_u8.mask(seed * 8 + k, hs, ws)) for k in range(self.shot + 1)]
        labels = get_val_labels(max(self.split, 0), self.dataset)
        cls = labels[seed % len(labels)]
        return pairs[:self.shot], pairs[self.shot:], cls


def eval_episodes(dcfg, shot, split, decoded=False):
    """The evaluation episode source of a command.  ``data.base_dir`` set: the PASCAL-5i directory it names, read as the
    reference reads it (data_kits/pascal_voc.py: class lists, deterministic task sampler; decoded uint8 episodes go through the
    device-side preprocessing).  Empty (the default: there is no dataset on either box): synthetic ``E(seed)`` episodes."""
    if dcfg.get("base_dir"):
        from ..data_kits.pascal_voc import load
        return load(dcfg, "test", split, shot, one_cls=dcfg.get("one_cls", 0))[0]
    cls = SyntheticDecodedEpisodes if decoded else SyntheticEpisodes
    return cls(dcfg["test_n"], dcfg["test_seed"], shot, split, dcfg["height"], dcfg["width"], dcfg["dataset"])


def train_batches(dcfg, shot, split, rank, steps_per_epoch, device):
    """-> ``batches(epoch)`` for ``TrainingLoop.start_training_loop``.  From ``data.base_dir`` when set: every epoch samples its
    tasks with the reference's sampler (rank r of a data-parallel job seeds it with ``data.seed + r``: the reference has one
    process), shuffles them, draws the augmentation on the host and preprocesses on the device, the next batch staged on a
    side stream; synthetic batches otherwise."""
    from .train_stage1 import device_batches, synthetic_batches
    if not dcfg.get("base_dir"):
        return lambda epoch: synthetic_batches(dcfg["bs"], shot, steps_per_epoch, dcfg["seed"] + 7919 * epoch, rank, dcfg["height"], dcfg["width"])
    import random
    from ..data_kits.episode import EpisodeLoader, EpisodeTransform
    from ..data_kits.pascal_voc import load
    ds = load(dict(dcfg, seed=dcfg["seed"] + rank), "train", split, shot, one_cls=dcfg.get("one_cls", 0))[0]
    rng = random.Random(dcfg["seed"] * 7919 + rank)
    tf = EpisodeTransform(dcfg["height"], dcfg["width"], mean=tuple(dcfg["mean"]), std=tuple(dcfg["std"]), device=device)

    def batches(epoch):
        ds.sample_tasks()
        return device_batches(EpisodeLoader(ds.train_batches(dcfg["bs"], rng), tf), dcfg["bs"], shot, dcfg["height"], dcfg["width"])
    return batches


def shard_indices(n, rank, world):
    """Episode indices of one rank: every rank builds the same task list and takes tasks[rank::world]."""
    return range(rank, n, world)


def allreduce_round(stat, loss_sum, count, device=None):
    """SUM over ranks of the (integer-valued) tp/fp/fn table, the per-episode loss sum and the episode
    count -- the only collective of the evaluation path.  Works on any backend (RCCL on GPU, gloo on CPU)."""
    pack = torch.from_numpy(np.concatenate([np.asarray(stat, np.float64).reshape(-1), [loss_sum, float(count)]]))
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and dist.get_backend() == "nccl":
            pack = pack.to(device)
        dist.all_reduce(pack, op=dist.ReduceOp.SUM)
        pack = pack.cpu()
    return pack[:-2].numpy().reshape(np.shape(stat)), float(pack[-2]), float(pack[-1])


class DeviceRoundTable:
    """The metric table of ONE evaluation round (reference core/base_trainer.py:70-100: ``FewShotMetric.update`` per episode,
    mIoU per round) kept where the statistics rows are produced: the rows of the fused tail (``pemp_eval_tail``: loss sum,
    pixel count, tp/fp/fn of background and foreground) are folded into a ``[C+1, 3]`` float64 table by the episode's class,
    the table + loss sum + episode count travel as ONE vector through ONE ``all_reduce(SUM)`` as it lies (RCCL on device
    memory, gloo on host memory; no staging copy) and reach the host once per round.  Every entry of the table is an integer
    below 2^53, so the sums -- on the device and across ranks -- are exact in any order and the mIoU does not depend on the
    number of ranks."""

    def __init__(self, num_classes, device):
        self.rows = num_classes + 1
        self.pack = torch.zeros(self.rows * 3 + 2, dtype=torch.float64, device=device)

    def reset(self):
        self.pack.zero_()

    def add(self, stats, classes):
        """``stats`` f64 [n, 8] (rows of the tail kernel), ``classes`` int64 [n] on the same device."""
        table = self.pack[:self.rows * 3].view(self.rows, 3)
        table[0] += stats[:, 2:5].sum(dim=0)
        table.index_add_(0, classes, stats[:, 5:8])
        self.pack[-2] += (stats[:, 0] / stats[:, 1].clamp_min(1.0)).sum()      # mean over episodes of the episode's CE mean
        self.pack[-1] += float(stats.shape[0])

    def allreduce(self):
        """The round's only collective; returns the number of bytes reduced (0 without a process group)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.pack, op=dist.ReduceOp.SUM)
            return self.pack.numel() * self.pack.element_size()
        return 0

    def fetch(self):
        """-> (stat [C+1, 3] float64 ndarray, loss sum, episode count) on the host (one synchronising copy)."""
        pack = self.pack.cpu().numpy()
        return pack[:-2].reshape(self.rows, 3).copy(), float(pack[-2]), float(pack[-1])


class Evaluator:
    """``lanes`` > 1: single-episode steps (the reference's test_bs = 1 protocol) are issued round-robin over that many
    engine replicas, each on its own HIP stream -- a one-episode step leaves most of the 256 CUs idle (M = 5202 rows
    give 1.3 waves per SIMD), several in flight fill them.  Per-episode results do not change (same kernels, same
    operands); ``test_steps_device`` is the entry point that overlaps them."""

    def __init__(self, model, device=None, use_graph=True, lanes=1, splitk=None):
        self.model = model
        #: True: one-episode steps may run the split-K conv variants (faster, equal to the exact path to rounding; see
        #: pemp_amd.ops.EVAL_SPLITK); None: the process-wide setting (off unless PEMP_EVAL_SPLITK=1)
        self.splitk = splitk
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.use_graph = use_graph
        self._ws = {}
        self.lanes = max(1, int(lanes))
        self._lane_streams = [torch.cuda.Stream(device=self.device) for _ in range(self.lanes)] if self.lanes > 1 else []
        self._lane_ws = [{} for _ in range(self.lanes)]

    def test_step_device(self, inputs, qry_msk):
        """-> (argmax uint8 [B,Ho,Wo], stats f64 [B,8]) both on the GPU, no host synchronisation."""
        tgt = qry_msk.view(-1, *qry_msk.shape[-2:]).to(self.device, non_blocking=True)
        with torch.no_grad():
            if self.use_graph and type(self)._lowres is Evaluator._lowres and not any(x.is_cuda for x in inputs):
                # host tensors (the reference's test_step body): straight into the graph's static input buffers
                with ops.eval_splitk(ops.EVAL_SPLITK if self.splitk is None else self.splitk):
                    pred = self.model.lowres_graphed(*inputs, device=self.device)[0]
            else:
                pred = self._lowres([x.to(self.device, non_blocking=True) for x in inputs])
            am, stats, _ = ops.eval_tail(pred, tgt, ws_cache=self._ws)
        return am, stats

    def test_steps_device(self, episodes):
        """``episodes``: list of (inputs, qry_msk), ONE episode each, evaluated one per step in the given order (the
        reference protocol).  -> stats f64 [len(episodes), 8] on the GPU, no host synchronisation.  With ``lanes`` > 1
        step i runs on lane i % lanes (own stream, own engine replica and graphs); the caller's stream waits for all
        lanes before the result is used."""
        n = len(episodes)
        rows = torch.empty((n, 8), dtype=torch.float64, device=self.device)
        if self.lanes == 1:
            for i, (inputs, qry_msk) in enumerate(episodes):
                rows[i:i + 1].copy_(self.test_step_device(inputs, qry_msk)[1])
            return rows
        cur = torch.cuda.current_stream()
        for s in self._lane_streams:
            s.wait_stream(cur)
        for i, (inputs, qry_msk) in enumerate(episodes):
            k = i % self.lanes
            with torch.cuda.stream(self._lane_streams[k]), self.model.lane(k), torch.no_grad():
                dev_in = [x.to(self.device, non_blocking=True) for x in inputs]
                tgt = qry_msk.view(-1, *qry_msk.shape[-2:]).to(self.device, non_blocking=True)
                pred = self._lowres(dev_in)
                _, stats, _ = ops.eval_tail(pred, tgt, ws_cache=self._lane_ws[k])
                rows[i:i + 1].copy_(stats)
        for s in self._lane_streams:
            cur.wait_stream(s)
        return rows

    def test_step_batch(self, episodes):
        """``episodes``: list of (inputs, qry_msk) with one episode each (any query-label sizes).  One encoder + head
        pass for all of them, then one fused tail launch per distinct label size.  -> stats f64 [len(episodes), 8] on
        the GPU, rows in the given order.  Per-episode results are identical to ``test_step_device`` on each episode
        alone (every image's rows are independent in the conv GEMMs and all kernel variants are bit-identical)."""
        dev_in = [torch.cat([ep[0][k].to(self.device, non_blocking=True) for ep in episodes]) for k in range(3)]
        labels = [ep[1].view(-1, *ep[1].shape[-2:]).to(self.device, non_blocking=True) for ep in episodes]
        with torch.no_grad():
            pred = self._lowres(dev_in)
            stats = torch.empty((len(episodes), 8), dtype=torch.float64, device=self.device)
            by_size = {}
            for i, lab in enumerate(labels):
                by_size.setdefault(tuple(lab.shape[-2:]), []).append(i)
            for idx in by_size.values():
                sel = torch.tensor(idx, device=self.device)
                _, st, _ = ops.eval_tail(pred.index_select(0, sel), torch.cat([labels[i] for i in idx]), ws_cache=self._ws)
                stats.index_copy_(0, sel, st)
        return stats

    def _lowres(self, dev_in):
        """Feature-resolution prediction of a batch of episodes (stage 2 overrides: stage-1 prior first)."""
        if self.splitk is None:
            return (self.model.lowres_graphed(*dev_in) if self.use_graph else self.model.lowres(*dev_in))[0]
        with ops.eval_splitk(self.splitk):
            return (self.model.lowres_graphed(*dev_in) if self.use_graph else self.model.lowres(*dev_in))[0]

    def test_step(self, inputs, qry_msk, **kwargs):
        """Reference contract (entry/pemp_stage1.py:48-53): -> (qry_pred numpy [B,H,W], loss float)."""
        am, stats = self.test_step_device(inputs, qry_msk)
        # both results through pinned buffers, ONE host synchronisation (the reference's body has two: loss.item(), .cpu())
        key = ("host", tuple(am.shape), tuple(stats.shape))
        host = self.__dict__.setdefault("_host_bufs", {})
        bufs = host.get(key)
        if bufs is None:
            bufs = host[key] = (torch.empty(am.shape, dtype=am.dtype, pin_memory=True), torch.empty(stats.shape, dtype=stats.dtype, pin_memory=True))
        bufs[0].copy_(am, non_blocking=True)
        bufs[1].copy_(stats, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        st = bufs[1].numpy()
        loss = float(st[:, 0].sum() / max(st[:, 1].sum(), 1.0))
        return bufs[0].numpy().copy(), loss

    def _episodes(self, dataset, indices):
        """(inputs, qry_msk, cls) per episode.  Datasets with ``decoded_task`` (uint8 sources) go through the
        double-buffered device-side input pipeline: episode i+1 is uploaded and preprocessed on a side stream
        while episode i is evaluated."""
        if not hasattr(dataset, "decoded_task"):
            for i in indices:
                yield dataset.task(i)
            return
        from ..data_kits.episode import EpisodeLoader, EpisodeTransform, test_samples
        H, W, S = dataset.height, dataset.width, dataset.shot
        classes = []

        def batches():
            for i in indices:
                sup, qry, cls = dataset.decoded_task(i)
                classes.append(cls)
                yield test_samples(sup, qry, H, W)

        for k, (img, planes, labels) in enumerate(EpisodeLoader(batches(), EpisodeTransform(H, W, device=self.device))):
            yield (img[:S][None], planes[None], img[S:][None]), labels[0][None, None], torch.tensor([classes[k]])

    def start_eval_loop(self, dataset, num_classes, split, te_epochs=5, logger=None, batch=1, dataset_name="PASCAL"):
        """Reference loop (core/base_trainer.py:59-102), sharded over ranks.  ``batch`` > 1 evaluates that many
        episodes per step (the reference uses test_bs = 1, data_kits/datasets.py:23); the metrics are identical."""
        self.model.eval()
        dataset.reset_sampler()
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        accum = Accumulator(loss=[], miou=[], biou=[])
        val_labels = get_val_labels(split, dataset_name)
        table = DeviceRoundTable(num_classes, self.device)
        timed, calls = 0.0, 0
        for epoch in range(1, te_epochs + 1):
            metric = FewShotMetric(num_classes)
            dataset.sample_tasks()
            rows, classes = [], []
            group = []
            pending = []
            for inputs, qry_msk, cls in self._episodes(dataset, shard_indices(len(dataset), rank, world)):
                classes += [int(c) for c in cls]
                if batch == 1 and self.lanes > 1:          # one episode per step, several steps in flight
                    pending.append((inputs, qry_msk))
                    if len(pending) == 4 * self.lanes:
                        t0 = time.time()
                        rows.append(self.test_steps_device(pending))
                        timed += time.time() - t0
                        calls += len(pending)
                        pending = []
                    continue
                if batch > 1:
                    group.append((inputs, qry_msk))
                    if len(group) < batch:
                        continue
                t0 = time.time()
                stats = self.test_step_batch(group) if batch > 1 else self.test_step_device(inputs, qry_msk)[1]
                timed += time.time() - t0
                calls += max(len(group), 1)
                group = []
                rows.append(stats)
            if group:
                t0 = time.time()
                rows.append(self.test_step_batch(group))
                timed += time.time() - t0
                calls += len(group)
            if pending:
                t0 = time.time()
                rows.append(self.test_steps_device(pending))
                timed += time.time() - t0
                calls += len(pending)
            # the round's aggregation stays on the device: rows -> [C+1, 3] table by class, ONE all-reduce, one fetch
            # (per-episode losses are averaged like the reference: mean over episodes of the episode's CE mean)
            t0 = time.time()
            table.reset()
            if rows:
                table.add(torch.cat(rows), torch.tensor(classes, dtype=torch.int64, device=self.device))
            table.allreduce()
            metric.stat, loss_tot, n_tot = table.fetch()
            timed += time.time() - t0
            miou_c, miou = metric.mIoU(val_labels)
            biou_c, biou = metric.mIoU(val_labels, binary=True)
            if logger is not None and rank == 0:
                logger.info(f"[round {epoch}/{te_epochs}] mIoU: {miou * 100:5.2f}  |  bIoU: {biou * 100:5.2f}")
            accum.update(loss=loss_tot / max(n_tot, 1.0), miou=miou_c, biou=biou_c)
        self.cps = calls / timed if timed > 0 else 0.0
        #: per-class IoU of every round, [te_epochs, len(val_labels)] / [te_epochs, 2] (the return value averages them)
        self.round_miou, self.round_biou = np.array(accum.values["miou"]), np.array(accum.values["biou"])
        return accum.mean(["loss", "miou", "biou"])


@ex.command
def test(_config, split, shot, seed, exp_id, ckpt):
    """``python -m pemp_amd.entry.pemp_stage1 test with split=0 exp_id=1`` on synthetic episodes: the checkpoint is looked up
    with the reference's rules (``exp_id`` / ``ckpt``, entry/pemp_stage1.py:157-158, utils/misc.py:123-147) and loaded; no
    checkpoint -> FileNotFoundError (``ckpt=wgen`` opts into synthetic weights explicitly)."""
    import logging
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    logger = logging.getLogger(NAME)
    if split < 0:
        raise ValueError("Argument `split` is required! For example: `python -m pemp_amd.entry.pemp_stage1 test with split=0`")
    torch.manual_seed(seed)
    from ..core.snapshots import load_for_eval
    model = ModelClass(logger)
    load_for_eval(model, _config, exp_id, ckpt, logger)
    model = model.cuda().eval()
    dcfg = _config["data"]
    data = eval_episodes(dcfg, shot, split)
    # the reference's protocol is one episode per step (data.test_bs = 1): keep it, with four steps in flight
    ev = Evaluator(model, lanes=4 if dcfg["test_bs"] == 1 else 1)
    nclass = num_classes(dcfg["dataset"])
    loss, miou, biou = ev.start_eval_loop(data, nclass, split, _config["te"]["epochs"], logger, batch=dcfg["test_bs"],
                                          dataset_name=dcfg["dataset"])
    return f"Loss: {loss:.4f}, mIoU: {np.mean(miou) * 100:.2f}, bIoU: {np.mean(biou) * 100:.2f}"


def run_training(_config, name, make_trainer, make_evaluator, split, shot, seed, exp_id):
    """Shared body of the ``train`` commands: process-group setup, broadcast of every model involved (the trained one and,
    for stage 2, the frozen stage-1 prior network), TrainingLoop on synthetic episodes.  Checkpoints go to a FRESH run
    directory ``<g.model_dir>/<tag>/<next id>`` (Sacred numbers its runs the same way); ``exp_id`` only ever names a run to
    LOAD from."""
    import logging
    import os
    from ..core.base_trainer import TrainingLoop
    from .train_stage1 import broadcast_model
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    logger = logging.getLogger(name)
    if split < 0:
        raise ValueError("Argument `split` is required! For example: `python -m pemp_amd.entry.pemp_stage1 train with split=0`")
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(seed + rank)
    d = _config["data"]
    from ..core.snapshots import next_run_id
    trainer = make_trainer(logger if rank == 0 else None, dev)
    broadcast_model(trainer.model)
    if getattr(trainer, "stage1", None) is not None:          # the frozen prior network must be the same on every rank
        broadcast_model(trainer.stage1)
    run_id = torch.tensor([next_run_id(_config)], device=dev)
    if world > 1:
        dist.broadcast(run_id, 0)                             # rank 0 names the run
    loop = TrainingLoop(_config, trainer, make_evaluator(trainer, dev), logger, run_id=int(run_id.item()))
    val = eval_episodes(d, shot, split)
    batches = train_batches(d, shot, split, rank, loop.steps_per_epoch, dev)
    hist = loop.start_training_loop(batches, val, num_classes(d["dataset"]), split)
    return f"best val mIoU {loop.best_iou * 100:.2f} at epoch {loop.best_epoch}; checkpoints in {loop.model_dir}" if hist else "no epochs"


@ex.command
def train(_config, split, shot, seed, loss, sigma, exp_id):
    """``python -m pemp_amd.entry.pemp_stage1 train with split=0 tr.total_epochs=3 data.train_n=5000``: the reference's
    training procedure (entry/pemp_stage1.py:68-113, core/base_trainer.py:183-294) on synthetic episodes: fused HIP train
    steps, per-epoch evaluation, ``ckpt.pth`` / ``bestckpt.pth`` under ``<g.model_dir>/<tag>/<id>``.  One process per GPU
    under torchrun (gradients all-reduced over RCCL, evaluation sharded)."""
    from .train_stage1 import Trainer

    def make_trainer(logger, dev):
        return Trainer(ModelClass(logger), lr=_config["tr"]["lr"], device=dev, loss=loss, sigma=sigma)

    return run_training(_config, NAME, make_trainer, lambda tr, dev: Evaluator(tr.model, device=dev), split, shot, seed, exp_id)


#: response-map palette of the reference's viewer (core/base_trainer.py:348-349, listed there in BGR for cv2): rows 0-2 =
#: background prototypes, 3-5 = foreground prototypes; RGB here (PIL writes RGB)
RESPONSE_PALETTE_RGB = np.array([[25, 70, 147], [30, 116, 179], [112, 172, 207],
                                 [100, 11, 12], [193, 32, 38], [247, 178, 78]], np.uint8)
def evaluate_and_save(model, dataset, out_dir, n_episodes, device=None):
    """Counterpart of ``evaluate_and_save`` (reference core/base_trainer.py:311-403, driven by ``visualize``,
    entry/pemp_stage1.py:199-222): per episode run ``model(..., ret_ind=True)`` and write, under
    ``<out_dir>/<i>_<cls>/``, the support / query images and label images, the binary prediction, the
    response map coloured with the viewer's palette and ``data.json`` (Dice "acc", class id / name, sample names).
    ``dataset`` provides ``decoded_task(i)`` (uint8 sources).  Returns the list of Dice scores."""
    import json
    from pathlib import Path
    from PIL import Image
    from ..data_kits.episode import EpisodeTransform, test_samples
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    tf = EpisodeTransform(dataset.height, dataset.width, device=device)
    dataset.reset_sampler()
    dataset.sample_tasks()
    S, accs = dataset.shot, []
    model.eval()
    for i in range(n_episodes):
        sup, qry, cls = dataset.decoded_task(i)
        img, planes, labels = tf(test_samples(sup, qry, dataset.height, dataset.width))
        label = labels[0]
        with torch.no_grad():
            logits, indices = model(img[:S][None], planes[None], img[S:][None], out_shape=tuple(label.shape), ret_ind=True)
        pred = logits.argmax(dim=1)[0]
        acc = float((pred * label).sum() * 2) / max(float(pred.sum() + label.sum()), 1.0)
        accs.append(acc)
        cname = PASCAL_CLASSES[cls].replace("/", "_") if 0 <= cls < len(PASCAL_CLASSES) else str(cls)
        save = Path(out_dir) / f"{i:03d}_{cls:02d}"
        save.mkdir(parents=True, exist_ok=True)
        names = dataset.names(i) if hasattr(dataset, "names") else None        # a dataset on disk: the samples' own names
        data = {"acc": str(round(acc, 3)), "cls_id": int(cls), "cls_name": cname, "qry": names[1][0] if names else f"q{i}"}
        for j, (im, lab) in enumerate(sup):
            key = "sup" if S == 1 else f"sup{j + 1}"
            data[key] = names[0][j] if names else f"s{i}_{j}"
            Image.fromarray(im).save(save / f"{cname}_sup_img_{data[key]}.jpg")
            Image.fromarray(lab).save(save / f"{cname}_sup_msk_{data[key]}.png")
        Image.fromarray(qry[0][0]).save(save / f"{cname}_qry_img_{data['qry']}.jpg")
        Image.fromarray(qry[0][1]).save(save / f"{cname}_qry_msk_{data['qry']}.png")
        Image.fromarray((pred.cpu().numpy() * 255).astype(np.uint8)).save(save / f"{cname}_qry_pred_{data['qry']}.png")
        Image.fromarray(RESPONSE_PALETTE_RGB[indices[0].cpu().numpy()]).save(save / f"{cname}_qry_color_{data['qry']}.png")
        (save / "data.json").write_text(json.dumps(data))
    return accs


@ex.command
def visualize(_config, split, shot, seed, tag, exp_id, ckpt, p):
    """``python -m pemp_amd.entry.pemp_stage1 visualize with split=0 test_n=20``: predictions and response maps of
    the first ``test_n`` evaluation episodes into ``http/static/<exp>`` (the layout the reference's html viewer reads)."""
    if split < 0:
        raise ValueError("Argument `split` is required!")
    torch.manual_seed(seed)
    from ..core.snapshots import load_for_eval
    model = ModelClass(None)
    load_for_eval(model, _config, exp_id, ckpt)                  # entry/pemp_stage1.py:208-209
    model = model.cuda().eval()
    dcfg = _config["data"]
    if p["cls"] > 0:            # one chosen episode (entry/pemp_stage1.py:198-201): only a dataset on disk has named samples
        if not dcfg.get("base_dir"):
            raise ValueError("visualize with p.cls=.. p.sup=.. p.qry=.. needs data.base_dir (sample names belong to a dataset on disk)")
        from ..data_kits.pascal_voc import OneExampleLoader
        data = OneExampleLoader(dcfg, split, shot).choose(p["cls"], p["sup"], p["qry"])
        dcfg = dict(dcfg, test_n=1)
    else:
        data = eval_episodes(dcfg, shot, split, decoded=True)
    out = f"http/static/{exp_id}_{dcfg['dataset'].lower()}_{shot}shot_{tag}_s{split}"
    accs = evaluate_and_save(model, data, out, dcfg["test_n"])
    return f"saved {len(accs)} episodes to {out}; mean Dice {np.mean(accs):.3f}"


if __name__ == "__main__":
    print(ex.run_commandline())
