"""PASCAL-5i episodes from a dataset directory: the host side of the reference's ``data_kits/pascal_voc.py``.

What the reference's ``PascalVOCTrain`` / ``PascalVOCTest`` do on the host is kept as it is -- the directory layout, the class
lists, the deterministic task sampler -- and everything that touches pixels goes to the device (``data_kits/episode.py``):

* layout (pascal_voc.py:103-107,262-264): ``<base_dir>/JPEGImages/<name>.jpg``, ``<base_dir>/Binary_map_aug/{train,val}/<cls>.txt``
  (one sample name per line) and ``.../{train,val}/<cls>/<name>.png`` (0 / 255 label images);
* classes (pascal_voc.py:114-116,270-272): evaluation = the split's five, training = the other fifteen;
* sampler (pascal_voc.py:118-135): ``np.random.RandomState(test_seed)`` for evaluation (``seed`` for training), per task one
  ``choice(classes)`` followed by one ``choice(n_cls, size=shot + query, replace=False)`` -- the same two calls in the same
  order, so a dataset on disk yields the reference's task list (its data tests pin the first five of split 0 under
  ``test_seed = 5678``: classes 5, 1, 5, 3, 1; data_kits/pascal_voc_test.py:59-65);
* an episode leaves this module as DECODED uint8 arrays (``Image.open`` only): ``decoded_task(i)`` for evaluation (resize /
  normalise / mask planes then run in ``pemp_episode_preprocess``), ``train_batches`` for training (the reference's
  augmentation draws -- scale 1..1.5, flip, ColorJitter order and factors, ``crop_obj`` window -- made on the host in the
  reference's order, pascal_voc.py:184-240, the pixel work on the device).

COCO-20i (data_kits/coco.py) reads its annotations through ``pycocotools``, which this image does not have: ``load`` raises
for it and names the reason.
"""
import random as _random
from pathlib import Path

import numpy as np

CLASS_NAMES = (("aeroplane", "bicycle", "bird", "boat", "bottle"), ("bus", "car", "cat", "chair", "cow"),
               ("diningtable", "dog", "horse", "motorbike", "person"), ("potted plant", "sheep", "sofa", "train", "tv/monitor"))


class PascalVOCEpisodes:
    """``PascalVOCTest`` (``train=False``) / ``PascalVOCTrain`` (``train=True``) of the reference on decoded uint8 data.
    ``cfg``: the ``data`` ingredient's keys (``base_dir``, ``height``, ``width``, ``seed``, ``test_seed``, ``train_n``,
    ``test_n``, ``cache``).  The evaluator's protocol: ``reset_sampler()``, ``sample_tasks()``, ``len()``, ``decoded_task(i)``."""

    def __init__(self, cfg, split, shot, query=1, train=False, one_cls=0):
        if query != 1:
            raise ValueError("query must stay 1 (networks/pemp_stage1.py:197,257 broadcast one query per episode)")
        if split not in (0, 1, 2, 3):
            raise ValueError(f"split = {split}: PASCAL-5i has splits 0..3")
        self.cfg, self.split, self.shot, self.query, self.train, self.one_cls = cfg, int(split), int(shot), int(query), bool(train), int(one_cls)
        self.height, self.width, self.dataset = int(cfg["height"]), int(cfg["width"]), "PASCAL"
        self.base_dir = Path(cfg["base_dir"])
        sub = "train" if train else "val"
        self.img_dir = self.base_dir / "JPEGImages"
        self.lab_dir = self.id_dir = self.base_dir / "Binary_map_aug" / sub
        if not self.img_dir.is_dir() or not self.id_dir.is_dir():
            raise FileNotFoundError(f"Dataset PASCAL is not found in {self.base_dir} (JPEGImages/ and Binary_map_aug/{sub}/ expected)")
        self.cache = bool(cfg.get("cache", True))
        self._images, self._labels = {}, {}
        self.sample_by_class = {c: (self.id_dir / f"{c}.txt").read_text().strip().splitlines() for c in self.classes}
        self.idx_by_class = {c: len(v) for c, v in self.sample_by_class.items()}
        self.tasks = []
        self.reset_sampler()

    @property
    def classes(self):
        own = list(range(self.split * 5 + 1, self.split * 5 + 6))
        return sorted(set(range(1, 21)) - set(own)) if self.train else own

    def __len__(self):
        return int(self.cfg["train_n"] if self.train else self.cfg["test_n"])

    def reset_sampler(self):
        self.sampler = np.random.RandomState(int(self.cfg["seed"] if self.train else self.cfg["test_seed"]))

    def sample_tasks(self):
        self.tasks = []
        for _ in range(len(self)):
            cls = self.one_cls if self.one_cls > 0 else self.sampler.choice(self.classes)
            indices = self.sampler.choice(self.idx_by_class[cls], size=self.shot + self.query, replace=False)
            self.tasks.append((int(cls), [self.sample_by_class[cls][j] for j in indices]))

    # -- decoding (the only pixel work left on the host) --------------------------------------------------------
    def get_image(self, name):
        """uint8 [H, W, 3] of ``JPEGImages/<name>.jpg`` (pascal_voc.py:158-164)."""
        from PIL import Image
        a = self._images.get(name)
        if a is None:
            with Image.open(self.img_dir / f"{name}.jpg") as im:
                a = np.ascontiguousarray(np.asarray(im.convert("RGB"), np.uint8))
            if self.cache:
                self._images[name] = a
        return a

    def get_label(self, cls, name):
        """uint8 [H, W] (0 / 255) of ``Binary_map_aug/<mode>/<cls>/<name>.png`` (pascal_voc.py:166-176)."""
        from PIL import Image
        key = (cls, name)
        a = self._labels.get(key)
        if a is None:
            with Image.open(self.lab_dir / f"{cls}/{name}.png") as im:
                a = np.ascontiguousarray(np.array(im, np.uint8))
            if a.ndim != 2:
                raise ValueError(f"label image {cls}/{name}.png has shape {a.shape}: a single-channel 0 / 255 image is expected")
            if self.cache:
                self._labels[key] = a
        return a

    def pairs(self, cls, names):
        return [(self.get_image(n), self.get_label(cls, n)) for n in names]

    def names(self, i):
        """(support names, query names) of task ``i`` (the reference's ``ret_name`` outputs)."""
        _, all_names = self.tasks[i]
        return all_names[:self.shot], all_names[self.shot:]

    def decoded_task(self, i):
        """-> (support [(image, label)] * shot, query [(image, label)], class id): what ``Evaluator`` feeds to the device-side
        preprocessing (evaluation transform, pascal_voc.py:200-229: resize only, the query label keeps its size)."""
        cls, all_names = self.tasks[i]
        return self.pairs(cls, all_names[:self.shot]), self.pairs(cls, all_names[self.shot:]), cls

    def train_batches(self, bs, rng=None, shuffle=None):
        """One epoch of training batches as Sample lists for ``EpisodeLoader``: ``sample_tasks()`` must have been called.  The
        reference's DataLoader shuffles the task list (``shuffle=True``, pascal_voc.py:511-516) and draws the augmentation in
        its workers; here the order comes from ``shuffle`` (a permutation of ``len(self)``; default: torch.randperm) and the
        draws from ONE Python ``random`` stream ``rng`` in the reference's per-sample order (episode.train_samples).  The last,
        short batch is dropped (the fused step's hipGraphs and BatchNorm statistics are per batch shape)."""
        from .episode import train_samples
        if shuffle is None:
            import torch
            shuffle = torch.randperm(len(self)).tolist()
        rng = rng if rng is not None else _random
        for b in range(len(shuffle) // bs):
            samples = []
            for i in shuffle[b * bs:(b + 1) * bs]:
                cls, all_names = self.tasks[i]
                p = self.pairs(cls, all_names)
                for img, lab in p:
                    if img.shape[0] < 1 or img.shape[:2] != lab.shape:
                        raise ValueError(f"image / label size mismatch in class {cls}: {img.shape[:2]} vs {lab.shape}")
                samples += train_samples(p[:self.shot], p[self.shot:], self.height, self.width, rng)
            yield samples


class OneExampleLoader(PascalVOCEpisodes):
    """``OneExampleLoader`` of the reference (pascal_voc.py:533-558; ``visualize with p.cls=<id> p.sup=<name> p.qry=<name>``,
    entry/pemp_stage1.py:198-201): ONE chosen episode of the validation directory, no sampler, no cache."""

    def __init__(self, cfg, split, shot, query=1):
        super().__init__(cfg, split, shot, query, train=False)
        self.cache = False

    def reset_sampler(self):
        pass

    def sample_tasks(self):
        pass

    def __len__(self):
        return len(self.tasks)

    def choose(self, cls, sup_names, qry_names):
        def as_list(v):
            v = [v] if isinstance(v, (str, int)) else list(v)
            if not all(isinstance(n, str) for n in v):     # 2007_000032 unquoted is an int literal to Python (and to Sacred's parser)
                raise ValueError(f"sample names must be strings, got {v}: quote them on the command line (p.sup='\"2007_000032\"')")
            return v
        sup_names, qry_names = as_list(sup_names), as_list(qry_names)
        if len(sup_names) != self.shot or len(qry_names) != self.query:
            raise ValueError(f"p.sup / p.qry: {self.shot} support and {self.query} query sample name(s) expected")
        self.tasks = [(int(cls), sup_names + qry_names)]
        return self


def load(cfg, train_mode, split, shot, query=1, one_cls=0):
    """``pascal_voc.load`` / ``datasets.load`` of the reference (data_kits/datasets.py:53-72, pascal_voc.py:462-531) for the modes
    this build runs: -> (dataset, num_classes).  ``train_mode``: "train" | "test" | "eval_online"."""
    if cfg.get("dataset", "PASCAL") != "PASCAL":
        raise NotImplementedError("COCO-20i from disk needs pycocotools (data_kits/coco.py:7), which this image does not have; "
                                  "COCO-shaped SYNTHETIC episodes run without data.base_dir")
    if train_mode not in ("train", "test", "eval_online"):
        raise ValueError(f"Not support training mode `{train_mode}`. Selected from [train, test, eval_online]")
    return PascalVOCEpisodes(cfg, split, shot, query, train=train_mode == "train", one_cls=one_cls), 20
