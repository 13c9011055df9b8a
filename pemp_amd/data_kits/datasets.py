"""``data`` ingredient and label helpers with the reference's key surface (data_kits/datasets.py:13-31,83-117).

There is no dataset on either box, so ``base_dir`` defaults to empty and the entry commands then run on the synthetic
episode sources of ``pemp_amd.entry`` (which read ``height / width / bs / test_bs / test_n / seed / test_seed / mean / std``
from here); a reference command line (``with data.test_n=1000 data.height=401 ...``) parses unchanged.  With
``data.base_dir=<.../VOC2012>`` the commands read PASCAL-5i from disk with the reference's lists and task sampler
(``pemp_amd.data_kits.pascal_voc``; the reference's config hook points base_dir at ``data/VOCdevkit/VOC2012`` itself,
data_kits/datasets.py:34-50).
"""
from ..config import Ingredient

data_ingredient = Ingredient("data", save_git_info=False)


@data_ingredient.config
def data_config():
    dataset = "PASCAL"              # str, dataset name [PASCAL, COCO]
    base_dir = ""                   # str, dataset directory (PASCAL: .../VOCdevkit/VOC2012); empty: synthetic episodes
    mean = [0.485, 0.456, 0.406]    # list, normalization mean in data preprocessing
    std = [0.229, 0.224, 0.225]     # list, normalization std in data preprocessing
    height = 401                    # int, input image height
    width = 401                     # int, input image width
    bs = 4                          # int, training batch size (episodes)
    test_bs = 1                     # int, episodes per evaluation step (metrics are identical for any value)
    num_workers = min(bs, 4)        # int, loader workers
    pin_memory = True               # bool, pinned host staging
    train_n = 5000                  # int, training episodes per epoch
    test_n = 1000                   # int, evaluation episodes per round
    seed = 1234                     # int, training sampler seed
    test_seed = 5678                # int, evaluation sampler seed
    one_cls = 0                     # int, restrict to one class (0 = all)
    cache = True                    # bool, cache decoded images


PASCAL_CLASSES = ("background", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
                  "diningtable", "dog", "horse", "motorbike", "person", "potted plant", "sheep", "sofa", "train", "tv/monitor")


#: COCO-20i class names per split, label id = split * 20 + position + 1 (reference data_kits/coco.py:20-35)
COCO_CLASSES = (
    ("person", "airplane", "boat", "parking meter", "dog", "elephant", "backpack", "suitcase", "sports ball", "skateboard",
     "wine glass", "spoon", "sandwich", "hot dog", "chair", "dining table", "mouse", "microwave", "refrigerator", "scissors"),
    ("bicycle", "bus", "traffic light", "bench", "horse", "bear", "umbrella", "frisbee", "kite", "surfboard",
     "cup", "bowl", "orange", "pizza", "couch", "toilet", "remote", "oven", "book", "teddy bear"),
    ("car", "train", "fire hydrant", "bird", "sheep", "zebra", "handbag", "skis", "baseball bat", "tennis racket",
     "fork", "banana", "broccoli", "donut", "potted plant", "tv", "keyboard", "toaster", "clock", "hair drier"),
    ("motorcycle", "truck", "stop sign", "cat", "cow", "giraffe", "tie", "snowboard", "baseball glove", "bottle",
     "knife", "apple", "carrot", "cake", "bed", "laptop", "cell phone", "sink", "vase", "toothbrush"),
)


def get_val_labels(split, dataset="PASCAL"):
    """Validation classes of a split (reference data_kits/datasets.py:83-104): 5 per PASCAL-5i split, 20 per COCO-20i."""
    from .. import synth
    return synth.val_labels(split, dataset)


def num_classes(dataset="PASCAL"):
    """Foreground classes of the metric table (entry/pemp_stage1.py:151-152): PASCAL 20, COCO 80 -> ``[C+1, 3]``."""
    from .. import synth
    return synth.num_classes(dataset)


def get_class_name(cls, dataset="PASCAL"):
    if dataset == "PASCAL" and 0 <= cls < len(PASCAL_CLASSES):
        return PASCAL_CLASSES[cls]
    if dataset == "COCO" and 1 <= cls <= 80:          # data_kits/coco.py:20-35, class_names[split][k] <-> split*20+k+1
        return COCO_CLASSES[(int(cls) - 1) // 20][(int(cls) - 1) % 20]
    return str(int(cls))
