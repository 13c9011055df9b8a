"""``data`` ingredient and label helpers with the reference's key surface (data_kits/datasets.py:13-31,83-117).

There is no dataset on either box: the keys are accepted and carried so that a reference command line
(``with data.test_n=1000 data.height=401 ...``) parses unchanged; the synthetic episode sources of
``pemp_amd.entry`` read ``height / width / bs / test_bs / test_n / seed / test_seed / mean / std`` from here.
"""
from ..config import Ingredient

data_ingredient = Ingredient("data", save_git_info=False)


@data_ingredient.config
def data_config():
    dataset = "PASCAL"              # str, dataset name [PASCAL, COCO]
    base_dir = ""                   # str, data directory (unused: synthetic episodes)
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


def get_val_labels(split, dataset="PASCAL"):
    """Validation classes of a split (reference data_kits/datasets.py:83-104): 5 per PASCAL-5i split, 20 per COCO-20i."""
    n = 5 if dataset == "PASCAL" else 20
    return list(range(split * n + 1, split * n + n + 1))


def get_class_name(cls, dataset="PASCAL"):
    if dataset == "PASCAL" and 0 <= cls < len(PASCAL_CLASSES):
        return PASCAL_CLASSES[cls]
    return str(int(cls))
