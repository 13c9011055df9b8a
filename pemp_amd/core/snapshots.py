"""Checkpoint lookup with the reference's rules (utils/misc.py:87-147) and run-directory numbering.

``find_snapshot(cfg, exp_id, ckpt)`` resolves, in this order,
  1. ``<g.model_dir>/<tag>/<exp_id>/{ckpt, bestckpt.pth, ckpt.pth}``            (exp_id >= 0)
  2. any ``<g.model_dir>/*/<exp_id>/...`` (a run of another tag, e.g. the stage-1 run a stage-2 job names with ``s1.id``)
  3. ``ckpt`` taken as a path
  4. the run with the LARGEST id under ``<g.model_dir>/<tag>/``
and returns ``(path, exp_id)`` (99999999 for a direct path) like the reference.  Where the reference falls back to an
interactive ``input("Cannot find checkpoints. Please input:")`` this raises ``FileNotFoundError``: the commands never
evaluate or freeze a randomly initialised model silently.  ``ckpt=wgen`` is the one explicit opt-in to synthetic
``Wgen`` weights (pemp_amd.synth) for boxes that hold no checkpoint at all.
"""
from pathlib import Path

WGEN = "wgen"           #: ``ckpt=wgen``: deterministic synthetic weights instead of a file (no dataset / checkpoint exists on the GPU boxes)
DIRECT = 99999999


def get_pth(exp_dir, ckpt):
    """utils/misc.py:90-99: the named file, else bestckpt.pth, else ckpt.pth inside ``exp_dir``."""
    for name in (ckpt, "bestckpt.pth", "ckpt.pth"):
        if name is None:
            continue
        p = Path(exp_dir) / name
        if p.exists() and p.is_file():
            return p
    return False


def try_possible_ckpt_names(base, exp_id, ckpt=None):
    """utils/misc.py:102-120."""
    base = Path(base)
    if isinstance(exp_id, int) and exp_id >= 0:
        p = get_pth(base / str(exp_id), ckpt)
        if p:
            return p, exp_id
        for exp_dir in sorted(base.parent.glob("*/[0-9]*")):
            if exp_dir.is_dir() and exp_dir.name.isdigit() and int(exp_dir.name) == exp_id:
                return get_pth(exp_dir, ckpt), exp_id
    if ckpt:
        f = Path(ckpt)
        if f.exists() and f.is_file():
            return f, DIRECT
    return False, DIRECT


def find_snapshot(cfg, exp_id=-1, ckpt=None):
    """utils/misc.py:123-147 without the interactive prompt (raises FileNotFoundError instead)."""
    if ckpt is None:
        ckpt = cfg["ckpt"]
    base = Path(cfg["g"]["model_dir"]) / str(cfg["tag"])
    found = try_possible_ckpt_names(base, exp_id, ckpt)
    if found[0]:
        return found
    ids = [int(d.name) for d in base.glob("[0-9]*") if d.is_dir() and d.name.isdigit()]
    if ids:
        found = try_possible_ckpt_names(base, max(ids), ckpt)
        if found[0]:
            return found
    raise FileNotFoundError(
        f"Cannot find checkpoints: exp_id={exp_id} ckpt={ckpt!r} under {base} (or as a path).  Pass `exp_id=<run>` / "
        f"`ckpt=/path/to/ckpt.pth`, or `ckpt={WGEN}` to run on synthetic Wgen weights explicitly.")


def load_for_eval(model, cfg, exp_id, ckpt, logger=None, wgen_seed=1234):
    """What the reference's ``test`` / ``visualize`` do before evaluating (entry/pemp_stage1.py:157-158):
    ``find_snapshot`` + ``model.load_weights``.  Returns the path loaded (or "wgen")."""
    if ckpt == WGEN:
        from .. import synth
        model.load_state_dict(synth.wgen_state_dict_for(model, wgen_seed))
        if logger is not None:
            logger.info(f"           ==> Model {model.__class__.__name__} initialized from synthetic Wgen({wgen_seed}) weights (ckpt={WGEN})")
        return WGEN
    path, _ = find_snapshot(cfg, exp_id, ckpt)
    import logging
    model.load_weights(path, logger if logger is not None else logging.getLogger("pemp_amd"))
    return path


def next_run_id(cfg):
    """A fresh run directory number under ``<g.model_dir>/<tag>/`` (Sacred's FileStorageObserver numbers runs 1, 2, ...;
    entry/pemp_stage1.py:113 ``_run._id``): never the id of an existing run."""
    base = Path(cfg["g"]["model_dir"]) / str(cfg["tag"])
    ids = [int(d.name) for d in base.glob("[0-9]*") if d.is_dir() and d.name.isdigit()]
    return max(ids, default=0) + 1
