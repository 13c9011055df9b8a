"""Double-precision evaluation of the train-step fixtures (test infrastructure).

The fp32 gradients the reference produced (``*_trainstep.npz``, made by make_golden.py) carry the
rounding of back-propagation through ~50 batch-statistics BatchNorms; to know how much of a difference
to the HIP path is rounding, the same step is evaluated in fp64 with the oracle (oracle/ref_cpu.py, which
is bit-equal to the reference in fp32 on every eval fixture) under torch autograd.  Writes
``<case>_trainstep_f64.npz`` with the same sampled tensors as the fp32 fixture (keys ``g64__<name>``)
and all gradient norms (``grad_norms64``).

    python tests/golden/make_f64.py stage1 | stage2 | stage1_full | stage2_full | baseline_rn50 | baseline_vgg16 | stage1_vgg16
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle import ref_cpu as R          # noqa: E402
from pemp_amd import synth              # noqa: E402
from tests import util                  # noqa: E402

CASES = {"stage1": ("stage1_rn50_trainstep", "stage1_rn50", 1234), "stage2": ("stage2_rn50cm_trainstep", "stage2_rn50cm", 4321),
         # the shapes BASELINE.json configs[2] / configs[3] run at (401x401; 4 one-shot episodes / one 5-shot episode)
         "stage1_full": ("stage1_rn50_trainstep_full", "stage1_rn50", 1234),
         "stage2_full": ("stage2_rn50cm_trainstep5_full", "stage2_rn50cm", 4321),
         "baseline_rn50": ("baseline_rn50_trainstep", "baseline_rn50", 1234),
         "baseline_vgg16": ("baseline_vgg16_trainstep", "baseline_vgg16", 1234),
         "stage1_vgg16": ("stage1_vgg16_trainstep", "stage1_vgg16", 1234)}


def main(case):
    fixture, wtag, wseed = CASES[case]
    torch.set_num_threads(8)
    g = util.gold(fixture)
    sd32 = util.wgen_state_dict(wtag, wseed)
    frozen = {str(n) for n, v in zip(g["grad_names"], g["grad_norms"]) if v < 0}
    sd = {}
    for k, v in sd32.items():
        if v.dtype != torch.float32:
            sd[k] = v.clone()
        else:
            leaf = "running" not in k and k not in frozen
            sd[k] = v.double().requires_grad_(leaf)
    seeds = [int(v) for v in g["seeds"]] if "seeds" in g.files else [31, 32]
    H = int(g["H"]) if "H" in g.files else 97
    shot = int(g["shot"]) if "shot" in g.files else 1
    b = synth.make_batch(seeds, shot=shot, height=H, width=H, out_hw=(H, H))
    t = lambda a: torch.from_numpy(a)
    R.TRAIN = True
    try:
        ins = (t(b["sup_img"]).double(), t(b["sup_mask"]).double(), t(b["qry_img"]).double())
        if case.startswith("baseline"):
            logits = R.baseline_forward(sd, *ins, (H, H), backbone="resnet50" if case.endswith("rn50") else "vgg16")
        elif case.startswith("stage1"):
            logits = R.stage1_forward(sd, *ins, (H, H), backbone="vgg16" if case.endswith("vgg16") else "resnet50")
        else:
            from tests.golden.make_golden import stage2_train_prior
            prior = t(stage2_train_prior(b["qry_mask"])).double()
            logits = R.stage2_forward(sd, *ins, prior, (H, H))
        loss = F.cross_entropy(logits, t(b["qry_mask"][:, 0]), ignore_index=255)
        names = [k for k, v in sd.items() if v.requires_grad]
        grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    finally:
        R.TRAIN = False
    print("loss fp64", float(loss), "reference fp32", float(g["loss"]))
    out = {"loss64": np.array(float(loss))}
    for key in [k for k in g.files if k.startswith("grad__")]:
        name = key[len("grad__"):]
        ref = torch.from_numpy(g[key])
        g64 = grads[name]
        g64 = (g64 if g64.numel() <= 40000 else g64.reshape(-1)[::37]).reshape(ref.shape)
        print(f"{name:55s} |ref32 - f64| / max|f64| = {float((ref.double() - g64).abs().max() / g64.abs().max()):.2e}")
        out["g64__" + name] = g64.numpy()
    out["grad_norms64"] = np.array([float(grads[str(n)].norm()) if str(n) in grads else -1.0 for n in g["grad_names"]])
    np.savez_compressed(ROOT / "tests" / "golden" / f"{fixture}_f64.npz", **out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "stage1")
