"""Size-independent properties of the path at the full evaluation shape (401 x 401): what must hold whatever the weights and
episodes are, with no reference output needed.

* episodes of a batch are independent: permuting them permutes the outputs, bit for bit (exact conv variants);
* the prototypes average over the shots: permuting the support shots of a 5-shot episode changes the logits by rounding only;
* the similarity map is a cosine: scaling the query features or the prototypes by a positive constant changes nothing but
  rounding; a NEGATED prototype flips the sign of its map;
* the evaluation statistics are additive: tp / fp / fn of a batch are the sums over its episodes evaluated alone."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
H = 401


@pytest.fixture(scope="module")
def model(hip_lib, dev):
    from pemp_amd.networks import pemp_stage1 as m
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    return net.to(dev).eval()


def _episodes(dev, seeds, shot):
    from pemp_amd import synth
    b = synth.make_batch(list(seeds), shot=shot, height=H, width=H, out_hw=(H, H))
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(b["sup_img"]), t(b["sup_mask"]), t(b["qry_img"]), t(b["qry_mask"][:, 0])


def test_permuting_the_episodes_permutes_the_outputs(model, dev, exact_eval_variants):
    from pemp_amd import ops
    sup, msk, qry, gt = _episodes(dev, range(900, 906), 1)
    perm = torch.tensor([4, 0, 5, 2, 1, 3], device=dev)
    with torch.no_grad():
        pred, _ = model.lowres(sup, msk, qry)
        am, st, _ = ops.eval_tail(pred, gt)
        pred2, _ = model.lowres(sup[perm].contiguous(), msk[perm].contiguous(), qry[perm].contiguous())
        am2, st2, _ = ops.eval_tail(pred2, gt[perm].contiguous())
    assert torch.equal(pred[perm], pred2) and torch.equal(am[perm], am2) and torch.equal(st[perm], st2)
    # additivity: every episode alone gives its row of the batch statistics
    for i in (0, 3):
        with torch.no_grad():
            p1, _ = model.lowres(sup[i:i + 1], msk[i:i + 1], qry[i:i + 1])
            _, s1, _ = ops.eval_tail(p1, gt[i:i + 1])
        assert torch.equal(p1[0], pred[i]) and torch.equal(s1[0], st[i])


def test_permuting_the_support_shots_changes_rounding_only(model, dev):
    sup, msk, qry, gt = _episodes(dev, (910, 911), 5)
    order = torch.tensor([3, 0, 4, 1, 2], device=dev)
    with torch.no_grad():
        a, _ = model.lowres(sup, msk, qry)
        b, _ = model.lowres(sup[:, order].contiguous(), msk[:, order].contiguous(), qry)
    d = (a - b).abs().max().item()
    print(f"5-shot, shots permuted: max |d logit| at feature resolution {d:.2e}")
    assert d < 2e-4                                                     # the trunk is per image (bit-equal); only the mean over shots reorders
    lead = (a[:, 1] - a[:, 0]).abs()
    assert torch.equal(a.argmax(1)[lead > 1e-3], b.argmax(1)[lead > 1e-3])


def test_similarity_map_is_scale_invariant_and_odd_in_the_prototype(hip_lib, dev):
    from pemp_amd import ops
    torch.manual_seed(3)
    B, h, w, c, p = 25, 51, 51, 512, 3
    qry = torch.randn(B, h, w, c, device=dev)
    pro = torch.randn(B, 2 * p, c, device=dev)
    base, resp = ops.cosine_proto_max(qry, pro, 20.0, want_resp=True)
    assert base.abs().max().item() <= 20.0 + 1e-4
    for k in (3.7, 1.0 / 64):
        assert (ops.cosine_proto_max(qry * k, pro, 20.0) - base).abs().max().item() < 5e-5
        assert (ops.cosine_proto_max(qry, pro * k, 20.0) - base).abs().max().item() < 5e-5
    # powers of two scale exactly: bit-identical maps and response indices
    p2, r2 = ops.cosine_proto_max(qry * 4.0, pro * 0.5, 20.0, want_resp=True)
    assert torch.equal(p2, base) and torch.equal(r2, resp)
    # one prototype per group (p = 1): the map is 20 cos, odd in the prototype
    one = pro[:, :2].contiguous()
    m1 = ops.cosine_proto_max(qry, one, 20.0)
    assert (ops.cosine_proto_max(qry, -one, 20.0) + m1).abs().max().item() < 5e-5
