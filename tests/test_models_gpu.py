"""End-to-end parity of the other models of the hot path (Baseline VGG16 / ResNet-50, stage-1 VGG16,
stage-1 plain-MAP branch, stage-2 ResNet-50+CM) against the reference's golden vectors.
Tolerances as in test_stage1_gpu.py (VGG accumulates K = 4608 products per output, 13 layers deep)."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _compare(g, e, logits, tgt, feats=None, H=97, ltol=util.LOGIT_TOL):
    """logits within ``ltol`` of the reference's; arg-max EXACT outside the 2 * ltol decision margin."""
    logits = logits.cpu()
    lref = torch.from_numpy(g[f"e{e}_logits_s7"])
    lerr = (logits[0, :, ::7, ::7] - lref).abs().max().item()
    assert lerr < ltol, f"logit err {lerr}"
    am = logits.argmax(1).numpy().astype(np.uint8)
    ref_bits = np.unpackbits(g[f"e{e}_argmax_bits"])[: am.size].reshape(am.shape)
    agree = 1.0 - util.assert_argmax_exact(logits, ref_bits, margin=2 * ltol, max_masked=0.02)
    print(f"|dlogit| {lerr:.2e}  pixels outside the margin {agree:.5f}")
    loss = torch.nn.functional.cross_entropy(logits, tgt.cpu(), ignore_index=255).item()
    assert abs(loss - float(g[f"e{e}_loss"])) < 2e-4
    if feats is not None:
        f = feats.cpu().permute(0, 3, 1, 2)
        fref = torch.from_numpy(g[f"e{e}_feat_c8"])
        fgot = f[:, ::8] if H <= 97 else f[:, ::8, ::5, ::5]
        ferr = ((fgot - fref).abs() / (1 + fref.abs())).max().item()
        assert ferr < 2e-3, f"feature err {ferr}"
    return lerr, agree


@pytest.mark.parametrize("backbone,keys,fixtures", [
    ("vgg16", "baseline_vgg16", ["baseline_vgg16_small", "baseline_vgg16_small5", "baseline_vgg16_full"]),
    ("resnet50", "baseline_rn50", ["baseline_rn50_small"])])
def test_baseline_matches_reference_golden(hip_lib, dev, backbone, keys, fixtures):
    from pemp_amd.networks import baseline as m
    net = m.Baseline(None, backbone=backbone)
    net.load_state_dict(util.wgen_state_dict(keys))
    net = net.to(dev).eval()
    for fx in fixtures:
        g = util.gold(fx)
        shot, H = int(g["shot"]), int(g["H"])
        for e, seed in enumerate(g["seeds"]):
            hw = tuple(int(v) for v in g[f"e{e}_out_hw"])
            t = util.episode_tensors(seed, shot, H, hw, dev)
            with torch.no_grad():
                out = net(t["sup_img"], t["sup_mask"], t["qry_img"], hw)
            _compare(g, e, out, t["qry_mask"], net._last_feats, H)


def test_stage1_vgg16_matches_reference_golden(hip_lib, dev):
    from pemp_amd.networks import pemp_stage1 as m
    net = m.PEMPStage1(None, backbone="vgg16")
    net.load_state_dict(util.wgen_state_dict("stage1_vgg16"))
    net = net.to(dev).eval()
    g = util.gold("stage1_vgg16_small")
    for e, seed in enumerate(g["seeds"]):
        hw = tuple(int(v) for v in g[f"e{e}_out_hw"])
        t = util.episode_tensors(seed, 1, 97, hw, dev)
        with torch.no_grad():
            out = net(t["sup_img"], t["sup_mask"], t["qry_img"], hw)
        _compare(g, e, out, t["qry_mask"], net._last_feats)


def test_stage1_resnet101_matches_reference_golden(hip_lib, dev):
    """The deeper trunk (23 blocks in layer3) through the same engines."""
    from pemp_amd.networks import pemp_stage1 as m
    net = m.PEMPStage1(None, backbone="resnet101")
    net.load_state_dict(util.wgen_state_dict("stage1_rn101"))
    net = net.to(dev).eval()
    g = util.gold("stage1_rn101_small")
    for e, seed in enumerate(g["seeds"]):
        hw = tuple(int(v) for v in g[f"e{e}_out_hw"])
        t = util.episode_tensors(seed, 1, 97, hw, dev)
        with torch.no_grad():
            out, resp = net(t["sup_img"], t["sup_mask"], t["qry_img"], hw, ret_ind=True)
        _compare(g, e, out, t["qry_mask"], net._last_feats)
        sd101 = util.wgen_state_dict("stage1_rn101")
        _, margin = util.response_reference(util.oracle_stage1_feats(sd101, t["sup_img"], t["qry_img"], "resnet101"), t["sup_mask"],
                                            sd101["ctr"], 1, 1, 3, 20, hw)
        util.assert_response_exact(resp[0, ::7, ::7], g[f"e{e}_resp_s7"], margin[0, ::7, ::7], what="rn101", max_masked=0.02)   # the oracle's share: 0.0046


def test_stage1_plain_map_branch_matches_reference_golden(hip_lib, dev):
    from pemp_amd.networks import pemp_stage1 as m
    net = m.PEMPStage1(None, protos=0)
    sd = util.wgen_state_dict("stage1_rn50")
    sd.pop("ctr")
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    g = util.gold("stage1_rn50_map_small")
    t = util.episode_tensors(11, 2, 97, (80, 120), dev)
    with torch.no_grad():
        out = net(t["sup_img"], t["sup_mask"], t["qry_img"], (80, 120), protos=0)
    assert (out[0].cpu() - torch.from_numpy(g["e0_logits"])).abs().max().item() < 5e-3


@pytest.mark.parametrize("fixture", ["stage2_rn50cm_small", "stage2_rn50cm_small5", "stage2_rn50cm_full5"])
def test_stage2_matches_reference_golden(hip_lib, dev, fixture):
    """``full5``: BASELINE.json configs[3] at its real shape -- one 5-shot episode at 401 x 401 (6 images through
    ResNet-50+CM, communication-module statistics at 101 x 101, prior from the reference's own stage-1 arg-max)."""
    from pemp_amd.networks import pemp_stage2 as m
    g = util.gold(fixture)
    shot, H = int(g["shot"]), int(g["H"])
    net = m.PEMPStage2(shot, 1, None)
    sd2 = util.wgen_state_dict("stage2_rn50cm", seed=4321)
    net.load_state_dict(sd2)
    net = net.to(dev).eval()
    for e, seed in enumerate(g["seeds"]):
        hw = tuple(int(v) for v in g[f"e{e}_out_hw"])
        t = util.episode_tensors(seed, shot, H, hw, dev)
        prior = torch.from_numpy(np.unpackbits(g[f"e{e}_prior_bits"])[: H * H].reshape(1, 1, H, H).astype(np.int64)).to(dev)
        with torch.no_grad():
            out, resp = net(t["sup_img"], t["sup_mask"], t["qry_img"], prior, hw, ret_ind=True)
        _compare(g, e, out, t["qry_mask"], net._last_feats, H)
        ap = net.adaptive_p.cpu()
        ref = torch.from_numpy(g[f"e{e}_adaptive_p"])
        assert ((ap - ref).abs() / (1 + ref.abs())).max().item() < 2e-3
        # margins from the ORACLE's features (its share of sampled pixels inside the margin: <= 0.0139)
        _, margin = util.response_reference(util.oracle_stage2_feats(sd2, t["sup_img"], t["sup_mask"], t["qry_img"], prior), t["sup_mask"], sd2["ctr"],
                                            1, shot, 3, 20, hw)
        util.assert_response_exact(resp[0, ::7, ::7], g[f"e{e}_resp_s7"], margin[0, ::7, ::7], what=fixture, max_masked=0.03)


@pytest.mark.parametrize("fixture", ["stage2_vgg16cm_small", "stage2_vgg16cm_small5"])
def test_stage2_vgg16cm_matches_reference_golden(hip_lib, dev, fixture):
    """stage 2 on VGG16CM (reference networks/backbones.py:424-533) -- communication channels concatenated in
    front of zero-padded 3x3 convs, four Linear(2c -> 2), no purifier -- against vectors the reference's own classes produced
    (built with pretrained = None: the shipped import step crashes)."""
    from pemp_amd.networks import pemp_stage2 as m
    g = util.gold(fixture)
    shot, H = int(g["shot"]), int(g["H"])
    net = m.PEMPStage2(shot, 1, None, backbone2="vgg16")
    sd2 = util.wgen_state_dict("stage2_vgg16cm", seed=4321)
    net.load_state_dict(sd2)
    net = net.to(dev).eval()
    for e, seed in enumerate(g["seeds"]):
        hw = tuple(int(v) for v in g[f"e{e}_out_hw"])
        t = util.episode_tensors(seed, shot, H, hw, dev)
        prior = torch.from_numpy(np.unpackbits(g[f"e{e}_prior_bits"])[: H * H].reshape(1, 1, H, H).astype(np.int64)).to(dev)
        with torch.no_grad():
            out, resp = net(t["sup_img"], t["sup_mask"], t["qry_img"], prior, hw, ret_ind=True)
            again, _ = net.lowres_graphed(t["sup_img"], t["sup_mask"], t["qry_img"], prior.float())
            again2, _ = net.lowres_graphed(t["sup_img"], t["sup_mask"], t["qry_img"], prior.float())
        _compare(g, e, out, t["qry_mask"], net._last_feats, H)
        ap = net.adaptive_p.cpu()
        ref = torch.from_numpy(g[f"e{e}_adaptive_p"])
        assert ((ap - ref).abs() / (1 + ref.abs())).max().item() < 2e-3
        # margins from the ORACLE's features (its share of sampled pixels inside the margin: 0.0602 on small/e0, <= 0.0077 elsewhere)
        _, margin = util.response_reference(util.oracle_stage2_feats(sd2, t["sup_img"], t["sup_mask"], t["qry_img"], prior, "vgg16"), t["sup_mask"],
                                            sd2["ctr"], 1, shot, 3, 20, hw)
        util.assert_response_exact(resp[0, ::7, ::7], g[f"e{e}_resp_s7"], margin[0, ::7, ::7], what=fixture, max_masked=0.08)
        assert torch.equal(again, again2)                      # hipGraph replay of the VGG16CM engine is stable


def test_stage2_prior_from_stage1_pipeline(hip_lib, dev):
    """Evaluator.test_step of stage 2 (entry/pemp_stage2.py:58-65): stage-1 argmax at 401-style full size
    feeds stage 2; the device-side argmax (eval_tail without target) must equal the logits' argmax."""
    from pemp_amd import ops
    from pemp_amd.networks import pemp_stage1 as m1
    s1 = m1.ModelClass(None)
    s1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    s1 = s1.to(dev).eval()
    t = util.episode_tensors(3, 1, 97, (80, 120), dev)
    with torch.no_grad():
        logits = s1(t["sup_img"], t["sup_mask"], t["qry_img"])
        pred, _ = s1.lowres(t["sup_img"], t["sup_mask"], t["qry_img"])
        am, _, _ = ops.eval_tail(pred, None, out_hw=(97, 97))
    assert torch.equal(am.long(), logits.argmax(1))
    g = util.gold("stage2_rn50cm_small")
    ref_prior = np.unpackbits(g["e0_prior_bits"])[: 97 * 97].reshape(1, 97, 97)
    util.assert_argmax_exact(logits, ref_prior, what="stage-1 prior")


def test_stage2_evaluator_pipeline_with_graphs(hip_lib, dev):
    """entry/pemp_stage2 Evaluator: stage-1 prior -> stage 2 -> fused tail, eager vs hipGraph replay."""
    from pemp_amd.entry import pemp_stage2 as e2
    from pemp_amd.networks import pemp_stage1 as m1, pemp_stage2 as m2
    s1 = m1.ModelClass(None)
    s1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    s2 = m2.PEMPStage2(1, 1, None)
    s2.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    s1, s2 = s1.to(dev).eval(), s2.to(dev).eval()
    t = util.episode_tensors(3, 1, 97, (80, 120))
    inputs = (t["sup_img"], t["sup_mask"], t["qry_img"])
    ev_e = e2.Evaluator(s1, s2, dev, use_graph=False)
    ev_g = e2.Evaluator(s1, s2, dev, use_graph=True)
    p_e, l_e = ev_e.test_step(inputs, t["qry_mask"][None])
    for _ in range(2):                                   # second call replays the captured graphs
        p_g, l_g = ev_g.test_step(inputs, t["qry_mask"][None])
    assert p_e.shape == (1, 80, 120) and (p_e == p_g).all() and l_e == l_g
    g = util.gold("stage2_rn50cm_small")
    ref_bits = np.unpackbits(g["e0_argmax_bits"])[: p_e.size].reshape(p_e.shape)
    assert (p_e == ref_bits).mean() > 0.995 and abs(l_e - float(g["e0_loss"])) < 5e-4


def test_real_pascal_episodes_stage1_prior_stage2(hip_lib, dev):
    """The two REAL PASCAL episodes the reference ships with its viewer (http/static/1005_pascal_1shot_pemp_stage2_s0: real
    textures, real object outlines, fg shares 33 % and 5 %), from decoded uint8 pixels to the stage-2 prediction entirely on
    the device: pemp_episode_preprocess (Pillow-exact resize / normalise / label planes) -> stage 1 -> arg-max prior -> stage 2,
    against what the reference's own classes produced from the same files (tests/golden/real_episodes.npz).  Logits within
    LOGIT_TOL, arg-max and response index exact outside the decision margin (margin shares printed), tp/fp/fn counts within the
    pixels inside the margin, CE loss within 2e-4; and the production evaluator (device-side prior, hipGraph replay) gives the
    same prediction."""
    from pemp_amd.data_kits.episode import EpisodeTransform, test_samples
    from pemp_amd.entry import pemp_stage2 as e2
    from pemp_amd.networks import pemp_stage1 as m1, pemp_stage2 as m2
    g = util.gold("real_episodes")
    s1 = m1.ModelClass(None)
    s1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    s2 = m2.PEMPStage2(1, 1, None)
    s2.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    s1, s2 = s1.to(dev).eval(), s2.to(dev).eval()
    tf = EpisodeTransform(401, 401, device=dev)
    ev = e2.Evaluator(s1, s2, dev)
    for e in range(2):
        sup = [(g[f"e{e}_sup_img_u8"], g[f"e{e}_sup_lab_u8"])]
        qry = [(g[f"e{e}_qry_img_u8"], g[f"e{e}_qry_lab_u8"])]
        img, planes, labels = tf(test_samples(sup, qry, 401, 401))
        gt = labels[0][None]
        hw = tuple(gt.shape[-2:])
        # the device-side preprocessing against the reference's tensors (Pillow + ToTensor + Normalize in the generator)
        assert torch.equal(img[0, :, ::5, ::5].cpu(), torch.from_numpy(g[f"e{e}_sup_rgb_s5"]))
        assert torch.equal(img[1, :, ::5, ::5].cpu(), torch.from_numpy(g[f"e{e}_qry_rgb_s5"]))
        assert np.array_equal(np.packbits(planes[0, 0].cpu().numpy().astype(np.uint8).reshape(-1)), g[f"e{e}_sup_fg_bits"])
        ins = (img[:1][None], planes[None], img[1:][None])
        with torch.no_grad():
            l1, r1 = s1(*ins, hw, ret_ind=True)
            p_logits = s1(*ins)
            prior = p_logits.argmax(dim=1, keepdim=True)
            ref_prior = np.unpackbits(g[f"e{e}_prior_bits"])[: 401 * 401].reshape(1, 401, 401)
            share_p = util.assert_argmax_exact(p_logits, ref_prior, what=f"real episode {e}: prior")
            # stage 2 is compared on the REFERENCE's prior (its own input), then run end to end on ours
            rp = torch.from_numpy(ref_prior.astype(np.int64))[:, None].to(dev)
            l2, r2 = s2(*ins, rp, hw, ret_ind=True)
            l2_own, _ = s2(*ins, prior, hw, ret_ind=True)
        for tag, logits in (("s1_", l1), ("s2_", l2)):
            k = f"e{e}_{tag}"
            lc = logits.cpu()
            lerr = (lc[0, :, ::3, ::3] - torch.from_numpy(g[k + "logits_s3"])).abs().max().item()
            assert lerr < util.LOGIT_TOL, (k, lerr)
            n = hw[0] * hw[1]
            ref_am = np.unpackbits(g[k + "argmax_bits"])[:n].reshape(1, *hw)
            share = util.assert_argmax_exact(lc, ref_am, what=k)
            cnt = util.counts(lc.argmax(1)[0].numpy(), gt[0].cpu().numpy())
            assert np.abs(cnt - g[k + "counts"]).max() <= share * n + 1e-9, (cnt, g[k + "counts"])
            loss = torch.nn.functional.cross_entropy(lc, gt.cpu(), ignore_index=255).item()
            assert abs(loss - float(g[k + "loss"])) < 2e-4
            print(f"real episode {e} {tag}: |dlogit| {lerr:.2e}, arg-max pixels inside the margin {share:.5f}, "
                  f"max count delta {int(np.abs(cnt - g[k + 'counts']).max())} of {n}, prior margin share {share_p:.5f}")
        # end to end on our own prior: the two priors differ only inside the margin, so the stage-2 logits stay close
        assert (l2_own - l2).abs().max().item() < 0.05 or share_p > 0
        pred_ev, loss_ev = ev.test_step(ins, gt[None])
        agree = (pred_ev[0] == l2_own.argmax(1)[0].cpu().numpy()).mean()
        assert agree > 0.9995 and abs(loss_ev - torch.nn.functional.cross_entropy(l2_own.cpu(), gt.cpu()).item()) < 2e-4
