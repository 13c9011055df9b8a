"""Episode input pipeline on the device (pemp_episode_preprocess) against the CPU oracle (oracle/pil_ops.py)
and against fixtures produced by Pillow itself (tests/golden/pil_ops.npz).  Byte / integer work and exactly
rounded fp32: every comparison is BIT-EXACT."""
import random
import zlib

import numpy as np
import pytest
import torch

from oracle import pil_ops as P
from tests import util

pytestmark = pytest.mark.gpu


def _tf(dev, h, w):
    from pemp_amd.data_kits.episode import EpisodeTransform
    return EpisodeTransform(h, w, device=dev)


def _u8(img_f32, mean, std):
    """invert ToTensor+Normalize exactly enough to recover the uint8 image (for comparisons in uint8)."""
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return np.rint((img_f32 * s + m) * 255).astype(np.uint8).transpose(1, 2, 0)


def test_resize_matches_pillow_fixtures(hip_lib, dev):
    from pemp_amd.data_kits.episode import MEAN, STD, Sample
    from pemp_amd.data_kits import synth_u8
    g = util.gold("pil_ops")
    for i in range(5):                                     # small cases: inputs stored in the fixture
        h, w = (int(v) for v in g[f"s{i}_hw"])
        img, planes, _ = _tf(dev, h, w)([Sample(g[f"s{i}_img"], g[f"s{i}_msk"], 1)])
        want = P.to_tensor_normalize(g[f"s{i}_bilinear"], MEAN, STD)
        assert np.array_equal(img[0].cpu().numpy(), want), i
        assert np.array_equal(planes[0].cpu().numpy(), P.support_mask_planes(g[f"s{i}_nearest"])), i
    for i in range(4):                                     # full-size cases: CRC of Pillow's output
        hs, ws, h, w = (int(v) for v in g[f"f{i}_dims"])
        src, msk = synth_u8.image(200 + i, hs, ws), synth_u8.mask(200 + i, hs, ws)
        img, planes, lab = _tf(dev, h, w)([Sample(src, msk, 1), Sample(None, msk, 2)])
        got = _u8(img[0].cpu().numpy(), MEAN, STD)
        assert zlib.crc32(got.tobytes()) == int(g[f"f{i}_bilinear_crc"]), i
        fg = (planes[0, 0].cpu().numpy() * 255).astype(np.uint8)
        assert zlib.crc32(fg.tobytes()) == int(g[f"f{i}_nearest_crc"]), i
        assert np.array_equal(lab[0].cpu().numpy(), (fg // 255).astype(np.int64))
        assert np.array_equal(img[0].cpu().numpy(), P.to_tensor_normalize(P.resize_bilinear(src, h, w), MEAN, STD))


def test_color_jitter_matches_pillow_fixtures(hip_lib, dev):
    from pemp_amd.data_kits.episode import MEAN, STD, Sample
    g = util.gold("pil_ops")
    src = g["j_img"]
    h, w = src.shape[:2]
    names = ("brightness", "contrast", "saturation")
    samples = []
    for i in range(4):
        order = tuple(names[t] for t in g[f"j{i}_order"])
        fac = dict(zip(names, (float(v) for v in g[f"j{i}_factors"])))
        samples.append(Sample(src, None, 0, (h, w), jitter=(order, fac)))
    img, _, _ = _tf(dev, h, w)(samples)
    for i in range(4):
        assert np.array_equal(img[i].cpu().numpy(), P.to_tensor_normalize(g[f"j{i}_out"], MEAN, STD)), i


def test_train_episode_matches_oracle_pipeline(hip_lib, dev):
    """Scale + jitter + flip + crop_obj window for supports, flip + jitter for the query, all in one batch of
    differently sized sources -- vs the oracle applying the same draws with numpy."""
    from pemp_amd.data_kits import synth_u8
    from pemp_amd.data_kits.episode import MEAN, STD, train_samples, _OPS
    H = W = 97
    rng = random.Random(7)
    sup = [(synth_u8.image(1, 120, 160), synth_u8.mask(1, 120, 160)), (synth_u8.image(2, 90, 75), synth_u8.mask(2, 90, 75))]
    qry = [(synth_u8.image(3, 200, 150), synth_u8.mask(3, 200, 150))]
    samples = train_samples(sup, qry, H, W, rng)
    assert any(s.flip for s in samples) or True
    img, planes, labels = _tf(dev, H, W)(samples)
    k = 0
    for n, s in enumerate(samples):
        sh, sw = s.scaled
        x = P.resize_bilinear(s.img, sh, sw)
        order, fac = s.jitter
        x = P.color_jitter(x, tuple(_OPS[o] - 1 for o in order), (fac["brightness"], fac["contrast"], fac["saturation"]))
        m = P.resize_nearest(s.mask, sh, sw)
        if s.flip:
            x, m = P.hflip(x), P.hflip(m)
        oy, ox = s.crop
        x, m = x[oy:oy + H, ox:ox + W], m[oy:oy + H, ox:ox + W]
        assert np.array_equal(img[n].cpu().numpy(), P.to_tensor_normalize(x, MEAN, STD)), n
        if s.mask_mode == 1:
            assert np.array_equal(planes[n].cpu().numpy(), P.support_mask_planes(m)), n
        else:
            assert np.array_equal(labels[k].cpu().numpy(), (m // 255).astype(np.int64)), n
            k += 1


def test_eval_episode_and_loader_overlap(hip_lib, dev):
    """test-time episode (query label at its own size) through the double-buffered loader; every batch equals
    the oracle, whatever the interleaving of the side stream."""
    from pemp_amd.data_kits import synth_u8
    from pemp_amd.data_kits.episode import MEAN, STD, EpisodeLoader, EpisodeTransform, test_samples as mk
    H = W = 97
    sizes = [(111, 140), (97, 97), (150, 100), (64, 200), (130, 131)]

    def batches():
        for b in range(5):
            eps = []
            for e in range(3):
                hs, ws = sizes[(b + e) % 5]
                s = 10 * b + e
                eps += mk([(synth_u8.image(s, hs, ws), synth_u8.mask(s, hs, ws))],
                          [(synth_u8.image(s + 500, ws, hs), synth_u8.mask(s + 500, ws, hs))], H, W)
            yield eps

    ref = list(batches())
    n = 0
    for (img, planes, labels), samples in zip(EpisodeLoader(batches(), EpisodeTransform(H, W, device=dev)), ref):
        torch.cuda.current_stream().synchronize()
        li = 0
        pi = 0
        for i, s in enumerate(samples):
            assert np.array_equal(img[i].cpu().numpy(), P.to_tensor_normalize(P.resize_bilinear(s.img, H, W), MEAN, STD))
            if s.mask_mode == 1:
                assert np.array_equal(planes[pi].cpu().numpy(), P.support_mask_planes(P.resize_nearest(s.mask, H, W)))
                pi += 1
            else:
                assert tuple(labels[li].shape) == s.mask.shape
                assert np.array_equal(labels[li].cpu().numpy(), (s.mask // 255).astype(np.int64))
                li += 1
        n += 1
    assert n == 5


def test_episode_bad_arguments(hip_lib, dev):
    from pemp_amd._lib import PempHipError
    from pemp_amd.data_kits import synth_u8
    from pemp_amd.data_kits.episode import Sample
    tf = _tf(dev, 97, 97)
    img = synth_u8.image(1, 50, 60)
    with pytest.raises(PempHipError, match="crop window"):
        tf([Sample(img, None, 0, (100, 100), crop=(10, 0))])
    with pytest.raises(PempHipError, match="contrast twice"):
        tf([Sample(img, None, 0, jitter=(("contrast", "contrast", "brightness"), dict(brightness=1., contrast=1., saturation=1.)))])
    with pytest.raises(PempHipError, match="shrinks"):
        tf([Sample(synth_u8.image(1, 97 * 40, 8), None, 0)])
    with pytest.raises(ValueError):
        tf([Sample(img.astype(np.float32), None, 0)])


def test_evaluator_on_decoded_episodes_equals_host_preprocessing(hip_lib, dev):
    """start_eval_loop over uint8 'decoded' episodes (device-side input pipeline, prefetch on a side stream)
    == the same loop fed with tensors the ORACLE preprocessed on the host: identical tp/fp/fn, loss, mIoU."""
    from pemp_amd.data_kits.episode import MEAN, STD
    from pemp_amd.entry import pemp_stage1 as e
    net = e.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()

    class Decoded(e.SyntheticDecodedEpisodes):
        SIZES = ((90, 120), (130, 101), (97, 97))

    class HostPreprocessed(Decoded):
        decoded_task = None

        def __getattribute__(self, name):                    # hide decoded_task -> the evaluator takes the tensor path
            if name == "decoded_task":
                raise AttributeError(name)
            return super().__getattribute__(name)

        def task(self, i):
            sup, qry, cls = Decoded.decoded_task(self, i)
            H, W = self.height, self.width
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
            sup_img = torch.stack([t(P.to_tensor_normalize(P.resize_bilinear(im, H, W), MEAN, STD)) for im, _ in sup])
            sup_msk = torch.stack([t(P.support_mask_planes(P.resize_nearest(lb, H, W))) for _, lb in sup])
            qry_img = torch.stack([t(P.to_tensor_normalize(P.resize_bilinear(im, H, W), MEAN, STD)) for im, _ in qry])
            return (sup_img[None], sup_msk[None], qry_img[None]), t((qry[0][1] // 255).astype(np.int64))[None, None], torch.tensor([cls])

    res = []
    for cls_ in (Decoded, HostPreprocessed):
        ev = e.Evaluator(net, device=dev)
        res.append(ev.start_eval_loop(cls_(6, 5678, 1, split=0, height=97, width=97), 20, 0, te_epochs=2))
    (l0, m0, b0), (l1, m1, b1) = res
    assert l0 == l1 and np.array_equal(m0, m1) and np.array_equal(b0, b1)


def test_training_main_on_decoded_episodes(hip_lib, dev):
    """Training harness fed by the device-side input pipeline (augmentation draws on the host, pixels on the GPU)."""
    from pemp_amd.entry import train_stage1 as t
    model = t.main(steps=3, bs=2, shot=1, lr=1e-3, seed=11, log_every=100, model="stage1", decoded=1, height=97, width=97)
    assert all(torch.isfinite(p).all() for p in model.parameters())
    # the generator alone: shapes and value ranges of one device batch
    from pemp_amd.data_kits.episode import EpisodeLoader, EpisodeTransform
    loader = EpisodeLoader(t.decoded_batches(2, 1, 1, 5, 0, 97, 97), EpisodeTransform(97, 97, device=dev))
    (sup, msk, qry), lab = next(t.device_batches(loader, 2, 1, 97, 97))
    assert tuple(sup.shape) == (2, 1, 3, 97, 97) and tuple(msk.shape) == (2, 1, 2, 97, 97) and tuple(lab.shape) == (2, 1, 97, 97)
    assert torch.equal(msk[:, :, 0] + msk[:, :, 1], torch.ones_like(msk[:, :, 0])) and set(lab.unique().tolist()) <= {0, 1}
    assert lab.dtype == torch.int64 and sup.abs().max() < 3.0


def test_visualize_writes_predictions_and_response_maps(hip_lib, dev, tmp_path):
    """evaluate_and_save (reference core/base_trainer.py:311-403): files, palette, Dice and ret_ind consistency."""
    import json
    from PIL import Image
    from pemp_amd.entry import pemp_stage1 as e
    net = e.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()

    class Small(e.SyntheticDecodedEpisodes):
        SIZES = ((90, 120), (97, 97))

    accs = e.evaluate_and_save(net, Small(2, 5678, 1, split=0, height=97, width=97), tmp_path, 2, device=dev)
    dirs = sorted(p for p in tmp_path.iterdir())
    assert len(dirs) == 2 and len(accs) == 2 and all(0.0 <= a <= 1.0 for a in accs)
    for d in dirs:
        meta = json.loads((d / "data.json").read_text())
        names = sorted(p.name for p in d.iterdir())
        assert len(names) == 7 and abs(float(meta["acc"]) - accs[dirs.index(d)]) < 1e-3
        cname = meta["cls_name"]
        pred = np.asarray(Image.open(d / f"{cname}_qry_pred_{meta['qry']}.png"))
        color = np.asarray(Image.open(d / f"{cname}_qry_color_{meta['qry']}.png"))
        lab = np.asarray(Image.open(d / f"{cname}_qry_msk_{meta['qry']}.png"))
        assert pred.shape == lab.shape and color.shape == lab.shape + (3,) and set(np.unique(pred)) <= {0, 255}
        # every pixel carries one of the six palette colours; foreground-prototype colours (rows 3..5) sit where the
        # prediction is foreground, up to the boundary band where nearest (response) and bilinear (logits) upsampling differ
        pal = e.RESPONSE_PALETTE_RGB
        assert (color[..., None, :] == pal[None, None]).all(-1).any(-1).all()
        is_fg_color = (color[..., None, :] == pal[3:][None, None]).all(-1).any(-1)
        assert (is_fg_color == (pred == 255)).mean() > 0.9
