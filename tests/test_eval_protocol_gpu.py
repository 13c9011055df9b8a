"""The north-star parity statement on the aggregated metric: mIoU / bIoU of an evaluation round computed by
the HIP path (Evaluator.start_eval_loop: hipGraph replay + fused tail + device-side tp/fp/fn) versus the
CPU oracle (reference forward + FewShotMetric on the host) on identical synthetic episodes.
Tolerance: |d mIoU| <= 1e-4, |d bIoU| <= 1e-4 (BASELINE.json north_star); loss 1e-4."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

N_EPISODES = 30


def test_eval_round_miou_matches_cpu_oracle(hip_lib, dev):
    from oracle import ref_cpu
    from pemp_amd.entry import pemp_stage1 as e1
    from pemp_amd.networks import pemp_stage1 as m
    sd = util.wgen_state_dict("stage1_rn50")
    net = m.ModelClass(None)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    data = e1.SyntheticEpisodes(N_EPISODES, 5678, shot=1, split=0)
    ev = e1.Evaluator(net, dev)
    loss, miou_c, biou_c = ev.start_eval_loop(data, 20, 0, te_epochs=1)
    # oracle: same episodes through the CPU restatement + host metric
    torch.set_num_threads(16)
    data.reset_sampler()
    data.sample_tasks()
    metric = ref_cpu.FewShotMetric(20)
    losses = []
    fwd = lambda a, b, c, hw: ref_cpu.stage1_forward(sd, a, b, c, hw)
    with torch.no_grad():
        for i in range(N_EPISODES):
            inputs, qry_msk, cls = data.task(i)
            pred, l, _ = ref_cpu.test_step(fwd, inputs, qry_msk[0])
            metric.update(pred, qry_msk[0].numpy(), cls.tolist())
            losses.append(l)
    labels = e1.get_val_labels(0)
    ref_miou = metric.miou(labels)[1]
    ref_biou = metric.miou(labels, binary=True)[1]
    got_miou, got_biou = float(np.mean(miou_c)), float(np.mean(biou_c))
    print(f"mIoU hip {got_miou:.6f} ref {ref_miou:.6f}  bIoU hip {got_biou:.6f} ref {ref_biou:.6f}  "
          f"loss hip {loss:.6f} ref {np.mean(losses):.6f}")
    assert abs(got_miou - ref_miou) <= 1e-4
    assert abs(got_biou - ref_biou) <= 1e-4
    assert abs(loss - float(np.mean(losses))) <= 1e-4


def test_batched_evaluation_gives_identical_metrics(hip_lib, dev, exact_eval_variants):
    """start_eval_loop(batch=4) (several episodes per encoder pass, one tail launch per label size, ragged last
    group) returns exactly the metrics of the reference-style one-episode-per-step loop."""
    from pemp_amd.entry import pemp_stage1 as e
    net = e.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()
    res = []
    for batch in (1, 4):
        ev = e.Evaluator(net, device=dev)
        res.append(ev.start_eval_loop(e.SyntheticEpisodes(10, 5678, 1, split=0, height=97, width=97), 20, 0, te_epochs=1, batch=batch))
    (l0, m0, b0), (l1, m1, b1) = res
    assert l0 == l1 and np.array_equal(m0, m1) and np.array_equal(b0, b1)


def test_single_episode_steps_in_flight_give_identical_statistics(hip_lib, dev, exact_eval_variants):
    """The reference protocol (one episode per test_step) with 1 and with 4 steps in flight (Evaluator(lanes=4): engine
    replicas on their own HIP streams): the per-episode statistics rows are bit-identical, in order, for ragged label sizes;
    the evaluation loop returns identical metrics; and both equal the batched step."""
    from pemp_amd.entry import pemp_stage1 as e
    net = e.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()
    data = e.SyntheticEpisodes(11, 5678, 1, split=0, height=97, width=97)
    data.sample_tasks()
    eps = [data.task(i)[:2] for i in range(11)]
    rows = [e.Evaluator(net, device=dev, lanes=k).test_steps_device(eps).cpu() for k in (1, 4)]
    rows.append(e.Evaluator(net, device=dev).test_step_batch(eps).cpu())
    assert torch.equal(rows[0], rows[1]) and torch.equal(rows[0], rows[2])
    again = e.Evaluator(net, device=dev, lanes=4)
    assert torch.equal(again.test_steps_device(eps).cpu(), rows[0]) and torch.equal(again.test_steps_device(eps[::-1]).cpu(), rows[0].flip(0))
    res = [e.Evaluator(net, device=dev, lanes=k).start_eval_loop(e.SyntheticEpisodes(10, 5678, 1, split=0, height=97, width=97), 20, 0, te_epochs=2)
           for k in (1, 3)]
    (l0, m0, b0), (l1, m1, b1) = res
    assert l0 == l1 and np.array_equal(m0, m1) and np.array_equal(b0, b1)


def test_stage2_batched_evaluation_gives_identical_metrics(hip_lib, dev, exact_eval_variants):
    """The stage-2 evaluator (stage-1 prior -> stage 2) through the same sharded loop: batch 3 == batch 1."""
    from pemp_amd.entry import pemp_stage2 as e2
    s1 = e2.PriorNet(None)
    s1.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    s1 = s1.to(dev).eval()
    net = e2.ModelClass(1, 1, None)
    net.load_state_dict(util.wgen_state_dict("stage2_rn50cm", seed=4321))
    net = net.to(dev).eval()
    res = []
    for batch in (1, 3):
        ev = e2.Evaluator(s1, net, device=dev)
        res.append(ev.start_eval_loop(e2.SyntheticEpisodes(10, 5678, 1, split=0, height=97, width=97), 20, 0, te_epochs=1, batch=batch))
    (l0, m0, b0), (l1, m1, b1) = res
    assert l0 == l1 and np.array_equal(m0, m1) and np.array_equal(b0, b1)


def test_coco20i_round_matches_cpu_oracle(hip_lib, dev, monkeypatch):
    """BASELINE.json configs[4] (COCO-20i 1-shot): a COCO-shaped round -- 40 episodes, every one of the split's 20
    validation labels drawn, ground truth at the COCO picture formats up to 640x640, metric table [81, 3]
    (reference data_kits/datasets.py:99-100, core/metrics.py:7) -- through the sharded evaluator (one episode per
    step and 8 per step) versus the CPU oracle + the reference-style host metric.  |d mIoU|, |d bIoU| <= 1e-4,
    every per-class IoU finite."""
    from oracle import ref_cpu
    from pemp_amd import synth
    from pemp_amd.entry import pemp_stage1 as e1
    from pemp_amd.networks import pemp_stage1 as m
    n_eps, split = 40, 1                                   # split 1: labels 21..40 -- PASCAL-style ids would all be NaN
    sd = util.wgen_state_dict("stage1_rn50")
    net = m.ModelClass(None)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    nclass = e1.num_classes("COCO")
    labels = e1.get_val_labels(split, "COCO")
    assert nclass == 80 and labels == list(range(21, 41))
    from pemp_amd import ops
    res, per_class = [], []
    for batch, sk in ((1, False), (8, False), (1, True)):
        # sk False (the default): every evaluation conv variant is bit-identical, one episode per step == eight per step exactly;
        # sk True (opt-in): a one-episode step may run the 3x3 layers split along K -- same metrics to rounding
        monkeypatch.setattr(ops, "EVAL_SPLITK", sk)
        data = e1.SyntheticEpisodes(n_eps, 5678, shot=1, split=split, dataset="COCO")
        ev = e1.Evaluator(net, dev)
        res.append(ev.start_eval_loop(data, nclass, split, te_epochs=1, batch=batch, dataset_name="COCO"))
        per_class.append(ev.round_miou[0])
    (loss, miou_c, biou_c), (loss8, miou8, biou8), (loss_sk, miou_sk, biou_sk) = res
    assert loss == loss8 and np.array_equal(miou_c, miou8) and np.array_equal(biou_c, biou8)
    assert np.array_equal(per_class[0], per_class[1]) and per_class[0].shape == (20,)
    assert abs(loss_sk - loss) <= 1e-5 and np.abs(np.asarray(miou_sk) - np.asarray(miou_c)).max() <= 1e-4
    # both paths are held to the oracle: the exact one through the split-K one's bounds above, the split-K one directly below
    miou_c, biou_c, loss, per_class[0] = miou_sk, biou_sk, loss_sk, per_class[2]
    assert np.isfinite(per_class[0]).all() and np.isfinite(miou_c) and np.isfinite(biou_c)
    torch.set_num_threads(16)
    data = e1.SyntheticEpisodes(n_eps, 5678, shot=1, split=split, dataset="COCO")
    data.reset_sampler()
    data.sample_tasks()
    metric = ref_cpu.FewShotMetric(nclass)
    assert metric.stat.shape == (81, 3)
    losses, seen, sizes = [], set(), set()
    fwd = lambda a, b, c, hw: ref_cpu.stage1_forward(sd, a, b, c, hw)
    with torch.no_grad():
        for i in range(n_eps):
            inputs, qry_msk, cls = data.task(i)
            pred, l, _ = ref_cpu.test_step(fwd, inputs, qry_msk[0])
            metric.update(pred, qry_msk[0].numpy(), cls.tolist())
            losses.append(l)
            seen.add(int(cls[0]))
            sizes.add(tuple(qry_msk.shape[-2:]))
    assert seen == set(labels) and (640, 640) in sizes and sizes <= set(synth.QUERY_SIZES_COCO)
    ref_c, ref_miou = metric.miou(labels)
    ref_biou = metric.miou(labels, binary=True)[1]
    got_miou, got_biou = float(np.mean(miou_c)), float(np.mean(biou_c))
    print(f"COCO-20i mIoU hip {got_miou:.6f} ref {ref_miou:.6f}  bIoU hip {got_biou:.6f} ref {ref_biou:.6f}  "
          f"loss hip {loss:.6f} ref {np.mean(losses):.6f}")
    assert np.isfinite(ref_c).all()
    assert abs(got_miou - ref_miou) <= 1e-4 and abs(got_biou - ref_biou) <= 1e-4
    assert np.abs(per_class[0] - ref_c).max() <= 2e-4
    assert abs(loss - float(np.mean(losses))) <= 1e-4


def test_one_episode_step_with_split_k_matches_the_batched_step(hip_lib, dev, monkeypatch):
    """The opt-in fast path at the real shape (401 x 401): a one-episode step (5202 feature rows; the autotuner may pick
    the split-K conv variants, pemp_amd.ops.EVAL_SPLITK = True) against the same episode inside an 8-episode step (unsplit
    variants): feature-resolution logits within LOGIT_TOL / 20 (measured ~1e-5: only the summation order of the K slices
    differs), pixel counts within 0.1 % of the label, loss within 1e-5; and the one-episode result is bit-stable across
    replays of its hipGraph."""
    from pemp_amd import ops, synth
    from pemp_amd.networks import pemp_stage1 as m
    monkeypatch.setattr(ops, "EVAL_SPLITK", True)
    net = m.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()
    b = synth.make_batch([5678 + i for i in range(8)], shot=1, out_hw=(366, 500))
    t = lambda k: torch.from_numpy(b[k]).to(dev)
    sup, msk, qry, lab = t("sup_img"), t("sup_mask"), t("qry_img"), t("qry_mask")[:, 0]
    with torch.no_grad():
        pred8 = net.lowres(sup, msk, qry)[0].clone()
        _, st8, _ = ops.eval_tail(pred8, lab)
        st8 = st8.cpu().numpy()
        for i in (0, 5):
            p1 = net.lowres_graphed(sup[i:i + 1], msk[i:i + 1], qry[i:i + 1])[0].clone()
            p1b = net.lowres_graphed(sup[i:i + 1], msk[i:i + 1], qry[i:i + 1])[0].clone()
            assert torch.equal(p1, p1b)
            d = (p1 - pred8[i:i + 1]).abs().max().item()
            assert d <= util.LOGIT_TOL / 20, d
            _, st1, _ = ops.eval_tail(p1, lab[i:i + 1])
            st1 = st1.cpu().numpy()[0]
            assert abs(st1[0] / st1[1] - st8[i, 0] / st8[i, 1]) <= 1e-5
            assert np.abs(st1[2:] - st8[i, 2:]).max() <= 1e-3 * lab[i].numel()


def test_reference_test_step_body_takes_host_tensors(hip_lib, dev):
    """``Evaluator.test_step`` as the reference calls it (entry/pemp_stage1.py:48-53: host tensors in, numpy prediction and a float
    loss out): host inputs go straight into the hipGraph's static input buffers (one H2D copy each) -- same prediction and loss,
    bit for bit, as the same episode handed over on the device, pageable or pinned."""
    from pemp_amd.entry import pemp_stage1 as e
    net = e.ModelClass(None)
    net.load_state_dict(util.wgen_state_dict("stage1_rn50"))
    net = net.to(dev).eval()
    ev = e.Evaluator(net, device=dev)
    for seed in (3, 4):
        t = util.episode_tensors(seed, 1, 97, (80, 120))
        host = (t["sup_img"], t["sup_mask"], t["qry_img"])
        p0, l0 = ev.test_step(tuple(x.to(dev) for x in host), t["qry_mask"].to(dev))
        p1, l1 = ev.test_step(host, t["qry_mask"])
        p2, l2 = ev.test_step(tuple(x.pin_memory() for x in host), t["qry_mask"])
        assert isinstance(p1, np.ndarray) and p1.shape == (1, 80, 120) and isinstance(l1, float)
        assert np.array_equal(p0, p1) and np.array_equal(p0, p2) and l0 == l1 == l2
