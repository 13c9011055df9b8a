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


def test_batched_evaluation_gives_identical_metrics(hip_lib, dev):
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


def test_stage2_batched_evaluation_gives_identical_metrics(hip_lib, dev):
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
