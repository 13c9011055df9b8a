"""N > 1 evaluation path on CPU: two gloo ranks shard a round of episodes, reduce the metric table
once, and must reproduce the single-process mIoU bit for bit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _episode_counts(i):
    rng = np.random.RandomState(1000 + i)
    tp, fp, fn = rng.randint(0, 50000, 3)
    tp2, fp2, fn2 = rng.randint(0, 50000, 3)
    return np.array([tp, fp, fn, tp2, fp2, fn2], np.float64), 1 + i % 5, rng.rand()


def _round(n, rank, world):
    from pemp_amd.core.metrics import FewShotMetric
    from pemp_amd.entry.pemp_stage1 import allreduce_round, shard_indices
    m = FewShotMetric(20)
    loss, cnt = 0.0, 0
    for i in shard_indices(n, rank, world):
        c, cls, l = _episode_counts(i)
        m.update_counts(c[None], [cls])
        loss += l
        cnt += 1
    m.stat, loss, cnt = allreduce_round(m.stat, loss, cnt)
    return m.mIoU([1, 2, 3, 4, 5])[1], m.mIoU([1, 2, 3, 4, 5], binary=True)[1], m.stat.copy(), loss, cnt


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _round(n, rank, world)))
    finally:
        dist.destroy_process_group()


def test_two_rank_eval_round_equals_single_process():
    n = 37                                   # ragged: ranks get 19 and 18 episodes
    single = _round(n, 0, 1)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        miou, biou, stat, loss, cnt = got[r]
        assert miou == single[0] and biou == single[1]          # integer tables: exact
        assert np.array_equal(stat, single[2]) and cnt == n
        assert abs(loss - single[3]) < 1e-9


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pemp_amd.train_engine import allreduce_gradients
        from pemp_amd.entry.train_stage1 import broadcast_model
        g = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank-specific "gradients"
        scale = allreduce_gradients(g)
        lin = torch.nn.Linear(4, 3)
        torch.manual_seed(100 + rank)
        with torch.no_grad():
            lin.weight.normal_()
        broadcast_model(lin)
        q.put((rank, g.numpy().copy(), scale, lin.weight.detach().numpy().copy()))   # plain arrays: no fd passing
    finally:
        dist.destroy_process_group()


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pemp_amd.train_engine import GradBuckets
        n = 10000
        g = torch.zeros(n)
        bk = GradBuckets(g, cuts=[100, 3000, 6000, 6100, 9000], min_bytes=4000)      # 1000-float minimum
        out = [list(bk.buckets)]
        for step in range(2):                              # two steps: the launch state resets in finish()
            g.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1) + step)
            bk.enabled = True
            bk.ready_from(9500)                            # nothing lies wholly above 9500
            launched = [bk.next]
            bk.ready_from(9000)                            # the backward pass reports progress from the end ...
            bk.ready_from(6000)
            launched.append(bk.next)
            scale = bk.finish()                            # ... and finish() sends the rest and waits
            out.append((launched, scale, g.numpy().copy()))
        bk.enabled = False
        g.fill_(1.0)
        bk.ready_from(0)                                   # hooks of a trainer that never calls finish(): no-ops
        out.append((bk.next, float(g.sum())))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_overlapped_gradient_buckets():
    """GradBuckets: buckets formed from the end of the flat buffer, each all-reduced exactly once, in the same order
    on every rank, some during "backward" (ready_from) and the rest in finish()."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(10000, dtype=np.float32)
    for r in (0, 1):
        buckets, s0, s1, off = got[r]
        assert buckets == [(9000, 10000), (6100, 9000), (3000, 6100), (0, 3000)]      # 6000 / 100 would make tiny buckets
        for step, (launched, scale, g) in enumerate((s0, s1)):
            assert launched == [0, 2] and scale == 0.5
            assert np.array_equal(g, base * 3 + 2 * step)                              # every element summed exactly once
        assert off == (0, 10000.0)


def test_two_rank_gradient_bucket_allreduce_and_broadcast():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        r, g, scale, w = q.get(timeout=120)
        got[r] = (g, scale, w)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(1000, dtype=np.float32)
    for r in (0, 1):
        g, scale, w = got[r]
        assert scale == 0.5 and np.array_equal(g, base * 3) and np.array_equal(g * scale, base * 1.5)   # mean of ranks
    assert np.array_equal(got[0][2], got[1][2])                                                        # same weights


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N starts its own ranks (no external launcher); PEMP_BENCH_DRYRUN stands in for the GPU step
# ---------------------------------------------------------------------------------------------
def _run_bench(extra_env, *argv):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PEMP_BENCH_DRYRUN="1", PEMP_BENCH_BACKEND="gloo")
    env.update(extra_env)
    if not env.get("PEMP_BENCH_DRYRUN"):
        env.pop("PEMP_BENCH_DRYRUN")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=240)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no WORLD_SIZE in the environment) spawns two ranks, they rendezvous on 127.0.0.1,
    and rank 0 prints exactly ONE JSON line with n_gpus = 2 -- for the eval and the train mode alike."""
    import json
    for mode in ("eval", "train"):
        r = _run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", mode)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["config"]["mode"] == mode
        # MAX over ranks: rank 1 sleeps twice as long per step as rank 0 (4 ms vs 2 ms)
        assert out["ms_per_step"] >= 3.9
        _check_comm(out, 2, mode)


def _check_comm(out, world, mode):
    """The self-verifying part of an N-rank line: who took part, what was exchanged inside the timed region, how evenly the
    ranks ran -- and that the process group was gone before rank 0 went on alone."""
    comm = out["comm"]
    assert comm["backend"] == "gloo" and comm["world"] == world and comm["rccl_version"] is None
    per = comm["rank_ms_per_step"]
    assert len(per["all"]) == world and per["min"] == min(per["all"]) and per["max"] == max(per["all"])
    # every rank's OWN step time (before the collective couples them): rank r sleeps 2 (r + 1) ms per step
    # (lower bound exact -- a sleep never returns early --, upper bound generous: on a loaded 8-core container a 4 ms sleep was seen
    # to take 7.3; what the line must show is each rank's OWN time, i.e. the ranks in the order of their sleeps)
    assert all(2.0 * (r + 1) <= v < 2.0 * (r + 1) + 25.0 for r, v in enumerate(per["all"])), per
    assert world < 2 or per["all"][-1] > per["all"][0], per
    assert 1 <= comm["host_threads_per_rank"] <= max(1, len(os.sched_getaffinity(0)) // world)
    assert out["process_group_alive_at_print"] is False
    if mode == "eval":
        assert comm["collectives_in_timed_region"] == 1 and comm["allreduce_bytes_per_round"] == (21 * 3 + 2) * 8
        assert comm["table_equals_sum_of_rank_shards"] is True and comm["allreduce_ms"] >= 0
        m = out["miou"]
        assert m["episodes"] == out["steps"] * 25 * world and m["episodes_per_rank"] == [out["steps"] * 25] * world
        assert 0 < m["miou"] < 1 and 0 < m["biou"] < 1 and m["classes"] == 5
        # the same table from the rank shards, built here: the printed mIoU is the one of the SUM over ranks
        import bench
        from pemp_amd.core.metrics import FewShotMetric
        fm = FewShotMetric(20)
        for r in range(world):
            rows, cls = bench.dry_rows(out["steps"] * 25, r)
            fm.update_counts(rows[:, 2:].numpy(), cls.tolist())
        assert abs(fm.mIoU([1, 2, 3, 4, 5])[1] - m["miou"]) < 1e-6 and abs(fm.mIoU([1, 2, 3, 4, 5], binary=True)[1] - m["biou"]) < 1e-6


def test_bench_eight_rank_eval_line_is_self_verifying():
    """`bench.py --gpus 8` (dry run, gloo): the round's metric table is all-reduced inside the timed region and equals the sum
    of the eight rank shards; per-rank step times, thread cap, no process group left when rank 0 prints."""
    import json
    r = _run_bench({}, "--gpus", "8", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8
    _check_comm(out, 8, "eval")


def test_device_round_table_equals_host_metric():
    """DeviceRoundTable (rows -> class table with index_add, one packed vector) == FewShotMetric.update_counts on the host."""
    import bench
    from pemp_amd.core.metrics import FewShotMetric
    from pemp_amd.entry.pemp_stage1 import DeviceRoundTable
    rows, cls = bench.dry_rows(57, 3)
    t = DeviceRoundTable(20, torch.device("cpu"))
    t.add(rows[:30], cls[:30])
    t.add(rows[30:], cls[30:])
    assert t.allreduce() == 0                                    # no process group: nothing to reduce
    stat, loss_sum, count = t.fetch()
    fm = FewShotMetric(20)
    fm.update_counts(rows[:, 2:].numpy(), cls.tolist())
    assert np.array_equal(stat, fm.stat) and count == 57
    assert abs(loss_sum - float((rows[:, 0] / rows[:, 1]).sum())) < 1e-9
    t.reset()
    assert float(t.pack.abs().sum()) == 0.0


def test_gradient_bucket_schedule_of_the_real_stage1_layout():
    """The all-reduce schedule DESIGN.md section 6 promises, derived on the CPU from the real model: 11 955 392 gradient floats,
    FIVE buckets cut at residual-block starts, launched from the END of the buffer (purifier / ASPP first, ctr .. layer2 last,
    from finish())."""
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import GradBuckets, flat_layout, stage1_bucket_cuts
    net = m.ModelClass(None)
    params, offs, n = flat_layout(net)
    assert n == 11955392 and len(params) == 148 and all(o % 4 == 0 for o in offs)
    blocks, tail = stage1_bucket_cuts(net, params, offs)
    assert len(blocks) == 3 + 4 + 6 and blocks == sorted(blocks) and tail == 8542656 and tail > blocks[-1]
    bk = GradBuckets(torch.empty(n), blocks + [tail])
    assert bk.buckets == [(8542656, 11955392),      # purifier + ASPPV2: 13.65 MB, ready when the tail's backward is enqueued
                          (6308288, 8542656),       # layer3 blocks 4-5: 8.94 MB
                          (4073920, 6308288),       # layer3 blocks 2-3: 8.94 MB
                          (1446336, 4073920),       # layer3 blocks 0-1 + its downsample: 10.51 MB
                          (0, 1446336)]             # ctr, stem, layer1, layer2: 5.79 MB, sent by finish()
    assert sum(hi - lo for lo, hi in bk.buckets) == n and all(b[0] in blocks + [tail, 0] for b in bk.buckets)
    assert all((hi - lo) * 4 >= 8 << 20 for lo, hi in bk.buckets[:-1])


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies makes the whole job exit non-zero (the surviving rank is blocked in the rendezvous / a barrier
    and is terminated by the launcher); nothing is printed as a result line."""
    r = _run_bench({"PEMP_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited" in r.stderr


def test_bench_train_control_flow_two_ranks_with_the_rank0_roofline_pass():
    """`bench.py --mode train --gpus 2` through main_train's REAL control flow (PEMP_BENCH_STUB: the real
    Stage1Trainer.train_step / reduce_gradients / GradBuckets over a small CPU buffer, gloo): warm-up + timed steps with
    bucketed all-reduces on both ranks, then the roofline pass that rank 0 runs ALONE -- it must be collective-free
    (round 2: its bucket all-reduces paired with the other rank's barrier).  The job completes, the line carries
    `roofline`, and the weights after the timed steps are the data-parallel result."""
    import json
    r = _run_bench({"PEMP_BENCH_STUB": "1", "PEMP_BENCH_DRYRUN": "", "PEMP_BENCH_SILENCE_S": "60"},
                   "--gpus", "2", "--steps", "3", "--warmup", "2", "--mode", "train")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["episodes_per_step"] == 4
    assert "roofline" in out and "error" not in out["roofline"], out.get("roofline")
    # rank 0: one rank-local forward + backward first (kernel variants are timed by rank 0 alone and broadcast,
    # ops.tuned_by_rank0) -- NO optimizer step, and the replicas are re-synchronised after it -- then 5 steps of lr 0.1 on the
    # MEAN gradient of the ranks ((1 + 2) / 2): the data-parallel result, every all-reduce of the warm-up and of the timed
    # part paired up
    assert abs(out["config"]["stub_weight"] - (-0.1 * 1.5 * 5)) < 1e-6
    assert out["config"]["last_loss"] == 6.0          # rank 0 made exactly 1 + 5 calls before the roofline pass
    assert out["process_group_alive_at_roofline"] is False      # the roofline pass ran after destroy_process_group
    comm = out["comm"]
    assert comm["backend"] == "gloo" and comm["world"] == 2 and len(comm["rank_ms_per_step"]["all"]) == 2
    assert comm["gradient_elements"] == 1 << 16 and comm["allreduce_bytes_per_step"] == 4 << 16
    assert comm["buckets"] == [[49152, 65536], [32768, 49152], [16384, 32768], [0, 16384]] and comm["collectives_per_step"] == 4
    assert sum(comm["bucket_bytes"]) == comm["allreduce_bytes_per_step"] and comm["exposed_comm_ms"] > 0
    # every rank's host time per step, next to its step time: what tells an enqueue-bound rank from a communication-bound one
    he = comm["host_enqueue_ms_per_step"]
    assert len(he["all"]) == 2 and he["max"] == max(he["all"]) > 0 and 0 < he["of_step"] <= 1.05


def test_bench_train_graph_flag_two_ranks():
    """`bench.py --gpus 2 --mode train --train-graph` end to end on gloo: the launcher hands the flag to both ranks, the line
    says so (``config.hipgraph``), the gradient exchange is the one all-reduce BEHIND the replayed chain that the graph step
    makes (Stage1Trainer.train_step: the buckets are switched off while a hipGraph chain replays), and the data-parallel
    result is unchanged.  (The CPU stub has no hipGraphs: what runs here is the control flow around them.)"""
    import json
    r = _run_bench({"PEMP_BENCH_STUB": "1", "PEMP_BENCH_DRYRUN": "", "PEMP_BENCH_SILENCE_S": "60"},
                   "--gpus", "2", "--steps", "3", "--warmup", "2", "--mode", "train", "--train-graph")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["hipgraph"] is True
    assert abs(out["config"]["stub_weight"] - (-0.1 * 1.5 * 5)) < 1e-6
    assert len(out["comm"]["host_enqueue_ms_per_step"]["all"]) == 2 and len(out["comm"]["rank_ms_per_step"]["all"]) == 2


def test_bench_train_stub_roofline_pass_with_collectives_left_on_would_mismatch():
    """The guard the test above relies on: a rank-local step issues NO collective (collectives = False), a normal step does."""
    import bench
    import torch.distributed as dist
    calls = []
    tr = bench.stub_trainer(0)
    real_active = type(tr.eng.buckets).active
    try:
        type(tr.eng.buckets).active = lambda self: self.enabled          # pretend a 2-rank group exists
        tr.eng.buckets._launch = lambda lo, hi: calls.append((lo, hi))
        tr.eng.buckets.finish = lambda: calls.append("finish") or 0.5
        tr.train_step(*[torch.zeros(1)] * 4)
        assert len([c for c in calls if c != "finish"]) == 4
        calls.clear()
        tr.collectives = False
        tr.train_step(*[torch.zeros(1)] * 4)
        assert calls == []
    finally:
        type(tr.eng.buckets).active = real_active
    assert not dist.is_initialized()


def test_bench_launcher_watchdog_stops_a_silent_job():
    """A rank that never reaches the collective the others wait in: no heartbeat for PEMP_BENCH_SILENCE_S seconds ->
    the launcher terminates every rank and exits 124."""
    import time
    t0 = time.time()
    r = _run_bench({"PEMP_BENCH_HANG_RANK": "1", "PEMP_BENCH_SILENCE_S": "4"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert "no rank has made progress" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 120


def test_bench_under_an_external_launcher_uses_its_world_size():
    """Under torchrun-style environment variables bench.py does not spawn: a single rank with WORLD_SIZE=1 reports n_gpus 1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PEMP_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _buffer_sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from pemp_amd.core.base_trainer import TrainingLoop
        from pemp_amd.entry.train_stage1 import broadcast_model
        torch.manual_seed(100 + rank)                       # every rank starts (and drifts) differently
        model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8))
        prior = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4))
        broadcast_model(model)
        broadcast_model(prior)                              # run_training does this for trainer.stage1
        model.train()
        model(torch.randn(4, 3, 9, 9) * (1 + rank))         # rank-local batches move the running statistics apart
        before = model[1].running_mean.clone()
        loop = SimpleNamespace(trainer=SimpleNamespace(model=model))
        TrainingLoop.sync_buffers(loop)                     # what evaluation() / try_snapshot() call first
        q.put((rank, before.numpy(), model[1].running_mean.numpy().copy(), model[1].num_batches_tracked.item(),
               [p.detach().numpy().copy() for p in prior.parameters()]))
    finally:
        dist.destroy_process_group()


def test_two_rank_buffers_and_prior_network_are_synchronised():
    """BN running statistics are rank 0's on every rank before an evaluation / snapshot (each rank updates them from its
    own batches), and the frozen stage-1 prior network is identical on all ranks."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_buffer_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, b0, a0, n0, pr0), (_, b1, a1, n1, pr1) = res
    assert not np.allclose(b0, b1)                          # they had drifted apart ...
    assert np.array_equal(a0, b0) and np.array_equal(a1, b0) and n0 == n1     # ... and both hold rank 0's afterwards
    assert all(np.array_equal(x, y) for x, y in zip(pr0, pr1))


# ---------------------------------------------------------------------------------------------
# eight ranks (the node BASELINE.json's configs run on): ragged episode shards, ranks without an episode in the last batch,
# the gradient buckets, the rank-0 autotune broadcast
# ---------------------------------------------------------------------------------------------
def _eight_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pemp_amd import ops
        from pemp_amd.entry.pemp_stage1 import shard_indices
        from pemp_amd.train_engine import GradBuckets, allreduce_gradients
        res = {"round": _round(n, rank, world), "mine": list(shard_indices(n, rank, world))}
        # gradient buckets: the flat buffer finished from its end, four buckets, every rank the same schedule
        nfl = 1 << 14
        flat = torch.full((nfl,), float(rank + 1))
        b = GradBuckets(flat, [nfl // 4, nfl // 2, 3 * nfl // 4], min_bytes=nfl)
        b.enabled = True
        for lo in (3 * nfl // 4, nfl // 2, nfl // 4, 0):
            b.ready_from(lo)
        res["scale"] = b.finish()
        res["bucket_sum"] = (float(flat.min()), float(flat.max()), len(b.buckets))
        whole = torch.full((1000,), float(rank + 1))
        res["whole_scale"] = allreduce_gradients(whole)
        res["whole_sum"] = float(whole[0])
        # kernel picks: only rank 0 "times", everybody ends up with its choices
        calls = []

        def warm():
            calls.append(len(ops._TILE_CACHE))
            if rank == 0:
                ops._TILE_CACHE[("test-shape", 1)] = 34
                ops.WGRAD_PICKS[("test-wgrad", 2)] = (2, 512)
        ops.tuned_by_rank0(warm)
        res["picks"] = (ops._TILE_CACHE.get(("test-shape", 1)), ops.WGRAD_PICKS.get(("test-wgrad", 2)), calls)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_eight_rank_eval_round_buckets_and_autotune_broadcast():
    n, world = 37, 8                            # ranks 0..4 evaluate 5 episodes, ranks 5..7 four: the last "batch" is ragged
    single = _round(n, 0, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = sorted(i for r in range(world) for i in got[r]["mine"])
    assert seen == list(range(n)) and [len(got[r]["mine"]) for r in range(world)] == [5, 5, 5, 5, 5, 4, 4, 4]
    for r in range(world):
        miou, biou, stat, loss, cnt = got[r]["round"]
        assert miou == single[0] and biou == single[1] and np.array_equal(stat, single[2]) and cnt == n
        assert abs(loss - single[3]) < 1e-9
        assert got[r]["scale"] == 1.0 / world and got[r]["bucket_sum"] == (36.0, 36.0, 4)          # 1 + 2 + ... + 8
        assert got[r]["whole_scale"] == 1.0 / world and got[r]["whole_sum"] == 36.0
        tile, wg, calls = got[r]["picks"]
        assert tile == 34 and tuple(wg) == (2, 512) and len(calls) == 1
    # the seven other ranks warmed AFTER the broadcast: rank 0's pick was already in their cache when warm() ran
    assert all(got[r]["picks"][2][0] >= 1 for r in range(1, world))


def test_eight_rank_eval_with_fewer_episodes_than_ranks():
    """5 episodes on 8 ranks: three ranks have nothing to evaluate and still take part in the round's all-reduce."""
    n, world = 5, 8
    single = _round(n, 0, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        miou, biou, stat, loss, cnt = got[r]
        assert cnt == n and np.array_equal(stat, single[2]) and (miou == single[0] or (np.isnan(miou) and np.isnan(single[0])))
