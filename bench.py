#!/usr/bin/env python3
"""Headline benchmark: episodes/sec of the PEMP stage-1 evaluation step (BASELINE.json configs[1]:
pemp_stage1, PASCAL-5i-shaped 1-shot episodes, ResNet-50, 401x401) on N MI355X.

A "step" is one pass of the hot path -- Evaluator.test_step's device work (encoder, meta-prototype
module, cosine map, upsample + argmax + CE + IoU counts; reference entry/pemp_stage1.py:48-53) --
over ``--batch`` synthetic episodes that are already resident in HBM.  Ranks are independent
(episodes shard; no data-path collective), scaling is weak.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel (the fp32-MFMA implicit-GEMM conv) against the 157.3 TFLOP/s
                dense fp32 matrix peak; durations from HIP events around every conv launch of the
                same workload, taken in bench.py on the launch stream;
  cpu_baseline  the CPU oracle (oracle/ref_cpu.py, verified bit-equal to the reference here) timed on
                the host cores for a bounded sample of the same episodes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PEMP_BENCH_BATCH", "25")),
                    help="episodes per step (the reference evaluates 1 per step); 25 -> 50 x 2601 feature rows, "
                         "which fill the 256-row conv tiles and the 256 CUs almost exactly")
    ap.add_argument("--shot", type=int, default=1)
    ap.add_argument("--model", choices=("stage1", "stage2", "baseline", "panet"), default="stage1",
                    help="stage1 = headline; stage2 = stage-1 prior + stage-2 (use with --shot 5 for configs[3]); "
                         "baseline = Baseline VGG-16 (configs[0]); panet = PANet VGG-16 (the Baseline step + the alignment branch)")
    ap.add_argument("--mode", choices=("eval", "train"), default="eval",
                    help="eval (headline metric, BASELINE.json configs[1]) or train (configs[2])")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-episodes", type=int, default=12, help="bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--train-graph", action="store_true", help="--mode train: replay forward/backward from a hipGraph")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the staging-inclusive end_to_end figure")
    ap.add_argument("--dataset", choices=("PASCAL", "COCO"), default="PASCAL",
                    help="COCO: BASELINE.json configs[4] -- COCO-20i label set and picture formats (ground truth up to 640x640)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# N > 1 without an external launcher: `python bench.py --gpus N` starts its own ranks
# ---------------------------------------------------------------------------------------------
def launch_ranks(n):
    """Start ``n`` copies of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    as torchrun would), wait for them and return the exit code: 0 only if every rank exited 0.  Runs BEFORE this
    process makes any GPU call (it never makes one): the children are fresh processes, nothing is re-exec'ed.  A rank
    that dies takes the job down -- the survivors (blocked in a barrier) are terminated by PID."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PEMP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    try:
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in alive:
                        q.terminate()
    finally:
        for q in alive:
            q.kill()
    return rc


def dry_run(args, world, rank):
    """PEMP_BENCH_DRYRUN=1 (tests, CPU): the control flow of an N-rank run -- rendezvous, barriers, K timed steps,
    MAX over ranks, ONE line on rank 0 -- with a sleep in place of the GPU step.  PEMP_BENCH_FAIL_RANK=r makes rank r
    exit non-zero before the first barrier (the launcher must then fail the whole job)."""
    if os.environ.get("PEMP_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group(os.environ.get("PEMP_BENCH_BACKEND", "gloo"))
        dist.barrier()
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "episodes/sec (dry run: no GPU work)", "value": round(args.steps * args.batch * world / dt, 2),
                          "unit": "episodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run", "mode": args.mode}}))


def build_model(dev):
    from pemp_amd import synth
    from pemp_amd.networks import pemp_stage1 as m
    net = m.ModelClass(None)
    sd = synth.wgen_state_dict_for(net)
    net.load_state_dict(sd)
    return net.to(dev).eval(), sd


def episode_pool(dev, shot, batch, rank, n_groups=5):
    """n_groups batches of `batch` episodes; all episodes of a batch share one query size so that one
    fused tail launch serves the batch.  Seeds follow the evaluation sampler (test_seed = 5678)."""
    from pemp_amd import synth
    pool = []
    for g in range(n_groups):
        hw = synth.QUERY_SIZES[g % len(synth.QUERY_SIZES)]
        seeds = [5678 + 1000 * rank + g * batch + b for b in range(batch)]
        b = synth.make_batch(seeds, shot=shot, out_hw=hw)
        pool.append(dict(
            sup_img=torch.from_numpy(b["sup_img"]).to(dev), sup_mask=torch.from_numpy(b["sup_mask"]).to(dev),
            qry_img=torch.from_numpy(b["qry_img"]).to(dev), qry_mask=torch.from_numpy(b["qry_mask"][:, 0]).to(dev),
            seeds=seeds, hw=hw))
    return pool


def conv_roofline(net, pool, reps=3):
    """Per-launch HIP-event timing of every conv launch of one step (eager pass, same stream)."""
    from pemp_amd import ops
    records = []
    orig = ops.conv2d

    def timed(x, p, out=None, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, p, out=out, **kw)
        e1.record()
        n, ho, wo, co = y.shape
        cin_real = 3 if (p.stem and p.cin == 4 and not getattr(p, "real4", False)) else p.cin
        nbytes = 4.0 * (x.shape[0] * x.shape[1] * x.shape[2] * p.cin + n * ho * wo * co * (2 if kw.get("residual") is not None else 1)
                        + p.w.numel())
        records.append((e0, e1, 2.0 * n * ho * wo * co * p.kh * p.kw * cin_real, nbytes,
                        (n * ho * wo, co, p.kh * p.kw * cin_real, kw.get("residual") is not None)))
        return y

    cos_rec, cos_last = [], []
    orig_cos = ops.cosine_proto_max

    def timed_cos(qry, protos, dist_scalar, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_cos(qry, protos, dist_scalar, **kw)
        e1.record()
        b, h, w, c = qry.shape
        j = protos.shape[1]
        cos_rec.append((e0, e1, 4.0 * (b * h * w * c + protos.numel() + 2 * b * h * w), 2.0 * b * h * w * c * j))
        cos_last[:] = [(qry, protos, dist_scalar, kw)]
        return out

    ops.conv2d = timed
    ops.cosine_proto_max = timed_cos
    import pemp_amd.engine as eng
    eng.ops.conv2d = timed
    try:
        with torch.no_grad():
            for r in range(reps + 1):
                if r == 1:
                    records.clear()
                ep = pool[r % len(pool)]
                net.lowres(ep["sup_img"], ep["sup_mask"], ep["qry_img"])
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        eng.ops.conv2d = orig
        ops.cosine_proto_max = orig_cos
    cos = None
    if cos_rec:
        # a 25 us kernel bracketed by its own pair of events also measures ~4 us of launch gap: time 20 launches of the
        # step's last call back to back instead (same operands, one event pair) -- this agrees with rocprofv3's duration
        qry, protos, ds, kw = cos_last[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.no_grad():
            orig_cos(qry, protos, ds, **kw)
            e0.record()
            for _ in range(20):
                orig_cos(qry, protos, ds, **kw)
            e1.record()
        torch.cuda.synchronize()
        nbytes, nflop = cos_rec[-1][2], cos_rec[-1][3]           # algorithmic bytes / useful flops of one launch
        cms = e0.elapsed_time(e1) / 20
        gbs = nbytes / (cms * 1e-3) / 1e9
        cos = {"bound": "hbm", "kernel": "cosine_mfma_kernel (pixel x prototype cosine, MFMA outer product)",
               "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
               "avg_launch_us": round(cms * 1e3, 2),
               "useful_mfma_tflops": round(nflop / (cms * 1e-3) / 1e12, 3)}
    ms = sum(r[0].elapsed_time(r[1]) for r in records)
    flops = sum(r[2] for r in records)
    abytes = sum(r[3] for r in records)
    n = len(records)
    ach = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch from the PMC counters cannot be collected in-process; they are measured with
    # rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied) on this same
    # command and committed under profiles/.  Reported only when that file matches the batch size.
    traffic = None
    tf = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")
    if os.path.exists(tf):
        with open(tf) as f:
            rec = json.load(f)
        if rec.get("episodes_per_step") == len(pool[0]["seeds"]):
            traffic = rec.get("hbm_bytes_per_launch")
    # MFMA pipe utilisation from the PMC counters (rocprofv3 --pmc MfmaUtil on this command), likewise committed
    mfma_util = None
    mf = os.path.join(ROOT, "profiles", "r01_mfma_util.json")
    if os.path.exists(mf):
        with open(mf) as f:
            rec = json.load(f)
        if rec.get("episodes_per_step") == len(pool[0]["seeds"]):
            mfma_util = rec.get("conv_mfma_util_pct_time_weighted")
    # where the time goes: the six GEMM shapes (rows M, Cout N, K, with shortcut) with the largest share of conv time
    by = {}
    for r in records:
        t = by.setdefault(r[4], [0.0, 0.0])
        t[0] += r[0].elapsed_time(r[1])
        t[1] += r[2]
    top = sorted(by.items(), key=lambda kv: -kv[1][0])[:6]
    by_layer = [{"M": k[0], "N": k[1], "K": k[2], "shortcut": k[3], "share": round(v[0] / ms, 3),
                 "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1)} for k, v in top]
    return {"bound": "mfma", "kernel": "conv_dma_kernel + conv_igemm_kernel (all conv launches of a step)",
            "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "mfma_util_pmc_pct": mfma_util,
            "algorithmic_bytes_per_launch": int(abytes / n),
            "launches_per_step": n // reps, "avg_launch_us": round(ms * 1e3 / n, 2),
            "gflop_per_step": round(flops / reps / 1e9, 2), "conv_ms_per_step": round(ms / reps, 4),
            "by_layer": by_layer, "cosine_kernel": cos}


def cpu_baseline(sd, shot, n_eps):
    """Oracle test_step (forward + CE + argmax) on the host cores, bounded sample."""
    from oracle import ref_cpu
    from pemp_amd import synth
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    # a 1-GPU box owns a 16-core share of its host; more threads than that only oversubscribe it
    cores = max(1, min(cores, int(os.environ.get("PEMP_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    times = []
    with torch.no_grad():
        for i in range(n_eps + 1):
            ep = synth.make_episode(5678 + i, shot=shot, index=i)
            t = lambda a: torch.from_numpy(a)[None]
            sup, msk, qry, gt = t(ep["sup_img"]), t(ep["sup_mask"]), t(ep["qry_img"]), torch.from_numpy(ep["qry_mask"])
            t0 = time.time()
            fwd = lambda a, b, c, hw: ref_cpu.stage1_forward(sd, a, b, c, hw)
            ref_cpu.test_step(fwd, (sup, msk, qry), gt)
            dt = time.time() - t0
            if i > 0:                      # first episode warms the allocator / oneDNN primitives
                times.append(dt)
            if sum(times) > 30.0:
                break
    tot = sum(times)
    return {"value": round(len(times) / tot, 3), "unit": "episodes/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} episodes (seeds 5679..), oracle/ref_cpu.py test_step, torch {torch.__version__} CPU, "
                      f"{cores} threads, median {np.median(times) * 1e3:.0f} ms/episode"}


def end_to_end(net, args, dev):
    """Episodes/s INCLUDING input staging (SURVEY.md §8d ii): decoded uint8 samples in host memory -> one
    pinned blob per step -> async H2D -> Pillow-exact resize/normalise on the device (pemp_episode_preprocess)
    -> the same eval step, double-buffered on a side stream.  JPEG decode is not included (no dataset)."""
    from pemp_amd import ops
    from pemp_amd.data_kits import synth_u8
    from pemp_amd.data_kits.episode import EpisodeLoader, EpisodeTransform, test_samples
    B, S = args.batch, args.shot
    sizes = [(375, 500), (333, 500), (500, 375), (366, 500), (457, 500)]
    raw = []
    for g in range(3):
        hs, ws = sizes[g]
        imgs = [(synth_u8.image(10 * g + k, hs, ws), synth_u8.mask(10 * g + k, hs, ws)) for k in range(S + 1)]
        batch = []
        for _ in range(B):
            batch += test_samples(imgs[:S], imgs[S:], 401, 401)
        raw.append(batch)
    steps = max(6, min(args.steps, 20))
    ws_cache = {}

    def batches():
        for i in range(steps + 2):
            yield raw[i % len(raw)]

    def consume(out):
        img, planes, labels = out
        img = img.view(B, S + 1, 3, 401, 401)
        with torch.no_grad():
            pred, _ = net.lowres_graphed(img[:, :S].contiguous(), planes.view(B, S, 2, 401, 401), img[:, S:].contiguous())
            return ops.eval_tail(pred, torch.stack(labels), ws_cache=ws_cache)[1]

    loader = EpisodeLoader(batches(), EpisodeTransform(401, 401, device=dev))
    stats = [consume(next(loader)) for _ in range(2)]                      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for out in loader:
        stats.append(consume(out))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = torch.stack(stats).cpu().numpy()
    assert np.isfinite(st).all() and (st[..., 1] > 0).all()
    per_ep = sum(s.img.size + (s.mask.size if s.mask is not None else 0) for s in raw[0]) / B
    return {"value": round(steps * B / dt, 2), "unit": "episodes/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "h2d_bytes_per_episode": int(per_ep),
            "path": "host uint8 (decoded) -> pinned blob -> async H2D -> device resize/normalise -> eval step; "
                    "double-buffered on a side stream; JPEG decode excluded"}


def main_train(args, world, rank, dev):
    """--mode train: Trainer.train_step (reference entry/pemp_stage1.py:57-65) on `--batch` episodes per
    rank (the reference's data.bs = 4), data-parallel: one flat-gradient all-reduce per step over RCCL."""
    from pemp_amd import synth
    from pemp_amd.networks import pemp_stage1 as m
    from pemp_amd.train_engine import Stage1Trainer
    net = m.ModelClass(None)
    net.load_state_dict(synth.wgen_state_dict_for(net))
    # eager by default: the weight-gradient kernels run on a side stream concurrently with the input-gradient chain
    # (21.4 ms/step); a hipGraph replay of the same two-stream capture does not overlap its branches (23.6 ms/step)
    use_graph = args.train_graph
    if args.model == "stage2":          # frozen stage-1 prior + stage-2 step (entry/pemp_stage2.py:72-83)
        from pemp_amd.networks import pemp_stage2 as m2
        from pemp_amd.train_stage2 import Stage2Trainer
        net2 = m2.ModelClass(args.shot, 1, None)
        net2.load_state_dict(synth.wgen_state_dict_for(net2, seed=4321))
        tr = Stage2Trainer(net.to(dev).eval(), net2, device=dev, use_graph=use_graph)
    else:
        tr = Stage1Trainer(net, device=dev, use_graph=use_graph)
    B = args.batch
    pool = []
    for g in range(3):
        b = synth.make_batch([1234 + 1000 * rank + g * B + i for i in range(B)], shot=args.shot, out_hw=(401, 401))
        pool.append(tuple(torch.from_numpy(b[k]).to(dev) for k in ("sup_img", "sup_mask", "qry_img")) +
                    (torch.from_numpy(b["qry_mask"][:, 0]).to(dev),))
    for i in range(args.warmup):
        tr.train_step(*pool[i % len(pool)])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    losses = []
    for i in range(args.steps):
        losses.append(tr.train_step(*pool[i % len(pool)]))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ls = torch.stack(losses).cpu().numpy()
    assert np.isfinite(ls).all()
    if rank == 0:
        gflop = 3 * 2 * 64.94 * B                       # fwd + dgrad + wgrad, 2 images/episode, GFLOP
        s2 = args.model == "stage2"
        print(json.dumps({
            "metric": "train episodes/sec (PEMP %s train_step, ResNet-50, 1-shot, 401x401)" % ("stage-2" if s2 else "stage-1"),
            "value": round(args.steps * B * world / dt, 2), "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("pemp_stage2 train_step (stage-1 prior, batch-stat BN, CM, Dropout2d 0.5, CE, SGD), "
                                    if s2 else "pemp_stage1 train_step (batch-stat BN, DropBlock 0.1, CE, clip 1.1, SGD), ") +
                                   "%d episodes/rank/step" % B, "episodes_per_step": B, "shot": args.shot,
                       "first_loss": round(float(ls[0]), 5), "last_loss": round(float(ls[-1]), 5),
                       "effective_tflops": round(gflop * world / (dt / args.steps) / 1e3, 2)}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.mode == "train" and "--batch" not in sys.argv:           # the reference trains with data.bs = 4
        args.batch = 4
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:              # no launcher around us: be the launcher
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    if os.environ.get("PEMP_BENCH_DRYRUN"):
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # PEMP_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share
    # devices round-robin; the numbers mean nothing then).  The driver's runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("PEMP_BENCH_BACKEND", "nccl")
    first_local = local == 0
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from pemp_amd import build, ops
    if first_local:
        build.build()                 # normally a no-op: the prebuilt .so travels with the snapshot
    if world > 1:
        dist.barrier()
    if args.mode == "train":
        return main_train(args, world, rank, dev)
    if args.model in ("baseline", "panet"):        # BASELINE.json configs[0]: Baseline, VGG-16, 1-shot (PANet: same encoder)
        from pemp_amd import synth
        from pemp_amd.networks import baseline as mb, panet as mp
        net = mb.Baseline(None, backbone="vgg16") if args.model == "baseline" else mp.PANet(None, backbone="vgg16")
        sd = synth.wgen_state_dict_for(net)
        net.load_state_dict(sd)
        net = net.to(dev).eval()
    else:
        net, sd = build_model(dev)
    pool = episode_pool(dev, args.shot, args.batch, rank)
    ws = {}
    stats_log = torch.zeros((args.steps, args.batch, 8), dtype=torch.float64, device=dev)

    aux_log, ws_align = [], {}
    stage2 = None
    if args.model == "stage2":          # BASELINE.json configs[3]: stage-1 prior + stage-2 (ResNet-50 + CM)
        from pemp_amd import synth
        from pemp_amd.networks import pemp_stage2 as m2
        stage2 = m2.PEMPStage2(args.shot, 1, None)
        stage2.load_state_dict(synth.wgen_state_dict_for(stage2, seed=4321))
        stage2 = stage2.to(dev).eval()

    def step(i, log=True):
        ep = pool[i % len(pool)]
        ins = (ep["sup_img"], ep["sup_mask"], ep["qry_img"])
        with torch.no_grad():
            pred, _ = net.lowres(*ins) if args.no_graph else net.lowres_graphed(*ins)
            if stage2 is not None:
                prior, _, _ = ops.eval_tail(pred, None, out_hw=ins[0].shape[-2:], ws_cache=ws)
                prior = prior.unsqueeze(1).float()
                pred, _ = stage2.lowres(*ins, prior) if args.no_graph else stage2.lowres_graphed(*ins, prior)
            if args.model == "panet":       # auxiliary prototype-alignment loss of every episode (entry/panet.py:51-57)
                from pemp_amd.networks.panet import align_forward
                aux_log.append(align_forward(net._last_feats, pred, ins[1], ins[0].shape[0], args.shot, 1, 20, ws_align)["loss"])
            am, stats, _ = ops.eval_tail(pred, ep["qry_mask"], ws_cache=ws)
        if log:
            stats_log[i].copy_(stats)
        return am

    for i in range(args.warmup):
        step(i, log=False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # sanity on the logged statistics (the work really happened): finite losses, counts add up
    st = stats_log.cpu().numpy()
    assert np.isfinite(st).all() and (st[..., 1] > 0).all(), "eval tail produced invalid statistics"
    mean_loss = float((st[..., 0] / st[..., 1]).mean())

    out = None
    if rank == 0:
        eps_total = args.steps * args.batch * world
        out = {
            "metric": "episodes/sec (%s eval step, PASCAL-5i-shaped %d-shot, %s)" % (
                "Baseline" if args.model == "baseline" else "PANet" if args.model == "panet" else "PEMP stage-1" if stage2 is None else "PEMP stage-1 prior + stage-2",
                args.shot, "VGG-16" if args.model in ("baseline", "panet") else "ResNet-50"),
            "value": round(eps_total / dt, 2), "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s eval test_step, %s, %d-shot, 401x401, %d episode(s)/step, "
                                   "synthetic E(seed) episodes + Wgen(1234) weights" % (
                                       args.model if args.model in ("baseline", "panet") else "pemp_" + args.model,
                                       "VGG-16" if args.model in ("baseline", "panet") else "ResNet-50", args.shot, args.batch),
                       "episodes_per_step": args.batch, "shot": args.shot, "hipgraph": not args.no_graph,
                       "mean_ce_loss": round(mean_loss, 6)},
        }
        # the auxiliary measurements must never cost the headline line: a failure is reported in place
        def guarded(key, fn):
            try:
                out[key] = fn()
            except Exception as exc:  # noqa: BLE001
                out[key] = {"error": f"{type(exc).__name__}: {exc}"}

        if not args.no_roofline and stage2 is None:
            guarded("roofline", lambda: conv_roofline(net, pool))
            if args.model != "stage1" and "traffic" in out["roofline"]:
                out["roofline"]["traffic"] = None        # the committed PMC run is the stage-1 workload
        if world == 1 and args.model == "stage1" and not args.no_graph and not args.no_e2e:
            guarded("end_to_end", lambda: end_to_end(net, args, dev))
        if world == 1 and args.cpu_episodes > 0 and args.model not in ("baseline", "panet"):
            guarded("cpu_baseline", lambda: cpu_baseline({k: v.cpu() for k, v in sd.items()}, args.shot, args.cpu_episodes))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
