#!/usr/bin/env python3
"""Headline benchmark: ICP iterations/sec (incl. kNN) of point-to-plane ICP on
synthetic clouds (BASELINE.json metric; BASELINE.md section 3 inputs).

    python bench.py [--gpus N --steps K --warmup W] [--points 10000000]

One "step" = one iteration of registration::RegistrationICP's loop
(registration.cu:155-163): 6x6 solve of the previous reduction -> compose T ->
radius 1-NN of every source point against the target LBVH under the new T ->
JtJ/Jtr reduction (+ its 256-byte D2H, + the RCCL all-reduce when N > 1).
Inputs are resident in HBM, LBVH built, before the timed region starts.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank holds
the full target and a spatial (Morton-contiguous) 1/N shard of the source;
the only collective is the per-iteration all-reduce of 32 doubles.  Total work
is fixed -> "scaling": "strong".

The timed region is K = --steps iterations between barriers; it is repeated --repeats times
(default 7) within one run and `value` is K / the MEDIAN window (min / max are reported too).
config.secondary carries what the headline does not show: the same loop on a noisy, partially
overlapping source, a cold whole call (tree build + staging + 30 iterations), the unseeded first
pass and the build times.  cpu_baseline.parity_sample compares the oracle's nearest neighbours
for its sample with the engine's, bit for bit.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synth(n, seed=42):
    """target U[0,1)^3 (seed 42), unit normals (seed 43),
    source = T_gt^-1 * target permuted (seed 44), r = 2 * n^(-1/3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tgt = rng.random((n, 3), dtype=np.float32)
    nrm = np.random.Generator(np.random.PCG64(seed + 1)).standard_normal((n, 3), dtype=np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    s = float(n) ** (-1.0 / 3.0)
    ang = 0.2 * s
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = 0.2 * s * np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0)
    Rinv, tinv = R.T, -R.T @ t
    src = (tgt.astype(np.float64) @ Rinv.T + tinv).astype(np.float32)
    perm = np.random.Generator(np.random.PCG64(seed + 2)).permutation(n)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = t
    return np.ascontiguousarray(src[perm]), tgt, nrm, T, 2.0 * s


def cpu_baseline(src, tgt, nrm, max_dist, n_total, engine_nn=None):
    """The oracle (a port, not the reference build) timed on this box's host cores
    over a bounded sample of the same workload.  engine_nn = (idx, d2) of the engine for the whole
    source under the identity: the sample's oracle neighbours are compared with it."""
    from oracle import oracle as orc
    # the WHOLE source (a 10M-point iteration of the port takes a couple of seconds on the GPU box's 128 threads);
    # only the single-thread figure is a sample, scaled
    n_whole = len(src)
    # (the one-thread figure too is MEASURED over the whole source since round 6 -- ~18 s at 10M points; until then it
    # was scaled up from the first 50k points)
    n_single = min(len(src), 10_000_000)
    build_s, iter_s, _, iter1_s = orc.bench_iteration(src, tgt, nrm, max_dist, n_whole, repeats=2,
                                                       n_single=n_single)
    per_iter = iter_s * (n_total / n_whole)
    per_iter1 = iter1_s * (n_total / n_single)
    n_sample = min(len(src), 1_000_000)     # (of the parity comparison below)
    out = {"value": round(1.0 / per_iter, 4), "unit": "iterations/s", "cores": orc.num_threads(),
           "kind": "port", "extrapolated": n_whole != n_total,
           # the reference's README quotes its CPU comparison single-threaded (README.md:124)
           "single_thread_value": round(1.0 / per_iter1, 5), "single_thread_extrapolated": n_single != n_total,
           "sample": "1 point-to-plane iteration (radius 1-NN + 6x6 accumulation + solve) over all %d source "
                     "points against the full %d-point target kd-tree, best of 2 (%.2f s); OpenMP over queries; "
                     "kd-tree build (%.1f s) excluded; single_thread_value: the same iteration on one thread "
                     "over the first %d points"
                     % (n_whole, len(tgt), iter_s, build_s, n_single)}
    if engine_nn is not None:
        # parity on the bench's own data: the oracle's neighbours of the sample (identity transform)
        # against the engine's -- d2 bit for bit, an index may differ only on an exact tie
        tree = orc.Tree(tgt)
        _, oi, od = tree.search_radius(src[:n_sample], max_dist, 1)
        tree.close()
        oi, od = oi[:, 0], od[:, 0]
        gi, gd = engine_nn[0][:n_sample], engine_nn[1][:n_sample]
        hit = oi >= 0
        diff = np.flatnonzero(gi != oi)
        dd = src[:n_sample][diff] - tgt[np.maximum(gi[diff], 0)]
        alt = (dd[:, 2] * dd[:, 2] + (dd[:, 1] * dd[:, 1] + dd[:, 0] * dd[:, 0])).astype(np.float32)
        out["parity_sample"] = {"n": int(n_sample), "transform": "identity", "matches": int(hit.sum()),
                                "hit_pattern_equal": bool(np.array_equal(gi < 0, oi < 0)),
                                "d2_bitexact": bool(np.array_equal(gd[hit], od[hit]) and np.isinf(gd[~hit]).all()),
                                "idx_mismatch": int(len(diff)),
                                "idx_mismatch_all_exact_ties": bool(np.array_equal(alt, od[diff]))}
    return out


def secondary(eng, src, tgt, nrm, d_tgt, d_nrm, d_src, max_dist, n, torch, _lib):
    """What the headline does not show (each a median of 3): noisy / partially overlapping source,
    cold whole call, first pass (no previous matches), build times.  Leaves the engine loaded with (tgt, src)."""
    med = lambda xs: float(np.median(xs))
    out = {}
    # -- cold call: tree build + source staging + 30 iterations, inputs resident on the device
    tt, ts, tc = [], [], []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.set_target(d_tgt, d_nrm)
        t1 = time.perf_counter()
        eng.set_source(d_src)
        t2 = time.perf_counter()
        eng.registration_icp(_lib.EST_POINT_TO_PLANE, max_dist, None, 0.0, 0.0, 30, -1.0)
        t3 = time.perf_counter()
        tt.append(t1 - t0), ts.append(t2 - t1), tc.append(t3 - t0)
    out["cold_30_iteration_call_ms"] = round(med(tc) * 1e3, 3)
    out["build_ms_target"] = round(med(tt) * 1e3, 3)
    out["build_ms_source"] = round(med(ts) * 1e3, 3)
    # -- a pass without previous matches (what every new pair of clouds pays once): from the root here -- the
    # exact clouds' loop never asked for halos --, from the queries' own seeds further down, once they exist
    kinds = {0: "from the root", 1: "seeded", 2: "own seeds (binary descent through the split planes) + seeded search"}

    def first_pass():
        fp = []
        for _ in range(3):
            eng.drop_seeds()
            p0 = eng.get_profile()
            eng.evaluate_registration(max_dist)
            p1 = eng.get_profile()
            fp.append(p1["nn_ms"] - p0["nn_ms"])
        return round(med(fp), 4), kinds.get(eng.last_search_kind(), "?")

    # -- VoxelDownSample of the target (voxel = 2.154 mean spacings: ~10 points per occupied
    # voxel, 0.01 at 10M points in the unit cube, the shape profiles/r06_voxel.txt is quoted on), clouds on the device,
    # host-visible wall time of the whole call (its one synchronisation included)
    vox = 2.154 * float(n) ** (-1.0 / 3.0)
    for key, nn_ in (("voxel_downsample_ms", None), ("voxel_downsample_with_normals_ms", d_nrm)):
        tv = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            v = eng.voxel_downsample(d_tgt, vox, nn_)
            torch.cuda.synchronize()
            tv.append(time.perf_counter() - t0)
        out[key] = round(med(tv[1:]) * 1e3, 4)
    out["voxel_downsample"] = "%d points, voxel %.4g -> %d voxels, %s path" % (
        n, vox, len(v[0]), {1: "dense-grid (voxel_dense.h)", 0: "general (radix passes)"}.get(eng._L.mi_icp_debug_last_voxel_path(eng._ctx), "?"))
    eng.set_profiling(True)
    out["first_pass_ms"], out["first_pass_kind"] = first_pass()
    # -- the same loop on data that looks like a sensor's: a random 60 % of the target as the source,
    # Gaussian noise of 0.15 mean spacings per coordinate (scripts/measure_noisy.py)
    rng = np.random.default_rng(5)
    keep = rng.random(n) < 0.6
    s = float(n) ** (-1.0 / 3.0)
    noisy = src[keep] + rng.normal(0.0, 0.15 * s, (int(keep.sum()), 3)).astype(np.float32)
    eng.set_source(torch.from_numpy(np.ascontiguousarray(noisy, np.float32)).cuda())
    eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
    eng.icp_iterate(10)
    rates, nn = [], []
    for _ in range(3):
        p0 = eng.get_profile()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.icp_iterate(20)
        torch.cuda.synchronize()
        rates.append(20.0 / (time.perf_counter() - t0))
        p1 = eng.get_profile()
        nn.append((p1["nn_ms"] - p0["nn_ms"]) / 20)
    out["noisy_sigma_0.15_it_per_s"] = round(med(rates), 1)
    out["noisy_sigma_0.15_nn_ms"] = round(med(nn), 4)
    out["noisy_sigma_0.15_source_points"] = int(keep.sum())
    # (that loop asked for the target's halos and got them: the first pass of the exact source again)
    eng.set_source(d_src)
    out["first_pass_with_halos_ms"], out["first_pass_with_halos_kind"] = first_pass()
    eng.set_profiling(False)
    # -- a TRANSIENT: what a caller of RegistrationICP sees on a new pair of clouds -- the whole source with the
    # same noise, started a rigid 1.5 spacings off (inside r = 2 s), 30 iterations with relative_* = 0: the
    # matches change for many iterations, the halos are not there yet when the loop starts.  Loop time only
    # (first pass, match-order re-sort, 30 iterations); clouds already staged.
    init = np.eye(4, dtype=np.float32)
    init[:3, 3] = (1.5 * s / np.sqrt(3.0)) * np.array([1.0, -1.0, 1.0], np.float32)
    ang = 0.5 * s
    init[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]], np.float32)
    rng = np.random.default_rng(6)
    noisy_all = (src + rng.normal(0.0, 0.15 * s, src.shape)).astype(np.float32)
    d_noisy = torch.from_numpy(np.ascontiguousarray(noisy_all)).cuda()
    tl, its = [], []
    for _ in range(3):
        eng.set_target(d_tgt, d_nrm)        # (a new pair: no halos, no previous matches)
        eng.set_source(d_noisy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = eng.registration_icp(_lib.EST_POINT_TO_PLANE, max_dist, init, 0.0, 0.0, 30, -1.0)
        torch.cuda.synchronize()
        tl.append(time.perf_counter() - t0)
        its.append(r.iterations)
    out["transient_30_iteration_loop_ms"] = round(med(tl) * 1e3, 3)
    out["transient_it_per_s"] = round(30.0 / med(tl), 1)
    out["transient"] = ("%d noisy source points (sigma = 0.15 spacings), init 1.5 spacings + %.3g rad off, %d iterations, "
                        "final fitness %.4f, rmse %.3g spacings" % (len(noisy_all), ang, int(its[-1]), r.fitness, r.inlier_rmse / s))
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_src)
    return out


def strong_big(eng, n_big, steps, warmup, rank, world, local, torch, dist, D, _lib):
    """The same strong-scaling bench at a size where an 8-way shard is still a large cloud:
    n_big-vs-n_big point-to-plane, BASELINE.md section 3's construction.  Generated on rank 0's GPU with torch's
    generator and broadcast (host memory and numpy would take minutes at 100M), sharded like the headline, timed the
    same way (5 windows of `steps`, median, max over ranks)."""
    dev = torch.device("cuda", local)
    s = float(n_big) ** (-1.0 / 3.0)

    class _StageFailed(Exception):
        pass

    def stage(name, fn):
        """Run one fallible stage; with several ranks AGREE on its outcome before anybody enters the next collective
        (a rank that ran out of memory here used to leave its peers waiting in a broadcast: ADVICE r4)."""
        err, val = "", None
        try:
            val = fn()
        except Exception as e:   # noqa: BLE001
            err = "%s: %s" % (name, e)
        if world > 1:
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                raise _StageFailed(err or "%s failed on another rank" % name)
        elif err:
            raise _StageFailed(err)
        return val

    def generate():
        if rank != 0:
            return tuple(torch.empty((n_big, 3), device=dev, dtype=torch.float32) for _ in range(3))
        g = torch.Generator(device=dev)
        g.manual_seed(4242)
        tgt = torch.rand((n_big, 3), generator=g, device=dev, dtype=torch.float32)
        nrm = torch.randn((n_big, 3), generator=g, device=dev, dtype=torch.float32)
        nrm /= torch.linalg.norm(nrm, dim=1, keepdim=True)
        ang = 0.2 * s
        ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        t = 0.2 * s * np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0)
        Rinv = torch.from_numpy(R.T.copy()).to(dev)
        tinv = torch.from_numpy(-R.T @ t).to(dev)
        src = torch.empty((n_big, 3), device=dev, dtype=torch.float32)
        for lo in range(0, n_big, 1 << 24):        # (in slices: the fp64 intermediate of the whole cloud is 2.4 GB)
            hi = min(n_big, lo + (1 << 24))
            src[lo:hi] = (tgt[lo:hi].double() @ Rinv.T + tinv).float()
        src = src[torch.randperm(n_big, generator=g, device=dev)]
        return tgt, nrm, src

    try:
        tgt, nrm, src = stage("generate", generate)
        if world > 1:
            for x in (tgt, nrm, src):
                dist.broadcast(x, src=0)

        def load():
            nonlocal src
            if world > 1:
                mine = D.device_shard_source(eng, src, rank, world)
                src = src[torch.from_numpy(mine).to(dev)].contiguous()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.set_target(tgt, nrm)
            eng.set_source(src)
            eng.synchronize()
            return (time.perf_counter() - t0) * 1e3

        build_ms = stage("build", load)

        def begin():
            if world > 1:
                eng.set_global_source_count(n_big)
            eng.set_profiling(False)
            eng.icp_begin(_lib.EST_POINT_TO_PLANE, 2.0 * s, None, -1.0)
            eng.icp_iterate(warmup)

        stage("begin", begin)
    except _StageFailed as e:
        return {"error": str(e)}
    # (five windows: a target that has been registered against for 40 iterations gets its halos built in the background --
    # at 100M points a build of tens of milliseconds that lands in the second and third window; the median stays clear)
    windows = []
    for _ in range(5):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = eng.icp_iterate(steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        w = time.perf_counter() - t0
        if world > 1:
            tw = torch.tensor([w], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            w = float(tw.item())
        windows.append(w)
    el = float(np.median(windows))
    return {"points": n_big, "n_gpus": world, "steps": steps, "value": round(steps / el, 2), "unit": "iterations/s",
            "ms_per_step": round(el / steps * 1e3, 4), "source_points_this_rank": int(len(src)), "build_ms": round(build_ms, 1),
            "final_fitness": round(float(res.fitness), 6),
            "note": "same bench, same sharding and exchange, %d-vs-%d points: the size at which one rank's share of an "
                    "8-way shard (%.1fM points) is still bandwidth-bound; compare with the N = 1 line's figure"
                    % (n_big, n_big, n_big / 8e6)}


_ABANDONED = []   # contexts whose RCCL call never returned: never closed (closing would wait for them)


def rccl_beside(D, Engine, local, rank, world, timeout_s=180.0):
    """The in-library RCCL path with `world` ranks -- ncclCommInitRank, the known-answer self-test and 200 timed
    ncclAllReduce of the 32 sums (mi_icp_comm_autotune) -- on a SCRATCH context and a helper thread, while the
    engine that runs the bench keeps its mailbox.  A communicator that does not form (or a collective that does not
    finish) within `timeout_s` is reported and left behind; the bench goes on, and leaves through os._exit at the end."""
    import threading
    from cupoch_amd.engine import comm_unique_id
    try:
        uid = D.exchange_unique_id(comm_unique_id, rank)   # (torch.distributed broadcast: on the main thread, every rank)
    except Exception as e:   # noqa: BLE001
        return {"error": "unique id: %s" % e}
    box = {}

    def work():
        try:
            e2 = Engine(local)
            box["engine"] = e2
            e2.comm_init(uid, world, rank)
            box["tune"] = e2.comm_autotune(200)
            e2.comm_destroy()
            e2.close()
            box["closed"] = True
        except Exception as e:   # noqa: BLE001
            box["error"] = str(e)

    os.environ["MI_ICP_NO_MAILBOX"] = "1"   # (read by mi_icp_comm_init at every call: RCCL alone on the scratch context)
    th = threading.Thread(target=work, daemon=True)
    t0 = time.perf_counter()
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        _ABANDONED.append((th, box))
        return {"error": "no answer from RCCL within %.0f s (communicator or collective): left behind" % timeout_s}
    os.environ.pop("MI_ICP_NO_MAILBOX", None)
    if "error" in box:
        if not box.get("closed") and "engine" in box:
            _ABANDONED.append((th, box))
        return {"error": box["error"]}
    tune = box["tune"]
    return {"latency_us": tune["latency_us"].get("rccl"), "comm_ranks": tune["rccl_comm_count"],
            "verified": tune["verified"], "exchanges": tune["exchanges"], "set_up_and_test_s": round(time.perf_counter() - t0, 2),
            "note": "ncclCommInitRank + known-answer self-test + timed ncclAllReduce of the 32 sums, in-library, on a scratch context"}


def self_launch(n_ranks):
    """Re-run this command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on a free local port
    (one rank per GPU; MI_ICP_BENCH_ONE_DEVICE=1: all on cuda:0).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--repeats", type=int, default=7, help="timed windows of --steps iterations; value = median")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--big-points", type=int, default=100_000_000,
                    help="size of secondary.strong_100M (the strong-scaling bench again at a size an 8-way shard still fills); 0: skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cupoch_amd import _lib
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own (the shape of the driver's N = 1 command): launch the N ranks here --
        # the same command the driver uses for N > 1 -- pass rank 0's single JSON line through, leave with their rc
        sys.exit(self_launch(args.gpus))
    # MI_ICP_BENCH_ONE_DEVICE=1: a rehearsal of the N > 1 path on a one-GPU box -- every rank on cuda:0, gloo for
    # the host-side collectives (RCCL cannot put two ranks on one device); the numbers mean nothing, the control flow does
    one_device = os.environ.get("MI_ICP_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))

    n = args.points
    src, tgt, nrm, T_gt, max_dist = synth(n)
    eng = Engine(local)
    if world > 1:
        # the rank's Morton-contiguous shard, cut from the order the engine computes on the device
        mine = D.device_shard_source(eng, torch.from_numpy(src).cuda(), rank, world)
        src_local = np.ascontiguousarray(src[mine])
    else:
        src_local = src
    # inputs resident in HBM before anything is timed
    d_tgt, d_nrm = torch.from_numpy(tgt).cuda(), torch.from_numpy(nrm).cuda()
    d_src = torch.from_numpy(src_local).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.set_target(d_tgt, d_nrm)
    eng.set_source(d_src)
    eng.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    host_allreduce = False
    begun = False
    exchange = None
    if world > 1:
        # How the ranks exchange the 32 sums is MEASURED (mi_icp_comm_autotune): the node's shared-memory mailbox
        # (host-memory words) and device inboxes over HIP IPC each run 200 exchanges of a known vector, checked exactly
        # and timed (max over ranks); the fastest that passed on every rank is used, one that fails is skipped.
        # Set-ups, in this order: the mailbox alone (shared memory + HIP IPC: nothing that can block on a network),
        # RCCL communicator + mailbox, and -- should neither come up on every rank -- a host-driven loop over
        # torch.distributed.  The in-library ncclAllReduce is self-tested and timed BESIDE the first one, on a scratch
        # context and under a watchdog (rccl_beside, below): its figure is reported, and a communicator that never
        # forms cannot hold the run.
        tried = []
        order = ("mailbox",) if one_device else ("mailbox", "rccl+mailbox")
        for attempt in order:
            ok, why, tune = 1, "", None
            try:
                if attempt == "mailbox":
                    D.init_engine_comm_local(eng, n)       # no RCCL in the library at all
                else:
                    D.init_engine_comm(eng, n)
                tune = eng.comm_autotune(200)
                eng.set_profiling(False)
                eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
                eng.icp_iterate(args.warmup)
            except Exception as e:   # noqa: BLE001 -- keep the scaling run alive, say what happened
                ok, why = 0, str(e)
            t = torch.tensor([ok], dtype=torch.int32, device=("cpu" if one_device else "cuda"))
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            tried.append({"setup": attempt, "ok": bool(int(t.item())), "why_not": why or None})
            if int(t.item()) == 1:
                begun = True
                exchange = dict(tune, setup=attempt)
                break
            if rank == 0:
                print("bench: exchange via %s unavailable (%s)" % (attempt, why or "failed on another rank"), file=sys.stderr)
            try:
                eng.comm_destroy()
            except Exception:   # noqa: BLE001
                pass
        if not begun:
            host_allreduce = True
            exchange = {"chosen": "host-driven torch.distributed all-reduce", "setup": "none"}
            eng.set_global_source_count(n)
            if rank == 0:
                print("bench: falling back to a host-driven loop with torch.distributed all-reduce", file=sys.stderr)
        elif (exchange.get("setup") == "mailbox" and os.environ.get("MI_ICP_BENCH_NO_RCCL") != "1" and
              (not one_device or os.environ.get("MI_ICP_BENCH_RCCL_BESIDE") == "1")):
            # (MI_ICP_BENCH_RCCL_BESIDE=1 in a one-device rehearsal: RCCL refuses several ranks on one GPU -- the error path)
            exchange["rccl_in_library"] = rccl_beside(D, Engine, local, rank, world,
                                                      float(os.environ.get("MI_ICP_BENCH_RCCL_TIMEOUT_S", "180")))
            if exchange["rccl_in_library"].get("latency_us") is not None:
                exchange["latency_us"]["rccl"] = exchange["rccl_in_library"]["latency_us"]
                exchange["rccl_comm_count"] = exchange["rccl_in_library"].get("comm_ranks")
        exchange["tried"] = tried
    elif os.environ.get("MI_ICP_BENCH_HOST_LOOP") == "1":
        host_allreduce = True      # exercises the fallback loop on one rank
    elif os.environ.get("MI_ICP_FORCE_COMM") == "2":
        # single-rank mailbox: the exchange's fixed cost (post, poll, read back through host memory) on a 1-GPU box
        os.environ["MI_ICP_MAILBOX_SOLO"] = "1"
        eng.comm_init_local("bench_solo_%d" % os.getpid(), 1, 0)
        eng.set_global_source_count(n)
    elif os.environ.get("MI_ICP_FORCE_COMM") == "1":
        # single-rank communicator: exercises the RCCL all-reduce path on a 1-GPU box
        from cupoch_amd.engine import comm_unique_id
        eng.comm_init(comm_unique_id(), 1, 0)
        eng.set_global_source_count(n)

    # det_thresh <= 0: at this size the fp32 determinant of JtJ overflows and the
    # reference's default check would reject every solve
    # The timed windows run WITHOUT per-kernel HIP events (two event pairs per iteration cost
    # ~12 us of a 0.18-ms step at N = 1 and far more of a sharded one); the kernels' average
    # durations come from one further window of the same K steps, bracketed by events on the
    # engine's stream, right after.
    eng.set_profiling(False)

    if host_allreduce:
        import ctypes as C

        class _Stepper:   # the host-driven loop behind the engine's stepping interface
            def __init__(self):
                self.loop = D.HostDrivenLoop(eng, _lib.EST_POINT_TO_PLANE, max_dist, n, world=world).begin()

            def iterate(self, k):
                self.loop.iterate(k)
                res = eng.evaluate_registration(max_dist, self.loop.T)
                res.fitness, res.inlier_rmse = self.loop.fitness, self.loop.inlier_rmse
                res.transformation = (C.c_float * 16)(*self.loop.T.T.reshape(-1))
                return res

        eng.icp_iterate = _Stepper().iterate           # same call shape below
        eng.icp_iterate(args.warmup)
    elif not begun:
        eng.icp_begin(_lib.EST_POINT_TO_PLANE, max_dist, None, -1.0)
        eng.icp_iterate(args.warmup)
    windows = []
    halo_builds_before = eng.get_profile()["halo_builds_by_loops"]
    for _ in range(max(1, args.repeats)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = eng.icp_iterate(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        w = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([w], dtype=torch.float64, device=("cpu" if one_device else "cuda"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        windows.append(w)
    elapsed = float(np.median(windows))
    # did the loop start a build of the target's halos (2 ms of GPU time at 10M points) inside the timed windows?
    halo_builds_in_windows = eng.get_profile()["halo_builds_by_loops"] - halo_builds_before
    eng.set_profiling(True)
    prof0 = eng.get_profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.icp_iterate(args.steps)
    torch.cuda.synchronize()
    profiled_window = time.perf_counter() - t0
    prof1 = eng.get_profile()
    eng.set_profiling(False)

    T = np.array(res.transformation, np.float32).reshape(4, 4).T
    err = float(np.linalg.norm(T - T_gt))
    nn_launches = prof1["nn_launches"] - prof0["nn_launches"]
    nn_ms = (prof1["nn_ms"] - prof0["nn_ms"]) / max(nn_launches, 1)
    red_launches = prof1["reduce_launches"] - prof0["reduce_launches"]
    # (sources of up to ~110k points run search + reduction + step as ONE kernel, timed under "search")
    red_ms = (prof1["reduce_ms"] - prof0["reduce_ms"]) / red_launches if red_launches > 0 else None
    ns_local, nt = len(src_local), len(tgt)
    alg_bytes = 20.0 * ns_local + 20.0 * nt       # kNN kernel, per launch
    achieved = alg_bytes / (nn_ms * 1e-3) / 1e9 if nn_ms > 0 else 0.0
    # HBM bytes of one launch of the search kernel from the PMC passes (separate rocprofv3 runs of this
    # very command, scripts/gpu_traffic.sh, with the FETCH_SIZE calibration applied); null when the
    # committed measurement is for another size / GPU count
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "nn_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("points") == n and tj.get("n_gpus", 1) == world:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_note = tj.get("note")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            # BASELINE.json's metric, verbatim, when run on its configuration (--points changes the name)
            "metric": ("ICP iterations/sec (incl. kNN) on 10M-pt point-to-plane, 1/2/4/8 MI355X" if n == 10_000_000 else
                       "ICP iterations/sec (incl. kNN), point-to-plane, %s-vs-%s points" % (_fmt(n), _fmt(n))),
            "value": round(args.steps / elapsed, 3),
            "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            # the K-step window was timed `repeats` times in this run; value / ms_per_step are its median
            "repeats": len(windows),
            "ms_per_step_min_max": [round(min(windows) / args.steps * 1e3, 4), round(max(windows) / args.steps * 1e3, 4)],
            "value_min_max": [round(args.steps / max(windows), 3), round(args.steps / min(windows), 3)],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s-vs-%s point-to-plane ICP, radius 1-NN on an 8-ary kd-cell tree, r=2*N^(-1/3), "
                                   "uniform random clouds (BASELINE.md section 3)" % (_fmt(n), _fmt(n)),
                       "points": n, "max_correspondence_distance": max_dist, "det_thresh": -1.0,
                       "parallelism": ("source sharded x%d (Morton-contiguous), target + tree replicated; 32 f64 summed over the "
                                       "ranks per iteration via %s -- chosen by a timed known-answer self-test of every "
                                       "available path, us per exchange (max over ranks): %s; in-library RCCL communicator of %s ranks"
                                       % (world, exchange.get("chosen"), json.dumps(exchange.get("latency_us")),
                                          exchange.get("rccl_comm_count")))
                       if world > 1 else "single GPU",
                       "exchange": exchange,
                       "accumulate": "f64", "build_ms": round(build_ms, 2),
                       "halo_builds_inside_the_timed_windows": int(halo_builds_in_windows),
                       "final_fitness": round(float(res.fitness), 6),
                       "final_rmse": float(res.inlier_rmse),
                       "T_error_fro_vs_ground_truth": err},
            "roofline": {"bound": "hbm", "kernel": "nn_packet_kernel",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms_avg": round(nn_ms, 4), "reduce_ms_avg": (round(red_ms, 4) if red_ms is not None else None),
                         "kernel_ms_source": "HIP events on the engine's stream around every search / reduction launch "
                                             "of one further window of the same %d steps (%.4f ms per step with the events; "
                                             "the timed windows run without them)" % (args.steps, profiled_window / args.steps * 1e3),
                         # a whole iteration moves 60 N_s + 20 N_t algorithmic bytes
                         "iteration": {"algorithmic_bytes": 60.0 * n + 20.0 * nt,
                                       "achieved": round((60.0 * n + 20.0 * nt) * args.steps / elapsed / 1e9, 2),
                                       "frac": round((60.0 * n + 20.0 * nt) * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5)}},
        }
        if world == 1 and not host_allreduce and not args.no_secondary:
            out["config"]["secondary"] = secondary(eng, src, tgt, nrm, d_tgt, d_nrm, d_src, max_dist, n, torch, _lib)
        if world == 1 and not args.no_cpu_baseline:
            eng.drop_seeds()
            gi, gd, _ = eng.search_radius_1nn(max_dist)        # the engine's neighbours under the identity
            out["cpu_baseline"] = cpu_baseline(src, tgt, nrm, max_dist, n, (gi, gd))
    # the strong-scaling bench once more at a size where an 8-way shard is still a large cloud (every rank takes part)
    big = None
    if args.big_points > 0 and n == 10_000_000 and not args.no_secondary and not host_allreduce and not one_device:
        try:
            big = strong_big(eng, args.big_points, args.steps, args.warmup, rank, world, local, torch, dist, D, _lib)
        except Exception as e:   # noqa: BLE001 -- a secondary figure must not cost the headline
            big = {"error": str(e)}
    if rank == 0:
        if big is not None:
            out["config"].setdefault("secondary", {})["strong_100M" if args.big_points == 100_000_000 else "strong_big"] = big
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if _ABANDONED:   # a helper thread is still inside RCCL: no teardown that could wait for it
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        eng.comm_destroy()
        dist.destroy_process_group()
    eng.close()


def _fmt(n):
    return "%dM" % (n // 1_000_000) if n % 1_000_000 == 0 else ("%dk" % (n // 1000) if n % 1000 == 0 else str(n))


if __name__ == "__main__":
    main()
