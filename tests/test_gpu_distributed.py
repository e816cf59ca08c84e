"""GPU: the N > 1 path on the HIP kernels.

Two PROCESSES share GPU 0, each holds the full target and its Morton-contiguous shard of the
source (cut from the order the engine computes on the device), every iteration all-reduces the
32 doubles over torch.distributed (gloo here -- RCCL cannot put two ranks on one device) and
solves the same 6x6: final transformation, statistics and the gathered correspondence set must
equal the single-process run.  Plus the in-library RCCL all-reduce on a one-rank communicator."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, make_pair

pytestmark = pytest.mark.gpu
PT2PL = 2
N, ITER = 60000, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    d = make_pair(N, seed=13, noise=0.03)
    eng = Engine(0)
    src_dev = torch.from_numpy(d["src"]).cuda()
    mine = D.device_shard_source(eng, src_dev, rank, world)          # Morton order computed on the device
    box = d["src"][mine]
    assert np.prod(box.max(0) - box.min(0)) < 0.8                      # a spatial shard, not a random half
    eng.set_target(torch.from_numpy(d["tgt"]).cuda(), torch.from_numpy(d["tgt_nrm"]).cuda())
    eng.set_source(src_dev[torch.from_numpy(mine).cuda()])
    loop = D.HostDrivenLoop(eng, PT2PL, d["max_dist"], N).begin()
    loop.iterate(ITER)
    allc = D.gather_correspondences(eng.get_correspondences(), mine)
    np.save(os.path.join(out_dir, "T_%d.npy" % rank), loop.T)
    np.save(os.path.join(out_dir, "stat_%d.npy" % rank), np.array([loop.fitness, loop.inlier_rmse]))
    np.save(os.path.join(out_dir, "corr_%d.npy" % rank), allc)
    np.save(os.path.join(out_dir, "idx_%d.npy" % rank), mine)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_equal_the_single_process_run(tmp_path):
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = make_pair(N, seed=13, noise=0.03)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    # the same host-driven arithmetic on one rank ...
    one = D.HostDrivenLoop(eng, PT2PL, d["max_dist"], N, world=1).begin()
    one.iterate(ITER)
    cor1 = eng.get_correspondences()
    # ... and the device-resident loop
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER, -1.0)
    Tdev = np.array(res.transformation, np.float32).reshape(4, 4).T
    T0, T1 = np.load(tmp_path / "T_0.npy"), np.load(tmp_path / "T_1.npy")
    np.testing.assert_array_equal(T0, T1)                                  # all ranks take identical steps
    assert np.linalg.norm(T0 - one.T) <= 1e-6 and np.linalg.norm(T0 - Tdev) <= 1e-6
    s0 = np.load(tmp_path / "stat_0.npy")
    assert s0[0] == pytest.approx(one.fitness, abs=1e-6) and s0[1] == pytest.approx(one.inlier_rmse, rel=1e-5)
    assert s0[0] == pytest.approx(res.fitness, abs=1e-6)
    c0, c1 = np.load(tmp_path / "corr_0.npy"), np.load(tmp_path / "corr_1.npy")
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(c0, cor1)                                 # = the single-process set, same order
    i0, i1 = np.load(tmp_path / "idx_0.npy"), np.load(tmp_path / "idx_1.npy")
    assert len(np.intersect1d(i0, i1)) == 0 and len(i0) + len(i1) == N
    eng.close()


def test_in_library_rccl_allreduce_on_a_one_rank_communicator():
    """mi_icp_comm_init + the per-iteration ncclAllReduce(double, 32) on the engine's stream: with a
    single rank the sum is the identity, so the loop must return exactly what it returns without."""
    from cupoch_amd.engine import Engine, comm_unique_id
    d = make_pair(40000, seed=5, noise=0.02)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    ref = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 6, -1.0)
    eng.comm_init(comm_unique_id(), 1, 0)
    eng.set_global_source_count(len(d["src"]))
    got = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 6, -1.0)
    np.testing.assert_array_equal(np.array(got.transformation), np.array(ref.transformation))
    assert got.fitness == ref.fitness and got.inlier_rmse == ref.inlier_rmse
    eng.comm_destroy()
    eng.close()


def _worker_mailbox(rank, world, job, out_dir, mode="host"):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MI_ICP_MAILBOX"] = mode
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine
    torch.cuda.set_device(0)
    d = make_pair(N, seed=13, noise=0.03)
    eng = Engine(0)
    src_dev = torch.from_numpy(d["src"]).cuda()
    mine = D.device_shard_source(eng, src_dev, rank, world)
    eng.set_target(torch.from_numpy(d["tgt"]).cuda(), torch.from_numpy(d["tgt_nrm"]).cuda())
    eng.set_source(src_dev[torch.from_numpy(mine).cuda()])
    eng.comm_init_local(job, world, rank)                  # the shared-memory mailbox, no RCCL
    assert eng.comm_kind() == (3 if mode == "device" else 2)
    eng.set_global_source_count(N)
    out = {}
    # the device-resident loop: point-to-plane (exchange + step inside the reduction's finishing block) ...
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER, -1.0)
    out["T_pl"] = np.array(res.transformation, np.float32)
    out["stat_pl"] = np.array([res.fitness, res.inlier_rmse, res.iterations])
    # ... point-to-point (generic reduction: the exchange opens the step kernel) ...
    res = eng.registration_icp(1, d["max_dist"], None, 0.0, 0.0, ITER, -1.0)
    out["T_pp"] = np.array(res.transformation, np.float32)
    out["stat_pp"] = np.array([res.fitness, res.inlier_rmse, res.iterations])
    # ... and a one-shot evaluation (the exchange as a kernel of its own)
    ev = eng.evaluate_registration(d["max_dist"], None)
    out["eval"] = np.array([ev.fitness, ev.inlier_rmse])
    np.savez(os.path.join(out_dir, "mail_%d.npz" % rank), **out)
    eng.comm_destroy()
    eng.close()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_mailbox_exchange_two_processes_on_one_gpu(tmp_path, mode):
    """The node-local communicator (csrc/mailbox.h): two processes share GPU 0, each runs the
    DEVICE-resident loop on its shard; every evaluation's 32 sums are exchanged by the kernels themselves --
    through the mailbox in shared host memory, or (MI_ICP_MAILBOX=device) by writing them into each other's
    inboxes in device memory, opened through HIP IPC.  Both ranks must end with bit-identical results, equal
    (to rounding of the sums' order) to the single-process loop."""
    from cupoch_amd.engine import Engine
    world = 2
    job = "test_%d_%d" % (os.getpid(), _free_port())
    mp.spawn(_worker_mailbox, args=(world, job, str(tmp_path), mode), nprocs=world, join=True)
    d = make_pair(N, seed=13, noise=0.03)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    r_pl = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER, -1.0)
    r_pp = eng.registration_icp(1, d["max_dist"], None, 0.0, 0.0, ITER, -1.0)
    ev = eng.evaluate_registration(d["max_dist"], None)
    eng.close()
    m0, m1 = np.load(tmp_path / "mail_0.npz"), np.load(tmp_path / "mail_1.npz")
    for k in m0.files:
        np.testing.assert_array_equal(m0[k], m1[k], err_msg=k)          # identical steps on every rank
    for key, ref in (("pl", r_pl), ("pp", r_pp)):
        T = m0["T_" + key]
        assert np.linalg.norm(T - np.array(ref.transformation, np.float32)) <= 1e-6, key
        st = m0["stat_" + key]
        assert st[0] == pytest.approx(ref.fitness, abs=1e-6) and st[1] == pytest.approx(ref.inlier_rmse, rel=1e-5)
        assert int(st[2]) == ref.iterations
    assert m0["eval"][0] == pytest.approx(ev.fitness, abs=1e-6) and m0["eval"][1] == pytest.approx(ev.inlier_rmse, rel=1e-5)


def _worker_big(rank, world, job, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine
    torch.cuda.set_device(0)
    d = make_pair(N_BIG, seed=29, noise=0.2)
    eng = Engine(0)
    src_dev = torch.from_numpy(d["src"]).cuda()
    mine = D.device_shard_source(eng, src_dev, rank, world)
    eng.set_target(torch.from_numpy(d["tgt"]).cuda(), torch.from_numpy(d["tgt_nrm"]).cuda())
    eng.set_source(src_dev[torch.from_numpy(mine).cuda()])
    eng.comm_init_local(job, world, rank)
    eng.set_global_source_count(N_BIG)
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 1e-6, 1e-6, 30, -1.0)   # stops by its criteria
    np.savez(os.path.join(out_dir, "big_%d.npz" % rank), T=np.array(res.transformation, np.float32),
             stat=np.array([res.fitness, res.inlier_rmse, res.iterations]))
    eng.comm_destroy()
    eng.close()


N_BIG = 1_100_000


def test_sharded_loop_with_halo_decisions_of_its_own_on_every_rank(tmp_path):
    """Shards of more than 500k noisy points: every rank's loop looks at ITS lanes' requests for halos and
    builds them (or not) when it sees fit, and the loop ends by its convergence criteria, not its budget.  The
    ranks must still take identical steps, stop at the same iteration, and end where the single-process loop ends
    (the number of evaluations a rank enqueues must not depend on what it decides: mi_icp.hip loop_run)."""
    from cupoch_amd.engine import Engine
    world = 2
    job = "testbig_%d_%d" % (os.getpid(), _free_port())
    mp.spawn(_worker_big, args=(world, job, str(tmp_path)), nprocs=world, join=True)
    d = make_pair(N_BIG, seed=29, noise=0.2)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    ref = eng.registration_icp(PT2PL, d["max_dist"], None, 1e-6, 1e-6, 30, -1.0)
    eng.close()
    m0, m1 = np.load(tmp_path / "big_0.npz"), np.load(tmp_path / "big_1.npz")
    np.testing.assert_array_equal(m0["T"], m1["T"])
    np.testing.assert_array_equal(m0["stat"], m1["stat"])
    assert int(m0["stat"][2]) == ref.iterations and ref.iterations < 30
    assert np.linalg.norm(m0["T"] - np.array(ref.transformation, np.float32)) <= 1e-6
    assert m0["stat"][0] == pytest.approx(ref.fitness, abs=1e-6) and m0["stat"][1] == pytest.approx(ref.inlier_rmse, rel=1e-5)


def _worker_lonely(rank, job, out_dir):
    import sys
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MI_ICP_MAIL_SPIN_LIMIT"] = "20000"          # ~50 ms instead of ~15 s
    from cupoch_amd.engine import Engine, MiIcpError
    torch.cuda.set_device(0)
    d = make_pair(20000, seed=3, noise=0.02)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    eng.comm_init_local(job, 2, rank)                       # both ranks attach ...
    if rank == 1:                                           # ... and rank 1 never posts anything
        time.sleep(3.0)
        eng.close()
        return
    msgs = []
    for call in (lambda: eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 5, -1.0),
                 lambda: eng.evaluate_registration(d["max_dist"], None)):
        try:
            call()
            msgs.append("no error")
        except MiIcpError as e:
            msgs.append(str(e))
    msgs.append("kind %d" % eng.comm_kind())
    eng.comm_destroy()
    # ... and the context is usable again on its own
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 5, -1.0)
    msgs.append("alone: fitness %.3f" % res.fitness)
    eng.close()
    with open(os.path.join(out_dir, "lonely.txt"), "w") as f:
        f.write("\n".join(msgs))


def test_mailbox_exchange_times_out_instead_of_hanging(tmp_path):
    """A peer that attaches and then never posts: the kernels give up after the spin limit, the call fails
    with MI_ICP_ERR_COMM (no hang, no garbage result), the communicator is void from then on (the mailbox is
    given up: a later post could be taken for another exchange's) and every further exchange fails at once;
    the context works again once the communicator is gone."""
    job = "lonely_%d_%d" % (os.getpid(), _free_port())
    mp.spawn(_worker_lonely, args=(job, str(tmp_path)), nprocs=2, join=True)
    lines = open(tmp_path / "lonely.txt").read().splitlines()
    assert "timed out" in lines[0] and "void" in lines[0], lines
    assert "void" in lines[1], lines
    assert lines[2] == "kind 0", lines
    assert lines[3].startswith("alone: fitness") and float(lines[3].split()[-1]) > 0.9, lines


def _worker_attach_alone(_index, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MI_ICP_MAIL_ATTACH_MS"] = "300"
    from cupoch_amd.engine import Engine, MiIcpError
    torch.cuda.set_device(0)
    eng = Engine(0)
    try:
        eng.comm_init_local("alone_%d" % os.getpid(), 2, 0)     # rank 1 never shows up
        msg = "no error"
    except MiIcpError as e:
        msg = str(e)
    kind = eng.comm_kind()
    eng.close()
    with open(os.path.join(out_dir, "alone.txt"), "w") as f:
        f.write("%s\nkind %d" % (msg, kind))


def test_mailbox_setup_fails_cleanly_without_its_peers(tmp_path):
    """Rank 0 waits for every rank to attach before the box is declared in use; without them the set-up
    fails after MI_ICP_MAIL_ATTACH_MS and leaves no communicator (and no shared-memory name) behind."""
    mp.spawn(_worker_attach_alone, args=(str(tmp_path),), nprocs=1, join=True)
    lines = open(tmp_path / "alone.txt").read().splitlines()
    assert "attached" in lines[0] and lines[1] == "kind 0", lines
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("mi_icp_alone_")]


N8, ITER8 = 64000, 6


def _worker_eight(rank, world, job, out_dir, mode="host"):
    import sys
    import time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MI_ICP_MAILBOX"] = mode
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine
    torch.cuda.set_device(0)
    d = make_pair(N8, seed=21, noise=0.03)
    eng = Engine(0)
    src_dev = torch.from_numpy(d["src"]).cuda()
    mine = D.device_shard_source(eng, src_dev, rank, world)
    eng.set_target(torch.from_numpy(d["tgt"]).cuda(), torch.from_numpy(d["tgt_nrm"]).cuda())
    eng.set_source(src_dev[torch.from_numpy(mine).cuda()])
    eng.comm_init_local(job, world, rank)
    assert eng.comm_kind() == (3 if mode == "device" else 2)
    eng.set_global_source_count(N8)
    if rank == 5:
        time.sleep(0.7)                                     # a deliberately slow rank: its peers' kernels wait for its posts
    out = {}
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER8, -1.0)
    out["T"] = np.array(res.transformation, np.float32)
    out["stat"] = np.array([res.fitness, res.inlier_rmse, res.iterations])
    if rank == 2:
        time.sleep(0.3)                                     # ... and another one, one loop later (the slots' parity at work)
    res = eng.registration_icp(1, d["max_dist"], None, 0.0, 0.0, ITER8 + 1, -1.0)
    out["T_pp"] = np.array(res.transformation, np.float32)
    ev = eng.evaluate_registration(d["max_dist"], None)
    out["eval"] = np.array([ev.fitness, ev.inlier_rmse])
    np.savez(os.path.join(out_dir, "eight_%d.npz" % rank), **out)
    eng.comm_destroy()
    eng.close()


@pytest.mark.parametrize("mode", ["host", "device"])
def test_mailbox_exchange_eight_processes_on_one_gpu(tmp_path, mode):
    """The exchange at the width of a node: eight processes share GPU 0, each with an eighth of the source;
    8 posts per exchange, odd and even numbers of exchanges (both slots), ranks that arrive late -- through the
    host-memory box and through device inboxes (56 IPC mappings among the eight).  All ranks end bit-identical,
    equal to the single-process loop."""
    from cupoch_amd.engine import Engine
    world = 8
    job = "eight_%d_%d" % (os.getpid(), _free_port())
    mp.spawn(_worker_eight, args=(world, job, str(tmp_path), mode), nprocs=world, join=True)
    d = make_pair(N8, seed=21, noise=0.03)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    ref = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER8, -1.0)
    ref_pp = eng.registration_icp(1, d["max_dist"], None, 0.0, 0.0, ITER8 + 1, -1.0)
    ev = eng.evaluate_registration(d["max_dist"], None)
    eng.close()
    m = [np.load(tmp_path / ("eight_%d.npz" % r)) for r in range(world)]
    for r in range(1, world):
        for k in m[0].files:
            np.testing.assert_array_equal(m[0][k], m[r][k], err_msg="%s rank %d" % (k, r))
    assert np.linalg.norm(m[0]["T"] - np.array(ref.transformation, np.float32)) <= 1e-6
    assert np.linalg.norm(m[0]["T_pp"] - np.array(ref_pp.transformation, np.float32)) <= 1e-6
    assert m[0]["stat"][0] == pytest.approx(ref.fitness, abs=1e-6) and int(m[0]["stat"][2]) == ref.iterations
    assert m[0]["eval"][0] == pytest.approx(ev.fitness, abs=1e-6) and m[0]["eval"][1] == pytest.approx(ev.inlier_rmse, rel=1e-5)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("mi_icp_eight_")]     # the name went once all had attached


def _worker_autotune(rank, world, job, out_dir, broken):
    import json
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.pop("MI_ICP_MAILBOX", None)                 # the default set-up: the box AND device inboxes
    if broken:
        os.environ["MI_ICP_SELFTEST_BREAK"] = broken
    from cupoch_amd import distributed as D
    from cupoch_amd.engine import Engine
    torch.cuda.set_device(0)
    d = make_pair(N8, seed=21, noise=0.03)
    eng = Engine(0)
    src_dev = torch.from_numpy(d["src"]).cuda()
    mine = D.device_shard_source(eng, src_dev, rank, world)
    eng.set_target(torch.from_numpy(d["tgt"]).cuda(), torch.from_numpy(d["tgt_nrm"]).cuda())
    eng.set_source(src_dev[torch.from_numpy(mine).cuda()])
    eng.comm_init_local(job, world, rank)
    assert eng.comm_kind() == 2                            # until measured otherwise: the host-memory words
    tune = eng.comm_autotune(64)
    tune["kind_after"] = eng.comm_kind()
    eng.set_global_source_count(N8)
    res = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER8, -1.0)
    tune["T"] = np.array(res.transformation, np.float32).tolist()
    tune["stat"] = [res.fitness, res.inlier_rmse, res.iterations]
    with open(os.path.join(out_dir, "tune_%d.json" % rank), "w") as f:
        json.dump(tune, f)
    eng.comm_destroy()
    eng.close()


@pytest.mark.parametrize("broken", ["", "wrong:2", "mute:1"])
def test_exchange_paths_are_self_tested_timed_and_chosen_alike_on_every_rank(tmp_path, broken):
    """mi_icp_comm_autotune, rehearsed with eight processes on one GPU: every available path --
    the box's host-memory words, device inboxes over HIP IPC (RCCL cannot put two ranks on one device) -- runs its
    exchanges of a known vector, checked exactly on the device and timed; the ranks gather the figures through the box
    and ALL keep the same, fastest path that passed everywhere.  A path on which one rank posts a wrong vector
    ("wrong:2": the device inboxes) or nothing at all ("mute:1": the host words; its peers' kernels time out after
    ~2 s) is reported as failed on every rank, skipped, and the exchange counters are re-aligned behind it: the loop
    that follows runs on the other path and still equals the single-process loop."""
    import json
    from cupoch_amd.engine import Engine
    world = 8
    job = "tune_%d_%d" % (os.getpid(), _free_port())
    mp.spawn(_worker_autotune, args=(world, job, str(tmp_path), broken), nprocs=world, join=True)
    t = [json.load(open(tmp_path / ("tune_%d.json" % r))) for r in range(world)]
    for r in range(1, world):
        assert t[r] == t[0], (r, t[r], t[0])                  # figures, choice and results identical on every rank
    lat = t[0]["latency_us"]
    assert lat["rccl"] is None and t[0]["rccl_comm_count"] == 0 and t[0]["exchanges"] == 64
    if not broken:
        assert t[0]["verified"] and all(isinstance(lat[k], float) and 0.0 < lat[k] < 1e5 for k in ("host mailbox", "device inboxes"))
        best = min(("host mailbox", "device inboxes"), key=lambda k: lat[k])
        assert t[0]["chosen"] == best and t[0]["kind_after"] == (2 if best == "host mailbox" else 3)
    elif broken == "wrong:2":
        assert not t[0]["verified"] and lat["device inboxes"] == "failed" and t[0]["chosen"] == "host mailbox"
    else:
        assert not t[0]["verified"] and lat["host mailbox"] == "failed" and t[0]["chosen"] == "device inboxes"
        assert t[0]["kind_after"] == 3
    d = make_pair(N8, seed=21, noise=0.03)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    ref = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, ITER8, -1.0)
    eng.close()
    assert np.linalg.norm(np.array(t[0]["T"], np.float32) - np.array(ref.transformation, np.float32)) <= 1e-6
    assert t[0]["stat"][0] == pytest.approx(ref.fitness, abs=1e-6) and int(t[0]["stat"][2]) == ref.iterations


def test_autotune_on_a_one_rank_rccl_communicator_reports_the_collective():
    """... and the third path where a one-GPU box can run it: a one-rank RCCL communicator (no box: the figure travels
    through the communicator itself); ncclCommCount is reported, the choice is RCCL, the loop unchanged."""
    from cupoch_amd.engine import Engine, comm_unique_id
    d = make_pair(40000, seed=5, noise=0.02)
    eng = Engine(0)
    eng.set_target(d["tgt"], d["tgt_nrm"])
    eng.set_source(d["src"])
    ref = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 6, -1.0)
    eng.comm_init(comm_unique_id(), 1, 0)
    tune = eng.comm_autotune(32)
    assert tune["chosen"] == "rccl" and tune["rccl_comm_count"] == 1 and tune["verified"]
    assert tune["latency_us"]["host mailbox"] is None and 0.0 < tune["latency_us"]["rccl"] < 1e5
    eng.set_global_source_count(len(d["src"]))
    got = eng.registration_icp(PT2PL, d["max_dist"], None, 0.0, 0.0, 6, -1.0)
    np.testing.assert_array_equal(np.array(got.transformation), np.array(ref.transformation))
    eng.comm_destroy()
    eng.close()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the shape of the driver's N = 1 command): the bench
    re-launches itself with one rank per GPU, rank 0 prints the one JSON line, the exit code is the ranks'.
    MI_ICP_BENCH_ONE_DEVICE=1: both ranks on cuda:0 (the control flow of the N > 1 path on a one-GPU box)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_ICP_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--points", "200000", "--steps", "5",
                        "--warmup", "2", "--repeats", "3", "--no-secondary", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric')]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["steps"] == 5
