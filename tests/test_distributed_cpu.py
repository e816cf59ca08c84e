"""N>1 path on CPU: world_size-2 gloo processes.  Checks the host logic the
multi-GPU run depends on -- spatial sharding, the unique-id exchange and that
all-reducing the per-shard systems reproduces the single-process system and
therefore the same update on every rank (compute stands in through the
oracle here; on GPUs the same 32 doubles come out of the HIP reduction)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, make_pair


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cupoch_amd import distributed as D
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = make_pair(6000, seed=13, noise=0.03)
    mine = D.shard_source(d["src"], rank, world)
    uid = D.exchange_unique_id(lambda: bytes(range(128)), rank)
    assert uid == bytes(range(128))
    ev = orc.evaluate_registration(d["src"][mine], d["tgt"], d["max_dist"])
    part = orc.compute_system(orc.EST_PT2PL, d["src"][mine], d["tgt"], ev.correspondence_set,
                              tgt_nrm=d["tgt_nrm"])
    total = D.allreduce_system(part)
    ok, T = orc.solve_system(total, -1.0)
    np.save(os.path.join(out_dir, "sys_%d.npy" % rank), total)
    np.save(os.path.join(out_dir, "T_%d.npy" % rank), T)
    np.save(os.path.join(out_dir, "idx_%d.npy" % rank), mine)
    # the job's correspondence set from the shard-local ones
    allc = D.gather_correspondences(ev.correspondence_set, mine)
    np.save(os.path.join(out_dir, "corr_%d.npy" % rank), allc)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_system_equals_single_process(tmp_path):
    from oracle import oracle as orc
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    d = make_pair(6000, seed=13, noise=0.03)
    ev = orc.evaluate_registration(d["src"], d["tgt"], d["max_dist"])
    full = orc.compute_system(orc.EST_PT2PL, d["src"], d["tgt"], ev.correspondence_set,
                              tgt_nrm=d["tgt_nrm"])
    s0, s1 = np.load(tmp_path / "sys_0.npy"), np.load(tmp_path / "sys_1.npy")
    np.testing.assert_array_equal(s0, s1)                       # every rank holds the same sum
    np.testing.assert_allclose(s0, full, rtol=1e-12, atol=1e-12 * np.abs(full).max())
    np.testing.assert_array_equal(np.load(tmp_path / "T_0.npy"), np.load(tmp_path / "T_1.npy"))
    c0, c1 = np.load(tmp_path / "corr_0.npy"), np.load(tmp_path / "corr_1.npy")
    np.testing.assert_array_equal(c0, c1)
    np.testing.assert_array_equal(c0, ev.correspondence_set)    # = the single-process set, same order
    i0, i1 = np.load(tmp_path / "idx_0.npy"), np.load(tmp_path / "idx_1.npy")
    assert len(np.intersect1d(i0, i1)) == 0 and len(i0) + len(i1) == 6000


def test_shard_bounds_and_spatial_coherence():
    from cupoch_amd import distributed as D
    for n, w in [(10, 3), (7, 8), (0, 2), (10_000_001, 8)]:
        cuts = [D.shard_bounds(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
    with pytest.raises(ValueError):
        D.shard_bounds(5, 2, 2)
    rng = np.random.default_rng(0)
    pts = rng.random((40000, 3), dtype=np.float32)
    vol = []
    for r in range(8):
        p = pts[D.shard_source(pts, r, 8)]
        vol.append(np.prod(p.max(0) - p.min(0)))
    assert sum(vol) < 3.0            # random sharding would give 8 x ~1.0: shards are spatial
