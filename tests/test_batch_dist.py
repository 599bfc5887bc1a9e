"""The N>1 path on CPU: world_size-2 gloo.  Utterances are scattered from rank
0, every rank scores its own block (here with the oracle standing in for the
GPU scorer -- the sharding logic is what is under test), rows are gathered
back and must equal the unsharded result bit for bit; the timing reduction is
a MAX over ranks.  No data-path collective exists to test: the path shards."""
import os
import socket

import numpy as np
import pytest

import pso
from pocketsphinx_amd import batch


def test_partition_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 4, 8):
        for _ in range(20):
            lens = rng.integers(0, 400, rng.integers(0, 40)).tolist()
            parts = batch.partition(lens, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == len(lens)
            for (a, b), (c, d) in zip(parts[:-1], parts[1:]):
                assert b == c and a <= b
            if sum(lens) and world > 1 and len(lens) >= 4 * world:
                fr = [sum(lens[a:b]) for a, b in parts]
                assert max(fr) <= sum(lens) / world + max(lens)      # balanced to within one utterance
    assert batch.partition([], 2) == [(0, 0), (0, 0)]
    assert batch.shard([5, 5, 5, 5], 1, 2) == (2, 4, 10, 20)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = pso.load_tables()
        g = np.load(os.path.join(pso.GOLDEN_DIR, "ptm_goforward.npz"))
        lens = [7, 0, 19, 11, 5, 23, 3]
        feats = g["feat"][:sum(lens)] if rank == 0 else None
        mine, all_lens, (ub, ue) = batch.scatter_feats(feats, lens if rank == 0 else None)
        o = pso.OraclePTM(t)
        rows, f0 = [], 0
        for n in all_lens[ub:ue]:
            if n:
                scr, _, _ = o.score_utt(mine[f0:f0 + n].numpy(), reset_hist=True, want_topn=False)
                rows.append(scr)
            f0 += n
        local = torch.from_numpy(np.concatenate(rows) if rows else np.zeros((0, o.n_sen), np.int16))
        full = batch.gather_rows(local, all_lens)
        tmax = batch.max_over_ranks(1.0 + rank)
        if rank == 0:
            want, f0 = [], 0
            for n in lens:
                if n:
                    scr, _, _ = o.score_utt(g["feat"][f0:f0 + n], reset_hist=True, want_topn=False)
                    want.append(scr)
                f0 += n
            ok = bool(np.array_equal(full.numpy(), np.concatenate(want)))
            q.put((ok, tmax, int(full.shape[0])))
    finally:
        dist.destroy_process_group()


def test_sharded_scoring_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok, tmax, nrows = q.get(timeout=10)
    assert ok and nrows == 68
    assert tmax == 2.0


def _worker_pcm(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        pcm_all = (np.arange(world * n) % 30011 - 15000).astype(np.int16) if rank == 0 else None
        mine = batch.scatter_pcm(torch.empty(n, dtype=torch.int16), pcm_all, n)
        want = (np.arange(rank * n, (rank + 1) * n) % 30011 - 15000).astype(np.int16)
        ok = bool(np.array_equal(mine.numpy(), want))
        # every rank's "hypothesis records": [4 utterances][3 words][4] int32, tagged with the rank
        rec = torch.full((4, 3, 4), rank, dtype=torch.int32) + torch.arange(4, dtype=torch.int32).reshape(4, 1, 1)
        full = batch.gather_records(rec)
        if rank == 0:
            ok = ok and tuple(full.shape) == (4 * world, 3, 4) and all(int(full[4 * r + u, 0, 0]) == r + u for r in range(world) for u in range(4))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_pcm_scatter_and_record_gather_gloo_world2():
    """what bench.py --gpus N moves: rank 0's PCM to the ranks before the timed region, the ranks' fixed-size hypothesis
    records back to rank 0 inside it"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_pcm, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(2))
    assert got == {0: True, 1: True}
