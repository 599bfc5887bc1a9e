"""The N > 1 path on the GPU: two ranks of one job on ONE MI355X (torch.distributed over gloo for the exchange -- RCCL does
not allow two ranks on one device -- and every rank's real device pipeline for the work): rank 0 owns the PCM of the batch,
scatters it, every rank decodes its block on the device (front end -> scorer -> phone loop -> lexicon-tree search ->
backtrace), the fixed-size hypothesis records are gathered back to rank 0, which holds them against the reference
decoder's recorded results.  What bench.py --gpus N does per step, with the collectives' device being the only difference."""
import os
import socket

import numpy as np
import pytest

import pso
from test_oracle_golden import _load

pytestmark = pytest.mark.gpu

NAMES = ["goforward", "numbers", "numbers", "goforward"]         # two utterances per rank


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pocketsphinx_amd as P
        from pocketsphinx_amd import batch
        torch.cuda.set_device(0)
        clips = _load("speech_clips.npz")
        per = len(NAMES) // world
        n_samp = max(clips[n].size for n in NAMES)                # equal shares: every utterance zero-padded to the longest
        pcm_all = None
        if rank == 0:
            pcm_all = np.zeros((len(NAMES), n_samp), np.int16)
            for i, n in enumerate(NAMES):
                pcm_all[i, :clips[n].size] = clips[n]
            pcm_all = pcm_all.reshape(-1)
        lens = torch.tensor([clips[n].size for n in NAMES], dtype=torch.int64)     # (known to every rank: the job's manifest)
        mine = batch.scatter_pcm(torch.empty(per * n_samp, dtype=torch.int16), pcm_all, per * n_samp)
        gt = _load("fwdtree_trace_goforward.npz")
        p = P.DecodePipeline(_load("mfcc_en_us_goforward.npz"), _load("en_us_ptm_tables.npz"), _load("fwdtree_static_en_us_turtle.npz"),
                             gt["par"], gt)
        pcms = [mine[i * n_samp:i * n_samp + int(lens[rank * per + i])].numpy() for i in range(per)]
        p.run(pcms)
        hn, hyp, res = p.fetch()
        rec = torch.zeros((per, 64, 4), dtype=torch.int32)
        cnt = torch.from_numpy(np.ascontiguousarray(hn[:, :2])).to(torch.int32)
        for u in range(per):
            k = int(hn[u, 0])
            rec[u, :k] = torch.from_numpy(np.ascontiguousarray(hyp[u, :k]))
        g_rec = batch.gather_records(rec)
        g_cnt = batch.gather_records(cnt)
        p.close()
        if rank == 0:
            q.put((g_rec.numpy(), g_cnt.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_decode_their_blocks():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rec, cnt = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert rec.shape[0] == len(NAMES)
    for u, n in enumerate(NAMES):
        g = _load("fwdtree_trace_%s.npz" % n)
        seg = [(int(a), int(b)) for a, b in g["seg"][:, :2]]
        assert int(cnt[u, 0]) == len(seg) and int(cnt[u, 1]) == int(g["hyp_score"][0]), (u, n)
        assert [(int(r[1]), int(r[2])) for r in rec[u, :len(seg)]] == seg, (u, n)


def test_bench_rccl_code_path_with_one_rank(tmp_path):
    """bench.py's N > 1 code path -- init_process_group("nccl") (= RCCL), the PCM scatter before the timed region, the per-step
    gather of the hypothesis records on a stream of its own, the MAX over ranks of the time -- run with ONE rank on this
    one-GPU box (PSGPU_BENCH_FORCE_DIST), small workload, the reference's parity check on: it must produce the bench line"""
    import json
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PSGPU_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--utts", "8", "--seconds", "5",
                          "--no-extras", "--scatter"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["value"] > 0 and j["config"]["utterances_per_gpu"] == 8
    if j.get("cpu_baseline"):
        assert j["parity"]["identical"] == j["parity"]["checked"] == 8
