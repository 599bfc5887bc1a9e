"""Utterance-batch sharding across ranks (one process per GPU).

Utterances are independent (reference: one ps_decoder_t per utterance stream,
no cross-utterance state except the top-N seed carry-over, which batch mode
resets per utterance -- SURVEY 8e), so the hot path shards with NO data-path
collective: every rank holds a replica of the model tables, scores its own
contiguous block of utterances, and only batch scatter / result gather touch
torch.distributed (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
import numpy as np


def partition(utt_lens, world):
    """Contiguous, frame-balanced split of utterances over `world` ranks.
    Returns [(u_begin, u_end)] per rank; every utterance is owned exactly once
    and order is preserved (so gathered rows concatenate back in input order)."""
    lens = np.asarray(utt_lens, np.int64)
    n = lens.size
    csum = np.concatenate([[0], np.cumsum(lens)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        # first boundary whose prefix reaches the target, never moving backwards
        k = int(np.searchsorted(csum, target, side="left"))
        k = min(max(k, cuts[-1]), n)
        cuts.append(k)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard(utt_lens, rank, world):
    """(u_begin, u_end, frame_begin, frame_end) of this rank's block."""
    lens = np.asarray(utt_lens, np.int64)
    ub, ue = partition(lens, world)[rank]
    fb = int(lens[:ub].sum())
    fe = fb + int(lens[ub:ue].sum())
    return ub, ue, fb, fe


def scatter_feats(feats, utt_lens, group=None, src=0, device="cpu"):
    """Rank `src` holds feats [T][veclen] + utt_lens; every rank receives its
    block (torch tensor on `device`) and the full length list."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    meta = [None]
    if rank == src:
        feats = np.ascontiguousarray(feats, np.float32)
        meta = [(list(map(int, utt_lens)), int(feats.shape[1]))]
    dist.broadcast_object_list(meta, src=src, group=group)
    lens, veclen = meta[0]
    parts = partition(lens, world)
    csum = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nfr = [int(csum[e] - csum[b]) for b, e in parts]
    cap = max(max(nfr), 1)
    recv = torch.empty((cap, veclen), dtype=torch.float32, device=device)
    if rank == src:
        chunks = []
        for (b, e), n in zip(parts, nfr):
            c = torch.zeros((cap, veclen), dtype=torch.float32, device=device)
            c[:n] = torch.from_numpy(feats[csum[b]:csum[e]]).to(device)
            chunks.append(c)
        dist.scatter(recv, chunks, src=src, group=group)
    else:
        dist.scatter(recv, None, src=src, group=group)
    ub, ue = parts[rank]
    return recv[:nfr[rank]], lens, (ub, ue)


def gather_rows(local_rows, utt_lens, group=None, dst=0):
    """Gather per-frame result rows (torch tensor [frames_of_my_block][W]) of
    every rank's block to `dst`, concatenated in utterance order.  Returns the
    full tensor on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lens = np.asarray(utt_lens, np.int64)
    parts = partition(lens, world)
    csum = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nfr = [int(csum[e] - csum[b]) for b, e in parts]
    assert local_rows.shape[0] == nfr[rank], "rank %d holds %d rows, owns %d frames" % (
        rank, local_rows.shape[0], nfr[rank])
    cap = max(max(nfr), 1)
    local_rows = local_rows.reshape(local_rows.shape[0], -1).contiguous()
    dtype, width = local_rows.dtype, local_rows.shape[1]
    # rows travel as raw bytes: int16/int32 rows are not a transport dtype of every backend
    pad = torch.zeros((cap, width * local_rows.element_size()), dtype=torch.uint8, device=local_rows.device)
    pad[:nfr[rank]] = local_rows.view(torch.uint8)
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst, group=group)
        return torch.cat([b[:n] for b, n in zip(bufs, nfr)], dim=0).view(dtype)
    dist.gather(pad, None, dst=dst, group=group)
    return None


def max_over_ranks(seconds, group=None, device="cpu"):
    """The timing reduction of bench.py: MAX of a host-measured duration."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def scatter_pcm(recv, pcm_all, n_per_rank, group=None, src=0, device="cpu"):
    """Rank `src` holds the 16-bit samples of the whole job back to back (numpy int16, world * n_per_rank); every rank's
    share lands in `recv` (torch int16 [n_per_rank] on `device`).  One rank's share at a time is staged on `src`'s
    device, so the source never needs more than two shares resident."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    assert recv.dtype == torch.int16 and recv.numel() == n_per_rank
    if rank == src:
        pcm_all = np.ascontiguousarray(pcm_all, np.int16).reshape(-1)
        assert pcm_all.size == world * n_per_rank
        recv.copy_(torch.from_numpy(pcm_all[src * n_per_rank:(src + 1) * n_per_rank]))
        for r in range(world):
            if r == src:
                continue
            stage = torch.from_numpy(pcm_all[r * n_per_rank:(r + 1) * n_per_rank]).to(device)
            dist.send(stage, dst=r, group=group)
    else:
        dist.recv(recv, src=src, group=group)
    return recv


def gather_records(local, group=None, dst=0, device="cpu"):
    """Fixed-size result records (torch tensor, same shape on every rank) of every rank to `dst`: returns
    [world * n][...] there (rank order = utterance order of a block-sharded job), None elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    local = local.contiguous()
    if rank == dst:
        bufs = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, bufs, dst=dst, group=group)
        return torch.cat(bufs, dim=0)
    dist.gather(local, None, dst=dst, group=group)
    return None
