"""Multi-GPU: one process per GPU, model replicated, frames sharded data-parallel, no data-path collective; the only
exchange is an all-gather of the decoded token ids / raw predictions (RCCL over xGMI via torch.distributed 'nccl';
'gloo' on CPU for tests). The reference has no distributed inference."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world) from torchrun's environment; initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # ORC_DIST_BACKEND=gloo: rehearsal with several ranks sharing one GPU (RCCL wants one GPU per rank)
            backend = os.environ.get('ORC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        dist.init_process_group(backend)
    return rank, world


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_indices(n, rank, world, groups=None):
    """Frame indices of this rank. Independent frames: round-robin (rank r takes r, r+world, ...), which keeps per-rank
    work within one frame of equal. With `groups` (take id per frame, temporal variant): whole takes are dealt to ranks
    greedily by size and stay in order inside a take, because frame t needs the predictions of frames < t of its take."""
    if groups is None:
        return list(range(rank, n, world))
    order = []
    for g in groups:
        if g not in order:
            order.append(g)
    sizes = {g: sum(1 for x in groups if x == g) for g in order}
    load = [0] * world
    owner = {}
    for g in sorted(order, key=lambda g: (-sizes[g], order.index(g))):
        r = min(range(world), key=lambda i: (load[i], i))
        owner[g] = r
        load[r] += sizes[g]
    return [i for i in range(n) if owner[groups[i]] == rank]


def gather_token_ids(tokens, lengths, world):
    """tokens (F, max_new) int32, lengths (F,) int32 on this rank -> fixed-shape buffer [F, 1+max_new] all-gathered to
    (world*F, 1+max_new) on every rank (one collective, KB-scale: latency-bound)."""
    buf = torch.cat([lengths.to(torch.int32)[:, None], tokens.to(torch.int32)], dim=1).contiguous()
    if world == 1:
        return buf
    out = torch.empty(world * buf.shape[0], buf.shape[1], dtype=torch.int32, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    return out


def gather_objects(obj, rank, world):
    """Merge per-rank {scan_id: predictions} dicts on rank 0 (host-side, used by the CLI)."""
    if world == 1:
        return obj
    parts = [None] * world
    dist.all_gather_object(parts, obj)
    merged = {}
    for p in parts:
        merged.update(p)
    return merged
