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
        if backend == 'nccl':   # one init path for RCCL (bench.py uses it too): the rank's GPU is bound to the group at creation
            local = int(os.environ.get('LOCAL_RANK', 0))
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return rank, world


def collective_device():
    """Where tensors of a collective must live: the rank's GPU under RCCL, the host under gloo."""
    return torch.device('cuda', torch.cuda.current_device()) if dist.is_initialized() and dist.get_backend() == 'nccl' else torch.device('cpu')


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


def rows_per_rank(n, world, groups=None):
    """Rows of the fixed-shape gather buffer: the largest shard (uneven shards -- 1378 frames over 8 ranks, or whole takes of
    different sizes -- are padded with length-0 rows)."""
    return max(len(shard_indices(n, r, world, groups)) for r in range(world)) if n else 0


def pack_token_rows(token_lists, n_rows, max_new):
    """This rank's answers, one per frame of its shard in shard order (lists of new-token ids), as the [n_rows, 1 + max_new] int32
    buffer of SURVEY 8e: column 0 = number of tokens (0 for the padding rows of a short shard), then the ids, zero-filled."""
    assert len(token_lists) <= n_rows
    toks = torch.zeros(n_rows, max_new, dtype=torch.int32)
    lens = torch.zeros(n_rows, dtype=torch.int32)
    for j, ids in enumerate(token_lists):
        ids = list(ids)[:max_new]
        lens[j] = len(ids)
        if ids:
            toks[j, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return toks, lens


def unpack_token_rows(allbuf, n, world, groups=None):
    """The all-gathered [world * n_rows, 1 + max_new] buffer -> {frame index: list of token ids} over the whole split (row j of
    rank r's block is frame shard_indices(n, r, world, groups)[j])."""
    allbuf = allbuf.cpu()
    n_rows = allbuf.shape[0] // world
    out = {}
    for r in range(world):
        for j, idx in enumerate(shard_indices(n, r, world, groups)):
            row = allbuf[r * n_rows + j]
            out[idx] = row[1:1 + int(row[0])].tolist()
    return out
