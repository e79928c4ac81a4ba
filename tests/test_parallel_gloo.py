"""CPU, world_size 2 over gloo: frame sharding (round-robin; whole takes for the temporal variant) and the fixed-shape
all-gather of token ids that is the only exchange of the multi-GPU path."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle_amd import parallel
    r, w = parallel.init_from_env(backend='gloo')
    n = 11
    mine = parallel.shard_indices(n, r, w)
    F = 6
    toks = torch.full((F, 5), -1, dtype=torch.int32)
    lens = torch.zeros(F, dtype=torch.int32)
    for j, idx in enumerate(mine):
        toks[j] = idx * 10 + torch.arange(5, dtype=torch.int32)
        lens[j] = 1 + idx % 5
    allbuf = parallel.gather_token_ids(toks, lens, w)
    groups = [4] * 5 + [8] * 6
    tmine = parallel.shard_indices(n, r, w, groups=groups)
    if r == 0:
        torch.save({'allbuf': allbuf, 'mine': mine, 'tmine': tmine}, out)
    dist.barrier()
    parallel.finalize()


def test_world2_sharding_and_gather(tmp_path):
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    buf = res['allbuf']
    assert buf.shape == (12, 6)
    seen = {}
    for row in buf.tolist():
        if row[0] > 0:
            idx = row[1] // 10
            assert row[1:] == [idx * 10 + k for k in range(5)] and row[0] == 1 + idx % 5
            seen[idx] = True
    assert sorted(seen) == list(range(11))
    assert res['mine'] == [0, 2, 4, 6, 8, 10]
    assert res['tmine'] == [5, 6, 7, 8, 9, 10]       # the larger take goes to rank 0, whole and in order


def _worker_rows(rank, world, port, out, n, groups):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle_amd import parallel
    r, w = parallel.init_from_env(backend='gloo')
    mine = parallel.shard_indices(n, r, w, groups)
    answers = [[1000 * idx + k for k in range(idx % 7)] for idx in mine]          # ragged answers, some empty, one over-long
    if mine:
        answers[-1] = answers[-1] + list(range(400))
    n_rows = parallel.rows_per_rank(n, w, groups)
    toks, lens = parallel.pack_token_rows(answers, n_rows, 300)
    dev = parallel.collective_device()
    allbuf = parallel.gather_token_ids(toks.to(dev), lens.to(dev), w)
    if r == 0:
        torch.save({'allbuf': allbuf, 'rows': parallel.unpack_token_rows(allbuf, n, w, groups), 'n_rows': n_rows}, out)
    dist.barrier()
    parallel.finalize()


def test_world2_token_rows_uneven_shards_and_takes(tmp_path):
    """The product path's exchange (CLI: oracle_amd/scene_graph_prediction/main.py): [rows per rank, 1 + 300] int32, length-prefixed,
    padding rows of the shorter shard have length 0; round-robin shards of 9 / 8 frames and take shards of 12 / 5."""
    for case, (n, groups) in enumerate([(17, None), (17, [4] * 12 + [8] * 5), (1, None)]):
        out = str(tmp_path / f'rows{case}.pt')
        mp.spawn(_worker_rows, args=(2, _free_port(), out, n, groups), nprocs=2, join=True)
        res = torch.load(out)
        from oracle_amd.parallel import shard_indices
        assert res['allbuf'].shape == (2 * res['n_rows'], 301) and res['allbuf'].dtype == torch.int32
        assert res['n_rows'] == max(len(shard_indices(n, r, 2, groups)) for r in range(2))
        assert sorted(res['rows']) == list(range(n))
        for r in range(2):
            shard = shard_indices(n, r, 2, groups)
            for j, idx in enumerate(shard):
                want = [1000 * idx + k for k in range(idx % 7)]
                if j == len(shard) - 1:
                    want = (want + list(range(400)))[:300]
                assert res['rows'][idx] == want, (case, idx)
            for j in range(len(shard), res['n_rows']):                             # padding rows: length 0
                assert int(res['allbuf'][r * res['n_rows'] + j, 0]) == 0


def test_shard_properties():
    from oracle_amd.parallel import shard_indices
    for n in (0, 1, 7, 1378):
        for w in (1, 2, 4, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    groups = [2] * 700 + [6] * 678
    parts = [shard_indices(1378, r, 8, groups=groups) for r in range(8)]
    assert sorted(sum(parts, [])) == list(range(1378))
    assert sum(1 for p in parts if p) == 2               # parallelism is bounded by the number of takes


def test_bench_launches_its_own_ranks_and_checks_world_size():
    """`bench.py --gpus N` starts N ranks itself (no external torchrun), max-over-ranks timing and the per-step
    all-gather run over gloo on the CPU (--dry-run: the engine step is a stand-in); --gpus != WORLD_SIZE is refused."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--dry-run'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['frames'] == 2 * 3 * 8
    assert res['ms_per_step'] >= res['slowest_rank_floor_ms']          # the slowest rank (rank 1 sleeps 20 ms) sets the time
    sp = res['extra']['split']                                         # the config-5 leg: sharded split, uneven shards, one final gather
    assert sp['frames'] == 45 and sp['frames_per_rank'] == [23, 22] and sp['gather_buffer'] == [46, 301]
    assert sp['all_frames_arrived_with_their_length'] and len(sp['per_rank_busy_s']) == 2
    assert sp['seconds'] >= max(sp['per_rank_busy_s'])                  # whole-split time = the slowest rank + the gather
    env2 = dict(env, WORLD_SIZE='4', RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True,
                       timeout=60, env=env2)
    assert r.returncode != 0 and 'does not match WORLD_SIZE' in r.stderr
