"""CLI of the scene-graph evaluation, drop-in for the reference's
`python -u -m scene_graph_prediction.main --config <name>.json --model_path <dir>`
(scene_graph_prediction/main.py:47-149): same two flags, same hard-coded mode / batch size rules
(batch 8; batch 1 and no shuffle for temporality == 'PRED'), config resolved under scene_graph_helpers/configs/
(JSON, comments allowed). The 4D-OR dataset is replaced by the synthetic dataset with the same item contract; with
several GPUs (torchrun) frames -- or whole takes for the temporal variant -- are sharded across ranks and raw
predictions all-gathered on rank 0 (oracle_amd/parallel.py)."""
import argparse
import json
import os
import random
import re
from pathlib import Path

import numpy as np
import torch

from .scene_graph_helpers.dataset.synthetic_dataset import SyntheticORDataset, batches
from .scene_graph_helpers.model.scene_graph_prediction_model_oracle import OracleWrapper

MAX_NEW_TOKENS = 300   # the reference's generate(max_new_tokens=300) (..._oracle.py:135-143): width of the gathered token buffer
CONFIG_DIRS = [Path('scene_graph_prediction/scene_graph_helpers/configs'), Path(__file__).parent / 'scene_graph_helpers' / 'configs']


def config_loader(config_path: str):
    for d in CONFIG_DIRS:
        p = d / config_path
        if p.exists():
            text = p.read_text()
            text = re.sub(r'^\s*(//|#).*$', '', text, flags=re.MULTILINE)      # whole-line comments
            return json.loads(text)
    raise FileNotFoundError(config_path)


def seed_everything(seed):
    """pl.seed_everything(42): python `random` (drives the temporal-memory shuffle), numpy and torch."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main(argv=None):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument('--config', type=str, default='example.json', help='configuration file name. Relative path under given path')
    ap.add_argument('--model_path', type=str, default=None, help='path to model checkpoint')
    ap.add_argument('--mode', type=str, default='evaluate', choices=['evaluate', 'infer'])
    ap.add_argument('--limit_frames', type=int, default=None, help='synthetic split size (default: 4D-OR split size)')
    ap.add_argument('--image_size', type=str, default='336x336', help='synthetic camera image size WxH (4D-OR: 2048x1536)')
    ap.add_argument('--image_format', type=str, default='png', choices=['png', 'jpg'], help='synthetic camera image files')
    ap.add_argument('--decode_workers', type=int, default=None, help='image-decode threads (0: decode in the main thread, as the reference)')
    ap.add_argument('--kv_slots', type=int, default=None,
                    help='KV-cache slots of the engine (default 32: 16 frames decode while 16 wait prefilled; 64: 32 + 32)')
    ap.add_argument('--max_seq', type=int, default=None,
                    help='positions per KV-cache slot (default 4096, LLaMA\'s: what the reference can run to). A plain-prompt split needs '
                         '~1100 (prompt 739 + 300 new tokens), the temporal / symbolic prompts ~2400; smaller regions decode up to 5 %% faster')
    args = ap.parse_args(argv)
    seed_everything(42)
    config = config_loader(args.config)
    mv_desc = False
    batch_size = 8
    temporal = config.get('temporality') == 'PRED'
    if temporal:
        print('Online temporality. Setting batch size to 1 and not shuffling')
        batch_size = 1
    name = args.config.replace('.json', '')
    from .. import parallel
    rank, world = parallel.init_from_env()
    split = 'val' if args.mode == 'evaluate' else 'test'
    print(f'Model path: {args.model_path}')
    train_dataset = SyntheticORDataset(config, 'train', n_frames=0, mv_desc=mv_desc)
    w_, h_ = (int(x) for x in args.image_size.lower().split('x'))
    eval_dataset = SyntheticORDataset(config, split, n_frames=args.limit_frames, for_eval=args.mode == 'infer', mv_desc=mv_desc,
                                      image_size=(w_, h_), image_format=args.image_format)
    model = OracleWrapper(config, num_class=len(eval_dataset.classNames), num_rel=len(eval_dataset.relationNames),
                          weights_obj=train_dataset.w_cls_obj, weights_rel=train_dataset.w_cls_rel,
                          relationNames=train_dataset.relationNames, model_path=args.model_path, mv_desc=mv_desc,
                          **({} if args.decode_workers is None else {'decode_workers': args.decode_workers}),
                          **({} if args.kv_slots is None else {'max_batch': args.kv_slots}),
                          **({} if args.max_seq is None else {'max_seq': args.max_seq}))
    takes = [int(s.split('_')[0]) for s in eval_dataset.scans]
    mine = parallel.shard_indices(len(eval_dataset), rank, world, groups=takes if temporal else None)
    results = model.validate(batches(eval_dataset, batch_size, mine), return_raw_predictions=True, report=world == 1)
    if world > 1:
        # the ONLY exchange of the multi-GPU path is one all-gather (RCCL over xGMI) of a fixed-shape int32 buffer
        # [frames per rank, 1 + 300] -- the answer's length, then its token ids, length 0 for the padding rows of a short shard.
        # Every rank saw only its shard; rank 0 detokenises, parses and scores the whole split in dataset order and prints the
        # reference's per-take and overall reports ONCE (N partial reports are not the metric the reference prints).
        import torch.distributed as dist
        groups = takes if temporal else None
        n = len(eval_dataset)
        n_rows = parallel.rows_per_rank(n, world, groups)
        toks, lens = parallel.pack_token_rows([model.last_new_tokens[eval_dataset.scans[i]] for i in mine], n_rows, MAX_NEW_TOKENS)
        dev = parallel.collective_device()
        allbuf = parallel.gather_token_ids(toks.to(dev), lens.to(dev), world)
        results = None
        if rank == 0:
            rows = parallel.unpack_token_rows(allbuf, n, world, groups)
            # a frame generated in a one-frame batch keeps its special tokens in the string (forward(), reference :144-147):
            # the temporal variant always, otherwise the last frame of a shard whose size is 1 mod the batch size
            single = {}
            for r in range(world):
                shard = parallel.shard_indices(n, r, world, groups)
                for j, idx in enumerate(shard):
                    single[idx] = batch_size == 1 or (len(shard) % batch_size == 1 and j == len(shard) - 1)
            results = model.score_token_ids([eval_dataset[i] for i in range(n)], [rows[i] for i in range(n)], [single[i] for i in range(n)])
        dist.barrier()
    if args.mode == 'infer' and rank == 0:
        out = f'scan_relations_{name}_{split}.json'
        with open(out, 'w') as f:
            json.dump(results, f)
        print('wrote', out)
    parallel.finalize()
    return results


if __name__ == '__main__':
    import subprocess
    try:   # the reference lists GPUs with nvidia-smi here; on ROCm the equivalent is rocm-smi
        subprocess.call(['rocm-smi', '--showid'])
    except FileNotFoundError:
        pass
    main()
