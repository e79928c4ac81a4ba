"""Synthetic stand-in for ORDataset (the 4D-OR data, open3d and timm are absent offline).

Emits the per-frame dict the hot path consumes (reference or_dataset.py:82-93,98-163): `image_paths` (6 camera
slots), `scan_id` "{take}_{pcd}_{split}", `take_idx`, `relations_tokenized` [(sub, rel, obj) vocabulary ids] and,
with USE_VIS_DESC, `vis_descriptor_embs` (7 tensors). Images are seeded noise written once as PNG files (default, 336 x 336)
or, with image_format='jpg', image-like content (smooth colour fields + mild noise) as JPEG at any size -- 2048 x 1536 is
what the 4D-OR cameras deliver, so the host-side decode cost of the drop-in CLI can be measured at the
real size. The split sizes follow 4D-OR (test: takes 2 and 6, 1378 frames; val: takes 4 and 8, 1332 frames)."""
import os
import random

import numpy as np
import torch
from PIL import Image

from .dataset_utils import CLASS_NAMES, RELATION_NAMES, TAKE_SPLIT, scene_graph_name_to_vocab_idx

SPLIT_FRAMES = {'train': 4024, 'val': 1332, 'test': 1378}


class SyntheticORDataset:
    def __init__(self, config, split='val', root='synthetic_4dor', n_frames=None, image_size=(336, 336), d_model=4096,
                 unique_images=8, for_eval=False, shuffle_objs=False, mv_desc=False, image_format='png'):
        self.config, self.split, self.root = config, split, root
        self.classNames, self.relationNames = list(CLASS_NAMES), list(RELATION_NAMES)
        self.w_cls_obj = torch.ones(len(self.classNames))
        self.w_cls_rel = torch.ones(len(self.relationNames))
        takes = TAKE_SPLIT[split]
        n = SPLIT_FRAMES[split] if n_frames is None else n_frames
        per = [n // len(takes) + (1 if i < n % len(takes) else 0) for i in range(len(takes))]
        self.scans = [f'{t}_{i:06d}_{split}' for t, k in zip(takes, per) for i in range(k)]
        self.image_size, self.unique_images = image_size, unique_images
        if image_format not in ('png', 'jpg'):
            raise ValueError(f'image_format {image_format!r}: png or jpg')
        self.image_format = image_format
        self.d_model = d_model
        self._written = set()
        self.use_vis = bool(config.get('USE_VIS_DESC'))
        if self.use_vis:
            g = torch.Generator().manual_seed(99)
            n_rows = [4, 4, 3, 4, 4, 4, 3] if mv_desc else [1] * 7
            self.vis_descriptor_embs = [(torch.randn(r, d_model, generator=g) * 0.02).squeeze(0) if r == 1 else
                                        torch.randn(r, d_model, generator=g) * 0.02 for r in n_rows]

    def __len__(self):
        return len(self.scans)

    def _image(self, idx, cam):
        slot = (idx % self.unique_images) * 6 + cam
        w, h = self.image_size
        tag = '' if (w, h) == (336, 336) else f'_{w}x{h}'
        path = os.path.join(self.root, f'frame{slot:05d}_camera0{cam + 1}{tag}.{self.image_format}')
        if path not in self._written and not os.path.exists(path):
            os.makedirs(self.root, exist_ok=True)
            rng = np.random.default_rng(1234 + slot)
            if self.image_format == 'png':
                Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(path)
            else:
                base = Image.fromarray(rng.integers(0, 256, size=(max(h // 64, 2), max(w // 64, 2), 3), dtype=np.uint8))
                img = np.asarray(base.resize((w, h), Image.BICUBIC)).astype(np.int16)
                img += rng.integers(-6, 7, size=(h, w, 1), dtype=np.int16)
                Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(path, quality=90)
        self._written.add(path)
        return path

    def __getitem__(self, index):
        scan_id = self.scans[index]
        rng = random.Random(index)
        objs = ['anesthesia_equipment', 'operating_table', 'instrument_table', 'patient', 'human_0', 'human_1', 'human_2']
        rels = ['assisting', 'closeto', 'cutting', 'holding', 'lyingon', 'touching', 'preparing']
        triplets = []
        for _ in range(rng.randint(3, 10)):
            s, o = rng.sample(objs, 2)
            triplets.append((scene_graph_name_to_vocab_idx[s], scene_graph_name_to_vocab_idx[rng.choice(rels)], scene_graph_name_to_vocab_idx[o]))
        sample = {'scan_id': scan_id, 'take_idx': int(scan_id.split('_')[0]),
                  'image_paths': [self._image(index, c) for c in range(6)], 'relations_tokenized': triplets}
        if self.use_vis:
            sample['vis_descriptor_embs'] = self.vis_descriptor_embs
        return sample

    def collate_fn(self, batch):
        for elem in batch:
            elem['take_idx'] = int(elem['scan_id'].split('_')[0])
        return batch


def batches(dataset, batch_size, indices=None):
    """DataLoader(collate_fn=list passthrough, shuffle=False) equivalent without worker processes."""
    idx = list(range(len(dataset))) if indices is None else list(indices)
    for i in range(0, len(idx), batch_size):
        yield dataset.collate_fn([dataset[j] for j in idx[i:i + batch_size]])
