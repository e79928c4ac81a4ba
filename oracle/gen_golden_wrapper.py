"""TEST INFRASTRUCTURE ONLY -- drives the REAL reference OracleWrapper (forward / validate / infer,
scene_graph_prediction_model_oracle.py) with a scripted stand-in for the LLaVA model, to pin the host-side glue:
prompt text, CLIP preprocessing, left-padded batching, answer decoding, symbolic back-mapping, triplet parsing,
temporal memory evolution and the relation labels handed to sklearn. Writes tests/golden/wrapper_*.json|npz.

Import shims (reference files untouched): heavy modules the wrapper imports but the scripted run never executes are
replaced by stubs in sys.modules (open3d, helpers.configurations -- which creates directories at import --, and
LLaVA.llava.model.builder whose load_pretrained_model is replaced by the scripted model factory)."""
import importlib
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

from oracle import ref_harness as rh
from oracle_amd.tokenizer import SynthTokenizer

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
REF = rh.REF_ROOT


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class ScriptedModel:
    """generate() returns input_ids ++ the scripted answer of each row (EOS-terminated, pad-filled)."""

    def __init__(self, tokenizer, answers):
        self.config = types.SimpleNamespace(mm_use_im_start_end=False, image_aspect_ratio='pad', mv_type=None, tokenizer_padding_side=None)
        self.device = torch.device('cpu')
        self.tok = tokenizer
        self.answers = answers          # callable(prompt_text, row_index) -> answer string
        self.calls = []

    def generate(self, input_ids, images=None, do_sample=False, use_cache=True, max_new_tokens=300, stopping_criteria=None,
                 vis_descriptor_embs=None):
        rows = []
        for b in range(input_ids.shape[0]):
            ids = [int(t) for t in input_ids[b] if int(t) not in (0,)]
            text = self.tok.decode([t for t in ids if t != -200])
            ans = self.answers(text, len(self.calls), b)
            rows.append(self.tok.encode(ans, add_special_tokens=False) + [2])
        n = max(len(r) for r in rows)
        new = torch.tensor([r + [0] * (n - len(r)) for r in rows])
        self.calls.append({'input_ids': input_ids.clone(), 'images_shape': list(images.shape) if torch.is_tensor(images) else [list(i.shape) for i in images],
                           'images_sum': float(images.float().sum()) if torch.is_tensor(images) else None,
                           'images': images, 'n_desc': None if vis_descriptor_embs is None else [len(v) for v in vis_descriptor_embs]})
        return torch.cat([input_ids, new], dim=1)


def load_wrapper_module():
    ns = rh.load_reference()
    _pkg('LLaVA', os.path.join(REF, 'LLaVA'))
    _pkg('LLaVA.llava', os.path.join(REF, 'LLaVA', 'llava'))
    _pkg('LLaVA.llava.model', os.path.join(REF, 'LLaVA', 'llava', 'model'))
    b = types.ModuleType('LLaVA.llava.model.builder')
    b.load_pretrained_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('replaced per run'))
    sys.modules['LLaVA.llava.model.builder'] = b
    sys.modules['open3d'] = types.ModuleType('open3d')
    _pkg('helpers', os.path.join(REF, 'helpers'))
    hc = types.ModuleType('helpers.configurations')
    from pathlib import Path
    hc.OR_4D_DATA_ROOT_PATH = Path('/nonexistent/4D-OR')
    hc.INSTANCE_LABELS_PATH = hc.INSTANCE_LABELS_PRED_PATH = hc.OR_4D_DATA_ROOT_PATH
    hc.OBJECT_LABEL_MAP = {}
    hc.TAKE_SPLIT = {'train': [1, 3, 5, 7, 9, 10], 'val': [4, 8], 'test': [2, 6]}
    sys.modules['helpers.configurations'] = hc
    for name, rel in [('scene_graph_prediction.scene_graph_helpers', 'scene_graph_prediction/scene_graph_helpers'),
                      ('scene_graph_prediction.scene_graph_helpers.dataset', 'scene_graph_prediction/scene_graph_helpers/dataset'),
                      ('scene_graph_prediction.scene_graph_helpers.model', 'scene_graph_prediction/scene_graph_helpers/model'),
                      ('scene_graph_prediction.data_processing', 'scene_graph_prediction/data_processing'),
                      ('scene_graph_prediction.utils', 'scene_graph_prediction/utils')]:
        _pkg(name, os.path.join(REF, rel))
    mod = importlib.import_module('scene_graph_prediction.scene_graph_helpers.model.scene_graph_prediction_model_oracle')
    du = importlib.import_module('scene_graph_prediction.scene_graph_helpers.dataset.dataset_utils')
    return mod, du, b


ENTS = ['head surgeon', 'assistant surgeon', 'circulator', 'nurse', 'anaesthetist', 'patient', 'instrument table', 'operating table',
        'secondary table', 'anesthesia equipment', 'instrument']
PREDS = ['assisting', 'cementing', 'cleaning', 'closeTo', 'cutting', 'drilling', 'hammering', 'holding', 'lyingOn', 'manipulating',
         'preparing', 'sawing', 'suturing', 'touching']
SYM_E = list('ABCDEFGHIJK')
SYM_P = list('αβγδεζηθικλμνξ')


def scripted_answer(kind, seed):
    rng = random.Random(seed)
    n = rng.randint(2, 7)
    trip = []
    for _ in range(n):
        s, o = rng.sample(range(len(ENTS)), 2)
        p = rng.randrange(len(PREDS))
        if kind == 'symbolic':
            trip.append(f'{SYM_E[s]},{SYM_E[o]},{SYM_P[p]}')
        else:
            trip.append(f'{ENTS[s]},{ENTS[o]},{PREDS[p]}')
    junk = rng.choice(['', ' broken,triplet;', ' anesthetist,head_surgent,holding;', ' tool,nonsense thing,flying;'])
    text = '; '.join(trip) + ';' + junk
    if kind == 'symbolic' or rng.random() < 0.3:
        text = f'<SG> {text} </SG>'
    return text


def make_dataset(tmp, n, takes, use_vis, seed=0):
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    paths = []
    for c in range(6):
        w, h = rng.choice([(336, 336), (400, 300), (300, 420)])
        p = os.path.join(tmp, f'cam{c}.png')
        Image.fromarray(nprng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    obj_ids = [0, 1, 2, 6, 7, 8, 9]
    rel_ids = [20, 23, 24, 27, 28, 29, 33]
    data = []
    for i in range(n):
        take = takes[i * len(takes) // n]
        trip = []
        for _ in range(rng.randint(0, 8)):
            s, o = rng.sample(obj_ids, 2)
            trip.append((s, rng.choice(rel_ids), o))
        elem = {'scan_id': f'{take}_{i * 3}_val', 'take_idx': take, 'image_paths': paths, 'relations_tokenized': trip}
        if use_vis:
            g = torch.Generator().manual_seed(7)
            elem['vis_descriptor_embs'] = [torch.randn(16, generator=g) for _ in range(7)]
        data.append(elem)
    return data, paths


def run_case(mod, builder, name, config, model_name, kind, n_frames, batch_size, relation_names):
    tok = SynthTokenizer()
    tmp = tempfile.mkdtemp(prefix='wrap_')
    data, paths = make_dataset(tmp, n_frames, [4, 8], bool(config.get('USE_VIS_DESC')))
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={'shortest_edge': 336}, crop_size={'height': 336, 'width': 336})
    model = ScriptedModel(tok, lambda text, call, b: scripted_answer(kind, call * 16 + b))
    mod.load_pretrained_model = lambda *a, **k: (tok, model, proc, 2048)
    random.seed(42)
    w = mod.OracleWrapper(config, num_class=11, num_rel=len(relation_names), weights_obj=None, weights_rel=None,
                          relationNames=relation_names, model_path=f'/x/{model_name}', mv_desc=False)
    captured = {}
    orig = w.evaluate_predictions

    def spy(epoch_loss, split, logging_information=None):
        captured['preds'] = {str(k): [int(x) for x in v] for k, v in w.val_take_rel_preds.items()}
        captured['gts'] = {str(k): [int(x) for x in v] for k, v in w.val_take_rel_gts.items()}
        captured['macro_f1'] = float(orig(epoch_loss, split, logging_information))
        return captured['macro_f1']

    w.evaluate_predictions = spy
    loader = [data[i:i + batch_size] for i in range(0, len(data), batch_size)]
    outputs = []
    orig_forward = w.forward

    def fwd(batch):
        o = orig_forward(batch)
        outputs.append(list(o))
        return o

    w.forward = fwd
    raw = w.infer(loader)
    rec = {
        'config': config, 'model_name': model_name, 'kind': kind, 'batch_size': batch_size,
        'image_sizes': [Image.open(p).size for p in paths],
        'dataset': [{k: (v if k != 'vis_descriptor_embs' else None) for k, v in e.items()} for e in data],
        'input_ids': [c['input_ids'].tolist() for c in model.calls],
        'images_shape': [c['images_shape'] for c in model.calls],
        'n_desc': [c['n_desc'] for c in model.calls],
        'forward_outputs': outputs,
        'raw_predictions': {k: [list(t) for t in v] for k, v in raw.items()},
        'take_rel_preds': captured['preds'], 'take_rel_gts': captured['gts'], 'macro_f1': captured['macro_f1'],
    }
    img0 = model.calls[0]['images']
    img0 = img0 if torch.is_tensor(img0) else torch.stack(list(img0))
    npz = {'images0_sub': img0.float().numpy()[..., ::23, ::29], 'images0_sum': np.float64(img0.double().sum().item()),
           'images0_shape': np.array(img0.shape)}
    return rec, npz


def main():
    mod, du, builder = load_wrapper_module()
    rel_names = ['Assisting', 'Cementing', 'Cleaning', 'CloseTo', 'Cutting', 'Drilling', 'Hammering', 'Holding', 'LyingOn',
                 'Operating', 'Preparing', 'Sawing', 'Suturing', 'Touching', 'none']
    base = {'LR': 0.00003, 'USE_VIS_DESC': False, 'MODEL': {'mv_type': 'learned'}}
    cases = {
        'plain_mv_b3': (dict(base, CAMERAS=[2, 1, 3, 5]), 'oracle_mv_learned', 'plain', 6, 3),
        'plain_single_b1': (dict(base, CAMERAS=[2], MODEL={'mv_type': 'concat'}), 'oracle', 'plain', 2, 1),
        'symbolic_b2': (dict(base, CAMERAS=[2, 1, 3, 5]), 'oracle_mv_learned_symbolic', 'symbolic', 4, 2),
        'visual_b2': (dict(base, CAMERAS=[2, 1, 3, 5], USE_VIS_DESC=True), 'oracle_mv_learned_symbolic_visual', 'symbolic', 4, 2),
        'temporal_b1': (dict(base, CAMERAS=[2, 1, 3, 5], temporality='PRED'), 'oracle_mv_learned_temporal_pred', 'plain', 10, 1),
    }
    allrec, allnpz = {}, {}
    for name, (cfg, mname, kind, n, bs) in cases.items():
        rec, npz = run_case(mod, builder, name, cfg, mname, kind, n, bs, rel_names)
        allrec[name] = rec
        for k, v in npz.items():
            allnpz[f'{name}.{k}'] = v
        print(name, 'frames', n, 'macro_f1', rec['macro_f1'])
    # vocabulary helpers of dataset_utils.py:292-359
    names = ['Patient', 'human_3', 'anaesthetist_station', 'close to', 'manipulating', 'tool', 'opertating_table', 'CloseTo', 'LyingOn']
    allrec['vocab'] = {'name_to_idx': {n: du.map_scene_graph_name_to_vocab_idx(n) for n in names},
                       'idx_to_name': {str(i): du.map_vocab_idx_to_scene_graph_name(i) for i in list(range(12)) + list(range(20, 34))},
                       'reversed_role_synonyms': du.reversed_role_synonyms}
    with open(os.path.join(OUT, 'wrapper_kat.json'), 'w') as f:
        json.dump(allrec, f, indent=1, ensure_ascii=False)
    np.savez_compressed(os.path.join(OUT, 'wrapper_images.npz'), **allnpz)
    print('wrote wrapper goldens')


if __name__ == '__main__':
    main()
