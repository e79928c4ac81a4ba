"""CPU: the OracleWrapper mirror against goldens produced by the REAL reference OracleWrapper driven by the same
scripted model (oracle/gen_golden_wrapper.py): token ids of the prompts, image tensors, decoded / back-mapped answers,
raw triplets, temporal-memory evolution and the labels given to sklearn must all be identical."""
import json
import os
import random
import tempfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.gen_golden_wrapper import ScriptedModel, make_dataset, scripted_answer
from oracle_amd.llava.mm_utils import ClipImageProcessor
from oracle_amd.scene_graph_prediction.scene_graph_helpers.dataset import dataset_utils as du
from oracle_amd.scene_graph_prediction.scene_graph_helpers.model import scene_graph_prediction_model_oracle as wmod
from oracle_amd.tokenizer import SynthTokenizer

KAT = json.load(open(os.path.join(GOLDEN, 'wrapper_kat.json')))
IMG = np.load(os.path.join(GOLDEN, 'wrapper_images.npz'))
REL = ['Assisting', 'Cementing', 'Cleaning', 'CloseTo', 'Cutting', 'Drilling', 'Hammering', 'Holding', 'LyingOn', 'Operating',
       'Preparing', 'Sawing', 'Suturing', 'Touching', 'none']


@pytest.mark.parametrize('name', ['plain_mv_b3', 'plain_single_b1', 'symbolic_b2', 'visual_b2', 'temporal_b1'])
def test_wrapper_case(name, monkeypatch, capsys):
    g = KAT[name]
    tok = SynthTokenizer()
    tmp = tempfile.mkdtemp(prefix='wrap_')
    data, paths = make_dataset(tmp, len(g['dataset']), [4, 8], bool(g['config'].get('USE_VIS_DESC')))
    model = ScriptedModel(tok, lambda text, call, b: scripted_answer(g['kind'], call * 16 + b))
    proc = ClipImageProcessor(336)
    monkeypatch.setattr(wmod, 'load_pretrained_model', lambda *a, **k: (tok, model, proc, 2048))
    random.seed(42)
    w = wmod.OracleWrapper(g['config'], num_class=11, num_rel=len(REL), weights_obj=None, weights_rel=None, relationNames=REL,
                           model_path=f'/x/{g["model_name"]}', mv_desc=False)
    outputs = []
    fwd = w.forward
    w.forward = lambda batch: (outputs.append(list(fwd(batch))) or outputs[-1])
    bs = g['batch_size']
    raw = w.infer([data[i:i + bs] for i in range(0, len(data), bs)])
    assert [c['input_ids'].tolist() for c in model.calls] == g['input_ids']
    assert [c['images_shape'] for c in model.calls] == g['images_shape']
    assert [c['n_desc'] for c in model.calls] == g['n_desc']
    assert outputs == g['forward_outputs']
    assert {k: [list(t) for t in v] for k, v in raw.items()} == g['raw_predictions']
    preds, gts = w.last_take_rel
    assert {str(k): [int(x) for x in v] for k, v in preds.items()} == g['take_rel_preds']
    assert {str(k): [int(x) for x in v] for k, v in gts.items()} == g['take_rel_gts']
    assert abs(w.last_macro_f1 - g['macro_f1']) < 1e-12
    img0 = model.calls[0]['images']
    img0 = img0.float().numpy()
    assert list(img0.shape) == list(IMG[f'{name}.images0_shape'])
    sub = IMG[f'{name}.images0_sub']
    assert np.abs(img0[..., ::23, ::29] - sub).max() <= 1e-2      # bf16 image tensor of the same PIL pipeline
    assert abs(float(img0.astype(np.float64).sum()) - float(IMG[f'{name}.images0_sum'])) <= 1e-3 * abs(float(IMG[f'{name}.images0_sum'])) + 50


def test_vocab_helpers():
    v = KAT['vocab']
    for n, i in v['name_to_idx'].items():
        assert du.map_scene_graph_name_to_vocab_idx(n) == i
    for i, n in v['idx_to_name'].items():
        assert du.map_vocab_idx_to_scene_graph_name(int(i)) == n
    assert du.reversed_role_synonyms == v['reversed_role_synonyms']
    with pytest.raises(KeyError):
        du.map_scene_graph_name_to_vocab_idx('flying')


def test_prompt_digests():
    import hashlib
    from oracle_amd.scene_graph_prediction import prompts
    kat = json.load(open(os.path.join(GOLDEN, 'text_kat.json')))['prompt_digest']
    got = {'symbolic': prompts.symbolic_prompt(), 'plain': prompts.plain_prompt(), 'visual': prompts.visual_prompt(False),
           'visual_mv': prompts.visual_prompt(True)}
    for k, text in got.items():
        assert len(text) == kat[k]['chars'] and hashlib.sha256(text.encode()).hexdigest() == kat[k]['sha256'], k
