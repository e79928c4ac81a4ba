"""TEST INFRASTRUCTURE ONLY -- drives the REAL reference AdaptabilityOracleWrapper (forward / validate,
scene_graph_prediction/scene_graph_helpers/model/adaptability_wrapper_oracle.py) and the REAL
map_triplets_to_4dor_eval (scene_graph_prediction/map_raw_prediction_to_4dorevaluator.py) with a scripted stand-in for
the LLaVA model, and writes tests/golden/adapt_kat.json: prompt token ids, number of visual-descriptor embeddings handed
to generate(), decoded / back-mapped answers, test-condition verdicts, the classification report text, and the
evaluator-format triplets.

Import shims are those of oracle/gen_golden_wrapper.py (reference files untouched). The wrapper reads descriptor
tensors from paths relative to the working directory, so the run happens inside a temporary directory that holds
tensors written by this script (torch.save of random tensors; nothing is loaded from the reference)."""
import contextlib
import importlib
import io
import json
import os
import random
import tempfile
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from oracle.gen_golden_wrapper import OUT, ScriptedModel, load_wrapper_module
from oracle_amd.tokenizer import SynthTokenizer

LABELS = ['cutting', 'drilling', 'kuka', 'mako', 'robotic_sawing']
VIS_NAMES = ['anesthesia equipment', 'cementing', 'cutting', 'drilling', 'hammering', 'sawing', 'suturing']
D_EMB = 16


def scripted_adapt_answer(kind, seed):
    """Answers that sometimes contain the new symbol / name so that both verdicts occur."""
    rng = random.Random(seed)
    if kind == 'plain':
        pool = ['head surgeon,patient,drilling', 'nurse,instrument table,closeTo', 'head surgeon,patient,cutting',
                'assistant surgeon,head surgeon,assisting', 'head surgeon,patient,sawing', 'circulator,kuka,touching']
    else:
        pool = ['A,F,ζ', 'D,G,δ', 'A,F,ε', 'B,A,α', 'A,F,μ', 'C,L,ξ', 'L,F,δ', 'A,F,ο']
    trip = rng.sample(pool, rng.randint(1, 4))
    text = '; '.join(trip) + ';'
    if kind != 'plain':
        text = f'<SG> {text} </SG>'
    return text


def make_cases(root, seed=0):
    """The item dicts AdaptabilityORDataset.__getitem__ yields (adaptability_or_dataset.py:36-41), on files made here."""
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    cases = []
    for label in LABELS:
        for k in range(2):
            d = Path(root) / 'adaptability_4dor' / label
            for sub in ('positives', 'negatives', 'visual_features'):
                (d / sub).mkdir(parents=True, exist_ok=True)
            name = f'{label}_{k}'
            for sub in ('positives', 'negatives'):
                Image.fromarray(nprng.integers(0, 256, size=(60, 80, 3), dtype=np.uint8)).save(d / sub / f'{name}.jpg')
            g = torch.Generator().manual_seed(100 + len(cases))
            torch.save(torch.randn(D_EMB, generator=g), d / 'visual_features' / f'{name}_vis_crop.pt')
            attrs = {'color': rng.choice(['red', 'silver']), 'size': rng.choice(['small', 'large']), 'shape': rng.choice(['round', 'L-shape']),
                     'texture': rng.choice(['metal', 'plastic']), 'object_type': rng.choice(['robot arm', 'power tool'])}
            target = 'sawing' if label == 'robotic_sawing' else label
            cond = f'<head surgeon>,<patient>,<{target}>' if label not in ('kuka', 'mako') else f'{target}'
            for is_pos in (True, False):
                img = (d / ('positives' if is_pos else 'negatives') / f'{name}.jpg').absolute()
                cases.append({'image_path': img, 'vis_desc_path': img.parent.parent / 'visual_features' / f'{name}_vis.jpg',
                              'test_condition': cond, 'textual_attributes': attrs, 'label': label, 'is_pos': is_pos})
    return cases


def write_original_crops(root):
    d = Path(root) / 'synthetic_or_generation' / 'original_crops'
    d.mkdir(parents=True, exist_ok=True)
    for i, n in enumerate(VIS_NAMES):
        g = torch.Generator().manual_seed(500 + i)
        torch.save(torch.randn(D_EMB, generator=g), d / f'{n}_cam2.pt')


def public_case(c, root):
    return {k: (str(Path(v).relative_to(root)) if k in ('image_path', 'vis_desc_path') else v) for k, v in c.items()}


def run_case(amod, config, model_name, kind):
    tok = SynthTokenizer()
    tmp = os.path.realpath(tempfile.mkdtemp(prefix='adapt_'))
    cases = make_cases(tmp)
    write_original_crops(tmp)
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={'shortest_edge': 336}, crop_size={'height': 336, 'width': 336})
    model = ScriptedModel(tok, lambda text, call, b: scripted_adapt_answer(kind, call))
    amod.load_pretrained_model = lambda *a, **k: (tok, model, proc, 2048)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        w = amod.AdaptabilityOracleWrapper(config, labels=LABELS, model_path=f'/x/{model_name}')
        verdicts = []
        fwd = w.forward
        w.forward = lambda batch: (verdicts.append(list(fwd(batch))) or tuple(verdicts[-1]))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            w.validate([[c] for c in cases])
    finally:
        os.chdir(cwd)
    return {'config': config, 'model_name': model_name, 'kind': kind,
            'cases': [public_case(c, tmp) for c in cases],
            'input_ids': [c['input_ids'].tolist() for c in model.calls],
            'images_shape': [c['images_shape'] for c in model.calls],
            'n_desc': [c['n_desc'] for c in model.calls],
            'verdicts': verdicts, 'report': buf.getvalue()}


def evaluator_cases():
    return {
        'plain': [['head surgeon', 'assisting', 'patient'], ['nurse', 'closeTo', 'instrument table'], ['patient', 'lyingOn', 'operating table']],
        'synonyms': [['head_surgent', 'manipulating', 'anesthesia equipment'], ['anesthetist', 'close to', 'tool'], ['circulator', 'touching', 'opertating_table']],
        'six_roles': [['head surgeon', 'assisting', 'assistant surgeon'], ['circulator', 'closeTo', 'nurse'], ['anaesthetist', 'preparing', 'patient'],
                      ['nurse', 'holding', 'instrument']],
        'self_and_unknown': [['nurse', 'touching', 'nurse'], ['nurse', 'flying', 'patient'], ['robot', 'closeTo', 'patient'], ['patient', 'lyingOn', 'operating table']],
        'empty': [],
    }


def main():
    mod, du, builder = load_wrapper_module()
    amod = importlib.import_module('scene_graph_prediction.scene_graph_helpers.model.adaptability_wrapper_oracle')
    base = {'USE_VIS_DESC': False, 'MODEL': {'mv_type': 'learned'}, 'CAMERAS': [2]}
    rec = {
        'plain': run_case(amod, dict(base), 'oracle_mv_learned', 'plain'),
        'symbolic': run_case(amod, dict(base), 'oracle_mv_learned_symbolic', 'symbolic'),
        'visual': run_case(amod, dict(base, USE_VIS_DESC=True), 'oracle_mv_learned_symbolic_visual', 'symbolic'),
    }
    for k, v in rec.items():
        print(k, 'cases', len(v['cases']), 'fulfilled', sum(1 for t in v['verdicts'] if t[0]))
    emod = importlib.import_module('scene_graph_prediction.map_raw_prediction_to_4dorevaluator')
    ev = {}
    for name, trip in evaluator_cases().items():
        with contextlib.redirect_stdout(io.StringIO()):
            ev[name] = {'in': trip, 'out': [list(t) for t in emod.map_triplets_to_4dor_eval([tuple(t) for t in trip])]}
    rec['evaluator_map'] = ev
    with open(os.path.join(OUT, 'adapt_kat.json'), 'w') as f:
        json.dump(rec, f, indent=1, ensure_ascii=False)
    print('wrote adapt_kat.json')


if __name__ == '__main__':
    main()
