"""CPU: the AdaptabilityOracleWrapper mirror, the adaptability dataset and the evaluator-format mapper against goldens
produced by the REAL reference code driven by the same scripted model (oracle/gen_golden_adapt.py): prompt token ids,
number and order of descriptor embeddings, back-mapped answers, verdicts and the printed report must be identical."""
import json
import os
import tempfile

import pytest
import torch

from conftest import GOLDEN
from oracle.gen_golden_adapt import LABELS, make_cases, public_case, scripted_adapt_answer, write_original_crops
from oracle.gen_golden_wrapper import ScriptedModel
from oracle_amd.llava.mm_utils import ClipImageProcessor
from oracle_amd.scene_graph_prediction import map_raw_prediction_to_4dorevaluator as emap
from oracle_amd.scene_graph_prediction.scene_graph_helpers.dataset.adaptability_or_dataset import AdaptabilityORDataset
from oracle_amd.scene_graph_prediction.scene_graph_helpers.model import adaptability_wrapper_oracle as amod
from oracle_amd.tokenizer import SynthTokenizer

KAT = json.load(open(os.path.join(GOLDEN, 'adapt_kat.json')))


@pytest.mark.parametrize('name', ['plain', 'symbolic', 'visual'])
def test_adaptability_case(name, monkeypatch):
    g = KAT[name]
    tok = SynthTokenizer()
    tmp = os.path.realpath(tempfile.mkdtemp(prefix='adapt_'))
    cases = make_cases(tmp)
    write_original_crops(tmp)
    assert [public_case(c, tmp) for c in cases] == g['cases']
    model = ScriptedModel(tok, lambda text, call, b: scripted_adapt_answer(g['kind'], call))
    monkeypatch.setattr(amod, 'load_pretrained_model', lambda *a, **k: (tok, model, ClipImageProcessor(336), 2048))
    monkeypatch.chdir(tmp)
    w = amod.AdaptabilityOracleWrapper(g['config'], labels=LABELS, model_path=f'/x/{g["model_name"]}')
    verdicts = []
    fwd = w.forward
    w.forward = lambda batch: (verdicts.append(list(fwd(batch))) or tuple(verdicts[-1]))
    report = w.validate([[c] for c in cases])
    assert [c['input_ids'].tolist() for c in model.calls] == g['input_ids']
    assert [c['images_shape'] for c in model.calls] == g['images_shape']
    assert [c['n_desc'] for c in model.calls] == g['n_desc']
    assert verdicts == g['verdicts']
    assert report + '\n' == g['report']


def test_visual_descriptor_order(monkeypatch):
    """The embeddings reach generate() in symbol order (entities, then predicates) with the new concept's crop in the
    place of its symbol: checked on tensor VALUES, which the golden's counts alone do not pin."""
    tok = SynthTokenizer()
    tmp = os.path.realpath(tempfile.mkdtemp(prefix='adapt_'))
    cases = make_cases(tmp)
    write_original_crops(tmp)
    seen = {}

    class Spy(ScriptedModel):
        def generate(self, input_ids, vis_descriptor_embs=None, **kw):
            seen['embs'] = vis_descriptor_embs
            return super().generate(input_ids, vis_descriptor_embs=vis_descriptor_embs, **kw)

    model = Spy(tok, lambda text, call, b: 'A,F,δ;')
    monkeypatch.setattr(amod, 'load_pretrained_model', lambda *a, **k: (tok, model, ClipImageProcessor(336), 2048))
    monkeypatch.chdir(tmp)
    w = amod.AdaptabilityOracleWrapper(KAT['visual']['config'], labels=LABELS, model_path='/x/oracle_symbolic_visual')
    load = lambda p: torch.load(p, weights_only=True).to(torch.bfloat16)
    crops = 'synthetic_or_generation/original_crops'
    # new entity 'kuka' -> symbol L, after anesthesia equipment (J); predicates beta, epsilon, zeta, eta, mu, nu
    kuka = next(c for c in cases if c['label'] == 'kuka')
    w.forward([kuka])
    want = [f'{crops}/anesthesia equipment_cam2.pt', str(kuka['vis_desc_path']).replace('.jpg', '_crop.pt')] + \
           [f'{crops}/{n}_cam2.pt' for n in ('cementing', 'cutting', 'drilling', 'hammering', 'sawing', 'suturing')]
    assert len(seen['embs']) == 8 and all(torch.equal(e, load(p)) for e, p in zip(seen['embs'], want))
    # redefined predicate 'drilling' keeps its symbol and takes the case's crop
    drill = next(c for c in cases if c['label'] == 'drilling')
    w.forward([drill])
    want = [f'{crops}/{n}_cam2.pt' for n in ('anesthesia equipment', 'cementing', 'cutting')] + \
           [str(drill['vis_desc_path']).replace('.jpg', '_crop.pt')] + [f'{crops}/{n}_cam2.pt' for n in ('hammering', 'sawing', 'suturing')]
    assert len(seen['embs']) == 7 and all(torch.equal(e, load(p)) for e, p in zip(seen['embs'], want))
    with pytest.raises(NotImplementedError):
        w.forward([dict(drill, label='welding')])


def test_adaptability_dataset(tmp_path):
    root = tmp_path / 'adaptability_4dor'
    for label in ('mako', 'cutting'):
        for sub in ('positives', 'negatives', 'visual_features'):
            (root / label / sub).mkdir(parents=True)
        (root / label / 'positives' / 'c0.json').write_text(json.dumps(
            {'vis_img_name': 'v0.jpg', 'test_condition': f'<x>,<y>,<{label}>', 'textual_attributes': {'color': 'red'}}))
    ds = AdaptabilityORDataset({}, root)
    assert ds.labels == ['cutting', 'mako'] and len(ds) == 4
    pos, neg = ds[0], ds[1]
    assert pos['is_pos'] and not neg['is_pos'] and pos['label'] == 'cutting'
    assert pos['image_path'] == (root / 'cutting' / 'positives' / 'c0.jpg').absolute()
    assert neg['image_path'] == (root / 'cutting' / 'negatives' / 'c0.jpg').absolute()
    assert pos['vis_desc_path'] == (root / 'cutting' / 'visual_features' / 'v0.jpg').absolute()
    assert pos['test_condition'] == '<x>,<y>,<cutting>' and pos['textual_attributes'] == {'color': 'red'}


def test_evaluator_format(tmp_path, capsys):
    for name, case in KAT['evaluator_map'].items():
        got = emap.map_triplets_to_4dor_eval([tuple(t) for t in case['in']])
        assert [list(t) for t in got] == case['out'], name
    assert 'Error:' in capsys.readouterr().out        # unknown names are reported and skipped, as in the reference
    src = tmp_path / 'scan_relations_x.json'
    src.write_text(json.dumps({'2_000001_test': KAT['evaluator_map']['plain']['in'], '2_000002_test': []}))
    out = emap.main(src)
    assert out.name == 'scan_relations_x_mapped.json'
    assert json.loads(out.read_text()) == {'2_000001_test': KAT['evaluator_map']['plain']['out'], '2_000002_test': []}
