"""Raw predicted triplets -> the naming the 4D-OR evaluator expects (reference
scene_graph_prediction/map_raw_prediction_to_4dorevaluator.py:37-80).

Per scan: role aliases are normalised; every human role that occurs (the patient counts as one here, :45-48) is numbered
`human_<i>` in alphabetical order of the roles present; names go through the vocabulary (synonyms, lower-casing) and come
back in the evaluator's spelling (objects lower_snake, predicates capitalised: `CloseTo`, `LyingOn`, `Operating`);
self-relations and triplets with a name outside the vocabulary (including a sixth human) are dropped, the latter with a
printed `Error:` line as in the reference."""
import json
from pathlib import Path

from .scene_graph_helpers.dataset.dataset_utils import RELATION_NAMES, map_scene_graph_name_to_vocab_idx, reversed_role_synonyms
from .scene_graph_helpers.dataset.dataset_utils import _OBJECTS

_ROLES = ('head surgeon', 'assistant surgeon', 'circulator', 'nurse', 'anaesthetist', 'patient')
# evaluator spelling: objects ids 0..11 as in the vocabulary, predicates ids 20..33 capitalised (reference :6-33)
vocab_idx_to_scene_graph_name = {**dict(enumerate(_OBJECTS)), **{20 + i: n for i, n in enumerate(RELATION_NAMES[:-1])}}
scene_graph_name_to_vocab_idx = {v: k for k, v in vocab_idx_to_scene_graph_name.items()}


def map_triplets_to_4dor_eval(triplets):
    normalised, roles = [], set()
    for sub, pred, obj in triplets:
        sub = reversed_role_synonyms.get(sub, sub)
        obj = reversed_role_synonyms.get(obj, obj)
        roles.update(x for x in (sub, obj) if x in _ROLES)
        normalised.append((sub, pred, obj))
    index = {role: f'human_{i}' for i, role in enumerate(sorted(roles))}

    def name(x):
        return vocab_idx_to_scene_graph_name[map_scene_graph_name_to_vocab_idx(index.get(x, x).replace(' ', '_'))]

    mapped = []
    for sub, pred, obj in normalised:
        try:
            s, o = name(sub), name(obj)
            p = vocab_idx_to_scene_graph_name[map_scene_graph_name_to_vocab_idx(pred)]
        except Exception as e:   # unknown name: the reference prints and skips
            print(f'Error: {e}')
            continue
        if s != o:
            mapped.append((s, p, o))
    return mapped


def main(raw_triplets_json_path='scan_relations_oracle_mv_learned_test_temporalmodelnottemporal.json'):
    """scan_id -> triplets JSON in, `<stem>_mapped.json` beside it out (reference :71-80; the path is a literal there)."""
    path = Path(raw_triplets_json_path)
    with path.open('r') as f:
        raw = json.load(f)
    mapped = {scan_id: map_triplets_to_4dor_eval(t) for scan_id, t in raw.items()}
    out = path.parent / (path.stem + '_mapped.json')
    with out.open('w') as f:
        json.dump(mapped, f)
    return out


if __name__ == '__main__':
    import sys
    main(*sys.argv[1:2])
