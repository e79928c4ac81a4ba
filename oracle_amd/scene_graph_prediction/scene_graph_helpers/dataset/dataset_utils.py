"""Scene-graph vocabulary of 4D-OR used by the evaluation glue (values: reference
scene_graph_prediction/scene_graph_helpers/dataset/dataset_utils.py:292-359)."""

_OBJECTS = ['anesthesia_equipment', 'operating_table', 'instrument_table', 'secondary_table', 'instrument', 'object', 'patient',
            'human_0', 'human_1', 'human_2', 'human_3', 'human_4']                       # ids 0..11
_PREDICATES = ['assisting', 'cementing', 'cleaning', 'closeto', 'cutting', 'drilling', 'hammering', 'holding', 'lyingon',
               'operating', 'preparing', 'sawing', 'suturing', 'touching']               # ids 20..33
scene_graph_name_to_vocab_idx = {**{n: i for i, n in enumerate(_OBJECTS)}, **{n: 20 + i for i, n in enumerate(_PREDICATES)}}
vocab_idx_to_scene_graph_name = {v: k for k, v in scene_graph_name_to_vocab_idx.items()}

synonyms = {
    'anesthesia_equipment': ['anaesthesia_equipment', 'anesthesia equipment', 'anaesthetist_station'],
    'closeto': ['close', 'close to'],
    'operating': ['manipulating'],
    'instrument': ['tool'],
    'operating_table': ['opertating_table'],
}
role_synonyms = {'head_surgeon': ['head_surgent'], 'anaesthetist': ['anesthetist']}


def reverse_synonym_mapping(table):
    return {alias: name for name, aliases in table.items() for alias in aliases}


reversed_synonyms = reverse_synonym_mapping(synonyms)
reversed_role_synonyms = reverse_synonym_mapping(role_synonyms)

RELATION_NAMES = ['Assisting', 'Cementing', 'Cleaning', 'CloseTo', 'Cutting', 'Drilling', 'Hammering', 'Holding', 'LyingOn',
                  'Operating', 'Preparing', 'Sawing', 'Suturing', 'Touching', 'none']    # data/relationships.txt + 'none'
CLASS_NAMES = ['anesthesia_equipment', 'human_0', 'human_1', 'human_2', 'human_3', 'human_4', 'human_5', 'instrument',
               'instrument_table', 'operating_table', 'secondary_table']
TAKE_SPLIT = {'train': [1, 3, 5, 7, 9, 10], 'val': [4, 8], 'test': [2, 6]}


def map_scene_graph_name_to_vocab_idx(name):
    name = name.lower()
    return scene_graph_name_to_vocab_idx[reversed_synonyms.get(name, name)]   # KeyError for unknown names, as the reference


def map_vocab_idx_to_scene_graph_name(vocab_idx):
    return vocab_idx_to_scene_graph_name[vocab_idx]
