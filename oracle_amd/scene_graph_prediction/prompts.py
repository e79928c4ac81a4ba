"""The user-turn prompts of OracleWrapper.forward (reference
scene_graph_prediction/scene_graph_helpers/model/scene_graph_prediction_model_oracle.py:71-81), composed from the
entity / predicate tables instead of being stored as literals. tests/golden/text_kat.json pins their SHA-256."""
from ..llava.constants import VIS_DESCRIPTOR_TOKEN

ENTITIES = ['head surgeon', 'assistant surgeon', 'circulator', 'nurse', 'anaesthetist', 'patient', 'instrument table',
            'operating table', 'secondary table', 'anesthesia equipment', 'instrument']
PREDICATES = ['assisting', 'cementing', 'cleaning', 'closeTo', 'cutting', 'drilling', 'hammering', 'holding', 'lyingOn',
              'manipulating', 'preparing', 'sawing', 'suturing', 'touching']
ENTITY_SYMBOLS = list('ABCDEFGHIJK')
PREDICATE_SYMBOLS = list('αβγδεζηθικλμνξ')
# symbol -> name, used to map symbolic answers back (reference ..._oracle.py:153-155)
SYMBOL_TO_NAME = dict(zip(ENTITY_SYMBOLS + PREDICATE_SYMBOLS, ENTITIES + PREDICATES))

TASK = ('Given the following scene graph memory representation, generate a scene graph for timepoint T. '
        'The output should strictly be a list of triplets, each in the format "entity1,entity2,predicate;". '
        'Do not provide a narrative or descriptive text.')

_ENTITY_KNOWLEDGE = [
    'primary operator in surgery, handles critical tasks', 'supports head surgeon, assists in surgical tasks',
    'coordinates OR activities and tools', 'assists in surgical prep and recovery', 'administers anesthesia, monitors patient',
    'undergoes surgical procedure', 'instrument table, blue, large, rectangular, matte',
    'operating table, black, large, rectangular, rubber', 'secondary table, gray, large, rectangular, metal',
    'anesthesia equipment, white, large, irregular, matte', 'instrument, handheld']
_TOOL = 'use of a tool: '
_PREDICATE_KNOWLEDGE = [
    'collaboration between staff', _TOOL + 'black, handheld, straight, metal, surgical bone cement gun',
    _TOOL + 'white, handheld, irregular, sanitization equipment', 'proximity of medical staff or equipment to each other in OR',
    _TOOL + 'white, small, straight, plastic, surgical scalpel', _TOOL + 'orange, handheld, L-shape, plastic, surgical drill',
    _TOOL + 'brown, handheld, T-shape, metal, surgical hammer', 'grasping surgical instruments',
    'patient positioned on the operating table', 'handling of medical objects like operating tables or anesthesia machines',
    'includes draping and sterilization', _TOOL + 'green, handheld, round, plastic, surgical bone saw',
    _TOOL + 'gray, small, straight, metal, surgical scissors', 'physical contact between entities']
# visual-descriptor variant: which definitions are replaced by image-crop embeddings, and by how many crops in the
# multi-view-descriptor form (order of crops: anesthesia machine, cementing, cutting, drilling, hammering, sawing, suturing)
_VIS_ENTITY = {'J': 4}
_VIS_PREDICATE = {'β': 4, 'ε': 3, 'ζ': 4, 'η': 4, 'μ': 4, 'ν': 3}


def _listing(names):
    return '[' + ', '.join(names) + ']'


def plain_prompt():
    return f'Entities: {_listing(ENTITIES)}. Predicates: {_listing(PREDICATES)}. {TASK}'


def _knowledge(entity_defs, predicate_defs):
    items = [f'{s}: {d}.' for s, d in zip(ENTITY_SYMBOLS, entity_defs)] + [f'{s}: {d}.' for s, d in zip(PREDICATE_SYMBOLS, predicate_defs)]
    return ' '.join(items)


def symbolic_prompt():
    head = f'Entities: {_listing(ENTITY_SYMBOLS)}. Predicates: {_listing(PREDICATE_SYMBOLS)}.'
    return f'{head} <knowledge_start> {_knowledge(_ENTITY_KNOWLEDGE, _PREDICATE_KNOWLEDGE)} <knowledge_end> {TASK}'


def visual_prompt(mv_desc=False):
    tok = lambda n: ' '.join([VIS_DESCRIPTOR_TOKEN] * (n if mv_desc else 1))
    ents = [tok(_VIS_ENTITY[s]) if s in _VIS_ENTITY else d for s, d in zip(ENTITY_SYMBOLS, _ENTITY_KNOWLEDGE)]
    ents[ENTITY_SYMBOLS.index('K')] = 'Instrument, handheld'
    preds = [_TOOL + tok(_VIS_PREDICATE[s]) if s in _VIS_PREDICATE else d for s, d in zip(PREDICATE_SYMBOLS, _PREDICATE_KNOWLEDGE)]
    head = f'Entities: {_listing(ENTITY_SYMBOLS)}. Predicates: {_listing(PREDICATE_SYMBOLS)}.'
    return f'{head} <knowledge_start> {_knowledge(ents, preds)} <knowledge_end> {TASK}'
