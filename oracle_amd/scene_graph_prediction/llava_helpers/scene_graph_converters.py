"""Answer parsing and the temporal scene-graph memory, with the reference's function names and results
(scene_graph_prediction/llava_helpers/scene_graph_converters.py:9-22,52-87,94-172). Pure string work on the host;
bit-exact against tests/golden/text_kat.json (outputs of the reference functions, incl. the `random.shuffle` order
under seed 42)."""
import random

_STRIP = ('.', '</s>', '<s>')


def triplet_strings(answer):
    """The ';'-separated fragments the reference parses: the text between <SG> and </SG> when both are present in
    that order, else the whole answer."""
    a, b = answer.find('<SG>'), answer.find('</SG>')
    if a != -1 and b != -1 and a < b:
        answer = answer.split('<SG>')[1].split('</SG>')[0].strip()
    return answer.split(';')


def parse_llava_sg(llava_sg):
    """'sub,obj,pred; ...' -> [(sub, pred, obj)]; fragments without exactly three fields are skipped."""
    out = []
    for frag in triplet_strings(llava_sg):
        for s in _STRIP:
            frag = frag.replace(s, '')
        fields = [f.strip() for f in frag.strip().split(',')]
        if frag.strip() == '' or len(fields) != 3:
            continue
        sub, obj, pred = fields
        out.append((sub, pred, obj))
    return out


def collapse_sgs(sgs):
    """Replay a change log [(timepoint, (sub, pred, obj))] into the current {(sub, obj): pred}."""
    state = {}
    for _, (sub, pred, obj) in sgs:
        if pred.startswith('not '):
            state.pop((sub, obj), None)
        else:
            state[(sub, obj)] = pred
    return state


def llava_sg_to_surgery_sg(llava_sgs, entity_of_interest=None, IRRELEVANT_PREDS=None):
    """History of predicted graphs -> change log with 'not <pred>' removals; the modifications of each timepoint are
    shuffled with the global `random` state exactly once per frame, as the reference does."""
    if entity_of_interest is not None:
        raise NotImplementedError('entity_of_interest filtering is not used on the inference path')
    irrelevant = IRRELEVANT_PREDS or []
    log = []
    for elem in llava_sgs:
        t = elem['timepoint_idx']
        before = collapse_sgs(log)
        now = {(sub, obj): pred for (sub, pred, obj) in elem['scene_graph']
               if pred not in irrelevant and sub != 'none' and obj != 'none'}
        changes = [(t, (sub, pred, obj)) for (sub, obj), pred in now.items() if (sub, obj) not in before]
        changes += [(t, (sub, f'not {pred}', obj)) for (sub, obj), pred in before.items() if (sub, obj) not in now]
        random.shuffle(changes)
        log.extend(changes)
    return log


def surgery_sg_to_memory_str(surgery_sg_triplets, current_timepoint, TEMPORAL_STYLE='longshort', DROP_HISTORY=False):
    """'Long: ' = first occurrence of every non-negated triplet before the last five changes, 'Short: ' = the last five
    changes verbatim; entries rendered 'sub,obj,pred; ' and the trailing '; ' removed."""
    def keep():
        return DROP_HISTORY is False or not (random.random() < DROP_HISTORY)

    text = ''
    if TEMPORAL_STYLE in ('long', 'longshort'):
        text += 'Long: '
        seen = set()
        for _, (sub, pred, obj) in surgery_sg_triplets[:-5]:
            if pred.startswith('not ') or (sub, obj, pred) in seen:
                continue
            seen.add((sub, obj, pred))
            if keep():
                text += f'{sub},{obj},{pred}; '
    if TEMPORAL_STYLE in ('short', 'longshort'):
        text += 'Short: '
        for _, (sub, pred, obj) in surgery_sg_triplets[-5:]:
            if keep():
                text += f'{sub},{obj},{pred}; '
    return text[:-2] if text else ''
