"""TEST / BENCH INFRASTRUCTURE -- emits tests/golden/answer_lengths_4dor_val.json (data only): the length of the ground-truth
answer string of every 4D-OR validation frame, from the reference's data file data/relationships_validation.json rendered
the way the reference's training-data generator renders a scene graph ("<SG> sub,obj,pred; ... </SG>",
scene_graph_prediction/llava_helpers/generate_dataset_format_for_llava.py:41-64). No Llama tokenizer exists offline, so
token counts are estimated: the SHAPE of the distribution comes from the data (characters per answer), the SCALE from
SURVEY.md 8d (mean answer ~150 tokens incl. EOS, max ~250): tokens = chars * 150 / mean(chars), at least 2.
bench.py --lengths dist draws per-frame length stops from this histogram. Run in the build container:
    python -m oracle.gen_answer_lengths"""
import json
import os

REF = '/root/reference/data/relationships_validation.json'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'answer_lengths_4dor_val.json')
MEAN_TOKENS = 150.0


def render(scan):
    objs = scan['objects']
    names = scan.get('human_idx_to_name', {})

    def name(i):
        n = objs[str(i)]
        if 'human_' in n:
            n = names.get(n, 'circulator')
        return n.replace('_', ' ').lower()
    out = '<SG> '
    for sub, obj, _, rel in scan['relationships']:
        rel = rel[0].lower() + rel[1:]
        if rel == 'operating':
            rel = 'manipulating'
        out += f'{name(sub)},{name(obj)},{rel}; '
    return out.rstrip('; ') + ' </SG>'


def main():
    scans = json.load(open(REF))['scans']
    chars = [len(render(s)) for s in scans]
    triplets = [len(s['relationships']) for s in scans]
    cpt = (sum(chars) / len(chars)) / MEAN_TOKENS
    tokens = [max(2, int(round(c / cpt))) for c in chars]
    res = {'source': 'egeozsoy/ORacle data/relationships_validation.json (1332 frames, takes 4 and 8)',
           'chars_per_token': cpt, 'scale_rule': 'mean tokens = 150 (SURVEY.md 8d)', 'n_frames': len(scans), 'triplets': triplets, 'chars': chars, 'tokens_est': tokens,
           'mean_tokens': sum(tokens) / len(tokens), 'max_tokens': max(tokens), 'mean_triplets': sum(triplets) / len(triplets)}
    json.dump(res, open(OUT, 'w'))
    print({k: v for k, v in res.items() if not isinstance(v, list)})


if __name__ == '__main__':
    main()
