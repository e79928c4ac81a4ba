"""TEST INFRASTRUCTURE ONLY -- emits tests/golden/*.npz|json by running the REAL reference glue.

Run in the build container (needs /root/reference):   python -m oracle.gen_golden [--only NAME]
Each fixture stores outputs of the reference (oracle/ref_harness.py) on seeded synthetic weights/inputs
(oracle_amd/synth.py regenerates the inputs bit-identically anywhere), never reference source text.
Large tensors are stored as a strided subset plus checksums.
"""
import argparse
import json
import os
import random

import numpy as np
import torch

from oracle import cpu_ref as cr
from oracle import ref_harness as rh
from oracle_amd.dims import TINY, Dims
from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN, install_chain, install_select_chains, synth_images, synth_state

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def _f32(x):
    return x.detach().float().numpy()


def sub2(x, s0, s1):
    """Strided subset over the last two dims + checksums of the full tensor."""
    x = x.detach().float()
    return {'sub': x[..., ::s0, ::s1].numpy(), 'sum': np.float64(x.double().sum().item()),
            'abssum': np.float64(x.double().abs().sum().item()), 'absmax': np.float64(x.abs().max().item()),
            'shape': np.array(x.shape)}


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f'{k}.{kk}'] = vv
        else:
            flat[k] = v
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **flat)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


# test prompts (synthetic ids; -200 = <image>, 18610 = visual descriptor placeholder, 0 = pad)
def ids_batch3():
    rows = [[1, 5, 6, -200, 7, 8, 9, 10, 11], [1, 21, -200, 22, 23], [1, 31, 32, 33, 34, 35, -200, 36, 37, 38, 39, 40]]
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def ids_desc(n_desc=7):
    row = [1, 50, -200, 51, 52]
    for j in range(n_desc):
        row += [18610, 60 + j, 70 + j]
    return torch.tensor([row + [90, 91]])


def gen_vision(dtype_name):
    """G1/G2: CLIP penultimate features, pooled and projected features, V in {1,3,4} (TINY depth) and the
    full 24-layer tower for 3 views."""
    dtype = getattr(torch, dtype_name)
    st = synth_state(TINY, seed=0, dtype=dtype)
    model, cfg = rh.build_reference_model(TINY, st, dtype=dtype)
    out = {}
    with torch.inference_mode():
        for V in (1, 3, 4):
            imgs = synth_images(2, V, TINY).to(dtype)                       # (2,V,3,H,W) or (2,3,H,W)
            flat = imgs if V == 1 else imgs.flatten(0, 1)
            vt = model.get_vision_tower()(flat)
            out[f'vt_V{V}'] = sub2(vt, 37, 41)
            if V == 1:
                feats = model.encode_images_pooled(imgs, None)
            else:
                feats = model.encode_images_pooled(flat, [V, V])
            out[f'learned_V{V}'] = sub2(feats, 37, 29)
        # ragged views per row (training-style list input): rows with 2 and 4 views
        a = synth_images(1, 2, TINY, seed=77).to(dtype)[0]
        b = synth_images(1, 4, TINY, seed=78).to(dtype)[0]
        feats = model.encode_images_pooled(torch.cat([a, b]), [2, 4])
        out['learned_ragged24'] = sub2(feats, 37, 29)
        # other mv types through the reference's own dispatch (llava_arch.py:203-239)
        imgs = synth_images(2, 3, TINY).to(dtype)
        ids = torch.tensor([[1, 5, -200, 6], [1, 7, -200, 8]])
        for mv in ('concat', 'concat_pooled', 'max', 'avg'):
            model.config.mv_type = mv
            r = model.prepare_inputs_labels_for_multimodal(ids, None, torch.ones_like(ids, dtype=torch.bool), None, None, imgs, None)
            emb = r[4]
            out[f'mv_{mv}'] = sub2(emb[:, 2:-1], 31, 29)     # the spliced image rows
            out[f'mv_{mv}_T'] = np.array(emb.shape[1])
        model.config.mv_type = 'learned'
        g = model.get_vision_tower().global_forward(synth_images(2, 1, TINY).to(dtype))
        out['global_pooler_output'] = _f32(g)
    save(f'vision_tiny_{dtype_name}', **out)


def gen_vision_full():
    """Full-depth CLIP (24 layers) for 3 views, bf16, through the reference tower."""
    d = Dims(n_layers=1, d_model=512, n_heads=4, d_ff=1408)     # LLaMA part irrelevant here
    st = synth_state(d, seed=3, dtype=torch.bfloat16)
    model, cfg = rh.build_reference_model(d, st, dtype=torch.bfloat16)
    imgs = synth_images(1, 3, d, seed=4321).to(torch.bfloat16)[0]
    with torch.inference_mode():
        vt = model.get_vision_tower()(imgs)
        feats = model.encode_images_pooled(imgs, [3])
    # the same bf16-valued weights through the reference in fp32: measures the reference's own bf16 noise at full depth
    model32, _ = rh.build_reference_model(d, st, dtype=torch.float32)
    with torch.inference_mode():
        vt32 = model32.get_vision_tower()(imgs.float())
        feats32 = model32.encode_images_pooled(imgs.float(), [3])
    out = dict(vt=sub2(vt, 19, 23), learned=sub2(feats, 19, 23))
    out['vt']['sub_fp32'] = vt32[..., ::19, ::23].numpy()
    out['learned']['sub_fp32'] = feats32[..., ::19, ::23].numpy()
    save('vision_full_bfloat16', **out)


def gen_splice():
    """G3: exact index work of prepare_inputs_labels_for_multimodal (fp32 so rows are exact copies)."""
    st = synth_state(TINY, seed=0, dtype=torch.float32)
    model, cfg = rh.build_reference_model(TINY, st, dtype=torch.float32)
    out = {}
    torch.manual_seed(5)
    with torch.inference_mode():
        imgs = synth_images(3, 3, TINY).float()
        feats = model.encode_images_pooled(imgs.flatten(0, 1), [3, 3, 3])
        cases = {
            'single': (ids_batch3()[2:3, :], None, 1),
            'batch3': (ids_batch3(), None, 3),
            'desc7': (ids_desc(7), [torch.randn(512) for _ in range(7)], 1),
            'desc_multirow': (ids_desc(2), [torch.randn(3, 512), torch.randn(512)], 1),
            'desc_missing': (ids_desc(7), None, 1),          # the llava_arch.py:305 quirk
        }
        for name, (ids, desc, nb) in cases.items():
            mask0 = ids != 0
            _, pos, am, _, emb, _ = model.prepare_inputs_labels_for_multimodal(ids, None, mask0, None, None, imgs[:nb], desc)
            # identify each output row by matching it against its possible sources
            out[f'{name}.ids'] = ids.numpy()
            out[f'{name}.mask'] = am.numpy().astype(np.uint8)
            out[f'{name}.emb_rowsum'] = emb.double().sum(-1).numpy()
            out[f'{name}.emb_sub'] = emb[:, :, ::61].numpy()
            # position_ids is None when not passed in (llava_arch.py:371-372); re-run passing positions
            p_in = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1)
            r = model.prepare_inputs_labels_for_multimodal(ids, p_in, mask0, None, None, imgs[:nb], desc)
            out[f'{name}.pos'] = r[1].numpy()
            if desc is not None:
                out[f'{name}.desc'] = torch.cat([x.reshape(-1, 512) for x in desc]).numpy()
        # right padding + truncation variant
        model.config.tokenizer_padding_side = 'right'
        model.config.tokenizer_model_max_length = 580
        ids = ids_batch3()
        p_in = torch.arange(ids.shape[1])[None].expand(ids.shape[0], -1)
        r = model.prepare_inputs_labels_for_multimodal(ids, p_in, ids != 0, None, None, imgs, None)
        out['right_trunc.ids'] = ids.numpy()
        out['right_trunc.mask'] = r[2].numpy().astype(np.uint8)
        out['right_trunc.pos'] = r[1].numpy()
        out['right_trunc.emb_rowsum'] = r[4].double().sum(-1).numpy()
    save('splice_tiny_float32', **out)


def gen_llm(dtype_name):
    """G4/G5: prefill last-token logits (top-16) and greedy tokens incl. EOS / ragged finishing."""
    dtype = getattr(torch, dtype_name)
    out = {}
    for tag, peaked in (('rand', False), ('peak', True)):
        st = synth_state(TINY, seed=0, dtype=dtype)
        if peaked:
            install_chain(st, TINY)
        model, cfg = rh.build_reference_model(TINY, st, dtype=dtype)
        imgs = synth_images(3, 3, TINY).to(dtype)
        ids = chain_prompts() if peaked else ids_batch3()
        with torch.inference_mode():
            mask0 = ids != 0
            # position ids as GenerationMixin.prepare_inputs_for_generation derives them from the mask
            p_in = mask0.long().cumsum(-1) - 1
            p_in.masked_fill_(~mask0, 1)
            o = model(input_ids=ids, attention_mask=mask0, position_ids=p_in, images=imgs, use_cache=True)
            logits = o.logits[:, -1].float()
            top = torch.topk(logits, 16, dim=-1)
            out[f'{tag}.ids'] = ids.numpy()
            out[f'{tag}.top16_val'] = top.values.numpy()
            out[f'{tag}.top16_idx'] = top.indices.numpy()
            out[f'{tag}.logits_sub'] = logits[:, ::97].numpy()
            out[f'{tag}.logits_absmax'] = logits.abs().max(-1).values.numpy()
        gen = rh.reference_generate(model, cfg, ids, imgs, 40)
        out[f'{tag}.generated'] = gen.numpy()
        # check the restatement reproduces the tokens in this very run (same threads / same machine)
        mine = cr.greedy_generate(st, TINY, ids, imgs, max_new_tokens=40)
        assert torch.equal(mine, gen), (tag, mine[:, ids.shape[1]:], gen[:, ids.shape[1]:])
    save(f'llm_tiny_{dtype_name}', **out)


WIDE2 = Dims(n_layers=2, clip_layers=2)      # full-width LLaMA (4096 / 32 heads / 11008) x 2 layers, 2 CLIP layers


def wide_prompts():
    """Two left-padded rows at the plain-prompt scale (~190 text tokens -> T ~ 765) and one long row at the
    temporal + symbolic-knowledge scale (BASELINE configs[3]: T ~ 2k)."""
    g = torch.Generator().manual_seed(99)

    def row(n_before, n_after):
        a = torch.randint(3, 31999, (n_before,), generator=g)
        b = torch.randint(3, 31999, (n_after,), generator=g)
        a[a == 18610] = 18611
        b[b == 18610] = 18611
        return [1] + a.tolist() + [-200] + b.tolist()
    r0, r1 = row(30, 160), row(30, 95)
    L = max(len(r0), len(r1))
    b2 = torch.tensor([[0] * (L - len(r)) + r for r in (r0, r1)])
    long = torch.tensor([row(30, 1470)])
    return b2, long


def gen_llm_wide():
    """Full-width LLaMA layers through the real reference in bf16: prefill logits and greedy tokens for a ragged
    batch of 2 (configs[2] prompt scale) and for one ~2k-token row (configs[3] scale)."""
    dtype = torch.bfloat16
    st = synth_state(WIDE2, seed=11, dtype=dtype)
    b2, long = wide_prompts()
    out = {}
    # the same bf16-valued weights through the reference in fp32: "exact math" logits, which measure the reference's own
    # bf16 rounding noise at this width (the GPU test states its tolerance against that floor)
    model32, _ = rh.build_reference_model(WIDE2, st, dtype=torch.float32)
    for tag, ids in (('b2', b2), ('long', long)):
        imgs = synth_images(ids.shape[0], 3, WIDE2, seed=555).float()
        with torch.inference_mode():
            mask0 = ids != 0
            p_in = mask0.long().cumsum(-1) - 1
            p_in.masked_fill_(~mask0, 1)
            o = model32(input_ids=ids, attention_mask=mask0, position_ids=p_in, images=imgs, use_cache=True)
        out[f'{tag}.logits_sub_fp32'] = o.logits[:, -1].float()[:, ::97].numpy()
    del model32
    model, cfg = rh.build_reference_model(WIDE2, st, dtype=dtype)
    for tag, ids, views, n_new in (('b2', b2, 3, 8), ('long', long, 3, 4)):
        imgs = synth_images(ids.shape[0], views, WIDE2, seed=555).to(dtype)
        with torch.inference_mode():
            mask0 = ids != 0
            p_in = mask0.long().cumsum(-1) - 1
            p_in.masked_fill_(~mask0, 1)
            o = model(input_ids=ids, attention_mask=mask0, position_ids=p_in, images=imgs, use_cache=True)
            logits = o.logits[:, -1].float()
            top = torch.topk(logits, 16, dim=-1)
        out[f'{tag}.ids'] = ids.numpy()
        out[f'{tag}.top16_val'] = top.values.numpy()
        out[f'{tag}.top16_idx'] = top.indices.numpy()
        out[f'{tag}.logits_sub'] = logits[:, ::97].numpy()
        out[f'{tag}.logits_absmax'] = logits.abs().max(-1).values.numpy()
        gen = rh.reference_generate(model, cfg, ids, imgs, n_new)
        out[f'{tag}.generated'] = gen.numpy()
        mine = cr.greedy_generate(st, WIDE2, ids, imgs, max_new_tokens=n_new)
        print(tag, 'reference tokens', gen[:, ids.shape[1]:].tolist(), 'restatement equal:', torch.equal(mine, gen))
        assert torch.equal(mine, gen)
    # peaked ("chain") weights at full width: free-running generate() must be token-identical, incl. ragged EOS / pad fill
    del model
    install_chain(st, WIDE2)
    model, cfg = rh.build_reference_model(WIDE2, st, dtype=dtype)
    ids = chain_prompts()
    imgs = synth_images(3, 3, WIDE2, seed=556).to(dtype)
    gen = rh.reference_generate(model, cfg, ids, imgs, 36)
    mine = cr.greedy_generate(st, WIDE2, ids, imgs, max_new_tokens=36)
    print('peak reference tokens', gen[:, ids.shape[1]:].tolist(), 'restatement equal:', torch.equal(mine, gen))
    assert torch.equal(mine, gen)
    out['peak.ids'] = ids.numpy()
    out['peak.generated'] = gen.numpy()
    save('llm_wide_bfloat16', **out)


def wide_batch_prompts(n_rows, seed):
    """n_rows ragged left-padded rows at the plain-prompt scale (T between ~620 and ~740 after the 576 image rows)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for i in range(n_rows):
        nb = int(torch.randint(8, 31, (1,), generator=g))
        na = int(torch.randint(12, 131, (1,), generator=g)) if i else 130          # row 0 is the longest: no padding
        a = torch.randint(3, 31999, (nb,), generator=g)
        b = torch.randint(3, 31999, (na,), generator=g)
        a[a == 18610] = 18611
        b[b == 18610] = 18611
        rows.append([1] + a.tolist() + [-200] + b.tolist())
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def gen_llm_wide_batch():
    """The headline's decode code path under the reference: full-width LLaMA layers, 8 and 16 ragged left-padded rows
    (un-split decode attention, B*H >= 192; persistent GEMVs with 8 / 16 live rows). Stored: prefill logits and the
    logits of 8 greedy decode steps from the REAL reference's generate() in bf16 (output_scores), and the same steps
    teacher-forced along those tokens in fp32 (the restatement, bit-identical to the reference in fp32) as the
    'exact math' floor."""
    dtype = torch.bfloat16
    st = synth_state(WIDE2, seed=11, dtype=dtype)
    n_new = 9
    out = {}
    model, cfg = rh.build_reference_model(WIDE2, st, dtype=dtype)
    toks = {}
    for tag, nrows, seed in (('b8', 8, 201), ('b16', 16, 202)):
        ids = wide_batch_prompts(nrows, seed)
        imgs = synth_images(nrows, 3, WIDE2, seed=600 + nrows).to(dtype)
        with torch.inference_mode():
            r = model.generate(ids, images=imgs, do_sample=False, use_cache=True, max_new_tokens=n_new,
                               past_key_values=rh.legacy_cache(cfg), output_scores=True, return_dict_in_generate=True)
        gen = r.sequences
        scores = torch.stack([s.float() for s in r.scores], dim=1)             # (B, n_new, V): raw logits of every step
        assert scores.shape[1] == n_new and torch.equal(scores.argmax(-1), gen[:, ids.shape[1]:])
        top = torch.topk(scores[:, 0], 16, dim=-1)
        out[f'{tag}.ids'] = ids.numpy()
        out[f'{tag}.generated'] = gen.numpy()
        out[f'{tag}.top16_val'] = top.values.numpy()
        out[f'{tag}.top16_idx'] = top.indices.numpy()
        out[f'{tag}.logits_sub'] = scores[:, :, ::97].numpy()
        out[f'{tag}.logits_absmax'] = scores.abs().max(-1).values.numpy()
        top2 = torch.topk(scores, 2, dim=-1).values
        out[f'{tag}.margin'] = (top2[..., 0] - top2[..., 1]).numpy()
        toks[tag] = (ids, imgs, gen)
        mine = cr.greedy_generate(st, WIDE2, ids, imgs, max_new_tokens=n_new)
        print(tag, 'T', ids.shape[1] - 1 + 576, 'restatement equal:', torch.equal(mine, gen))
        assert torch.equal(mine, gen)
    del model
    st32 = {k: v.float() for k, v in st.items()}
    for tag, (ids, imgs, gen) in toks.items():
        new = gen[:, ids.shape[1]:]
        _, trace = cr.greedy_generate(st32, WIDE2, ids, imgs.float(), max_new_tokens=n_new, forced_tokens=new,
                                      return_trace=True, trace_logits_stride=97)
        out[f'{tag}.logits_sub_fp32'] = torch.stack([t['logits_sub'] for t in trace], dim=1).numpy()
    save('llm_wide_batch_bfloat16', **out)


def select_prompts(n, seed, tail=0):
    """n left-padded prompts `BOS text(3..14 tokens) <image> [tail text tokens]`: with tail == 0 the image placeholder is
    the last prompt token (the first answer token is selected by the last image-token row, install_select_chains)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for _ in range(n):
        k = int(torch.randint(3, 15, (1,), generator=g))
        rows.append([1] + torch.randint(3, 19000, (k,), generator=g).tolist() + [-200]
                    + torch.randint(3, 19000, (tail,), generator=g).tolist())
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def select_center(st):
    """Mean lm_head input of the last prefill position over 8 calibration frames (fp32 math on the bf16-valued weights)."""
    st32 = {k: v.float() for k, v in st.items()}
    ids = select_prompts(8, 300)
    imgs = synth_images(8, 3, TINY, seed=6000).float()
    with torch.inference_mode():
        feats = cr.encode_image_features(st32, TINY, imgs, 'learned')
        emb, mask, pos, _ = cr.splice(st32, ids, ids != 0, feats)
        h, _ = cr.llama_forward(st32, TINY, emb, mask, pos, return_hidden=True)
    return h.mean(0)


SELECT_FRAMES, SELECT_BATCH = 128, 16


def gen_select():
    """Token identity as a RATE over a population of frames whose answer depends on the image:
    128 frames (8 ragged batches of 16) through the REAL reference's generate() on install_select_chains weights; stored:
    prompts, generated tokens, the first step's top-2 logits (the margin the selection was made with)."""
    dtype = torch.bfloat16
    base = synth_state(TINY, seed=0, dtype=dtype)
    center = select_center(base)
    st = install_select_chains(base, TINY, center=center)
    model, cfg = rh.build_reference_model(TINY, st, dtype=dtype)
    out = {'center': center.numpy()}
    for tag, n, tail, seed in (('last', SELECT_FRAMES, 0, 310),):
        ids_all, gen_all, top2_all, absmax_all = [], [], [], []
        for b0 in range(0, n, SELECT_BATCH):
            ids = select_prompts(SELECT_BATCH, seed * 100 + b0, tail)
            imgs = synth_images(SELECT_BATCH, 3, TINY, seed=7000 + (0 if tail == 0 else 5000) + b0).to(dtype)
            with torch.inference_mode():
                r = model.generate(ids, images=imgs, do_sample=False, use_cache=True, max_new_tokens=24,
                                   past_key_values=rh.legacy_cache(cfg), output_scores=True, return_dict_in_generate=True)
            new = r.sequences[:, ids.shape[1]:]
            new = torch.cat([new, new.new_zeros(new.shape[0], 24 - new.shape[1])], dim=1)       # pad to a fixed width
            s0 = r.scores[0].float()
            ids_all.append(torch.cat([ids.new_zeros(ids.shape[0], 16 + tail - ids.shape[1]), ids], dim=1))
            gen_all.append(new)
            top2_all.append(torch.topk(s0, 2, dim=-1).values)
            absmax_all.append(s0.abs().max(-1).values)
            if b0 == 0:
                mine = cr.greedy_generate(st, TINY, ids, imgs, max_new_tokens=24)
                assert torch.equal(mine[:, ids.shape[1]:], r.sequences[:, ids.shape[1]:]), tag
            print(tag, b0, 'first tokens', new[:, 0].tolist())
        out[f'{tag}.ids'] = torch.cat(ids_all).numpy()
        out[f'{tag}.generated'] = torch.cat(gen_all).numpy()
        out[f'{tag}.top2'] = torch.cat(top2_all).numpy()
        out[f'{tag}.absmax'] = torch.cat(absmax_all).numpy()
    save('select_tiny_bfloat16', **out)


def chain_prompts():
    ends = [CHAIN_BASE + CHAIN_LEN - 9, CHAIN_BASE + CHAIN_LEN - 30, CHAIN_BASE + CHAIN_LEN - 17]
    rows = [[1, 5, 6, -200, 7, 8, 9, 10, ends[0]], [1, 21, -200, 22, ends[1]], [1, 31, 32, 33, 34, 35, -200, 36, 37, 38, 39, ends[2]]]
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def gen_text():
    """G6/G7: pure-python scene-graph helpers of the reference (scene_graph_converters.py)."""
    ns = rh.load_reference()
    cv = ns.converters
    kat = {}
    raw = [
        'head surgeon,patient,cutting; assistant surgeon,head surgeon,assisting; nurse,instrument table,closeTo;',
        '<SG> head surgeon, patient, drilling ; anaesthetist,anesthesia equipment,manipulating;</SG> trailing words',
        'circulator,operating table,touching;broken,triplet;;  ;a,b,c,d; patient,operating table,lyingOn.</s>',
        '</SG> x,y,z; <SG> q,r,s;',
        '<s> nurse , secondary table , preparing ;</s>',
        '',
    ]
    kat['parse'] = [{'in': s, 'out': [list(t) for t in cv.parse_llava_sg(s)]} for s in raw]
    # temporal memory: a scripted 12-frame history under random.seed(42) (main.py:52 seeds `random` via pl.seed_everything)
    ents = ['head surgeon', 'assistant surgeon', 'circulator', 'nurse', 'anaesthetist', 'patient', 'instrument table', 'operating table']
    preds = ['assisting', 'cementing', 'cleaning', 'closeTo', 'cutting', 'drilling', 'holding', 'lyingOn', 'touching']
    rng = random.Random(7)
    hist = []
    cur = set()
    for t in range(12):
        for _ in range(rng.randint(0, 3)):
            s, o = rng.sample(ents, 2)
            cur.add((s, rng.choice(preds), o))
        if cur and rng.random() < 0.5:
            cur.discard(rng.choice(sorted(cur)))
        hist.append({'timepoint_idx': t * 3, 'scene_graph': sorted(cur)})
    steps = []
    random.seed(42)
    for n in range(1, 13):
        sg = cv.llava_sg_to_surgery_sg(hist[:n], entity_of_interest=None, IRRELEVANT_PREDS=['closeto', 'closeTo'])
        tp = hist[n - 1]['timepoint_idx'] + 3
        sg_f = [e for e in sg if e[0] < tp]
        steps.append({'n': n, 'timepoint': tp,
                      'surgery_sg': [[a, list(b)] for a, b in sg],
                      'longshort': cv.surgery_sg_to_memory_str(sg_f, current_timepoint=tp, TEMPORAL_STYLE='longshort'),
                      'short': cv.surgery_sg_to_memory_str(sg_f, current_timepoint=tp, TEMPORAL_STYLE='short'),
                      'long': cv.surgery_sg_to_memory_str(sg_f, current_timepoint=tp, TEMPORAL_STYLE='long')})
    kat['history'] = [{'timepoint_idx': h['timepoint_idx'], 'scene_graph': [list(x) for x in h['scene_graph']]} for h in hist]
    kat['memory'] = steps
    # conversation template (conversation.py conv_vicuna_v1) with a marker message
    conv = ns.conversation.conv_templates['vicuna_v1'].copy()
    conv.append_message(conv.roles[0], '<image>\nHELLO')
    conv.append_message(conv.roles[1], None)
    kat['vicuna_v1_prompt'] = conv.get_prompt()
    kat['default_conversation_is_vicuna_v1'] = ns.conversation.default_conversation.name if hasattr(ns.conversation.default_conversation, 'name') else None
    kat['sep2'] = conv.sep2
    # the user-turn prompt literals of OracleWrapper.forward (…_oracle.py:72-81): only their digests are stored
    import hashlib
    import re
    src = open(os.path.join(rh.REF_ROOT, 'scene_graph_prediction/scene_graph_helpers/model/scene_graph_prediction_model_oracle.py')).read()
    lits = re.findall(r"inp = f?'(Entities.*?)'\n", src)
    names = ['symbolic', 'plain', 'visual', 'visual_mv']
    kat['prompt_digest'] = {n: {'sha256': hashlib.sha256(l.replace('{VIS_DESCRIPTOR_TOKEN}', ns.constants.VIS_DESCRIPTOR_TOKEN).encode()).hexdigest(),
                                'chars': len(l.replace('{VIS_DESCRIPTOR_TOKEN}', ns.constants.VIS_DESCRIPTOR_TOKEN))}
                            for n, l in zip(names, lits)}
    path = os.path.join(OUT, 'text_kat.json')
    with open(path, 'w') as f:
        json.dump(kat, f, indent=1, ensure_ascii=False)
    print('wrote', path)


ALL = {
    'vision_f32': lambda: gen_vision('float32'),
    'vision_bf16': lambda: gen_vision('bfloat16'),
    'vision_full': gen_vision_full,
    'splice': gen_splice,
    'llm_f32': lambda: gen_llm('float32'),
    'llm_bf16': lambda: gen_llm('bfloat16'),
    'llm_wide': gen_llm_wide,
    'llm_batchwide': gen_llm_wide_batch,
    'select': gen_select,
    'text': gen_text,
}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    for k, fn in ALL.items():
        if a.only and a.only not in k:
            continue
        print('==', k)
        fn()
