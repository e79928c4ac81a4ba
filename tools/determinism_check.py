"""GPU-box diagnostic: the same full-width forward in engines of different capacity, and twice in one engine, must give
bit-identical logits (no race, no capacity-dependent numerics)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle_amd.dims import Dims  # noqa: E402
from oracle_amd.engine import Engine  # noqa: E402
from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402
from oracle_amd.synth import synth_images, synth_state  # noqa: E402

d = Dims(n_layers=2, clip_layers=2)
st = synth_state(d, seed=11)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'llm_wide_bfloat16.npz'))
ids = torch.from_numpy(g['b2.ids'])
imgs = synth_images(2, 3, d, seed=555).cuda()
mask = ids != 0
outs = []
for mb, mi, ms in ((2, 6, 2176), (16, 48, 2176), (2, 6, 1000)):
    eng = Engine(d, max_batch=mb, max_seq=ms, max_images=mi, max_new_tokens=16)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(d, tokenizer_padding_side='left'), eng)
    for rep in range(2):
        o = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), images=imgs, use_cache=True).logits[:, -1].float().cpu()
        f = model.encode_images_pooled(imgs.flatten(0, 1), [3, 3]).float().cpu()
        outs.append(((mb, mi, ms, rep), o, f))
    eng.close()
ref = outs[0]
for tag, o, f in outs[1:]:
    print(tag, 'logits identical:', torch.equal(o, ref[1]), 'max diff', float((o - ref[1]).abs().max()),
          '| vision identical:', torch.equal(f, ref[2]), float((f - ref[2]).abs().max()))
