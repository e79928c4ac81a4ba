"""GPU-box diagnostic: where the time of the slot-refill stream goes at full size (single-row prefill, per-batch vision +
splice, a chunk of decode steps + state read)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle_amd import synth  # noqa: E402
from oracle_amd.dims import FULL  # noqa: E402
from oracle_amd.engine import Engine  # noqa: E402
from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402
from oracle_amd.tokenizer import SynthTokenizer  # noqa: E402

dev = torch.device('cuda', 0)
state = synth.synth_state(FULL, seed=0, device=dev, device_fast=True)
tok = SynthTokenizer(FULL.vocab)
eng = Engine(FULL, max_batch=16, max_seq=739 + 308, max_images=24, max_new_tokens=300, pad_token_id=0, eos_token_id=-1, device=dev)
eng.load_state(state)
del state
model = LlavaLlamaForCausalLM(LlavaConfig(FULL, tokenizer_padding_side='left', mv_type='learned', eos_token_id=-1), eng)
ids, images = bench.build_inputs(tok, 8, 3, FULL, dev)


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


am = torch.ones_like(ids, dtype=torch.bool)
inp = model.prepare_inputs_for_generation(ids, attention_mask=am, images=images)
prep = lambda: model.prepare_inputs_labels_for_multimodal(inp['input_ids'], inp['position_ids'], inp['attention_mask'], None, None, images, None)
print(f'vision + pooler + projector + splice, 8 frames x 3 views: {t(prep):.2f} ms')
_, pos, mask, _, emb, _ = prep()
emb = emb.contiguous()
print(f'batch prefill 8 x {emb.shape[1]}: {t(lambda: eng.prefill(emb, pos.to(torch.int32).contiguous(), [0] * 8)):.2f} ms')
eng.stream_begin(8)
p32 = pos.to(torch.int32).contiguous()
print(f'single-row prefill T={emb.shape[1]} into a slot: {t(lambda: eng.prefill_slots([11], emb[3:4], p32[3:4], [0], [300])):.2f} ms')
print(f'4-row prefill into slots: {t(lambda: eng.prefill_slots([8, 9, 10, 11], emb[:4], p32[:4], [0] * 4, [300] * 4)):.2f} ms')
print(f'8-row prefill into slots 8..15: {t(lambda: eng.prefill_slots(list(range(8, 16)), emb, p32, [0] * 8, [300] * 8)):.2f} ms')
eng.prefill_slots(list(range(8)), emb, p32, [0] * 8, [300] * 8)
for r in range(8):
    eng.stream_attach(r, r)
print(f'8 decode steps + read_slots: {t(lambda: (eng.decode_steps(8), eng.read_slots())):.2f} ms  (8 x {t(lambda: eng.decode_steps(8)) / 8:.3f} ms/step without the read)')
one = images[:1]
prep1 = lambda: model.prepare_inputs_labels_for_multimodal(inp['input_ids'][:1], inp['position_ids'][:1], inp['attention_mask'][:1], None, None, one, None)
print(f'vision + pooler + projector + splice, 1 frame x 3 views: {t(prep1):.2f} ms')
eng.close()
