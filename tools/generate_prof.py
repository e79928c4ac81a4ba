"""GPU-box diagnostic: where the wall time of ONE generate() call of the headline workload goes (full size, batch 8, 3 views, 150 new
tokens), phase by phase, with a device synchronisation at every phase boundary -- and the same call untouched, for the difference."""
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
NEW = 150
state = synth.synth_state(FULL, seed=0, device=dev, device_fast=True)
tok = SynthTokenizer(FULL.vocab)
eng = Engine(FULL, max_batch=8, max_seq=739 + NEW + 8, max_images=24, max_new_tokens=NEW, pad_token_id=0, eos_token_id=-1, device=dev)
eng.load_state(state)
del state
model = LlavaLlamaForCausalLM(LlavaConfig(FULL, tokenizer_padding_side='left', mv_type='learned', eos_token_id=-1), eng)
ids, images = bench.build_inputs(tok, 8, 3, FULL, dev)


def gen():
    return model.generate(ids, images=images, do_sample=False, use_cache=True, max_new_tokens=NEW, chunk=NEW)


for _ in range(2):
    gen()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    gen()
torch.cuda.synchronize()
plain = (time.perf_counter() - t0) / 3 * 1e3

acc = {}
order = []


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrap(*a, **k):
        torch.cuda.synchronize()
        s = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - s
        if label not in order:
            order.append(label)
        return out
    setattr(obj, name, wrap)


timed(model, 'prepare_inputs_labels_for_multimodal', 'vision + fuse + splice (prepare_inputs_labels_for_multimodal)')
timed(eng, 'prefill')
timed(eng, 'decode_steps')
timed(eng, 'read_state')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    gen()
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 3 * 1e3
print(f'generate(), untouched: {plain:.2f} ms per call; with a sync at every phase boundary: {total:.2f} ms')
s = 0.0
for k in order:
    print(f'  {k:70s} {acc[k] / 3 * 1e3:8.2f} ms')
    s += acc[k] / 3 * 1e3
print(f'  {"everything else in generate() (host)":70s} {total - s:8.2f} ms')

# what the bench's live HIP-event sampling of the dominant kernel costs (start / stop events carried by every 8th layer's gate/up launch)
import importlib  # noqa: E402
for name in ('prepare_inputs_labels_for_multimodal',):
    pass
eng.profile_enable(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    gen()
torch.cuda.synchronize()
withp = (time.perf_counter() - t0) / 3 * 1e3
ms, n, by = eng.profile_read()
eng.profile_enable(False)
print(f'generate() with the event sampling on (and the phase syncs): {withp:.2f} ms per call; {n} sampled launches, {ms / max(n, 1) * 1e3:.2f} us each')
