"""GPU-box diagnostic: the vision stage alone (CLIP x 24 images, pooler, projector) a few times, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle_amd import synth  # noqa: E402
from oracle_amd.dims import Dims  # noqa: E402
from oracle_amd.engine import Engine  # noqa: E402
from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402

d = Dims(n_layers=1)                      # full CLIP / pooler / projector, one LLaMA layer (not run)
dev = torch.device('cuda', 0)
st = synth.synth_state(d, seed=0, device=dev, device_fast=True)
eng = Engine(d, max_batch=8, max_seq=1024, max_images=24, max_new_tokens=16, device=dev)
eng.load_state(st)
model = LlavaLlamaForCausalLM(LlavaConfig(d, tokenizer_padding_side='left'), eng)
imgs = synth.synth_images(8, 3, d).to(dev).flatten(0, 1)
import time
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f = model.encode_images_pooled(imgs, [3] * 8)
    torch.cuda.synchronize()
    print(f'vision stage {1e3 * (time.perf_counter() - t0):.2f} ms')
eng.close()
