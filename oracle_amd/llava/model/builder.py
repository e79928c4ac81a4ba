"""load_pretrained_model with the reference's signature and return value
(LLaVA/llava/model/builder.py:26-185): (tokenizer, model, image_processor, context_len).

Offline there are no released weights, so two sources are supported:
  * 'synthetic[:tiny|:full][:seed=N][:chain]' -- random-init weights from oracle_amd/synth.py;
  * a local directory holding `config.json` (LlavaConfig fields) and `*.safetensors` / `*.bin`-free weight
    shards whose tensor names are the reference checkpoint keys (already LoRA-merged).
Hub names (e.g. 'liuhaotian/llava-v1.5-7b') cannot be fetched and raise a clear error."""
import glob
import json
import os

import torch

from ... import synth
from ...dims import FULL, TINY, Dims
from ...engine import Engine
from ...tokenizer import SynthTokenizer
from .language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM


def _parse_synthetic(spec):
    dims, seed, chain = FULL, 0, False
    for part in spec.split(':')[1:]:
        if part == 'tiny':
            dims = TINY
        elif part == 'full':
            dims = FULL
        elif part == 'chain':
            chain = True
        elif part.startswith('seed='):
            seed = int(part[5:])
        else:
            raise ValueError(f'unknown synthetic model option {part!r}')
    return dims, seed, chain


def load_pretrained_model(model_path, model_base=None, model_name=None, load_8bit=False, load_4bit=False,
                          device_map='auto', device='cuda', max_batch=8, max_seq=2400, max_images=32,
                          max_new_tokens=512):
    if load_8bit or load_4bit:
        raise NotImplementedError('the native engine runs bf16 (the reference default, builder.py:43)')
    name = os.path.basename(str(model_path).rstrip('/'))
    if name.startswith('synthetic'):
        dims, seed, chain = _parse_synthetic(name)
        fast = dims is FULL
        state = synth.synth_state(dims, seed=seed, device=device if fast else 'cpu', device_fast=fast)
        if chain:
            synth.install_chain(state, dims)
        cfg = LlavaConfig(dims)
        tokenizer = SynthTokenizer(dims.vocab)
    elif os.path.isdir(model_path):
        with open(os.path.join(model_path, 'config.json')) as f:
            raw = json.load(f)
        dims = Dims(**{k: raw[k] for k in Dims.__dataclass_fields__ if k in raw})
        cfg = LlavaConfig(dims, **{k: v for k, v in raw.items() if k not in Dims.__dataclass_fields__})
        from safetensors.torch import load_file
        state = {}
        for shard in sorted(glob.glob(os.path.join(model_path, '*.safetensors'))):
            state.update(load_file(shard))
        if os.path.exists(os.path.join(model_path, 'tokenizer.model')) or os.path.exists(os.path.join(model_path, 'tokenizer.json')):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(model_path, use_fast=False)
        else:
            tokenizer = SynthTokenizer(dims.vocab)
    else:
        raise FileNotFoundError(f'{model_path!r} is neither "synthetic..." nor a local checkpoint directory; hub downloads '
                                '(model_base, openai/clip-vit-large-patch14-336) are unavailable offline')
    engine = Engine(dims, max_batch=max_batch, max_seq=max_seq, max_images=max_images, max_new_tokens=max_new_tokens,
                    pad_token_id=cfg.pad_token_id, eos_token_id=cfg.eos_token_id, device=None if device == 'cuda' else device)
    engine.load_state(state)
    del state
    model = LlavaLlamaForCausalLM(cfg, engine)
    vt = model.get_vision_tower()
    if not vt.is_loaded:
        vt.load_model()
    context_len = getattr(cfg, 'max_sequence_length', 2048)
    return tokenizer, model, vt.image_processor, context_len
