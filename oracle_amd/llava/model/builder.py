"""load_pretrained_model with the reference's signature and return value
(LLaVA/llava/model/builder.py:26-185): (tokenizer, model, image_processor, context_len).

Offline there are no released weights, so two sources are supported:
  * 'synthetic[:tiny|:full][:seed=N][:chain]' -- random-init weights from oracle_amd/synth.py;
  * a local directory holding `config.json` (LlavaConfig fields) and `*.safetensors` / `pytorch_model*.bin` weight
    shards whose tensor names are the reference checkpoint keys (already LoRA-merged);
  * the reference's released layout (builder.py:50-91): `model_path` = a LoRA checkpoint directory (`config.json`,
    `adapter_config.json`, `adapter_model.bin`, `non_lora_trainables.bin`) whose name contains "lora", `model_base` = a
    LOCAL directory with the base LLaVA weights, `mm_vision_tower` in the config = a LOCAL CLIP directory; the merge is
    done by checkpoint.py (no peft).
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
                          device_map='auto', device='cuda', max_batch=32, max_seq=4096, max_images=64,
                          max_new_tokens=512):
    """max_seq defaults to LLaMA's 4096 positions (max_position_embeddings), the length the reference can run to: the
    temporal + symbolic prompt (5000-char memory) plus 300 new tokens then always fits; KV cache = 1 MiB per slot per row at
    7B. max_batch = KV-cache slots: 32 = 16 frames decoding + the next 16 prefilled ahead (generate_stream decodes two of
    the reference's batches of 8 side by side: the same tokens, 1.5 x the frames per second), 68 GB of KV cache at 4096
    positions out of 288 GB. generate() stops at the capacity (with a
    warning) if a caller sizes max_seq smaller."""
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
        from .checkpoint import assemble_state, dims_from_config, load_dir_state
        lname = (model_name or name).lower()
        if 'lora' in lname and model_base is not None:
            # reference LoRA branch (builder.py:50-91): base + non_lora_trainables + merged adapter + CLIP + pooler
            if not os.path.isdir(model_base):
                raise FileNotFoundError(f'model_base {model_base!r} must be a local directory (hub downloads are unavailable offline)')
            state, raw = assemble_state(model_path, model_base)
            tok_dir = model_base
        else:
            with open(os.path.join(model_path, 'config.json')) as f:
                raw = json.load(f)
            state = load_dir_state(model_path)
            tok_dir = model_path
        dims = dims_from_config(raw)
        skip = set(Dims.__dataclass_fields__) | {'vocab_size', 'hidden_size', 'num_hidden_layers', 'num_attention_heads',
                                                 'intermediate_size', 'rms_norm_eps', 'rope_theta'}
        cfg = LlavaConfig(dims, **{k: v for k, v in raw.items() if k not in skip})
        if os.path.exists(os.path.join(tok_dir, 'tokenizer.model')) or os.path.exists(os.path.join(tok_dir, 'tokenizer.json')):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tok_dir, use_fast=False)
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
