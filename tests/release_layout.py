"""Writes the reference's released checkpoint layout (builder.py:50-91,150-178) from a synthetic state: base LLaVA
safetensors + local CLIP dir + LoRA checkpoint dir (config.json, adapter_config.json, adapter_model.bin,
non_lora_trainables.bin with peft-wrapped names). Shared by the CPU ingest tests and the GPU load test."""
import json
import os

import torch

from oracle_amd.synth import CLIP_PREFIX, POOL_PREFIX, synth_state


def write_release(tmp, D, r=4, alpha=8, pooler_complete=True, full=None):
    full = synth_state(D, seed=3) if full is None else full
    base = {k: v for k, v in full.items() if not k.startswith(CLIP_PREFIX) and not k.startswith(POOL_PREFIX)}
    clip = {k[len('model.vision_tower.vision_tower.'):]: v for k, v in full.items() if k.startswith(CLIP_PREFIX)}
    base_dir, clip_dir, lora_dir = (os.path.join(tmp, n) for n in ('base', 'clip', 'oracle_lora_ckpt'))
    for d in (base_dir, clip_dir, lora_dir):
        os.makedirs(d)
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in base.items()}, os.path.join(base_dir, 'model.safetensors'))
    torch.save(clip, os.path.join(clip_dir, 'pytorch_model.bin'))
    g = torch.Generator().manual_seed(5)
    # non-LoRA trainables as the reference's SaveCallback writes them (train.py:1174-1187): peft-wrapped names
    new_proj = torch.randn(D.d_model, D.clip_d, generator=g) * 0.02
    tuned_clip_key = CLIP_PREFIX + 'encoder.layers.0.mlp.fc1.bias'
    tuned_clip = torch.randn(D.clip_ff, generator=g) * 0.02
    non_lora = {'base_model.model.model.mm_projector.0.weight': new_proj,
                'base_model.model.' + tuned_clip_key: tuned_clip}
    pool = {k: v for k, v in full.items() if k.startswith(POOL_PREFIX)}
    if not pooler_complete:
        pool.pop(POOL_PREFIX + 'encoder.layer.1.output.dense.bias')
    for k, v in pool.items():
        non_lora['base_model.model.' + k] = v.float()
    non_lora['base_model.model.model.image_pooler.bert.embeddings.position_ids'] = torch.arange(D.pool_maxpos)[None]
    torch.save(non_lora, os.path.join(lora_dir, 'non_lora_trainables.bin'))
    adapter = {}
    for mod, (n_out, n_in) in {'model.layers.0.self_attn.q_proj': (D.d_model, D.d_model),
                               'model.layers.0.mlp.down_proj': (D.d_model, D.d_ff)}.items():
        adapter[f'base_model.model.{mod}.lora_A.weight'] = torch.randn(r, n_in, generator=g) * 0.05
        adapter[f'base_model.model.{mod}.lora_B.weight'] = torch.randn(n_out, r, generator=g) * 0.05
    torch.save(adapter, os.path.join(lora_dir, 'adapter_model.bin'))
    json.dump({'r': r, 'lora_alpha': alpha, 'peft_type': 'LORA', 'target_modules': ['q_proj', 'down_proj']},
              open(os.path.join(lora_dir, 'adapter_config.json'), 'w'))
    json.dump({'model_type': 'llava', 'vocab_size': D.vocab, 'hidden_size': D.d_model, 'num_hidden_layers': D.n_layers,
               'num_attention_heads': D.n_heads, 'intermediate_size': D.d_ff, 'rms_norm_eps': D.rms_eps, 'clip_layers': D.clip_layers,
               'mm_vision_tower': clip_dir, 'mv_type': 'learned', 'mm_projector_type': 'mlp2x_gelu'},
              open(os.path.join(lora_dir, 'config.json'), 'w'))
    return full, base_dir, lora_dir, adapter, new_proj, tuned_clip_key, tuned_clip


