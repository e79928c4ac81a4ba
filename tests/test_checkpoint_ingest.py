"""CPU: weight ingest of the reference's released checkpoint layout (reference
LLaVA/llava/model/builder.py:50-91,150-178): base + non_lora_trainables prefix handling + LoRA merge + CLIP remap +
strict pooler load. peft is absent offline: the merge is checked against a direct fp32 evaluation of the published
formula W + (alpha / r) * B @ A (parity with peft 0.4.0 itself is unpinned)."""
import json
import os

import pytest
import torch

from oracle_amd.dims import Dims
from oracle_amd.llava.model import checkpoint as ck
from oracle_amd.synth import CLIP_PREFIX, POOL_PREFIX, synth_state, weight_spec

D = Dims(vocab=256, d_model=128, n_layers=1, n_heads=1, d_ff=192, clip_layers=1)


def _write_release(tmp, r=4, alpha=8, pooler_complete=True):
    full = synth_state(D, seed=3)
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


def test_prefix_stripping_follows_the_reference():
    sd = {'base_model.model.model.mm_projector.0.weight': 1, 'base_model.model.lm_head.weight': 2}
    out = ck.strip_non_lora_prefixes(sd)
    assert set(out) == {'model.mm_projector.0.weight', 'lm_head.weight'}
    # keys that never had the double prefix keep their single "model."
    assert set(ck.strip_non_lora_prefixes({'model.mm_projector.0.weight': 1})) == {'model.mm_projector.0.weight'}


def test_released_layout_is_assembled_and_merged(tmp_path):
    full, base_dir, lora_dir, adapter, new_proj, tuned_clip_key, tuned_clip = _write_release(str(tmp_path))
    state, cfg = ck.assemble_state(lora_dir, base_dir)
    assert ck.dims_from_config(cfg) == D
    assert set(state) == set(weight_spec(D))                       # exactly the keys the engine ingests
    assert all(v.dtype == torch.bfloat16 for v in state.values())
    # LoRA: bf16( fp32(W) + alpha/r * B @ A ), untouched modules bit-identical to the base
    for mod in ('model.layers.0.self_attn.q_proj', 'model.layers.0.mlp.down_proj'):
        a = adapter[f'base_model.model.{mod}.lora_A.weight'].double()
        b = adapter[f'base_model.model.{mod}.lora_B.weight'].double()
        want = (full[mod + '.weight'].double() + 2.0 * (b @ a)).to(torch.bfloat16)
        got = state[mod + '.weight']
        assert (got != want).float().mean() < 1e-3                  # fp32 vs fp64 evaluation: ties only
        assert not torch.equal(got, full[mod + '.weight'])
    assert torch.equal(state['model.layers.0.self_attn.k_proj.weight'], full['model.layers.0.self_attn.k_proj.weight'])
    # non-LoRA trainables overwrite the base; the fine-tuned CLIP entry overwrites the stock tower; the rest of CLIP is stock
    assert torch.equal(state['model.mm_projector.0.weight'], new_proj.to(torch.bfloat16))
    assert torch.equal(state[tuned_clip_key], tuned_clip.to(torch.bfloat16))
    other = CLIP_PREFIX + 'encoder.layers.0.mlp.fc2.bias'
    assert torch.equal(state[other], full[other])
    assert torch.equal(state[POOL_PREFIX + 'encoder.layer.1.output.dense.weight'], full[POOL_PREFIX + 'encoder.layer.1.output.dense.weight'])


def test_pooler_load_is_strict_and_clip_must_be_local(tmp_path):
    _, base_dir, lora_dir, *_ = _write_release(str(tmp_path), pooler_complete=False)
    with pytest.raises(KeyError, match='strict'):
        ck.assemble_state(lora_dir, base_dir)
    with pytest.raises(FileNotFoundError, match='local'):
        ck.assemble_state(lora_dir, base_dir, clip_path='openai/clip-vit-large-patch14-336')
