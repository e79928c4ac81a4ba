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
    from release_layout import write_release
    return write_release(tmp, D, r, alpha, pooler_complete)


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
