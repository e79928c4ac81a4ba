"""Weight ingest for released-format ORacle checkpoints from LOCAL paths.

Restates what `load_pretrained_model` does to weights in the reference's LoRA branch
(LLaVA/llava/model/builder.py:50-91,150-178) without transformers / peft (peft is absent offline):

  1. base LLaVA weights from `model_base` (a local HF-format directory: `*.safetensors` or `pytorch_model*.bin`);
  2. `non_lora_trainables.bin` from `model_path`: strip the `base_model.` prefix, then one leading `model.` when keys
     look like `model.model.*` (builder.py:81-83); its entries overwrite the base (`load_state_dict(strict=False)`, :84);
  3. LoRA merge (builder.py:86-90, peft 0.4.0 `merge_and_unload`): W <- bf16( fp32(W) + (alpha/r) * B @ A ) for every
     `<module>.lora_A.weight` / `<module>.lora_B.weight` pair in `adapter_model.bin`;
  4. CLIP tower from the local directory named by `mm_vision_tower` (clip_encoder.py:22-27), then overwritten by the
     `model.vision_tower.vision_tower.*` entries of non_lora_trainables (builder.py:156-163; the last 12 layers were
     fine-tuned, LLaVA/slurm_config_multiview.conf:58);
  5. `model.image_pooler.*` entries loaded STRICTLY (builder.py:166-178): every pooler tensor must be present.

The result is one flat {reference checkpoint key: bf16 tensor} dict, which is exactly what the engine's
`orc_engine_load_weight` takes. Files are read with loaders that execute nothing from the file
(safetensors, `torch.load(weights_only=True)`).
Parity note: peft 0.4.0 is not installed and the reference holds no merged-weight fixture, so the merge arithmetic is
pinned only by its published formula (tests/test_checkpoint_ingest.py checks it against a direct fp32 evaluation).
"""
import glob
import json
import os

import torch

CLIP_KEY = 'model.vision_tower.vision_tower.'
POOL_KEY = 'model.image_pooler.'


def _load_file(path):
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location='cpu', weights_only=True)


def load_dir_state(path):
    """All weight shards of a HF-format directory, merged into one dict."""
    files = sorted(glob.glob(os.path.join(path, '*.safetensors')))
    if not files:
        files = sorted(glob.glob(os.path.join(path, 'pytorch_model*.bin')))
    if not files:
        raise FileNotFoundError(f'no *.safetensors or pytorch_model*.bin under {path!r}')
    state = {}
    for f in files:
        state.update(_load_file(f))
    return state


def strip_non_lora_prefixes(sd):
    """builder.py:81-83, verbatim semantics."""
    sd = {(k[11:] if k.startswith('base_model.') else k): v for k, v in sd.items()}
    if any(k.startswith('model.model.') for k in sd):
        sd = {(k[6:] if k.startswith('model.') else k): v for k, v in sd.items()}
    return sd


def merge_lora(state, adapter, r, alpha, fan_in_fan_out=False):
    """In place: W <- bf16(fp32(W) + (alpha / r) * B @ A). Adapter keys follow peft's saved layout
    `base_model.model.<module path>.lora_{A,B}.weight` (an adapter name segment, `.default`, is tolerated)."""
    scaling = float(alpha) / float(r)
    pairs = {}
    for k, v in adapter.items():
        for tag in ('lora_A', 'lora_B'):
            marker = f'.{tag}.'
            if marker in k:
                mod = k[:k.index(marker)]
                if mod.startswith('base_model.model.'):
                    mod = mod[len('base_model.model.'):]
                pairs.setdefault(mod, {})[tag] = v
    merged = []
    for mod, ab in pairs.items():
        if 'lora_A' not in ab or 'lora_B' not in ab:
            raise KeyError(f'LoRA adapter for {mod!r} lacks its A or B matrix')
        key = mod + '.weight'
        if key not in state:
            raise KeyError(f'LoRA target {key!r} is not in the base model')
        delta = ab['lora_B'].float() @ ab['lora_A'].float()
        if fan_in_fan_out:
            delta = delta.t()
        w = state[key]
        if delta.shape != w.shape:
            raise ValueError(f'LoRA delta {tuple(delta.shape)} does not match {key} {tuple(w.shape)}')
        state[key] = (w.float() + delta * scaling).to(w.dtype)
        merged.append(key)
    return merged


_HF_TO_DIMS = {'vocab_size': 'vocab', 'hidden_size': 'd_model', 'num_hidden_layers': 'n_layers', 'num_attention_heads': 'n_heads',
               'intermediate_size': 'd_ff', 'rms_norm_eps': 'rms_eps', 'rope_theta': 'rope_theta'}


def dims_from_config(cfg):
    """Dims from a config.json written either with this package's field names or with LlavaConfig / LlamaConfig names
    (reference llava_llama.py:27-28 inherits LlamaConfig). CLIP and pooler shapes are fixed by the reference
    (CLIP ViT-L/14-336, multimodal_projector/builder.py:62-71) unless the file overrides them."""
    from ...dims import Dims
    kw = {k: cfg[k] for k in Dims.__dataclass_fields__ if k in cfg}
    for hf, mine in _HF_TO_DIMS.items():
        if hf in cfg and mine not in kw:
            kw[mine] = cfg[hf]
    if cfg.get('num_key_value_heads', kw.get('n_heads', 32)) != kw.get('n_heads', 32):
        raise NotImplementedError('grouped-query attention checkpoints are outside the ORacle path (Vicuna-7B is MHA)')
    return Dims(**kw)


def assemble_state(model_path, model_base, clip_path=None, dtype=torch.bfloat16):
    """-> (state dict with reference checkpoint keys, raw config dict of model_path/config.json)."""
    with open(os.path.join(model_path, 'config.json')) as f:
        cfg = json.load(f)
    state = {k: v.to(dtype) for k, v in load_dir_state(model_base).items()}
    non_lora = {}
    nl_path = os.path.join(model_path, 'non_lora_trainables.bin')
    if os.path.exists(nl_path):
        non_lora = strip_non_lora_prefixes(_load_file(nl_path))
        non_lora = {k: v.to(dtype) for k, v in non_lora.items()}
    # (2) everything that is neither the vision tower nor the pooler goes straight onto the LLaVA model
    for k, v in non_lora.items():
        if not k.startswith('model.vision_tower.') and not k.startswith(POOL_KEY):
            state[k] = v
    # (3) LoRA
    ad_cfg_path = os.path.join(model_path, 'adapter_config.json')
    if os.path.exists(ad_cfg_path):
        with open(ad_cfg_path) as f:
            ad_cfg = json.load(f)
        ad_file = os.path.join(model_path, 'adapter_model.bin')
        if not os.path.exists(ad_file):
            ad_file = os.path.join(model_path, 'adapter_model.safetensors')
        merge_lora(state, _load_file(ad_file), ad_cfg['r'], ad_cfg['lora_alpha'], ad_cfg.get('fan_in_fan_out', False))
    # (4) CLIP tower: hub names cannot be fetched; a local directory is required
    clip_path = clip_path or cfg.get('mm_vision_tower')
    if not clip_path or not os.path.isdir(clip_path):
        raise FileNotFoundError(f'mm_vision_tower {clip_path!r} is not a local directory (hub downloads are unavailable offline); '
                                'pass a local CLIP ViT-L/14-336 checkpoint directory')
    for k, v in load_dir_state(clip_path).items():
        if k.startswith('vision_model.'):
            state[CLIP_KEY + k] = v.to(dtype)
    for k, v in non_lora.items():
        if k.startswith(CLIP_KEY):
            state[k] = v
    # (5) pooler, strict
    pool = {k: v for k, v in non_lora.items() if k.startswith(POOL_KEY)}
    pool.pop(POOL_KEY + 'bert.embeddings.position_ids', None)
    pool.pop(POOL_KEY + 'bert.embeddings.token_type_ids', None)
    if pool:
        from ...synth import POOL_PREFIX, weight_spec
        dims = dims_from_config(cfg)
        need = {k for k in weight_spec(dims) if k.startswith(POOL_PREFIX)}   # includes the (unused) BertPooler head
        have = set(pool)
        if need - have:
            raise KeyError(f'image pooler weights missing from non_lora_trainables.bin (strict load, builder.py:178): {sorted(need - have)[:4]} ...')
        if have - need:
            raise KeyError(f'unexpected image pooler weights (strict load, builder.py:178): {sorted(have - need)[:4]} ...')
        state.update(pool)
    return state, cfg
