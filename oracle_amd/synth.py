"""Deterministic synthetic weights and frames (no released ORacle/LLaVA weights exist offline).

Tensor names are the reference's checkpoint keys (transformers-4.31 layout), i.e. what
`load_pretrained_model` (reference LLaVA/llava/model/builder.py:26-185) would hand over:
`model.layers.N.self_attn.q_proj.weight`, `model.vision_tower.vision_tower.vision_model.*`,
`model.image_pooler.bert.*`, `model.mm_projector.{0,2}.*`, `lm_head.weight`.

Values come from numpy's counter-based Philox keyed by (seed, crc32(name)), so any machine with
this image regenerates the same bits without torch's RNG; `device_fast=True` uses torch.randn on
the GPU instead (bench-only: full 7B weights, no bit-reproducibility needed).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .dims import Dims

CLIP_PREFIX = 'model.vision_tower.vision_tower.vision_model.'
POOL_PREFIX = 'model.image_pooler.bert.'
PROJ_PREFIX = 'model.mm_projector.'


def weight_spec(d: Dims):
    """Ordered {name: (shape, kind)}; kind in {'w','b','ln_w','ln_b','emb'} picks the distribution."""
    s = OrderedDict()
    # --- CLIP vision tower
    p = CLIP_PREFIX
    s[p + 'embeddings.class_embedding'] = ((d.clip_d,), 'emb')
    s[p + 'embeddings.patch_embedding.weight'] = ((d.clip_d, 3, d.clip_patch, d.clip_patch), 'w')
    s[p + 'embeddings.position_embedding.weight'] = ((d.clip_tokens, d.clip_d), 'emb')
    s[p + 'pre_layrnorm.weight'] = ((d.clip_d,), 'ln_w')
    s[p + 'pre_layrnorm.bias'] = ((d.clip_d,), 'ln_b')
    for i in range(d.clip_layers):
        q = f'{p}encoder.layers.{i}.'
        for ln in ('layer_norm1', 'layer_norm2'):
            s[q + ln + '.weight'] = ((d.clip_d,), 'ln_w')
            s[q + ln + '.bias'] = ((d.clip_d,), 'ln_b')
        for nm in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
            s[q + f'self_attn.{nm}.weight'] = ((d.clip_d, d.clip_d), 'w')
            s[q + f'self_attn.{nm}.bias'] = ((d.clip_d,), 'b')
        s[q + 'mlp.fc1.weight'] = ((d.clip_ff, d.clip_d), 'w')
        s[q + 'mlp.fc1.bias'] = ((d.clip_ff,), 'b')
        s[q + 'mlp.fc2.weight'] = ((d.clip_d, d.clip_ff), 'w')
        s[q + 'mlp.fc2.bias'] = ((d.clip_d,), 'b')
    s[p + 'post_layernorm.weight'] = ((d.clip_d,), 'ln_w')
    s[p + 'post_layernorm.bias'] = ((d.clip_d,), 'ln_b')
    # --- learned multi-view pooler (BERT)
    p = POOL_PREFIX
    s[p + 'embeddings.word_embeddings.weight'] = ((1, d.clip_d), 'emb')
    s[p + 'embeddings.position_embeddings.weight'] = ((d.pool_maxpos, d.clip_d), 'emb')
    s[p + 'embeddings.token_type_embeddings.weight'] = ((2, d.clip_d), 'emb')
    s[p + 'embeddings.LayerNorm.weight'] = ((d.clip_d,), 'ln_w')
    s[p + 'embeddings.LayerNorm.bias'] = ((d.clip_d,), 'ln_b')
    for i in range(d.pool_layers):
        q = f'{p}encoder.layer.{i}.'
        for nm in ('query', 'key', 'value'):
            s[q + f'attention.self.{nm}.weight'] = ((d.clip_d, d.clip_d), 'w')
            s[q + f'attention.self.{nm}.bias'] = ((d.clip_d,), 'b')
        s[q + 'attention.output.dense.weight'] = ((d.clip_d, d.clip_d), 'w')
        s[q + 'attention.output.dense.bias'] = ((d.clip_d,), 'b')
        s[q + 'attention.output.LayerNorm.weight'] = ((d.clip_d,), 'ln_w')
        s[q + 'attention.output.LayerNorm.bias'] = ((d.clip_d,), 'ln_b')
        s[q + 'intermediate.dense.weight'] = ((d.pool_ff, d.clip_d), 'w')
        s[q + 'intermediate.dense.bias'] = ((d.pool_ff,), 'b')
        s[q + 'output.dense.weight'] = ((d.clip_d, d.pool_ff), 'w')
        s[q + 'output.dense.bias'] = ((d.clip_d,), 'b')
        s[q + 'output.LayerNorm.weight'] = ((d.clip_d,), 'ln_w')
        s[q + 'output.LayerNorm.bias'] = ((d.clip_d,), 'ln_b')
    s[p + 'pooler.dense.weight'] = ((d.clip_d, d.clip_d), 'w')
    s[p + 'pooler.dense.bias'] = ((d.clip_d,), 'b')
    # --- projector mlp2x_gelu
    s[PROJ_PREFIX + '0.weight'] = ((d.d_model, d.clip_d), 'w')
    s[PROJ_PREFIX + '0.bias'] = ((d.d_model,), 'b')
    s[PROJ_PREFIX + '2.weight'] = ((d.d_model, d.d_model), 'w')
    s[PROJ_PREFIX + '2.bias'] = ((d.d_model,), 'b')
    # --- LLaMA
    s['model.embed_tokens.weight'] = ((d.vocab, d.d_model), 'emb')
    for i in range(d.n_layers):
        q = f'model.layers.{i}.'
        s[q + 'input_layernorm.weight'] = ((d.d_model,), 'ln_w')
        for nm in ('q_proj', 'k_proj', 'v_proj', 'o_proj'):
            s[q + f'self_attn.{nm}.weight'] = ((d.d_model, d.d_model), 'w')
        s[q + 'post_attention_layernorm.weight'] = ((d.d_model,), 'ln_w')
        s[q + 'mlp.gate_proj.weight'] = ((d.d_ff, d.d_model), 'w')
        s[q + 'mlp.up_proj.weight'] = ((d.d_ff, d.d_model), 'w')
        s[q + 'mlp.down_proj.weight'] = ((d.d_model, d.d_ff), 'w')
    s['model.norm.weight'] = ((d.d_model,), 'ln_w')
    s['lm_head.weight'] = ((d.vocab, d.d_model), 'w')
    return s


_STD = {'w': 0.02, 'b': 0.02, 'ln_b': 0.02, 'emb': 0.02}


def _philox_normal(seed, name, shape):
    key = np.array([seed & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())], dtype=np.uint64)
    g = np.random.Generator(np.random.Philox(key=key))
    return g.standard_normal(size=shape, dtype=np.float32)


def synth_tensor(name, shape, kind, seed=0, dtype=torch.bfloat16, device='cpu', device_fast=False):
    if device_fast:
        x = torch.randn(shape, device=device, dtype=torch.float32)
    else:
        x = torch.from_numpy(_philox_normal(seed, name, shape))
    if kind == 'ln_w':
        x = 1.0 + 0.1 * x
    else:
        x = _STD[kind] * x
    x = x.to(dtype)
    return x if device_fast else x.to(device)


def synth_state(d: Dims, seed=0, dtype=torch.bfloat16, device='cpu', device_fast=False, only=None):
    """{name: tensor} for every weight of the path (or the names in `only`)."""
    out = OrderedDict()
    for name, (shape, kind) in weight_spec(d).items():
        if only is not None and name not in only:
            continue
        out[name] = synth_tensor(name, shape, kind, seed, dtype, device, device_fast)
    return out


CHAIN_BASE = 20000
CHAIN_LEN = 400


def install_chain(state, d: Dims, base=CHAIN_BASE, length=CHAIN_LEN, logit=30.0):
    """Make greedy decoding numerically robust ("peaked logits", SURVEY.md §7 hard parts).

    Tokens base..base+length form a chain: the embedding rows of chain tokens are boosted so the last
    token dominates the residual stream, and lm_head row v+1 gets a component along embed[v], so the
    token after v is v+1 by a margin far above bf16 noise; the last chain token is followed by EOS (id 2).
    A prompt ending n tokens before the end of the chain therefore produces exactly n tokens then EOS,
    while all layers, the KV cache and the image tokens still take part in every logit."""
    import math
    emb = state['model.embed_tokens.weight'].float()
    w = state['lm_head.weight'].float()
    lo, hi = base, base + length
    u = emb[lo:hi + 1]
    u = u / u.norm(dim=-1, keepdim=True)
    boost = 4.0 + 3.0 * 0.0004 * d.d_model * math.sqrt(2 * d.n_layers)
    emb[lo:hi + 1] = u * (boost * math.sqrt(d.d_model))
    g = logit / math.sqrt(d.d_model)
    w[lo + 1:hi + 1] += g * u[:-1]
    w[2] += g * u[-1]
    state['model.embed_tokens.weight'] = emb.to(state['model.embed_tokens.weight'].dtype)
    state['lm_head.weight'] = w.to(state['lm_head.weight'].dtype)
    return state


def synth_images(n_frames, views, d: Dims, seed=1234, dtype=torch.bfloat16):
    """Frames ~ N(0,1) clipped to the CLIP-normalised range; seed = 1234 + frame index.
    Returns (n_frames, views, 3, H, W) (or (n_frames, 3, H, W) when views == 1 -- the shape
    `process_images` gives, reference LLaVA/llava/mm_utils.py:38-39)."""
    frames = []
    for f in range(n_frames):
        x = _philox_normal(seed + f, 'frame', (views, 3, d.clip_image, d.clip_image))
        frames.append(torch.from_numpy(np.clip(x, -1.8, 2.2)))
    x = torch.stack(frames).to(dtype)
    return x[:, 0] if views == 1 else x


SELECT_DIRS = 2          # image-dependent directions; 2 * SELECT_DIRS candidate chains
SELECT_CHAIN_LEN = 24    # tokens reserved per candidate chain inside [CHAIN_BASE, ...)


def select_chain_layout(n_dirs=SELECT_DIRS, base=CHAIN_BASE, stride=SELECT_CHAIN_LEN):
    """[(first token, last token)] of the 2*n_dirs candidate chains; chain k emits 6 + 3k tokens, then EOS."""
    return [(base + k * stride, base + k * stride + 5 + 3 * k) for k in range(2 * n_dirs)]


def install_select_chains(state, d: Dims, center=None, n_dirs=SELECT_DIRS, g_sel=60.0, logit=60.0, boost=12.0, seed=77, dirs=None):
    """Greedy decoding whose answer is SELECTED BY THE IMAGE: with the <image> placeholder as the last
    prompt token, the final hidden state of the prefill is the LLaMA image of the last pooled + projected image-token row,
    i.e. ~70 % of it varies from frame to frame. lm_head rows of the 2*n_dirs chain-start tokens get +-g_sel along n_dirs
    fixed random unit directions r_j of that state, so the FIRST answer token is argmax_k (+-<h, r_j>): a function of
    the vision tower, the pooler (mask, positions), the projector, the splice and the LLaMA layers, with a top-2 margin
    that is a continuous random variable (moderately peaked: some frames land within bf16 noise, most do not). The
    decode then continues along the selected chain (token v -> v+1 by a wide margin, as install_chain) for 6 + 3k tokens
    and ends with EOS, so rows of a batch finish at different steps. Chain-token embeddings are made orthogonal to the
    r_j, and their boost keeps the selection rows quiet at the later positions. `center` (d_model floats) is the mean of
    that hidden state over a few calibration frames (computed by the test infrastructure and stored with the golden
    vectors); the r_j are made orthogonal to it so the frame-independent part does not decide the argmax.
    `dirs` (K <= 2 * n_dirs rows of d_model floats): CALIBRATED directions instead of random ones -- chain k starts where
    <h, dirs[k]> is largest (no +- pairs). With dirs[k] = the displacement of image class k's mean hidden state from the
    centre, the whole signal lies along the direction while rounding noise is spread over d_model dims: the first-token
    margin is then far above the noise of a 32-layer bf16 stack (tests/test_full_size_gpu.py: the 99 % identity rate)."""
    import math
    emb = state['model.embed_tokens.weight'].float()
    w = state['lm_head.weight'].float()
    dm = d.d_model
    r = torch.from_numpy(_philox_normal(seed, 'select_dirs', (n_dirs, dm))) if dirs is None else torch.as_tensor(dirs, dtype=torch.float32).clone()
    if center is not None:
        c = torch.as_tensor(center, dtype=torch.float32)
        c = c / c.norm()
        r = r - (r @ c)[:, None] * c
    if dirs is not None:
        assert r.shape[0] <= 2 * n_dirs and r.shape[1] == dm
        start = r / r.norm(dim=-1, keepdim=True)                       # chain-start rows: the class displacements themselves (centre removed)
    r = torch.linalg.qr(r.T).Q.T.contiguous()                          # orthonormal rows (still orthogonal to center)
    g = logit / math.sqrt(dm)
    for k, (lo, hi) in enumerate(select_chain_layout(n_dirs)):
        if dirs is not None and k >= start.shape[0]:
            break
        u = emb[lo:hi + 1]
        u = u - (u @ r.T) @ r                                           # orthogonal to the selection directions
        u = u / u.norm(dim=-1, keepdim=True)
        emb[lo:hi + 1] = u * (boost * math.sqrt(dm))
        w[lo + 1:hi + 1] += g * u[:-1]                                  # v -> v + 1 inside the chain
        w[2] += g * u[-1]                                               # last token of the chain -> EOS
        if dirs is not None:
            w[lo] += g_sel * start[k]                                   # chain start: <h, class-k displacement>
        else:
            w[lo] += (g_sel if k % 2 == 0 else -g_sel) * r[k // 2]      # chain start: +- <h, r_j>
    state['model.embed_tokens.weight'] = emb.to(state['model.embed_tokens.weight'].dtype)
    state['lm_head.weight'] = w.to(state['lm_head.weight'].dtype)
    return state
