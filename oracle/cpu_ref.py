"""TEST INFRASTRUCTURE ONLY -- CPU restatement of ORacle's multi-view LLaVA inference hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (oracle_amd/) never does. It restates, in plain torch on the CPU, what the reference
computes for one generate() call (scene_graph_prediction_model_oracle.py:133-143):

  CLIPVisionTower.forward / feature_select      LLaVA/llava/model/multimodal_encoder/clip_encoder.py:29-51
  ImageEmbeddingPooler.forward                  LLaVA/llava/model/multimodal_projector/builder.py:56-82
  build_vision_projector('mlp2x_gelu')          LLaVA/llava/model/multimodal_projector/builder.py:41-48
  encode_images* / pad_embeddings               LLaVA/llava/model/llava_arch.py:103-183
  prepare_inputs_labels_for_multimodal          LLaVA/llava/model/llava_arch.py:188-374
  LlavaLlamaForCausalLM.forward                 LLaVA/llava/model/language_model/llava_llama.py:54-100
  prepare_inputs_for_generation + greedy search llava_llama.py:102-112 (+ transformers GenerationMixin)

The arithmetic of CLIP / BERT / LLaMA lives in the third-party `transformers` (pinned 4.31.0 in
LLaVA/pyproject.toml:17, not vendored, absent offline); it is restated here from the published
algorithm as executed by the container's transformers 5.15 eager path.
PINNING: oracle/gen_golden.py runs the real reference glue (oracle/ref_harness.py) on the same seeded
weights/inputs and tests/test_oracle_vs_golden.py checks this file against those outputs; the reference
itself ships no tests or golden vectors, so beyond "equal to the reference run in this
container" parity is unpinned against the original transformers-4.31 numerics.

Weights: dict name -> tensor with the reference's checkpoint key names (oracle_amd/synth.py).
Every function computes in the dtype of its inputs (fp32 for tight checks, bf16 for the stated tolerance).
"""
import math

import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200          # LLaVA/llava/constants.py:8
VIS_DESCRIPTOR_TOKEN_INDEX = 18610  # LLaVA/llava/constants.py:10

CLIP = 'model.vision_tower.vision_tower.vision_model.'
POOL = 'model.image_pooler.bert.'
PROJ = 'model.mm_projector.'


def _lin(x, st, name, bias=True):
    return F.linear(x, st[name + '.weight'], st[name + '.bias'] if bias else None)


def _ln(x, st, name, eps):
    return F.layer_norm(x, (x.shape[-1],), st[name + '.weight'], st[name + '.bias'], eps)


def _eager_attention(q, k, v, add_mask, scaling, softmax_fp32):
    """q,k,v (B,H,S,hd). scores = (q k^T) * scaling (+ mask); softmax; @ v."""
    w = torch.matmul(q, k.transpose(-1, -2)) * scaling
    if add_mask is not None:
        w = w + add_mask
    if softmax_fp32:
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    else:
        w = F.softmax(w, dim=-1)
    return torch.matmul(w, v)


# ----------------------------------------------------------------------------- CLIP ViT (A3)
def clip_hidden_states(st, d, images, upto=None):
    """Hidden states after `upto` encoder layers (default: clip_layers-1 == hidden_states[-2])."""
    upto = d.clip_layers - 1 if upto is None else upto
    x = images.to(st[CLIP + 'embeddings.patch_embedding.weight'].dtype)
    n = x.shape[0]
    patch = F.conv2d(x, st[CLIP + 'embeddings.patch_embedding.weight'], None, stride=d.clip_patch)
    patch = patch.flatten(2).transpose(1, 2)
    cls = st[CLIP + 'embeddings.class_embedding'].expand(n, 1, -1)
    h = torch.cat([cls, patch], dim=1) + st[CLIP + 'embeddings.position_embedding.weight'][None]
    h = _ln(h, st, CLIP + 'pre_layrnorm', d.clip_eps)
    hd = d.clip_d // d.clip_heads
    for i in range(upto):
        p = f'{CLIP}encoder.layers.{i}.'
        r = h
        y = _ln(h, st, p + 'layer_norm1', d.clip_eps)
        shp = (n, -1, d.clip_heads, hd)
        q = _lin(y, st, p + 'self_attn.q_proj').view(shp).transpose(1, 2)
        k = _lin(y, st, p + 'self_attn.k_proj').view(shp).transpose(1, 2)
        v = _lin(y, st, p + 'self_attn.v_proj').view(shp).transpose(1, 2)
        a = _eager_attention(q, k, v, None, hd ** -0.5, softmax_fp32=True)
        a = a.transpose(1, 2).reshape(n, -1, d.clip_d)
        h = r + _lin(a, st, p + 'self_attn.out_proj')
        r = h
        y = _ln(h, st, p + 'layer_norm2', d.clip_eps)
        y = _lin(y, st, p + 'mlp.fc1')
        y = y * torch.sigmoid(1.702 * y)          # QuickGELU
        h = r + _lin(y, st, p + 'mlp.fc2')
    return h


def vision_tower(st, d, images):
    """clip_encoder.py:40-51: hidden_states[select_layer=-2][:, 1:] cast back to the image dtype."""
    return clip_hidden_states(st, d, images)[:, 1:].to(images.dtype)


def vision_tower_global(st, d, images):
    """clip_encoder.py:53-64 (visual-descriptor side path): pooler_output = post_layernorm(h_last[:, 0])."""
    h = clip_hidden_states(st, d, images, upto=d.clip_layers)
    return _ln(h[:, 0], st, CLIP + 'post_layernorm', d.clip_eps)


# ----------------------------------------------------------------------------- learned pooler (A4)
def pad_embeddings(feats_per_row):
    """llava_arch.py:143-170: list of (V_b,576,1024) -> (B, Vmax*576, 1024) zero padded + bool mask."""
    b = len(feats_per_row)
    img_len, dim = feats_per_row[0].shape[1:]
    vmax = max(f.shape[0] for f in feats_per_row)
    out = torch.zeros(b, vmax, img_len, dim, dtype=feats_per_row[0].dtype)
    mask = torch.zeros(b, vmax * img_len, dtype=torch.bool)
    for i, f in enumerate(feats_per_row):
        out[i, :f.shape[0]] = f
        mask[i, :f.shape[0] * img_len] = True
    return out.flatten(1, 2), mask


def image_pooler(st, d, emb, mask):
    """multimodal_projector/builder.py:74-82 on BertModel(inputs_embeds, attention_mask): 2 post-LN layers."""
    b, s, _ = emb.shape
    dt = emb.dtype
    x = emb + st[POOL + 'embeddings.token_type_embeddings.weight'][0]
    x = x + st[POOL + 'embeddings.position_embeddings.weight'][:s]
    x = _ln(x, st, POOL + 'embeddings.LayerNorm', d.pool_eps)
    add = torch.zeros(b, 1, 1, s, dtype=dt)
    add.masked_fill_(~mask[:, None, None, :], torch.finfo(dt).min)
    hd = d.clip_d // d.pool_heads
    for i in range(d.pool_layers):
        p = f'{POOL}encoder.layer.{i}.'
        shp = (b, s, d.pool_heads, hd)
        q = _lin(x, st, p + 'attention.self.query').view(shp).transpose(1, 2)
        k = _lin(x, st, p + 'attention.self.key').view(shp).transpose(1, 2)
        v = _lin(x, st, p + 'attention.self.value').view(shp).transpose(1, 2)
        a = _eager_attention(q, k, v, add, hd ** -0.5, softmax_fp32=False)
        a = a.transpose(1, 2).reshape(b, s, d.clip_d)
        x = _ln(_lin(a, st, p + 'attention.output.dense') + x, st, p + 'attention.output.LayerNorm', d.pool_eps)
        y = F.gelu(_lin(x, st, p + 'intermediate.dense'))
        x = _ln(_lin(y, st, p + 'output.dense') + x, st, p + 'output.LayerNorm', d.pool_eps)
    return x.to(dt)[:, :576]


# ----------------------------------------------------------------------------- projector (A5)
def mm_projector(st, x):
    return _lin(F.gelu(_lin(x, st, PROJ + '0')), st, PROJ + '2')


# ----------------------------------------------------------------------------- view fusion (A6, llava_arch.py:203-239)
def encode_image_features(st, d, images, mv_type):
    """images: (B,V,3,H,W) / list of (V_b,3,H,W) (5-D branch) or (B,3,H,W) (4-D branch). -> (B, N, d_model)."""
    five_d = isinstance(images, (list, tuple)) or images.ndim == 5
    if five_d:
        rows = [im for im in images]
        cat = torch.cat(rows, dim=0)
        split = [im.shape[0] for im in rows]
        if mv_type == 'learned':
            feats = torch.split(vision_tower(st, d, cat), split, dim=0)
            emb, mask = pad_embeddings(feats)
            return mm_projector(st, image_pooler(st, d, emb, mask))
        if mv_type == 'concat_pooled':
            f = torch.split(_concat_pool(st, d, cat), split, dim=0)
            return torch.stack([x.flatten(0, 1) for x in f])
        f = torch.split(mm_projector(st, vision_tower(st, d, cat)), split, dim=0)
        if mv_type == 'concat':
            return torch.stack([x.flatten(0, 1) for x in f])
        if mv_type == 'max':
            return torch.stack([x.max(dim=0)[0] for x in f])
        if mv_type == 'avg':
            return torch.stack([x.mean(dim=0) for x in f])
        raise ValueError(mv_type)
    if mv_type == 'concat_pooled':
        return _concat_pool(st, d, images)
    if mv_type == 'learned':
        f = vision_tower(st, d, images)
        mask = torch.ones(f.shape[0], f.shape[1], dtype=torch.bool)
        return mm_projector(st, image_pooler(st, d, f, mask))
    return mm_projector(st, vision_tower(st, d, images))


def _concat_pool(st, d, images):
    """llava_arch.py:128-141: projector then 2x2 max-pool of the 24x24 token grid."""
    f = mm_projector(st, vision_tower(st, d, images)).permute(0, 2, 1)
    f = f.reshape(f.shape[0], f.shape[1], d.clip_grid, d.clip_grid)
    f = F.max_pool2d(f, kernel_size=2, stride=2)
    return f.reshape(f.shape[0], f.shape[1], -1).permute(0, 2, 1)


# ----------------------------------------------------------------------------- splice (A7, llava_arch.py:245-374)
def splice(st, input_ids, attention_mask, image_features, vis_descriptor_embs=None,
           padding_side='left', max_length=None):
    """Returns (inputs_embeds (B,T,D), attention_mask (B,T) bool, position_ids (B,T) int64, row_source).

    row_source[b] lists, for every spliced row of batch element b, where it came from:
    ('tok', id) | ('img', image_idx, row) | ('desc', j, row) -- used to pin the index work exactly."""
    embed = st['model.embed_tokens.weight']
    b = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    attention_mask = attention_mask.bool()
    rows, srcs = [], []
    img_idx = 0
    for bi in range(b):
        ids = input_ids[bi][attention_mask[bi]]
        n_img = int((ids == IMAGE_TOKEN_INDEX).sum())
        n_desc = int((ids == VIS_DESCRIPTOR_TOKEN_INDEX).sum())
        if n_img == 0:
            rows.append(embed[ids])
            srcs.append([('tok', int(t)) for t in ids])
            img_idx += 1
            continue
        special = [-1] + torch.where((ids == IMAGE_TOKEN_INDEX) | (ids == VIS_DESCRIPTOR_TOKEN_INDEX))[0].tolist() + [ids.shape[0]]
        chunks = [ids[special[i] + 1:special[i + 1]] for i in range(len(special) - 1)]
        parts, src = [], []
        for i in range(n_img + 1):
            parts.append(embed[chunks[i]])
            src += [('tok', int(t)) for t in chunks[i]]
            if i < n_img:
                f = image_features[img_idx]
                parts.append(f)
                src += [('img', img_idx, r) for r in range(f.shape[0])]
                img_idx += 1
        if vis_descriptor_embs is not None:
            ve = vis_descriptor_embs
            if not isinstance(ve[0], list):      # batch size 1 (llava_arch.py:306-307)
                ve = [ve]
            for j in range(n_desc):
                f = ve[bi][j]
                if f.ndim == 1:
                    f = f.unsqueeze(0)
                parts.append(f)
                src += [('desc', j, r) for r in range(f.shape[0])]
                parts.append(embed[chunks[n_img + j + 1]])
                src += [('tok', int(t)) for t in chunks[n_img + j + 1]]
        # NB (llava_arch.py:305): with descriptor tokens but no embeddings the trailing chunks vanish.
        rows.append(torch.cat(parts))
        srcs.append(src)
    if max_length is not None:
        rows = [r[:max_length] for r in rows]
        srcs = [s[:max_length] for s in srcs]
    t = max(r.shape[0] for r in rows)
    dim = rows[0].shape[1]
    out = torch.zeros(b, t, dim, dtype=rows[0].dtype)
    mask = torch.zeros(b, t, dtype=torch.bool)
    pos = torch.zeros(b, t, dtype=torch.long)
    for bi, r in enumerate(rows):
        n = r.shape[0]
        if n == 0:
            continue
        if padding_side == 'left':
            out[bi, t - n:] = r
            mask[bi, t - n:] = True
            pos[bi, t - n:] = torch.arange(n)
        else:
            out[bi, :n] = r
            mask[bi, :n] = True
            pos[bi, :n] = torch.arange(n)
    return out, mask, pos, srcs


# ----------------------------------------------------------------------------- LLaMA (A8)
def _rms(x, w, eps):
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_cos_sin(d, position_ids, dtype):
    hd = d.head_dim
    inv = 1.0 / (d.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    fr = (inv[None, :, None].expand(position_ids.shape[0], -1, 1) @ position_ids[:, None, :].float()).transpose(1, 2)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def llama_forward(st, d, inputs_embeds, mask_full, position_ids, kv=None, all_logits=False, return_hidden=False):
    """One LlamaForCausalLM.forward. inputs_embeds (B,T,D) are the NEW positions; mask_full (B, past+T) bool
    covers cached + new keys; kv = list of (k,v) per layer (K cached after RoPE) or None.
    Returns (logits fp32 (B,V) at the last position or (B,T,V), new_kv)."""
    b, t, _ = inputs_embeds.shape
    dt = inputs_embeds.dtype
    past = 0 if kv is None else kv[0][0].shape[2]
    s = past + t
    cos, sin = rope_cos_sin(d, position_ids, dt)
    cos, sin = cos[:, None], sin[:, None]
    # additive mask: causal over absolute indices AND key padding; masked = finfo.min
    qi = torch.arange(past, s)[:, None]
    kj = torch.arange(s)[None, :]
    allowed = (kj <= qi)[None, None] & mask_full[:, None, None, :].bool()
    add = torch.zeros(b, 1, t, s, dtype=dt).masked_fill_(~allowed, torch.finfo(dt).min)
    h = inputs_embeds
    new_kv = []
    hd = d.head_dim
    for i in range(d.n_layers):
        p = f'model.layers.{i}.'
        r = h
        y = _rms(h, st[p + 'input_layernorm.weight'], d.rms_eps)
        shp = (b, t, d.n_heads, hd)
        q = _lin(y, st, p + 'self_attn.q_proj', False).view(shp).transpose(1, 2)
        k = _lin(y, st, p + 'self_attn.k_proj', False).view(shp).transpose(1, 2)
        v = _lin(y, st, p + 'self_attn.v_proj', False).view(shp).transpose(1, 2)
        q = (q * cos) + (_rot_half(q) * sin)
        k = (k * cos) + (_rot_half(k) * sin)
        if kv is not None:
            k = torch.cat([kv[i][0], k], dim=2)
            v = torch.cat([kv[i][1], v], dim=2)
        new_kv.append((k, v))
        a = _eager_attention(q, k, v, add, hd ** -0.5, softmax_fp32=True)
        a = a.transpose(1, 2).reshape(b, t, d.d_model)
        h = r + _lin(a, st, p + 'self_attn.o_proj', False)
        r = h
        y = _rms(h, st[p + 'post_attention_layernorm.weight'], d.rms_eps)
        y = _lin(F.silu(_lin(y, st, p + 'mlp.gate_proj', False)) * _lin(y, st, p + 'mlp.up_proj', False),
                 st, p + 'mlp.down_proj', False)
        h = r + y
    h = _rms(h, st['model.norm.weight'], d.rms_eps)
    if not all_logits:
        h = h[:, -1:]
    if return_hidden:          # the lm_head input (final RMSNorm output) of the last position: calibration of synthetic weights
        return h[:, -1].float(), new_kv
    logits = F.linear(h, st['lm_head.weight']).float()
    return (logits if all_logits else logits[:, 0]), new_kv


# ----------------------------------------------------------------------------- greedy search (A9/A10)
def greedy_generate(st, d, input_ids, images, mv_type='learned', max_new_tokens=300, vis_descriptor_embs=None,
                    pad_token_id=0, eos_token_id=2, padding_side='left', max_length=None,
                    forced_tokens=None, return_trace=False, image_features=None, trace_logits_stride=None):
    """model.generate(..., do_sample=False, use_cache=True) as called at scene_graph_prediction_model_oracle.py:135-143.

    forced_tokens (B, n) teacher-forces the fed-back token (the argmax is still recorded) so that GPU and CPU
    runs can be compared step by step without one flip changing all later steps."""
    b, l = input_ids.shape
    mask0 = (input_ids != pad_token_id) if bool((input_ids == pad_token_id).any()) else torch.ones_like(input_ids, dtype=torch.bool)
    if image_features is None:
        image_features = encode_image_features(st, d, images, mv_type)
    emb, mask, pos, _ = splice(st, input_ids, mask0, image_features, vis_descriptor_embs, padding_side, max_length)
    logits, kv = llama_forward(st, d, emb, mask, pos)
    unfinished = torch.ones(b, dtype=torch.long)
    cur = input_ids
    trace = []
    attn = mask0
    for step in range(max_new_tokens):
        nxt = torch.argmax(logits, dim=-1)
        if return_trace:
            top2 = torch.topk(logits, 2, dim=-1)
            trace.append({'argmax': nxt.clone(), 'top2_val': top2.values.clone(), 'top2_idx': top2.indices.clone()})
            if trace_logits_stride:
                trace[-1]['logits_sub'] = logits[:, ::trace_logits_stride].clone()
        nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        cur = torch.cat([cur, nxt[:, None]], dim=1)
        attn = torch.cat([attn, torch.ones(b, 1, dtype=torch.bool)], dim=1)
        unfinished = unfinished * (nxt != eos_token_id).long()
        if int(unfinished.max()) == 0 or step == max_new_tokens - 1:
            break
        feed = nxt if forced_tokens is None else forced_tokens[:, step]
        # decode-step branch llava_arch.py:192-201: mask extended with ones to past+1, position = sum(mask)-1
        past = kv[0][0].shape[2]
        m = torch.cat([attn, torch.ones(b, past + 1 - attn.shape[1], dtype=torch.bool)], dim=1) if past + 1 > attn.shape[1] else attn
        p = m.sum(dim=1, keepdim=True) - 1
        e = st['model.embed_tokens.weight'][feed][:, None]
        logits, kv = llama_forward(st, d, e, m, p, kv)
    return (cur, trace) if return_trace else cur
