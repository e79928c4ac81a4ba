"""Multi-view image-token fusion and the splice into the prompt, with the reference's class and method
names (LLaVA/llava/model/llava_arch.py:27-374). All tensor work runs in the native engine; the host only
turns the integer prompt into a row-source plan (one int32 code per spliced row)."""
import numpy as np
import torch

from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX, VIS_DESCRIPTOR_TOKEN_INDEX
from .multimodal_encoder.builder import build_vision_tower
from .multimodal_projector.builder import build_image_pooler, build_vision_projector

KIND_PAD, KIND_TOK, KIND_IMG, KIND_DESC = 0, 1, 2, 3


class _EmbedTokens:
    """model.get_model().embed_tokens(ids): rows of the engine-resident embedding table."""

    def __init__(self, engine):
        self.engine = engine

    def __call__(self, ids):
        flat = ids.reshape(-1).to(torch.int32)
        codes = (flat | (KIND_TOK << 28)).to(device=self.engine.device, dtype=torch.int32).contiguous()
        out = self.engine.splice(codes, None, None)
        return out.reshape(*ids.shape, self.engine.dims.d_model)


class LlavaMetaModel:
    def __init__(self, config, engine):
        self.config = config
        self.engine = engine
        self.embed_tokens = _EmbedTokens(engine)
        if hasattr(config, 'mm_vision_tower'):
            self.vision_tower = build_vision_tower(config, engine=engine, delay_load=True)
            self.mm_projector = build_vision_projector(config, engine=engine)
            self.image_pooler = build_image_pooler(config, engine=engine)

    def get_vision_tower(self):
        vt = getattr(self, 'vision_tower', None)
        return vt[0] if type(vt) is list else vt

    def get_image_pooler(self):
        return self.image_pooler


def plan_splice(input_ids, attention_mask, feature_rows, desc_rows, padding_side='left', max_length=None):
    """Row-source plan of prepare_inputs_labels_for_multimodal (reference llava_arch.py:262-359).

    input_ids (B,L) int array with IMAGE_TOKEN_INDEX / VIS_DESCRIPTOR_TOKEN_INDEX placeholders,
    attention_mask (B,L) bool, feature_rows = rows per image-feature block, desc_rows = None or per batch row a
    list of row counts of its descriptor embeddings.
    Returns (codes (B,T) int32, mask (B,T) bool, position (B,T) int64, label_src (B,T) int64) where
    label_src is the index into the unpadded row's labels, or -1 for rows that get IGNORE_INDEX."""
    ids = np.asarray(input_ids)
    am = np.asarray(attention_mask).astype(bool)
    rows, lsrc = [], []
    img_idx = 0
    desc_base = 0
    for b in range(ids.shape[0]):
        keep = np.nonzero(am[b])[0]
        r = ids[b][keep]
        n_img = int((r == IMAGE_TOKEN_INDEX).sum())
        n_desc = int((r == VIS_DESCRIPTOR_TOKEN_INDEX).sum())
        tok = lambda lo, hi: ((KIND_TOK << 28) | r[lo:hi].astype(np.int64)).astype(np.int64)
        if n_img == 0:
            rows.append(tok(0, len(r)))
            lsrc.append(np.arange(len(r)))
            img_idx += 1                      # the reference still consumes one feature block
            continue
        special = np.nonzero((r == IMAGE_TOKEN_INDEX) | (r == VIS_DESCRIPTOR_TOKEN_INDEX))[0]
        cut = np.concatenate([[-1], special, [len(r)]])
        chunk = lambda i: (cut[i] + 1, cut[i + 1])
        parts, lparts = [], []
        for i in range(n_img + 1):
            lo, hi = chunk(i)
            parts.append(tok(lo, hi))
            lparts.append(np.arange(lo, hi))
            if i < n_img:
                parts.append((KIND_IMG << 28) | (img_idx * feature_rows + np.arange(feature_rows, dtype=np.int64)))
                lparts.append(np.full(feature_rows, -1))
                img_idx += 1
        if desc_rows is not None:
            counts = desc_rows[b]
            off = desc_base
            for j in range(n_desc):
                parts.append((KIND_DESC << 28) | (off + np.arange(counts[j], dtype=np.int64)))
                lparts.append(np.full(counts[j], -1))
                off += counts[j]
                lo, hi = chunk(n_img + j + 1)
                parts.append(tok(lo, hi))
                lparts.append(np.arange(lo, hi))
            desc_base += int(sum(counts))
        # without descriptor embeddings the text after the first descriptor token is dropped (llava_arch.py:305)
        rows.append(np.concatenate(parts))
        lsrc.append(np.concatenate(lparts))
    if max_length is not None:
        rows = [x[:max_length] for x in rows]
        lsrc = [x[:max_length] for x in lsrc]
    t = max(len(x) for x in rows)
    bsz = len(rows)
    codes = np.zeros((bsz, t), dtype=np.int64)
    mask = np.zeros((bsz, t), dtype=bool)
    pos = np.zeros((bsz, t), dtype=np.int64)
    lab = np.full((bsz, t), -1, dtype=np.int64)
    for b, x in enumerate(rows):
        n = len(x)
        if n == 0:
            continue
        sl = slice(t - n, t) if padding_side == 'left' else slice(0, n)
        codes[b, sl] = x
        mask[b, sl] = True
        pos[b, sl] = np.arange(n)
        lab[b, sl] = lsrc[b]
    return codes.astype(np.uint32).view(np.int32), mask, pos, lab


class LlavaMetaForCausalLM:
    # subclasses provide get_model(), .config, .engine, .device

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def get_image_pooler(self):
        return self.get_model().get_image_pooler()

    def encode_images(self, images):
        return self.get_model().mm_projector(self.get_model().get_vision_tower()(images))

    def encode_images_concat_pool(self, images):
        f = self.encode_images(images)
        return self.engine.pool2x2(f.contiguous())

    def pad_embeddings(self, embeddings, padding_value=0):
        """list of (V_b,576,D) -> ((B, Vmax*576, D), bool mask) as the reference (llava_arch.py:143-170)."""
        b = len(embeddings)
        n, d = embeddings[0].shape[1:]
        vmax = max(e.shape[0] for e in embeddings)
        out = torch.full((b, vmax, n, d), padding_value, dtype=embeddings[0].dtype, device=embeddings[0].device)
        mask = torch.zeros(b, vmax * n, dtype=torch.bool, device=embeddings[0].device)
        for i, e in enumerate(embeddings):
            out[i, :e.shape[0]] = e
            mask[i, :e.shape[0] * n] = True
        return out.flatten(1, 2), mask

    def encode_images_pooled(self, images, split_sizes):
        feats = self.get_model().get_vision_tower()(images)
        if split_sizes is not None:
            pooled = self.engine.pool_views(feats.contiguous(), [int(v) for v in split_sizes])
        else:
            pooled = self.engine.pool_views(feats.contiguous(), [1] * feats.shape[0])
        return self.get_model().mm_projector(pooled)

    def _fuse_views(self, images):
        mv = getattr(self.config, 'mv_type')
        if type(images) is list or images.ndim == 5:
            rows = [im for im in images]
            cat = torch.cat(rows, dim=0)
            split = [int(im.shape[0]) for im in rows]
            if mv == 'learned':
                return self.encode_images_pooled(cat, split)
            if mv == 'concat_pooled':
                f = self.encode_images_concat_pool(cat)
            elif mv in ('concat', 'max', 'avg'):
                f = self.encode_images(cat)
            else:
                raise ValueError(f'unknown mv_type {mv}')
            per = torch.split(f, split, dim=0)
            if mv in ('concat', 'concat_pooled'):
                return torch.stack([x.flatten(0, 1) for x in per])
            if len(set(split)) == 1:
                return self.engine.view_reduce(f.reshape(len(split), split[0], *f.shape[1:]).contiguous(), mv)
            return torch.cat([self.engine.view_reduce(x[None].contiguous(), mv) for x in per])
        if mv == 'concat_pooled':
            return self.encode_images_concat_pool(images)
        if mv == 'learned':
            return self.encode_images_pooled(images, None)
        return self.encode_images(images)

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, vis_descriptor_embs):
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            # decode-step branch (llava_arch.py:192-201): extend the mask to past+1, position = valid count - 1
            if past_key_values is not None and vision_tower is not None and images is not None and input_ids.shape[1] == 1:
                target = past_key_values[-1][-1].shape[-2] + 1
                extra = torch.ones((attention_mask.shape[0], target - attention_mask.shape[1]), dtype=attention_mask.dtype,
                                   device=attention_mask.device)
                attention_mask = torch.cat((attention_mask, extra), dim=1)
                position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
            return input_ids, position_ids, attention_mask, past_key_values, None, labels

        image_features = self._fuse_views(images).to(self.device)
        if getattr(self.config, 'tune_mm_mlp_adapter', False) and getattr(self.config, 'mm_use_im_start_end', False):
            raise NotImplementedError

        dev = input_ids.device
        ids_h = input_ids.detach().cpu().numpy()
        am_h = np.ones_like(ids_h, dtype=bool) if attention_mask is None else attention_mask.detach().cpu().numpy().astype(bool)
        desc_flat, desc_rows = None, None
        if vis_descriptor_embs is not None:
            ve = vis_descriptor_embs if type(vis_descriptor_embs[0]) is list else [vis_descriptor_embs]
            per_row, blocks = [], []
            for b in range(ids_h.shape[0]):
                row = ids_h[b][am_h[b]]
                has_img = bool((row == IMAGE_TOKEN_INDEX).any())
                n_desc = int((row == VIS_DESCRIPTOR_TOKEN_INDEX).sum()) if (b < len(ve) and has_img) else 0
                counts = []
                for j in range(n_desc):
                    f = ve[b][j]
                    f = f.unsqueeze(0) if f.dim() == 1 else f
                    counts.append(int(f.shape[0]))
                    blocks.append(f.to(device=self.device, dtype=torch.bfloat16))
                per_row.append(counts)
            desc_rows = per_row
            if blocks:
                desc_flat = torch.cat(blocks).contiguous()
        feats = image_features.to(torch.bfloat16).contiguous()
        codes, mask, pos, lab = plan_splice(ids_h, am_h, feats.shape[1], desc_rows,
                                            getattr(self.config, 'tokenizer_padding_side', 'right'),
                                            getattr(self.config, 'tokenizer_model_max_length', None))
        b, t = codes.shape
        codes_d = torch.from_numpy(codes.reshape(-1)).to(self.device)
        embeds = self.engine.splice(codes_d, feats.reshape(-1, feats.shape[-1]), desc_flat).reshape(b, t, -1)

        new_labels = None
        if labels is not None:
            lab_t = torch.from_numpy(lab).to(labels.device)
            new_labels = torch.full((b, t), IGNORE_INDEX, dtype=labels.dtype, device=labels.device)
            for i in range(b):
                src = labels[i][torch.from_numpy(am_h[i]).to(labels.device)]
                sel = lab_t[i] >= 0
                new_labels[i, sel] = src[lab_t[i][sel]]
        new_mask = None if attention_mask is None else torch.from_numpy(mask).to(device=dev, dtype=attention_mask.dtype)
        new_pos = None if position_ids is None else torch.from_numpy(pos).to(device=dev, dtype=position_ids.dtype)
        self._last_plan = (mask, pos)
        return None, new_pos, new_mask, past_key_values, embeds, new_labels
