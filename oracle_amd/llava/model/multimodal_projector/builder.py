"""Projector and learned multi-view pooler with the reference's builder names
(LLaVA/llava/model/multimodal_projector/builder.py:35-86); both run in the native engine."""
import re


class MlpProjector:
    """mlp2x_gelu: Linear(1024->D) + GELU(erf) + Linear(D->D) as two MFMA GEMMs (orc_project)."""

    def __init__(self, engine):
        self.engine = engine

    def __call__(self, x):
        return self.engine.project(x.contiguous())


def build_vision_projector(config, delay_load=False, engine=None, **kwargs):
    projector_type = getattr(config, 'mm_projector_type', 'linear')
    m = re.match(r'^mlp(\d+)x_gelu$', projector_type)
    if m and int(m.group(1)) == 2:
        return MlpProjector(engine)
    if projector_type in ('linear', 'identity') or m:
        raise NotImplementedError(f'projector {projector_type}: the native engine implements ORacle\'s mlp2x_gelu')
    raise ValueError(f'Unknown projector type: {projector_type}')


class ImageEmbeddingPooler:
    """2-layer BERT over the concatenated view tokens, first 576 outputs kept (orc_pool_views).
    forward(embeddings (B,S,1024), attention_mask (B,S) bool) as in the reference; the mask must mark a prefix
    of whole views per row (what pad_embeddings produces)."""
    embedding_dim = 1024

    def __init__(self, engine):
        self.engine = engine

    def forward(self, embeddings, attention_mask):
        b, s, d = embeddings.shape
        patches = self.engine.dims.clip_patches
        valid = attention_mask.sum(dim=1).tolist()
        views = [int(v) // patches for v in valid]
        if any(int(v) % patches or int(v) == 0 for v in valid):
            raise ValueError('attention_mask must cover whole views')
        vmax = s // patches
        if all(v == vmax for v in views):
            feats = embeddings.reshape(b * vmax, patches, d)
        else:
            import torch
            feats = torch.cat([embeddings[i, :views[i] * patches].reshape(views[i], patches, d) for i in range(b)])
        return self.engine.pool_views(feats.contiguous(), views).to(embeddings.dtype)

    __call__ = forward

    def pool_list(self, feats_per_row):
        """Ragged fast path: list of (V_b,576,1024) -> (B,576,1024) without materialising the padded tensor."""
        import torch
        views = [int(f.shape[0]) for f in feats_per_row]
        return self.engine.pool_views(torch.cat(list(feats_per_row)).contiguous(), views)


def build_image_pooler(config, engine=None):
    return ImageEmbeddingPooler(engine)
