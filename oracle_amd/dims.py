"""Model dimensions of the ORacle multi-view LLaVA hot path.

Full-size values follow SURVEY.md §8(a): CLIP ViT-L/14-336 (24 layers, d=1024, 16 heads x 64,
MLP 4096, 577 tokens), the learned multi-view pooler (2-layer BERT, d=1024, 8 heads x 128,
reference LLaVA/llava/model/multimodal_projector/builder.py:56-72), the mlp2x_gelu projector
(builder.py:41-48) and Vicuna-7B-v1.5 (32 layers, d=4096, 32 heads x 128, MLP 11008, vocab 32000).
Reduced sizes keep every fixed quantity the reference hard-codes (CLIP width 1024, 24x24 patches,
pooler shape) and only shrink depths and the LLaMA width so the CPU oracle finishes in seconds.
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class Dims:
    # LLaMA
    vocab: int = 32000
    d_model: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    d_ff: int = 11008
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    # CLIP vision tower (total depth; the path stops after layer clip_layers-1, select_layer=-2)
    clip_layers: int = 24
    clip_d: int = 1024
    clip_heads: int = 16
    clip_ff: int = 4096
    clip_image: int = 336
    clip_patch: int = 14
    clip_eps: float = 1e-5
    # learned multi-view pooler (fixed by the reference)
    pool_layers: int = 2
    pool_heads: int = 8
    pool_ff: int = 4096
    pool_maxpos: int = 2304
    pool_eps: float = 1e-12

    @property
    def head_dim(self):
        return self.d_model // self.n_heads

    @property
    def clip_grid(self):
        return self.clip_image // self.clip_patch

    @property
    def clip_patches(self):
        return self.clip_grid ** 2

    @property
    def clip_tokens(self):
        return self.clip_patches + 1

    def to_dict(self):
        return asdict(self)


FULL = Dims()

# Reduced model used by parity tests: LLaMA narrowed (4 heads x 128) and 2 layers deep, CLIP 3 layers.
TINY = Dims(vocab=32000, d_model=512, n_layers=2, n_heads=4, d_ff=1408, clip_layers=3)

# One full-width LLaMA layer + 2 CLIP layers: per-layer parity at the BASELINE widths.
WIDE1 = Dims(n_layers=1, clip_layers=2)
