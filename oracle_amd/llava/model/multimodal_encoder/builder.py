"""build_vision_tower (reference LLaVA/llava/model/multimodal_encoder/builder.py:5-12)."""
import os

from .clip_encoder import CLIPVisionTower


def build_vision_tower(vision_tower_cfg, engine=None, **kwargs):
    vision_tower = getattr(vision_tower_cfg, 'mm_vision_tower', getattr(vision_tower_cfg, 'vision_tower', None))
    known = vision_tower is not None and (os.path.exists(str(vision_tower)) or str(vision_tower).startswith(('openai', 'laion', 'synthetic')))
    if known:
        return CLIPVisionTower(vision_tower, args=vision_tower_cfg, engine=engine, **kwargs)
    raise ValueError(f'Unknown vision tower: {vision_tower}')
