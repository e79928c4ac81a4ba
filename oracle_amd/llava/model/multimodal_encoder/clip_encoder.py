"""CLIPVisionTower with the reference's surface (LLaVA/llava/model/multimodal_encoder/clip_encoder.py:7-91);
the 24-layer ViT itself runs in the native engine (orc_vision_encode / orc_vision_global)."""
import types

import torch

from ...mm_utils import ClipImageProcessor


class CLIPVisionTower:
    def __init__(self, vision_tower, args, delay_load=False, engine=None):
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, 'mm_vision_select_feature', 'patch')
        self.engine = engine
        d = engine.dims
        self.cfg_only = types.SimpleNamespace(hidden_size=d.clip_d, image_size=d.clip_image, patch_size=d.clip_patch,
                                              num_hidden_layers=d.clip_layers, num_attention_heads=d.clip_heads,
                                              intermediate_size=d.clip_ff)
        if not delay_load:
            self.load_model()

    def load_model(self):
        # weights already live in the engine (orc_engine_load_weight); only the processor is host state
        self.image_processor = ClipImageProcessor(size=self.cfg_only.image_size)
        self.is_loaded = True

    def _check(self):
        if self.select_feature not in ('patch', 'cls_patch'):
            raise ValueError(f'Unexpected select feature: {self.select_feature}')
        if self.select_layer != -2 or self.select_feature != 'patch':
            raise NotImplementedError('the native tower implements ORacle\'s setting: select_layer=-2, select_feature="patch"')

    def _encode(self, images):
        x = images.to(device=self.device, dtype=torch.bfloat16).contiguous()
        return self.engine.vision_encode(x)

    def forward(self, images):
        self._check()
        if type(images) is list:
            return [self._encode(im.unsqueeze(0)).to(im.dtype) for im in images]
        return self._encode(images).to(images.dtype)

    __call__ = forward

    def global_forward(self, images):
        """pooler_output = post_layernorm(CLS of the last layer) -- the visual-descriptor side path."""
        if type(images) is list:
            return [self.engine.vision_global(im.unsqueeze(0).to(device=self.device, dtype=torch.bfloat16).contiguous()) for im in images]
        return self.engine.vision_global(images.to(device=self.device, dtype=torch.bfloat16).contiguous())

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return torch.bfloat16

    @property
    def device(self):
        return self.engine.device

    @property
    def config(self):
        return self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2
