"""Visual-descriptor embedding producer (reference synthetic_or_generation/embed_visual_prompts.py:46-48):
one d_model vector per crop = mm_projector(CLIP.pooler_output), where pooler_output = post_layernorm of the CLS token
after ALL clip layers (clip_encoder.py:53-64) -- the side path that, unlike the main path, runs the last CLIP layer.
The vectors are what `vis_descriptor_embs` carries into the splice (llava_arch.py:294-315)."""
import torch


def process_and_extract_features(img_batch, model, image_processor=None):
    """img_batch (n,3,S,S) (a leading singleton, as the reference's loader produces, is squeezed away) -> (n, d_model)."""
    x = img_batch.squeeze()
    if x.dim() == 3:
        x = x[None]
    x = x.to(device=model.device, dtype=torch.bfloat16).contiguous()
    pooled = model.get_vision_tower().global_forward(x)
    return model.model.mm_projector(pooled)
