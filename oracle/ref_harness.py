"""TEST INFRASTRUCTURE ONLY -- imports the real reference (read-only, /root/reference) to pin the oracle.

Used in the build container to (i) validate oracle/cpu_ref.py against the reference's own glue and
(ii) emit the golden fixtures under tests/golden/ (oracle/gen_golden.py). /root/reference does not
exist on the GPU box, so nothing in `-m gpu` tests, smoke() or bench.py may import this module.

What runs here is: reference glue files (LLaVA/llava/{constants,conversation,mm_utils}.py,
llava/model/llava_arch.py, multimodal_encoder/{builder,clip_encoder}.py,
multimodal_projector/builder.py, language_model/llava_llama.py) on top of the container's
transformers 5.15 with eager attention, CPU. The reference pins transformers==4.31.0
(LLaVA/pyproject.toml:17), absent offline; the import shims follow SURVEY.md Appendix B:
package __init__s skipped (they import MPT), AutoConfig/AutoModel registration made a no-op,
and a DynamicCache subclass restoring the tuple indexing llava_arch.py:194 relies on.
"""
import importlib
import json
import os
import sys
import tempfile
import types

import torch

REF_ROOT = '/root/reference'
REF_LLAVA = os.path.join(REF_ROOT, 'LLaVA')

_loaded = None


def reference_available():
    return os.path.isdir(os.path.join(REF_LLAVA, 'llava'))


def load_reference():
    """Import the reference's hot-path modules; returns a namespace of modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError('reference tree not present')
    os.environ.setdefault('HF_HUB_OFFLINE', '1')
    os.environ.setdefault('TRANSFORMERS_OFFLINE', '1')
    if REF_LLAVA not in sys.path:
        sys.path.insert(0, REF_LLAVA)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for name, rel in [('llava', 'llava'), ('llava.model', 'llava/model'),
                      ('llava.model.language_model', 'llava/model/language_model')]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_LLAVA, rel)]
        sys.modules[name] = m
    from transformers import AutoConfig, AutoModelForCausalLM
    AutoConfig.register = staticmethod(lambda *a, **k: None)
    AutoModelForCausalLM.register = staticmethod(lambda *a, **k: None)
    ns = types.SimpleNamespace()
    ns.llava_llama = importlib.import_module('llava.model.language_model.llava_llama')
    ns.llava_arch = importlib.import_module('llava.model.llava_arch')
    ns.mm_utils = importlib.import_module('llava.mm_utils')
    ns.constants = importlib.import_module('llava.constants')
    ns.conversation = importlib.import_module('llava.conversation')
    ns.clip_encoder = importlib.import_module('llava.model.multimodal_encoder.clip_encoder')
    ns.projector = importlib.import_module('llava.model.multimodal_projector.builder')
    # pure-python helpers of the scene-graph side (no heavy deps)
    for name, rel in [('scene_graph_prediction', 'scene_graph_prediction'),
                      ('scene_graph_prediction.llava_helpers', 'scene_graph_prediction/llava_helpers')]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_ROOT, rel)]
        sys.modules[name] = m
    ns.converters = importlib.import_module('scene_graph_prediction.llava_helpers.scene_graph_converters')
    _loaded = ns
    return ns


def legacy_cache(cfg):
    from transformers import DynamicCache

    class LegacyIndexCache(DynamicCache):
        def __getitem__(self, i):
            layer = self.layers[i]
            return (layer.keys, layer.values)

    return LegacyIndexCache(config=cfg)


def _save_local_clip(dims, out_dir):
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(hidden_size=dims.clip_d, intermediate_size=dims.clip_ff,
                           num_hidden_layers=dims.clip_layers, num_attention_heads=dims.clip_heads,
                           image_size=dims.clip_image, patch_size=dims.clip_patch,
                           layer_norm_eps=dims.clip_eps, hidden_act='quick_gelu')
    CLIPVisionModel(cfg).save_pretrained(out_dir)
    CLIPImageProcessor(size={'shortest_edge': dims.clip_image},
                       crop_size={'height': dims.clip_image, 'width': dims.clip_image}).save_pretrained(out_dir)
    return out_dir


def build_reference_model(dims, state, mv_type='learned', dtype=torch.float32, padding_side='left',
                          max_length=None):
    """Instantiate the reference LlavaLlamaForCausalLM with `state` (names as in oracle_amd/synth.py)."""
    ns = load_reference()
    tmp = tempfile.mkdtemp(prefix='ref_clip_')
    _save_local_clip(dims, tmp)
    cfg = ns.llava_llama.LlavaConfig(
        vocab_size=dims.vocab, hidden_size=dims.d_model, intermediate_size=dims.d_ff,
        num_hidden_layers=dims.n_layers, num_attention_heads=dims.n_heads, num_key_value_heads=dims.n_heads,
        rms_norm_eps=dims.rms_eps, rope_theta=dims.rope_theta, max_position_embeddings=4096,
        mm_vision_tower=tmp, mm_projector_type='mlp2x_gelu', mm_hidden_size=dims.clip_d,
        mm_vision_select_layer=-2, mm_vision_select_feature='patch', mv_type=mv_type, pretraining_tp=1,
        image_aspect_ratio='pad', tokenizer_padding_side=padding_side, pad_token_id=0, bos_token_id=1,
        eos_token_id=2, attn_implementation='eager', tie_word_embeddings=False)
    if max_length is not None:
        cfg.tokenizer_model_max_length = max_length
    model = ns.llava_llama.LlavaLlamaForCausalLM(cfg).eval()
    model.get_vision_tower().load_model()
    # eager numerics on the sub-models too
    model.get_vision_tower().vision_tower.config._attn_implementation = 'eager'
    model.get_model().image_pooler.bert.config._attn_implementation = 'eager'
    sd = {}
    for k, v in state.items():
        # transformers 5.15's CLIPVisionModel dropped the `vision_model.` level
        k2 = k.replace('vision_tower.vision_tower.vision_model.', 'vision_tower.vision_tower.')
        sd[k2] = v.to(dtype)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if 'position_ids' not in m and 'token_type_ids' not in m and 'inv_freq' not in m]
    if missing or unexpected:
        raise RuntimeError(f'state mismatch: missing={missing[:8]} unexpected={unexpected[:8]}')
    # from_pretrained(torch_dtype=bf16) (reference builder.py:43) leaves the non-persistent rotary
    # inv_freq buffer in fp32; a blanket .to(bf16) would round it, so put the fp32 values back.
    rot = model.get_model().rotary_emb
    inv = rot.inv_freq.detach().clone().float()
    model = model.to(dtype)
    rot.inv_freq = inv
    if hasattr(rot, 'original_inv_freq'):
        rot.original_inv_freq = inv.clone()
    return model, cfg


def reference_generate(model, cfg, input_ids, images, max_new_tokens, vis_descriptor_embs=None,
                       stopping_criteria=None):
    """The call at scene_graph_prediction_model_oracle.py:135-143."""
    kw = {}
    if stopping_criteria is not None:
        kw['stopping_criteria'] = stopping_criteria
    with torch.inference_mode():
        return model.generate(input_ids, images=images, do_sample=False, use_cache=True,
                              max_new_tokens=max_new_tokens, vis_descriptor_embs=vis_descriptor_embs,
                              past_key_values=legacy_cache(cfg), **kw)


if __name__ == '__main__':
    ns = load_reference()
    print(json.dumps({k: getattr(v, '__file__', None) for k, v in vars(ns).items()}, indent=1))
