"""LlavaConfig / LlavaLlamaModel / LlavaLlamaForCausalLM with the reference's surface
(LLaVA/llava/model/language_model/llava_llama.py:27-112): `forward`, `prepare_inputs_for_generation` and the
greedy `generate()` the ORacle wrapper calls (scene_graph_prediction_model_oracle.py:135-143). LLaMA prefill,
the KV cache, KV-cached decode and greedy sampling all run inside the native engine; this file reproduces
transformers' greedy-search bookkeeping around it."""
import types
import warnings

import torch

from ....dims import Dims
from ..llava_arch import LlavaMetaForCausalLM, LlavaMetaModel


class LlavaConfig:
    model_type = 'llava'

    def __init__(self, dims: Dims = None, **kw):
        d = dims or Dims()
        self.dims = d
        self.vocab_size, self.hidden_size, self.intermediate_size = d.vocab, d.d_model, d.d_ff
        self.num_hidden_layers, self.num_attention_heads = d.n_layers, d.n_heads
        self.rms_norm_eps, self.rope_theta = d.rms_eps, d.rope_theta
        self.mm_vision_tower = 'synthetic-clip-vit-large-patch14-336'
        self.mm_projector_type = 'mlp2x_gelu'
        self.mm_hidden_size = d.clip_d
        self.mm_vision_select_layer = -2
        self.mm_vision_select_feature = 'patch'
        self.mm_use_im_start_end = False
        self.mm_use_im_patch_token = False
        self.image_aspect_ratio = 'pad'
        self.mv_type = 'learned'
        self.tokenizer_padding_side = 'right'
        self.pad_token_id, self.bos_token_id, self.eos_token_id = 0, 1, 2
        self.pretraining_tp = 1
        for k, v in kw.items():
            setattr(self, k, v)


class LlavaLlamaModel(LlavaMetaModel):
    config_class = LlavaConfig


class _EngineCache:
    """Stand-in for past_key_values: the KV cache lives in the engine. Indexing gives objects with the `.shape`
    the decode-step branch of prepare_inputs_labels_for_multimodal reads (past_key_values[-1][-1].shape[-2])."""

    def __init__(self, model, length):
        self.model, self.length = model, length

    def __getitem__(self, i):
        d = self.model.config.dims
        kv = types.SimpleNamespace(shape=(self.model.engine.batch, d.n_heads, self.length, d.head_dim))
        return (kv, kv)


class LlavaLlamaForCausalLM(LlavaMetaForCausalLM):
    config_class = LlavaConfig

    def __init__(self, config, engine):
        self.config = config
        self.engine = engine
        self.model = LlavaLlamaModel(config, engine)
        self.vocab_size = config.vocab_size

    def get_model(self):
        return self.model

    @property
    def device(self):
        return self.engine.device

    @property
    def dtype(self):
        return torch.bfloat16

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ forward (step 0 or one decode step)
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None, vis_descriptor_embs=None):
        """Returns an object with `.logits` (B,1,V) fp32 of the LAST position (the reference computes all T
        positions and greedy search reads only the last), `.past_key_values` and `modified_labels`."""
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values, labels,
                                                          images, vis_descriptor_embs)
        if inputs_embeds is not None:
            b, t, _ = inputs_embeds.shape
            if attention_mask is None:
                attention_mask = torch.ones(b, t, dtype=torch.bool, device=inputs_embeds.device)
            left_pad = self._left_pad(attention_mask)
            if position_ids is None:
                position_ids = (attention_mask.long().cumsum(-1) - 1).clamp_min(0)
            logits = self.engine.prefill(inputs_embeds.to(torch.bfloat16).contiguous(),
                                         position_ids.to(device=self.device, dtype=torch.int32).contiguous(), left_pad,
                                         want_logits=True)
            self._cache_len = t
        else:
            if not isinstance(past_key_values, _EngineCache):
                raise ValueError('decode-step forward needs the past_key_values returned by the prefill forward')
            logits = self.engine.decode_step(input_ids.reshape(-1).to(self.device), want_logits=True)
            self._cache_len += 1
        out = types.SimpleNamespace(logits=logits[:, None, :], past_key_values=_EngineCache(self, self._cache_len),
                                    modified_labels=labels)
        return out

    __call__ = forward

    @staticmethod
    def _left_pad(mask):
        m = mask.bool().cpu()
        left = (~m).long().cumsum(-1)
        n_pad = (~m).sum(-1)
        # every padding slot must precede every real slot
        first_real = m.long().argmax(-1)
        ok = (first_real == n_pad) | (m.sum(-1) == 0)
        if not bool(ok.all()):
            raise NotImplementedError('the native LLaMA path takes left-padded (or unpadded) batches, as ORacle builds them '
                                      '(scene_graph_prediction_model_oracle.py:122-128)')
        del left
        return n_pad.tolist()

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        images = kwargs.pop('images', None)
        vis = kwargs.pop('vis_descriptor_embs', None)
        attention_mask = kwargs.get('attention_mask')
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        position_ids = None
        if attention_mask is not None:
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values is not None:
                position_ids = position_ids[:, -1:]
        out = {'input_ids': input_ids, 'past_key_values': past_key_values, 'attention_mask': attention_mask,
               'position_ids': position_ids, 'use_cache': kwargs.get('use_cache')}
        if images is not None:
            out['images'] = images
        if vis is not None:
            out['vis_descriptor_embs'] = vis
        return out

    # ------------------------------------------------------------------ greedy generate
    @torch.no_grad()
    def generate(self, input_ids, images=None, do_sample=False, use_cache=True, max_new_tokens=20, stopping_criteria=None,
                 vis_descriptor_embs=None, attention_mask=None, chunk=32, **kwargs):
        """Greedy search with the semantics of transformers' generate(do_sample=False) (A.6): returns
        LongTensor (B, L+n) = input_ids (placeholders intact) ++ new tokens; finished rows are filled with pad.
        Decode steps run back to back on the device in chunks of `chunk` steps; stopping criteria are then replayed
        on the host over the produced prefix, which gives the same stop step as checking after every token because
        greedy tokens do not depend on how far the device ran ahead."""
        if do_sample:
            raise NotImplementedError('only greedy decoding (do_sample=False) is implemented, as ORacle uses')
        if max_new_tokens > self.engine.max_new_tokens:
            raise ValueError(f'max_new_tokens {max_new_tokens} exceeds the engine capacity {self.engine.max_new_tokens}')
        pad, eos = self.config.pad_token_id, self.config.eos_token_id
        b, l = input_ids.shape
        if attention_mask is None:
            has_pad = pad is not None and bool((input_ids == pad).any()) and pad != eos
            attention_mask = (input_ids != pad) if has_pad else torch.ones_like(input_ids, dtype=torch.bool)
        inp = self.prepare_inputs_for_generation(input_ids, attention_mask=attention_mask, images=images,
                                                 vis_descriptor_embs=vis_descriptor_embs, use_cache=use_cache)
        _, pos, mask, _, embeds, _ = self.prepare_inputs_labels_for_multimodal(
            inp['input_ids'], inp['position_ids'], inp['attention_mask'], None, None, images, vis_descriptor_embs)
        if embeds is None:          # text-only prompt
            embeds = self.get_model().embed_tokens(input_ids)
            mask, pos = attention_mask, inp['position_ids']
        t = embeds.shape[1]
        if t >= self.engine.max_seq:
            raise ValueError(f'prompt ({t}) exceeds the KV capacity {self.engine.max_seq}')
        if t + max_new_tokens > self.engine.max_seq:
            # the engine's KV cache plays the part of transformers' max_length: generation stops at the capacity
            # (the reference would run on to the model's 4096 positions; size max_seq for that, see builder.py)
            warnings.warn(f'prompt ({t}) + max_new_tokens ({max_new_tokens}) exceeds the KV capacity {self.engine.max_seq}: '
                          f'generating at most {self.engine.max_seq - t} tokens')
            max_new_tokens = self.engine.max_seq - t
        self.engine.prefill(embeds.to(torch.bfloat16).contiguous(), pos.to(device=self.device, dtype=torch.int32).contiguous(),
                            self._left_pad(mask))
        while True:
            toks, done, n_unf, _ = self.engine.read_state()
            if n_unf == 0 or done >= max_new_tokens:
                break
            self.engine.decode_steps(min(chunk, max_new_tokens - done))
        toks = toks[:, :max_new_tokens].to(input_ids.device)
        # replay: stop after step s if every row has emitted EOS, the length cap is hit, or a user criterion fires
        n = toks.shape[1]
        is_eos = toks == eos
        finished_at = torch.where(is_eos.any(-1), is_eos.long().argmax(-1) + 1, torch.full((b,), n + 1, device=toks.device))
        stop = min(int(finished_at.max()), n)
        if stopping_criteria:
            for s in range(1, stop):
                seq = torch.cat([input_ids, toks[:, :s]], dim=1)
                if any(bool(c(seq, None)) for c in stopping_criteria):
                    stop = s
                    break
        return torch.cat([input_ids, toks[:, :stop].to(input_ids.dtype)], dim=1)
