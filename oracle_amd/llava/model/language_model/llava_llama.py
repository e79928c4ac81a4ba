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
                 vis_descriptor_embs=None, attention_mask=None, chunk=16, limits=None, **kwargs):
        """Greedy search with the semantics of transformers' generate(do_sample=False) (A.6): returns
        LongTensor (B, L+n) = input_ids (placeholders intact) ++ new tokens; finished rows are filled with pad.
        Decode steps run back to back on the device in chunks of `chunk` steps (the host looks at the unfinished count
        between chunks: at most chunk - 1 steps run past the stop, with finished rows frozen); stopping criteria are then
        replayed on the host over the produced prefix, which gives the same stop step as checking after every token
        because greedy tokens do not depend on how far the device ran ahead. limits: optional per-row length stops
        (synthetic answer-length workloads; a row counts as finished after that many tokens)."""
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
        if limits is not None:
            self.engine.set_row_limits([min(int(x), max_new_tokens) for x in limits])
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
        if limits is not None:
            finished_at = torch.minimum(finished_at, torch.as_tensor([int(x) for x in limits], device=toks.device))
        stop = min(int(finished_at.max()), n)
        if stopping_criteria:
            for s in range(1, stop):
                seq = torch.cat([input_ids, toks[:, :s]], dim=1)
                if any(bool(c(seq, None)) for c in stopping_criteria):
                    stop = s
                    break
        return torch.cat([input_ids, toks[:, :stop].to(input_ids.dtype)], dim=1)

    # ------------------------------------------------------------------ continuous batching
    @torch.no_grad()
    def generate_stream(self, batches, max_new_tokens=20, n_rows=None, chunk=8, do_sample=False, use_cache=True,
                        stopping_criteria=None):
        """generate() over a stream of batches with continuous batching. The reference decodes batch
        after batch (scene_graph_prediction_model_oracle.py:251-260) and every batch runs as long as its longest answer;
        frames are independent, so here a decode row whose frame has finished (EOS / length stop) continues with the next
        frame. The engine has max_batch KV-cache slots and decodes n_rows rows (default: half the slots, at most 32): the
        next incoming batch is prefilled AHEAD, as one left-padded batch (same kernels and shapes as generate()), into the
        free slots while the current frames decode, and a finished row is simply re-attached to a waiting slot. Vision,
        pooling, projection and the splice run per incoming batch, exactly as in generate().

        batches: iterable of dicts {'input_ids': (B, L) LongTensor, 'images': ..., 'vis_descriptor_embs': ... or None,
        'limits': optional per-row length stops (synthetic workloads), 'stopping_criteria': optional, per batch}. Yields,
        in input order, what generate() returns for each batch: (B, L + n) = input_ids ++ new tokens, rows that finished
        early filled with pad, n = the longest row of that batch. A row's greedy tokens do not depend on its neighbours
        (the decode GEMVs are bit-invariant to the batch), so the results equal generate() batch by batch."""
        if do_sample:
            raise NotImplementedError('only greedy decoding (do_sample=False) is implemented, as ORacle uses')
        eng = self.engine
        if max_new_tokens > eng.max_new_tokens:
            raise ValueError(f'max_new_tokens {max_new_tokens} exceeds the engine capacity {eng.max_new_tokens}')
        n_slots = eng.max_batch
        n_rows = max(1, min(32, n_slots // 2)) if n_rows is None else n_rows
        if not 1 <= n_rows <= min(32, n_slots):
            raise ValueError(f'n_rows {n_rows} outside 1..{n_slots} (the engine\'s KV-cache slots)')
        pad, eos = self.config.pad_token_id, self.config.eos_token_id
        it = iter(batches)
        info = {}                       # batch index -> {'ids', 'rows': {row: tokens}, 'n', 'stopping'}
        queue = []                      # prepared frames waiting for a slot: (batch index, row, embeds (T, D), length stop)
        free = list(range(n_slots))     # KV-cache slots holding nothing
        waiting = []                    # prefilled, not yet decoding: (slot, batch index, row)
        rows = [None] * n_rows          # decoding: (slot, batch index, row) per decode row
        next_in, next_out, exhausted = 0, 0, False
        eng.stream_begin(n_rows)

        def prepare():
            nonlocal next_in, exhausted
            try:
                item = next(it)
            except StopIteration:
                exhausted = True
                return
            ids = item['input_ids']
            images, vis = item.get('images'), item.get('vis_descriptor_embs')
            limits = item.get('limits')
            has_pad = pad is not None and bool((ids == pad).any()) and pad != eos
            am = (ids != pad) if has_pad else torch.ones_like(ids, dtype=torch.bool)
            inp = self.prepare_inputs_for_generation(ids, attention_mask=am, images=images, vis_descriptor_embs=vis, use_cache=use_cache)
            _, _, mask, _, embeds, _ = self.prepare_inputs_labels_for_multimodal(inp['input_ids'], inp['position_ids'],
                                                                                 inp['attention_mask'], None, None, images, vis)
            if embeds is None:          # text-only prompt
                embeds, mask = self.get_model().embed_tokens(ids), am
            embeds = embeds.to(torch.bfloat16).contiguous()
            m = mask.bool().cpu()
            self._left_pad(m)           # refuses right-padded batches, as generate() does
            counts = m.sum(-1).tolist()
            t = embeds.shape[1]
            for b in range(ids.shape[0]):
                lim = max_new_tokens if limits is None else min(int(limits[b]), max_new_tokens)
                if t + lim > eng.max_seq:      # t: the row may be prefilled at its batch's padded length
                    if t >= eng.max_seq:
                        raise ValueError(f'prompt ({t}) exceeds the KV capacity {eng.max_seq}')
                    warnings.warn(f'prompt ({t}) + max_new_tokens ({lim}) exceeds the KV capacity {eng.max_seq}: '
                                  f'generating at most {eng.max_seq - t} tokens')
                    lim = eng.max_seq - t
                queue.append((next_in, b, embeds[b, t - counts[b]:], lim))
            info[next_in] = {'ids': ids, 'rows': {}, 'n': ids.shape[0], 'stopping': item.get('stopping_criteria', stopping_criteria)}
            next_in += 1

        def prefill(k):
            """the first k queued frames, as one left-padded batch, into k free slots"""
            frames = [queue.pop(0) for _ in range(k)]
            slots = [free.pop(0) for _ in range(k)]
            lens = [f[2].shape[0] for f in frames]
            t = max(lens)
            emb = torch.zeros(k, t, frames[0][2].shape[1], dtype=torch.bfloat16, device=frames[0][2].device)
            pos = torch.zeros(k, t, dtype=torch.int32)
            for i, f in enumerate(frames):
                emb[i, t - lens[i]:] = f[2]
                pos[i, t - lens[i]:] = torch.arange(lens[i], dtype=torch.int32)
            lims = [min(f[3], eng.max_seq - t) for f in frames]
            if any(l_ < f[3] for l_, f in zip(lims, frames)):
                # frames of different incoming batches share this prefill: the padded length t can exceed the frame's own
                # batch length, and the KV capacity then cuts its answer shorter than generate() would (which warns too)
                warnings.warn(f'stream prefill padded to {t} tokens: length stops lowered to the KV capacity {eng.max_seq} '
                              f'({[f[3] for f in frames]} -> {lims})')
            eng.prefill_slots(slots, emb, pos, [t - n for n in lens], lims)
            waiting.extend((s_, f[0], f[1]) for s_, f in zip(slots, frames))

        def schedule():
            """attach waiting slots to idle rows; prefill ahead when a batch worth of slots is free, or at once when a row
            would otherwise idle"""
            while True:
                for r in range(n_rows):
                    if rows[r] is None and waiting:
                        rows[r] = waiting.pop(0)
                        eng.stream_attach(r, rows[r][0])
                if not queue and not exhausted:
                    prepare()
                k = min(len(free), len(queue), n_rows)
                starving = any(x is None for x in rows) and not waiting
                if k == 0 or (not starving and k < min(n_rows, len(queue))):
                    return
                prefill(k)

        def finished_batches():
            nonlocal next_out
            while next_out in info and len(info[next_out]['rows']) == info[next_out]['n']:
                rec = info.pop(next_out)
                ids = rec['ids']
                n = max(len(t) for t in rec['rows'].values())
                new = torch.full((ids.shape[0], n), pad if pad is not None else 0, dtype=ids.dtype)
                for b, t in rec['rows'].items():
                    new[b, :len(t)] = torch.as_tensor(t, dtype=ids.dtype)
                out = torch.cat([ids, new.to(ids.device)], dim=1)
                if rec['stopping']:        # replayed over the finished batch exactly as generate() does
                    for s_ in range(1, n):
                        if any(bool(c(out[:, :ids.shape[1] + s_], None)) for c in rec['stopping']):
                            out = out[:, :ids.shape[1] + s_]
                            break
                next_out += 1
                yield out

        while True:
            schedule()
            if all(x is None for x in rows):
                break
            eng.decode_steps(chunk)
            if len(queue) < n_rows and not exhausted:      # host work of the next batch overlaps the decode chunk
                prepare()
            toks, steps, unf = eng.read_slots()
            for r in range(n_rows):
                if rows[r] is not None and not unf[rows[r][0]]:
                    slot, bi, row = rows[r]
                    info[bi]['rows'][row] = toks[slot, :steps[slot]].tolist()
                    eng.stream_attach(r, -1)
                    free.append(slot)
                    rows[r] = None
            yield from finished_batches()
        yield from finished_batches()
