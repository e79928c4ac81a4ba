"""OracleWrapper: the task harness around generate(), with the reference's constructor, `forward`, `validate`,
`infer`, `compute_optimal_human_indices` and `evaluate_predictions`
(scene_graph_prediction/scene_graph_helpers/model/scene_graph_prediction_model_oracle.py:21-398).

Per batch: load the configured camera views, CLIP-preprocess, build the Vicuna-v1 prompt (plain / symbolic /
visual-descriptor variants, optional temporal memory), tokenise with the image placeholder, left-pad, run the
native generate(), decode, map symbols back to names, parse triplets, and accumulate the relation labels that
sklearn scores per take."""
import itertools
import os
import re
from collections import defaultdict

import numpy as np
import torch
from PIL import Image

from ....llava.constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
from ....llava.conversation import SeparatorStyle, default_conversation
from ....llava.mm_utils import KeywordsStoppingCriteria, get_model_name_from_path, process_images, tokenizer_image_token
from ....llava.model.builder import load_pretrained_model
from ... import prompts
from ...llava_helpers.scene_graph_converters import llava_sg_to_surgery_sg, surgery_sg_to_memory_str, triplet_strings
from ..dataset.dataset_utils import map_scene_graph_name_to_vocab_idx, map_vocab_idx_to_scene_graph_name, reversed_role_synonyms

HUMAN_ROLES = ('head surgeon', 'assistant surgeon', 'circulator', 'nurse', 'anaesthetist')
_SYMBOL_RE = re.compile('|'.join(map(re.escape, prompts.SYMBOL_TO_NAME)))
MEMORY_CHAR_LIMIT = 5000


class OracleWrapper:
    def __init__(self, config, num_class, num_rel, weights_obj, weights_rel, relationNames, model_path,
                 model_base='liuhaotian/llava-v1.5-7b', load_8bit=False, load_4bit=False, mv_desc=False,
                 device_preprocess=True, stream=True, decode_workers=None, **engine_kw):
        self.config = config
        self.mconfig = config['MODEL']
        self.n_object_types = 6
        self.weights_obj, self.weights_rel = weights_obj, weights_rel
        self.relationNames = relationNames
        self.relation_names_lower_case = [r.lower() for r in relationNames]
        self.lr = float(config['LR']) if 'LR' in config else 0.0
        self.reset_metrics()
        self.model_name = get_model_name_from_path(model_path)
        if config.get('temporality') == 'PRED':
            engine_kw.setdefault('max_batch', 2)      # online temporal: one frame at a time (main.py:58-61), no spare KV slots
        self.tokenizer, self.model, self.image_processor, self.context_len = load_pretrained_model(
            model_path, model_base, self.model_name, load_8bit, load_4bit, **engine_kw)
        self.model.config.mv_type = self.mconfig['mv_type']
        self.model.config.tokenizer_padding_side = 'left'
        self.mv_desc = mv_desc
        self.device_preprocess, self._preproc = device_preprocess, None
        self.stream = stream          # validate(): continuous batching across dataloader batches (identical outputs)
        # camera images of the NEXT batches are opened + decoded by a thread pool while the GPU works on the current one
        # (the reference decodes 4-6 x 2048x1536 JPEGs per frame serially in the main process, ..._oracle.py:60-63; Pillow's
        # decoder releases the GIL). 0 = decode in the main thread at use, as the reference does.
        self.decode_workers = min(8, max(1, (os.cpu_count() or 2) // 2)) if decode_workers is None else int(decode_workers)
        self._pool = None
        self.last_new_tokens = {}       # scan_id -> token ids of the answer (what a multi-GPU run all-gathers: oracle_amd/parallel.py)
        self.temporal_online_prediction = config.get('temporality') == 'PRED'
        if self.temporal_online_prediction:
            print('Preparing temporality PRED')
            self.take_to_history = defaultdict(list)

    # ------------------------------------------------------------------ prompt
    def _is_symbolic(self):
        return '_symbolic' in self.model_name or '_synthetic' in self.model_name

    def _use_vis_desc(self):
        return bool(self.config.get('USE_VIS_DESC'))

    def _user_message(self, elem):
        if self._use_vis_desc():
            inp = prompts.visual_prompt(self.mv_desc)
        elif self._is_symbolic():
            inp = prompts.symbolic_prompt()
        else:
            inp = prompts.plain_prompt()
        image_tok = DEFAULT_IMAGE_TOKEN
        if self.model.config.mm_use_im_start_end:
            image_tok = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN
        inp = image_tok + '\n' + inp
        if 'temporality' in self.config:
            if self.config['temporality'] == 'GT':
                take_idx, timepoint_idx, _ = elem['scan_id'].split('_')
                memory_str = self.take_timepoint_to_memory_str[f'{int(take_idx)}_{int(timepoint_idx)}']
            elif self.config['temporality'] == 'PRED':
                timepoint_idx = int(elem['scan_id'].split('_')[1])
                log = llava_sg_to_surgery_sg(self.take_to_history[elem['take_idx']], entity_of_interest=None,
                                             IRRELEVANT_PREDS=['closeto', 'closeTo'])
                log = [entry for entry in log if entry[0] < timepoint_idx]
                memory_str = surgery_sg_to_memory_str(log, current_timepoint=timepoint_idx, TEMPORAL_STYLE='longshort')
            else:
                raise NotImplementedError()
            if len(memory_str) > MEMORY_CHAR_LIMIT:
                print(f'Warning: memory string is too long ({len(memory_str)} chars)')
                memory_str = '...' + memory_str[-MEMORY_CHAR_LIMIT:]
            inp = inp.replace(f'{DEFAULT_IMAGE_TOKEN}\n', f'{DEFAULT_IMAGE_TOKEN}\n<memory_start>: {memory_str}<memory_end>.\n')
        return inp

    def build_prompt(self, elem):
        conv = default_conversation.copy()
        conv.append_message(conv.roles[0], self._user_message(elem))
        conv.append_message(conv.roles[1], None)
        return conv, conv.get_prompt()

    @staticmethod
    def _open_rgb(path):
        return Image.open(path).convert('RGB')

    def _prefetched(self, dataloader, depth=2):
        """Yields the dataloader's batches; the camera views of the next `depth` batches are being decoded meanwhile."""
        if self.decode_workers <= 0:
            yield from dataloader
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.decode_workers, thread_name_prefix='orc-decode')
        it = iter(dataloader)
        ahead = []

        def push():
            try:
                batch = next(it)
            except StopIteration:
                return
            for elem in batch:
                elem['_views'] = [self._pool.submit(self._open_rgb, elem['image_paths'][cam - 1]) for cam in self.config['CAMERAS']]
            ahead.append(batch)
        for _ in range(depth):
            push()
        while ahead:
            batch = ahead.pop(0)
            push()
            yield batch

    def load_views(self, elem):
        futures = elem.pop('_views', None)
        views = [f.result() for f in futures] if futures is not None else \
            [self._open_rgb(elem['image_paths'][cam - 1]) for cam in self.config['CAMERAS']]
        if (self.device_preprocess and torch.device(self.model.device).type == 'cuda' and len({v.size for v in views}) == 1
                and getattr(self.model.config, 'image_aspect_ratio', None) == 'pad'):
            # pad + PIL-exact bicubic resize + normalisation on the device (oracle_amd/preprocess.py); same bits as
            # process_images() followed by the bf16 cast, same shapes ((1,V,3,S,S) for V > 1, (1,3,S,S) for one view)
            if self._preproc is None:
                from ....preprocess import DevicePreprocessor
                self._preproc = DevicePreprocessor(self.image_processor, self.model.device)
            t = self._preproc([np.asarray(v) for v in views])
            return t if len(views) == 1 else t[None]
        t = process_images(views, self.image_processor, self.model.config)
        if type(t) is list:
            return [x.to(self.model.device, dtype=torch.bfloat16) for x in t]
        return t.to(self.model.device, dtype=torch.bfloat16)

    # ------------------------------------------------------------------ forward
    def _prepare(self, batch):
        """Host side of forward() before generate(): views -> image tensor, prompts -> left-padded ids, stopping criterion."""
        images, prompt_texts, vis_desc = [], [], []
        conv = None
        for elem in batch:
            images.append(self.load_views(elem))
            if self._use_vis_desc():
                vis_desc.append([d.to(self.model.device, dtype=torch.bfloat16) for d in elem['vis_descriptor_embs']])
            conv, text = self.build_prompt(elem)
            prompt_texts.append(text)
        if len(batch) == 1:
            input_ids = tokenizer_image_token(prompt_texts[0], self.tokenizer, IMAGE_TOKEN_INDEX, return_tensors='pt')[None].to(self.model.device)
            image_tensor = images[0]
        else:
            rows = [tokenizer_image_token(p, self.tokenizer, return_tensors='pt') for p in prompt_texts]
            width = max(r.shape[0] for r in rows)
            input_ids = torch.full((len(rows), width), self.tokenizer.pad_token_id, dtype=torch.long)
            for i, r in enumerate(rows):
                input_ids[i, width - r.shape[0]:] = r          # left padding
            input_ids = input_ids.to(self.model.device)
            image_tensor = torch.cat(images)
        stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
        stopping = KeywordsStoppingCriteria([stop_str], self.tokenizer, input_ids)
        return {'input_ids': input_ids, 'images': image_tensor, 'stopping_criteria': [stopping],
                'vis_descriptor_embs': vis_desc if self._use_vis_desc() else None}

    def _finish(self, batch, input_ids, output_ids):
        """Host side of forward() after generate(): token ids -> strings (+ symbolic back-map)."""
        new = output_ids[:, input_ids.shape[1]:]
        rows = new.tolist()
        pad = self.tokenizer.pad_token_id
        for elem, ids in zip(batch, rows):     # what the multi-GPU path exchanges (oracle_amd/parallel.py): the answer's token ids
            n = len(ids)
            while n > 0 and ids[n - 1] == pad:
                n -= 1
            self.last_new_tokens[elem['scan_id']] = ids[:n]
        return self.outputs_from_token_ids(rows, single=len(batch) == 1)

    def outputs_from_token_ids(self, rows, single):
        """Token ids of the answers -> the strings forward() returns (reference ..._oracle.py:144-166): one-frame batches keep
        special tokens (`tokenizer.decode(ids).strip()`), larger ones drop them; symbolic back-map where the model uses symbols.
        Trailing pad of a finished row changes neither form for the parser (pad is a special token / stripped with the tags)."""
        if single:
            outputs = [self.tokenizer.decode(torch.as_tensor(r)).strip() for r in rows]
        else:
            outputs = self.tokenizer.batch_decode([list(r) for r in rows], skip_special_tokens=True)
        if self._is_symbolic() or self.config['USE_VIS_DESC']:
            outputs = [self.symbols_to_names(o) for o in outputs]
        return outputs

    def forward(self, batch):
        p = self._prepare(batch)
        with torch.inference_mode():
            output_ids = self.model.generate(p['input_ids'], images=p['images'], do_sample=False, use_cache=True, max_new_tokens=300,
                                             stopping_criteria=p['stopping_criteria'], vis_descriptor_embs=p['vis_descriptor_embs'])
        return self._finish(batch, p['input_ids'], output_ids)

    def forward_stream(self, dataloader):
        """forward() over a whole dataloader with slot refill (LlavaLlamaForCausalLM.generate_stream): a row whose frame has
        finished is given the next frame instead of idling until the longest answer of its batch ends. Yields
        (batch, outputs) in dataloader order with the outputs forward(batch) returns. Not for the online-temporal variant
        (frame t needs the predictions of frames < t)."""
        assert not self.temporal_online_prediction
        pending = []

        def items():
            for batch in dataloader:
                p = self._prepare(batch)
                pending.append((batch, p['input_ids']))
                yield p
        with torch.inference_mode():
            for output_ids in self.model.generate_stream(items(), max_new_tokens=300):
                batch, input_ids = pending.pop(0)
                yield batch, self._finish(batch, input_ids, output_ids)

    @staticmethod
    def symbols_to_names(output):
        """A..K / alpha..xi -> entity / predicate names in ONE regex pass (so substituted text is never re-substituted),
        re-wrapped in <SG> tags (reference ..._oracle.py:149-165)."""
        body = output.replace('<SG>', '').replace('</SG>', '')
        return '<SG>' + _SYMBOL_RE.sub(lambda m: prompts.SYMBOL_TO_NAME[m.group(0)], body) + '</SG>'

    # ------------------------------------------------------------------ metrics plumbing
    def reset_metrics(self, split=None):
        if split in (None, 'train'):
            self.train_take_rel_preds, self.train_take_rel_gts = defaultdict(list), defaultdict(list)
        if split in (None, 'val'):
            self.val_take_rel_preds, self.val_take_rel_gts = defaultdict(list), defaultdict(list)

    def infer(self, dataloader):
        return self.validate(dataloader, return_raw_predictions=True)

    @torch.no_grad()
    def compute_optimal_human_indices(self, pred, gt):
        """Permutation of the human indices 7..11 occurring in pred/gt that maximises relation accuracy (first best wins)."""
        if len(gt) == 0 or len(pred) == 0:
            return {}
        gt = torch.as_tensor(gt)
        gt_h = set(gt[(gt >= 7) & (gt <= 11)].tolist())
        flat = torch.tensor(pred).flatten()
        pred_h = set(flat[(flat >= 7) & (flat <= 11)].tolist())
        humans = sorted(gt_h | pred_h)
        gt_list = gt.tolist()
        best, best_acc = None, -1
        for perm in itertools.permutations(humans):
            m = dict(zip(humans, perm))
            hits = 0
            for g_sub, g_rel, g_obj in gt_list:
                found = -1
                for p_sub, p_rel, p_obj in pred:
                    if m.get(p_sub, p_sub) == g_sub and m.get(p_obj, p_obj) == g_obj:
                        found = p_rel
                        break
                hits += int(found == g_rel)
            acc = float(hits) / float(len(gt_list))
            if acc > best_acc:
                best_acc, best = acc, m
        return best

    @staticmethod
    def parse_answer(output):
        """-> (raw_triplets [(sub,pred,obj)] as generated, triplets with role synonyms resolved, human roles seen)."""
        output = re.sub(r'""".*?"""', '', output, flags=re.DOTALL)
        raw, resolved, humans = [], [], set()
        for frag in triplet_strings(output):
            frag = frag.replace('.', '').replace('</s>', '').replace('<s>', '').strip()
            if frag == '':
                continue
            fields = [f.strip() for f in frag.split(',')]
            if len(fields) != 3:
                continue
            sub, obj, pred = fields
            raw.append((sub, pred, obj))
            sub = reversed_role_synonyms.get(sub, sub)
            obj = reversed_role_synonyms.get(obj, obj)
            humans.update(x for x in (sub, obj) if x in HUMAN_ROLES)
            resolved.append((sub, pred, obj))
        return raw, resolved, humans

    def validate(self, dataloader, limit_val_batches=None, logging_information=None, return_raw_predictions=False, report=True):
        """report=False (multi-rank runs, oracle_amd/parallel.py): skip the per-shard classification reports; the caller
        merges `last_frame_rel` of all ranks and calls `report_merged` once, so the printed per-take and overall reports
        and the macro-F1 are those of the whole split, as the reference's single process prints them."""
        take_rel_preds, take_rel_gts = defaultdict(list), defaultdict(list)
        frame_rel = {}
        scan_id_to_raw_predictions = {}
        limit = None
        if isinstance(limit_val_batches, int):
            limit = limit_val_batches
        elif isinstance(limit_val_batches, float):
            limit = int(limit_val_batches * len(dataloader))
        none_id = self.relation_names_lower_case.index('none')

        def limited():
            left = limit
            for batch in self._prefetched(dataloader):
                if left is not None:
                    if left <= 0:
                        return
                    left -= 1
                yield batch

        def run():
            if self.stream and not self.temporal_online_prediction and hasattr(self.model, 'generate_stream'):
                yield from self.forward_stream(limited())
            else:
                for batch in limited():
                    yield batch, self.forward(batch)

        for batch, outputs in run():
            assert len(batch) == 1 or not self.temporal_online_prediction
            for elem, output in zip(batch, outputs):
                self._score_frame(elem, output, take_rel_preds, take_rel_gts, frame_rel, scan_id_to_raw_predictions, none_id)
        return self._close_validation(take_rel_preds, take_rel_gts, frame_rel, scan_id_to_raw_predictions, report, logging_information,
                                      return_raw_predictions)

    def _score_frame(self, elem, output, take_rel_preds, take_rel_gts, frame_rel, scan_id_to_raw_predictions, none_id, keep_history=True):
        """One answer string -> raw triplets, relation labels against the frame's ground truth (reference ..._oracle.py:261-342)."""
        timepoint = int(elem['scan_id'].split('_')[1])
        take_idx = elem['take_idx']
        raw, triplets, humans = self.parse_answer(output)
        scan_id_to_raw_predictions[elem['scan_id']] = raw
        if self.temporal_online_prediction and keep_history:
            self.take_to_history[take_idx].append({'timepoint_idx': timepoint, 'scene_graph': raw})
        role_to_human = {role: f'human_{i}' for i, role in enumerate(sorted(humans))}
        rel_preds = []
        for sub, pred, obj in triplets:
            try:
                rel_preds.append((map_scene_graph_name_to_vocab_idx(role_to_human.get(sub, sub).replace(' ', '_')),
                                  map_scene_graph_name_to_vocab_idx(pred),
                                  map_scene_graph_name_to_vocab_idx(role_to_human.get(obj, obj).replace(' ', '_'))))
            except Exception as e:   # unknown entity / predicate names are dropped, as in the reference
                print(e)
        # the reference maps (sub, pred, obj) -> indices in the order (sub, pred, obj); keep that tuple order
        rel_labels = torch.tensor(elem['relations_tokenized'])
        hmap = self.compute_optimal_human_indices(rel_preds, rel_labels)
        rel_preds = [(hmap.get(s, s), r, hmap.get(o, o)) for s, r, o in rel_preds]
        labels = rel_labels.tolist()
        gt_objects = sorted({x for s, _, o in labels for x in (s, o)}) if labels else []
        n0 = len(take_rel_gts[take_idx])
        for a in gt_objects:
            for b in gt_objects:
                if a == b:
                    continue
                gt_rel = next((r for s, r, o in labels if s == a and o == b), None)
                take_rel_gts[take_idx].append(none_id if gt_rel is None else
                                              self.relation_names_lower_case.index(map_vocab_idx_to_scene_graph_name(gt_rel)))
                pr = next((r for s, r, o in rel_preds if s == a and o == b), None)
                if pr is None:
                    take_rel_preds[take_idx].append(none_id)
                else:
                    try:
                        take_rel_preds[take_idx].append(self.relation_names_lower_case.index(map_vocab_idx_to_scene_graph_name(pr)))
                    except Exception:   # a nonsense relation counts as 'none'
                        take_rel_preds[take_idx].append(none_id)
        frame_rel[elem['scan_id']] = (take_idx, take_rel_preds[take_idx][n0:], take_rel_gts[take_idx][n0:])

    def _close_validation(self, take_rel_preds, take_rel_gts, frame_rel, scan_id_to_raw_predictions, report, logging_information,
                          return_raw_predictions):
        self.last_frame_rel = frame_rel
        self.val_take_rel_preds, self.val_take_rel_gts = take_rel_preds, take_rel_gts
        self.last_take_rel = (dict(take_rel_preds), dict(take_rel_gts))
        if report:
            self.last_macro_f1 = self.evaluate_predictions(None, 'val', logging_information=logging_information)
        self.reset_metrics(split='val')
        if return_raw_predictions:
            return scan_id_to_raw_predictions

    def score_token_ids(self, elems, token_rows, single_flags, logging_information=None):
        """Rank 0 of a multi-GPU run: the whole split's answers arrive as token ids (one RCCL all-gather, oracle_amd/parallel.py);
        detokenise, back-map, parse and score them in dataset order and print the reference's per-take and overall reports --
        what validate() does in one process. elems: the dataset items in dataset order; token_rows: their answers' token ids;
        single_flags[i]: the frame was generated in a one-frame batch (its string keeps special tokens, as forward() returns it).
        Returns {scan_id: raw triplets}."""
        take_rel_preds, take_rel_gts = defaultdict(list), defaultdict(list)
        frame_rel, raw_predictions = {}, {}
        none_id = self.relation_names_lower_case.index('none')
        for elem, ids, single in zip(elems, token_rows, single_flags):
            output = self.outputs_from_token_ids([ids], single=single)[0]
            self._score_frame(elem, output, take_rel_preds, take_rel_gts, frame_rel, raw_predictions, none_id, keep_history=False)
        return self._close_validation(take_rel_preds, take_rel_gts, frame_rel, raw_predictions, True, logging_information, True)

    def report_merged(self, frame_rel, scan_order, logging_information=None):
        """Per-take and overall reports over per-frame labels gathered from all ranks ({scan_id: (take, preds, gts)}),
        concatenated in dataset order (what one process iterating the whole split accumulates)."""
        preds, gts = defaultdict(list), defaultdict(list)
        for scan_id in scan_order:
            if scan_id in frame_rel:
                take_idx, p, g = frame_rel[scan_id]
                preds[take_idx].extend(p)
                gts[take_idx].extend(g)
        self.val_take_rel_preds, self.val_take_rel_gts = preds, gts
        self.last_take_rel = (dict(preds), dict(gts))
        self.last_macro_f1 = self.evaluate_predictions(None, 'val', logging_information=logging_information)
        self.reset_metrics(split='val')
        return self.last_macro_f1

    def evaluate_predictions(self, epoch_loss, split, logging_information=None):
        from sklearn.metrics import classification_report
        if split == 'train':
            preds, gts = self.train_take_rel_preds, self.train_take_rel_gts
        elif split == 'val':
            preds, gts = self.val_take_rel_preds, self.val_take_rel_gts
        else:
            raise NotImplementedError()
        labels = list(range(len(self.relationNames)))
        all_g, all_p = [], []
        for take_idx in sorted(preds.keys()):
            all_g.extend(gts[take_idx])
            all_p.extend(preds[take_idx])
            if logging_information is not None:
                rep = classification_report(gts[take_idx], preds[take_idx], labels=labels, target_names=self.relationNames, output_dict=True)
                for rel_name in self.relationNames:
                    for score in ('precision', 'recall', 'f1-score'):
                        logging_information['logger'].log_metrics({f'{rel_name}/{take_idx}_{score[:2].upper()}': rep[rel_name][score]},
                                                                  step=logging_information['checkpoint_id'])
            print(f'\nTake {take_idx}\n')
            print(classification_report(gts[take_idx], preds[take_idx], labels=labels, target_names=self.relationNames))
        if not all_g:
            return 0.0
        results = classification_report(all_g, all_p, labels=labels, target_names=self.relationNames, output_dict=True)
        if logging_information is not None:
            for k, name in (('precision', 'precision'), ('recall', 'recall'), ('f1-score', 'macro_f1')):
                logging_information['logger'].log_metrics({f'{logging_information["split"]}_{name}': results['macro avg'][k]},
                                                          step=logging_information['checkpoint_id'])
        print(f'{split} Results:\n')
        print(classification_report(all_g, all_p, labels=labels, target_names=self.relationNames))
        return results['macro avg']['f1-score']
