"""AdaptabilityOracleWrapper: the same generate() hot path driven by a per-case prompt that introduces ONE new entity or
predicate through its textual attributes or through an image-crop embedding (reference
scene_graph_prediction/scene_graph_helpers/model/adaptability_wrapper_oracle.py:16-254). One frame per call (batch 1,
single view); the verdict is whether the test condition occurs in the back-mapped answer.

Knowledge prompt (reference :43-120): every entity / predicate gets the next free symbol (A.. / alpha..) in table order,
new names are appended, redefined names keep their place; in the visual variant the definitions of the seven
crop-described concepts (and of the new one) are the `<vis_descriptor>` placeholder and their embedding files are
collected in symbol order -- entities first -- which is the order generate() consumes them in."""
import re
from copy import deepcopy

import torch
from PIL import Image

from ....llava.constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX, VIS_DESCRIPTOR_TOKEN
from ....llava.conversation import SeparatorStyle, default_conversation
from ....llava.mm_utils import KeywordsStoppingCriteria, get_model_name_from_path, process_images, tokenizer_image_token
from ....llava.model.builder import load_pretrained_model
from ... import prompts

ENTITY_SYMBOLS = list('ABCDEFGHIJKLMNOPQRSTUVWXYZ')
PREDICATE_SYMBOLS = list('αβγδεζηθικλμνξοπρστυφχψω')
TOOL_PREDICATES = ('cementing', 'cutting', 'drilling', 'hammering', 'robotic_sawing', 'sawing', 'suturing')
NEW_ENTITIES = ('kuka', 'mako')
CROP_DIR = 'synthetic_or_generation/original_crops'
CROP_DESCRIBED = ('anesthesia equipment', 'cementing', 'cutting', 'drilling', 'hammering', 'sawing', 'suturing')


def _load_tensor(path):
    return torch.load(path, map_location='cpu', weights_only=True)


class AdaptabilityOracleWrapper:
    def __init__(self, config, labels, model_path, model_base='liuhaotian/llava-v1.5-7b', load_8bit=False, load_4bit=False, **engine_kw):
        self.config = config
        self.mconfig = config['MODEL']
        self.labels = labels
        self.model_name = get_model_name_from_path(model_path)
        self.tokenizer, self.model, self.image_processor, self.context_len = load_pretrained_model(
            model_path, model_base, self.model_name, load_8bit, load_4bit, **engine_kw)
        self.model.config.mv_type = self.mconfig['mv_type']
        self.model.config.tokenizer_padding_side = 'left'
        if self.config['USE_VIS_DESC']:
            self.vis_knowledge_paths = {name: f'{CROP_DIR}/{name}_cam2.pt' for name in CROP_DESCRIBED}
            self.vis_descriptor_embs = {p: _load_tensor(p) for p in self.vis_knowledge_paths.values()}

    def _use_vis(self):
        return bool(self.config.get('USE_VIS_DESC'))

    def _is_symbolic(self):
        return '_symbolic' in self.model_name or '_synthetic' in self.model_name

    # ------------------------------------------------------------------ prompt
    def _symbolic_prompt_maker(self, customizations):
        """customizations: [(name, definition, 'e' | 'p', crop image path or None)]
        -> (user prompt, entity name -> symbol, predicate name -> symbol, embedding paths in symbol order | None)"""
        is_vis = self._use_vis()
        definitions = {'e': {n: d + '.' for n, d in zip(prompts.ENTITIES, prompts._ENTITY_KNOWLEDGE)},
                       'p': {n: d + '.' for n, d in zip(prompts.PREDICATES, prompts._PREDICATE_KNOWLEDGE)}}
        emb_path = {}
        if is_vis:
            emb_path = dict(self.vis_knowledge_paths)
            for name in CROP_DESCRIBED:
                kind = 'e' if name in definitions['e'] else 'p'
                definitions[kind][name] = f'{VIS_DESCRIPTOR_TOKEN}.' if kind == 'e' else f'use of a tool: {VIS_DESCRIPTOR_TOKEN}.'
        for name, definition, kind, crop in customizations:
            if kind not in definitions:
                raise NotImplementedError()
            definitions[kind][name] = definition
            if is_vis:
                emb_path[name] = str(crop).replace('.jpg', '_crop.pt')

        def assign(table, symbols):
            name_to_symbol = dict(zip(table, symbols))
            rows = sorted((name_to_symbol[n], d) for n, d in table.items())
            return name_to_symbol, rows, sorted((name_to_symbol[n], emb_path[n]) for n in table if n in emb_path)

        ent_sym, ent_rows, ent_paths = assign(definitions['e'], ENTITY_SYMBOLS)
        pred_sym, pred_rows, pred_paths = assign(definitions['p'], PREDICATE_SYMBOLS)
        text = f'Entities: [{", ".join(s for s, _ in ent_rows)}]. Predicates: [{", ".join(s for s, _ in pred_rows)}]. <knowledge_start> '
        text += ''.join(f'{s}: {d} ' for s, d in ent_rows + pred_rows)
        text += f'<knowledge_end> {prompts.TASK}'
        paths = [p for _, p in ent_paths + pred_paths] if is_vis else None
        return text, ent_sym, pred_sym, paths

    def _customization(self, label, textual_attributes, vis_desc_path):
        if self._use_vis():
            body, crop = f'{VIS_DESCRIPTOR_TOKEN}.', vis_desc_path
        else:
            t = textual_attributes
            body, crop = f'{t["color"]}, {t["size"]}, {t["shape"]}, {t["texture"]}, {t["object_type"]}.', None
        if label in TOOL_PREDICATES:
            return [('sawing' if label == 'robotic_sawing' else label, f'use of a tool: {body}', 'p', crop)]
        if label in NEW_ENTITIES:
            return [(label, body, 'e', crop)]
        raise NotImplementedError(label)

    # ------------------------------------------------------------------ forward
    def forward(self, batch):
        assert len(batch) == 1
        elem = batch[0]
        conv = deepcopy(default_conversation)
        images = [Image.open(elem['image_path']).convert('RGB')]
        image_tensor = process_images(images, self.image_processor, self.model.config)
        if type(image_tensor) is list:
            image_tensor = [im.to(self.model.device, dtype=torch.bfloat16) for im in image_tensor]
        else:
            image_tensor = image_tensor.to(self.model.device, dtype=torch.bfloat16)

        ent_sym = pred_sym = None
        if self._use_vis() or self._is_symbolic():
            custom = self._customization(elem['label'], elem['textual_attributes'], elem['vis_desc_path'])
            inp, ent_sym, pred_sym, emb_paths = self._symbolic_prompt_maker(custom)
        else:
            inp, emb_paths = prompts.plain_prompt(), None
        if self.model.config.mm_use_im_start_end:
            inp = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN + '\n' + inp
        else:
            inp = DEFAULT_IMAGE_TOKEN + '\n' + inp
        conv.append_message(conv.roles[0], inp)
        conv.append_message(conv.roles[1], None)
        input_ids = tokenizer_image_token(conv.get_prompt(), self.tokenizer, IMAGE_TOKEN_INDEX, return_tensors='pt').unsqueeze(0).to(self.model.device)
        stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
        stopping = KeywordsStoppingCriteria([stop_str], self.tokenizer, input_ids)

        embs = None
        if emb_paths is not None:
            embs = [(self.vis_descriptor_embs[p] if p in self.vis_descriptor_embs else _load_tensor(p)).to(self.model.device, dtype=torch.bfloat16)
                    for p in emb_paths]
        with torch.inference_mode():
            output_ids = self.model.generate(input_ids, images=image_tensor, do_sample=False, use_cache=True, max_new_tokens=300,
                                             stopping_criteria=[stopping], vis_descriptor_embs=embs if self._use_vis() else None)
        output = self.tokenizer.decode(output_ids[0, input_ids.shape[1]:]).strip()

        if self._is_symbolic() or self.config['USE_VIS_DESC']:
            # all symbols in ONE regex pass, so substituted text is never substituted again (reference :209-217)
            back = {sym: name for name, sym in list(ent_sym.items()) + list(pred_sym.items())}
            regex = re.compile('|'.join(map(re.escape, back)))
            cleaned = regex.sub(lambda m: back[m.group(0)], output.replace('<SG>', '').replace('</SG>', ''))
            output = f'<SG>{cleaned}</SG>'
        fulfilled = elem['test_condition'].lower().replace('<', '').replace('>', '') in output.lower()
        return fulfilled, output

    def validate(self, dataloader):
        """Per case: ground truth = the label for a positive image, 'none' for a negative; prediction = the label if the
        test condition was met, else 'none'. Prints (and returns) sklearn's report over the real labels (reference :227-254)."""
        from sklearn.metrics import classification_report
        idx = {label: i for i, label in enumerate(self.labels)}
        none = len(idx)
        preds, gts = [], []
        for batch in dataloader:
            assert len(batch) == 1
            elem = batch[0]
            fulfilled, _ = self.forward(batch)
            gts.append(idx[elem['label']] if elem['is_pos'] == True else none)   # noqa: E712 (reference compares to True)
            preds.append(idx[elem['label']] if fulfilled else none)
        results = classification_report(gts, preds, labels=list(range(none)), target_names=self.labels, output_dict=False, zero_division=1.0)
        print(results)
        return results
