"""Host-side image / token utilities with the reference's names and argument meaning
(LLaVA/llava/mm_utils.py:14-105): expand2square, process_images, tokenizer_image_token,
get_model_name_from_path, KeywordsStoppingCriteria; plus the CLIP image processor the reference obtains
from transformers (CLIPImageProcessor: bicubic shortest-edge resize, centre crop, /255, normalise)."""
import numpy as np
import torch
from PIL import Image

from .constants import IMAGE_TOKEN_INDEX

OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


class ClipImageProcessor:
    """preprocess(image, return_tensors='pt')['pixel_values'] -> (1,3,S,S) float32, as CLIPImageProcessor does."""

    def __init__(self, size=336, image_mean=None, image_std=None):
        self.size = {'shortest_edge': size}
        self.crop_size = {'height': size, 'width': size}
        self.image_mean = list(OPENAI_CLIP_MEAN if image_mean is None else image_mean)
        self.image_std = list(OPENAI_CLIP_STD if image_std is None else image_std)

    def _one(self, image):
        if not isinstance(image, Image.Image):
            image = Image.fromarray(np.asarray(image))
        image = image.convert('RGB')
        s = self.size['shortest_edge']
        w, h = image.size
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = s, int(s * long / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        image = image.resize((nw, nh), resample=Image.BICUBIC)
        ch, cw = self.crop_size['height'], self.crop_size['width']
        top, left = (nh - ch) // 2, (nw - cw) // 2
        image = image.crop((left, top, left + cw, top + ch))
        x = np.asarray(image, dtype=np.float32) * np.float32(1.0 / 255.0)
        x = (x - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
        return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))

    def preprocess(self, images, return_tensors='pt'):
        if isinstance(images, (list, tuple)):
            px = torch.stack([self._one(im) for im in images])
        else:
            px = self._one(images)[None]
        return {'pixel_values': px}

    __call__ = preprocess


def expand2square(pil_img, background_color):
    """Centre the image on a square canvas of the longer side (mm_utils.py:14-25)."""
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


def process_images(images, image_processor, model_cfg):
    """PIL images of one frame -> (1,V,3,S,S) for V>1 views, (1,3,S,S) for a single view (mm_utils.py:28-40)."""
    if getattr(model_cfg, 'image_aspect_ratio', None) != 'pad':
        return image_processor(images, return_tensors='pt')['pixel_values']
    fill = tuple(int(c * 255) for c in image_processor.image_mean)
    views = [image_processor.preprocess(expand2square(im, fill), return_tensors='pt')['pixel_values'][0] for im in images]
    if all(v.shape == views[0].shape for v in views):
        stacked = torch.stack(views, dim=0)
        return stacked if len(views) == 1 else stacked[None]
    return views


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    """Tokenise the text around every '<image>' and join the pieces with the image placeholder id, keeping a
    single BOS (mm_utils.py:43-62)."""
    pieces = [tokenizer(chunk).input_ids for chunk in prompt.split('<image>')]
    has_bos = bool(pieces) and len(pieces[0]) > 0 and pieces[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    ids = [pieces[0][0]] if has_bos else []
    for i, piece in enumerate(pieces):
        if i > 0:
            ids.append(image_token_index)
        ids.extend(piece[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == 'pt':
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f'Unsupported tensor type: {return_tensors}')


def get_model_name_from_path(model_path):
    parts = model_path.strip('/').split('/')
    return parts[-2] + '_' + parts[-1] if parts[-1].startswith('checkpoint-') else parts[-1]


class KeywordsStoppingCriteria:
    """Stop when EVERY row ends with a keyword: last-k ids equal the keyword ids, or the keyword appears in the
    decoded tail (mm_utils.py:74-105). Callable as criteria(output_ids, scores) like a transformers StoppingCriteria."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.tokenizer = tokenizer
        self.keyword_ids = []
        for kw in self.keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.keyword_ids.append(torch.tensor(ids))
        self.max_keyword_len = max([len(k) for k in self.keyword_ids] + [0])
        self.start_len = input_ids.shape[1]

    def call_for_batch(self, output_ids, scores, **kwargs):
        row = output_ids[0]
        for kid in self.keyword_ids:
            if bool((row[-kid.shape[0]:].cpu() == kid).all()):
                return True
        offset = min(output_ids.shape[1] - self.start_len, self.max_keyword_len)
        text = self.tokenizer.batch_decode(output_ids[:, -offset:], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, output_ids, scores=None, **kwargs):
        return all(self.call_for_batch(output_ids[i:i + 1], scores) for i in range(output_ids.shape[0]))
