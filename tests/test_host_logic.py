"""CPU: host-side logic of the product (splice plan, tokenizer, prompt template, C-ABI surface)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import cpu_ref as cr
from oracle_amd.llava.model.llava_arch import KIND_DESC, KIND_IMG, KIND_TOK, plan_splice


def _codes_from_srcs(srcs, t, feature_rows, desc_counts, side='left'):
    out = np.zeros((len(srcs), t), dtype=np.int64)
    base = 0
    for b, src in enumerate(srcs):
        starts = np.concatenate([[0], np.cumsum(desc_counts[b])]) if desc_counts else None
        row = []
        for s in src:
            if s[0] == 'tok':
                row.append((KIND_TOK << 28) | s[1])
            elif s[0] == 'img':
                row.append((KIND_IMG << 28) | (s[1] * feature_rows + s[2]))
            else:
                row.append((KIND_DESC << 28) | (base + int(starts[s[1]]) + s[2]))
        if desc_counts:
            base += int(sum(desc_counts[b]))
        n = len(row)
        if side == 'left':
            out[b, t - n:] = row
        else:
            out[b, :n] = row
    return out


@pytest.mark.parametrize('name', ['single', 'batch3', 'desc7', 'desc_multirow', 'desc_missing'])
def test_plan_splice_matches_oracle_and_golden(name):
    g = np.load(os.path.join(GOLDEN, 'splice_tiny_float32.npz'))
    ids = g[f'{name}.ids']
    desc, counts = None, None
    if f'{name}.desc' in g.files:
        d = torch.from_numpy(g[f'{name}.desc'])
        desc = [d[0:3], d[3]] if name == 'desc_multirow' else [d[j] for j in range(d.shape[0])]
        counts = [[3, 1]] if name == 'desc_multirow' else [[1] * d.shape[0]]
    codes, mask, pos, lab = plan_splice(ids, ids != 0, 576, counts, 'left', None)
    assert np.array_equal(mask.astype(np.uint8), g[f'{name}.mask'])
    assert np.array_equal(pos, g[f'{name}.pos'])
    st = {'model.embed_tokens.weight': torch.zeros(32000, 4)}
    feats = torch.zeros(ids.shape[0], 576, 4)
    dd = None if desc is None else [x[..., :4] for x in desc]
    _, _, _, srcs = cr.splice(st, torch.from_numpy(ids), torch.from_numpy(ids != 0), feats, dd)
    want = _codes_from_srcs(srcs, mask.shape[1], 576, counts)
    assert np.array_equal(codes.view(np.uint32).astype(np.int64), want)


def test_plan_splice_right_truncated():
    g = np.load(os.path.join(GOLDEN, 'splice_tiny_float32.npz'))
    ids = g['right_trunc.ids']
    codes, mask, pos, _ = plan_splice(ids, ids != 0, 576, None, 'right', 580)
    assert np.array_equal(mask.astype(np.uint8), g['right_trunc.mask'])
    assert np.array_equal(pos, g['right_trunc.pos'])


def test_prompt_template_and_tokenizer():
    from oracle_amd.llava.conversation import SeparatorStyle, default_conversation
    from oracle_amd.llava.mm_utils import KeywordsStoppingCriteria, tokenizer_image_token
    from oracle_amd.tokenizer import SynthTokenizer
    kat = json.load(open(os.path.join(GOLDEN, 'text_kat.json')))
    c = default_conversation.copy()
    c.append_message(c.roles[0], '<image>\nHELLO')
    c.append_message(c.roles[1], None)
    assert c.get_prompt() == kat['vicuna_v1_prompt']
    assert c.sep_style == SeparatorStyle.TWO and c.sep2 == kat['sep2']
    tok = SynthTokenizer()
    ids = tokenizer_image_token(c.get_prompt(), tok, return_tensors='pt')
    assert ids[0] == 1 and int((ids == -200).sum()) == 1 and int((ids == 1).sum()) == 1
    text = 'J: ***. K: *** ***. α β head surgeon,patient,cutting;'
    enc = tok(text).input_ids
    assert enc.count(18610) == 3 and tok.decode(enc[1:]) == text
    crit = KeywordsStoppingCriteria(['</s>'], tok, ids[None])
    assert crit.keyword_ids[0].tolist() == [2]
    assert crit(torch.tensor([[5, 2], [7, 2]])) and not crit(torch.tensor([[5, 2], [7, 9]]))


def test_c_abi_exports_every_declared_symbol():
    from oracle_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'oracle_hip.h')).read()
    declared = set(re.findall(r'\b(orc_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('library not built (run __graft_entry__.build())')
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(l, name), name
    l.orc_abi_version.restype = ctypes.c_int
    assert l.orc_abi_version() == 1


def test_product_fails_loudly_without_library_or_device(monkeypatch):
    """No CPU fallback anywhere in the product: a missing shared library or a missing HIP device is an error that says so,
    and nothing under oracle_amd/ imports the oracle."""
    import torch
    from oracle_amd import _lib
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', os.path.join(ROOT, 'oracle_amd', 'csrc', 'no_such_library.so'))
    with pytest.raises(_lib.OracleHipError, match='no fallback'):
        _lib.lib()
    if not torch.cuda.is_available():
        with pytest.raises(_lib.OracleHipError, match='no CPU fallback'):
            Engine(TINY)
        from oracle_amd.llava.model.builder import load_pretrained_model
        with pytest.raises(_lib.OracleHipError):
            load_pretrained_model('synthetic:tiny', None, 'synthetic:tiny')
    for base, _, files in os.walk(os.path.join(ROOT, 'oracle_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle(\s|\.|$)', src, flags=re.MULTILINE), os.path.join(base, f)
