"""Synthetic stand-in for the Llama tokenizer (no tokenizer model exists offline).

A deterministic word-level vocabulary with byte fallback, with the ids the hot path depends on kept at
their Llama values: <unk>/pad 0, <s> 1, </s> 2, and "***" -> 18610 (VIS_DESCRIPTOR_TOKEN_INDEX,
reference LLaVA/llava/constants.py:9-10). Interface subset used by the reference: `tok(text).input_ids`,
`decode`, `batch_decode(..., skip_special_tokens=)`, `bos_token_id`, `eos_token_id`, `pad_token_id`.
A real tokenizer directory (tokenizer.model / tokenizer.json) is used instead when one is supplied to
`load_pretrained_model`."""
import re

BYTE_BASE = 3          # ids 3..258 are raw bytes
WORD_BASE = 1000       # word pieces start here
_SPECIAL = {'<unk>': 0, '<s>': 1, '</s>': 2}
_PIECE = re.compile(r' ?\*\*\*| ?[A-Za-z]+| ?[0-9]| ?[^\sA-Za-z0-9]|\s')

# Word list: every word of the ORacle prompts / entity / predicate vocabulary, so answers and prompts
# tokenise to realistic lengths (about one token per word or punctuation mark).
_CORPUS = """
A chat between a curious user and an artificial intelligence assistant The gives helpful detailed polite answers to
the questions USER ASSISTANT image Entities Predicates head surgeon circulator nurse anaesthetist patient instrument
table operating secondary anesthesia equipment assisting cementing cleaning closeTo cutting drilling hammering holding
lyingOn manipulating preparing sawing suturing touching Given following scene graph memory representation generate for
timepoint T output should strictly be list of triplets each in format entity1 entity2 predicate Do not provide
narrative or descriptive text knowledge start end primary operator surgery handles critical tasks supports assists
surgical coordinates OR activities tools prep recovery administers monitors undergoes procedure blue large rectangular
matte black rubber gray metal white irregular handheld collaboration staff use tool straight bone cement gun
sanitization proximity medical other small plastic scalpel orange L shape drill brown hammer grasping instruments
positioned on handling objects like tables machines includes draping sterilization green round saw scissors physical
contact entities Long Short SG Instrument none human object close tool B C D E F G H I J K
"""


class _Encoding:
    def __init__(self, ids):
        self.input_ids = ids
        self.attention_mask = [1] * len(ids)


class SynthTokenizer:
    bos_token_id = 1
    eos_token_id = 2
    pad_token_id = 0
    unk_token_id = 0
    bos_token, eos_token, pad_token, unk_token = '<s>', '</s>', '<unk>', '<unk>'
    padding_side = 'left'

    def __init__(self, vocab_size=32000):
        self.vocab_size = vocab_size
        words = sorted(set(_CORPUS.split()))
        self.piece_to_id = {}
        nid = WORD_BASE
        for w in words:
            for piece in (w, ' ' + w):
                if nid == 18610:
                    nid += 1
                self.piece_to_id[piece] = nid
                nid += 1
        self.piece_to_id['***'] = 18610
        self.id_to_piece = {v: k for k, v in self.piece_to_id.items()}
        for k, v in _SPECIAL.items():
            self.id_to_piece[v] = k

    def __len__(self):
        return self.vocab_size

    # --- encode
    def encode(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if add_special_tokens else []
        pos = 0
        while pos < len(text):
            hit = None
            for sp in ('</s>', '<s>', '<unk>'):
                if text.startswith(sp, pos):
                    hit = sp
                    break
            if hit:
                ids.append(_SPECIAL[hit])
                pos += len(hit)
                continue
            m = _PIECE.match(text, pos)
            piece = m.group(0) if m else text[pos]
            if piece == ' ***':            # the descriptor placeholder always maps to the single id 18610
                ids += [BYTE_BASE + 0x20, 18610]
            elif piece in self.piece_to_id:
                ids.append(self.piece_to_id[piece])
            else:
                ids += [BYTE_BASE + b for b in piece.encode('utf-8')]
            pos += len(piece)
        return ids

    def __call__(self, text, add_special_tokens=True, **kw):
        return _Encoding(self.encode(text, add_special_tokens))

    # --- decode
    def decode(self, ids, skip_special_tokens=False, **kw):
        if hasattr(ids, 'tolist'):
            ids = ids.tolist()
        out = bytearray()
        for i in ids:
            i = int(i)
            if i in (0, 1, 2):
                if not skip_special_tokens:
                    out += self.id_to_piece[i].encode()
            elif BYTE_BASE <= i < BYTE_BASE + 256:
                out.append(i - BYTE_BASE)
            elif i in self.id_to_piece:
                out += self.id_to_piece[i].encode()
            else:
                out += f'<{i}>'.encode()          # id outside the synthetic vocabulary
        return out.decode('utf-8', errors='replace')

    def batch_decode(self, seqs, skip_special_tokens=False, **kw):
        if hasattr(seqs, 'tolist'):
            seqs = seqs.tolist()
        return [self.decode(s, skip_special_tokens=skip_special_tokens) for s in seqs]
