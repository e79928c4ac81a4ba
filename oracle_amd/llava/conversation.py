"""Prompt template of the hot path: Vicuna v1 ("{system} USER: {msg} ASSISTANT:"), the reference's
`default_conversation` (LLaVA/llava/conversation.py:253-263,361). Only the separator styles that template
family needs are kept (SINGLE, TWO); prompt text is pinned by tests/golden/text_kat.json."""
import copy
import enum


class SeparatorStyle(enum.Enum):
    SINGLE = enum.auto()
    TWO = enum.auto()


class Conversation:
    def __init__(self, system, roles, messages=(), offset=0, sep_style=SeparatorStyle.SINGLE, sep='###', sep2=None,
                 version='Unknown'):
        self.system = system
        self.roles = tuple(roles)
        self.messages = [list(m) for m in messages]
        self.offset = offset
        self.sep_style = sep_style
        self.sep = sep
        self.sep2 = sep2
        self.version = version

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        seps = (self.sep, self.sep) if self.sep_style == SeparatorStyle.SINGLE else (self.sep, self.sep2)
        out = [self.system, seps[0]]
        for turn, (role, text) in enumerate(self.messages):
            if text:
                out += [role, ': ', text, seps[turn % 2]]
            else:
                out += [role, ':']          # open turn: the model continues from here
        return ''.join(out)

    def copy(self):
        return copy.deepcopy(self)


conv_vicuna_v1 = Conversation(
    system=('A chat between a curious user and an artificial intelligence assistant. '
            "The assistant gives helpful, detailed, and polite answers to the user's questions."),
    roles=('USER', 'ASSISTANT'), version='v1', sep_style=SeparatorStyle.TWO, sep=' ', sep2='</s>')

default_conversation = conv_vicuna_v1
conv_templates = {'default': conv_vicuna_v1, 'v1': conv_vicuna_v1, 'vicuna_v1': conv_vicuna_v1}
