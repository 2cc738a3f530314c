"""The token-id layout of CogView's unified tokenizer (data_utils/unified_tokenizer.py:23-68) without the tokenizers
themselves: image codes [0, 8192), text pieces [8192, 58192), then 27 command tokens -> 58219 ids.  The sampling loop
only needs the boundaries and the command ids (`tokenizer['[BOI1]']`, `tokenizer.img_tokenizer.num_tokens`, ...)."""

_COMMANDS = ['[PAD]', '[BOI1]', '[BOI2]', '[BOI3]', '[EOI1]', '[EOI2]', '[EOI3]', '[ROI1]', '[ROI2]', '[ROI3]', '[SEP]',
             '[MASK]', '[CLS]', '[ENC]', '[TINY]', '[SMALL]', '[BASE]', '[BIG]', '[POS0]', '[POS1]', '[POS2]', '[POS3]',
             '[POS4]', '[POS5]', '[POS6]', '[POS7]', '[POS8]']


class _Sized:
    def __init__(self, num_tokens):
        self.num_tokens = int(num_tokens)

    def __len__(self):
        return self.num_tokens


class IdSpace:
    def __init__(self, img_tokens=8192, txt_tokens=50000):
        self.img_tokenizer, self.txt_tokenizer = _Sized(img_tokens), _Sized(txt_tokens)
        base = img_tokens + txt_tokens
        self.command_tokens = {name: base + i for i, name in enumerate(_COMMANDS)}
        self.num_tokens = base + len(_COMMANDS)

    def __getitem__(self, command_token):
        return self.command_tokens[command_token]

    def wrap_code(self, code, idx=1):
        """data_utils/unified_tokenizer.py:125-150: [size token] [BOIidx] code [EOIidx]; the size token follows from the
        code length (8x8 [TINY], 16x16 [SMALL], 32x32 [BASE], 64x64 [BIG])."""
        import numpy as np
        s = int(round(len(code) ** 0.5))
        assert s * s == len(code)
        prefix = {8: '[TINY]', 16: '[SMALL]', 32: '[BASE]', 64: '[BIG]'}[s]
        head = [self.command_tokens[prefix], self.command_tokens['[BOI%d]' % idx]]
        tail = [self.command_tokens['[EOI%d]' % idx]]
        if isinstance(code, list):
            return head + code + tail
        return np.concatenate((np.array(head), np.asarray(code), np.array(tail)), axis=0)

    def __len__(self):
        return self.num_tokens
