"""A deterministic stand-in for transformers.BertTokenizer with exactly the interface the datasets of src/data/data.py use
(`__call__(text=, text_pair=, max_length=, padding=, truncation=)` -> object with `.data`, `tokenize`, `convert_tokens_to_ids`, the
special-token attributes).  Whitespace tokens looked up in a fixed vocabulary, [CLS] a [SEP] (b [SEP]) layout, BERT token types,
longest-first truncation, right padding to max_length.  Used by oracle/gen_collates.py (which runs the REFERENCE's dataset classes
on it) and by tests/test_collate_golden.py (which runs this repo's) -- so the fixture does not depend on the installed transformers
version."""
import types

SPECIALS = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]", "<S>"]
WORDS = ["手机", "红色", "蓝色", "大号", "小号", "棉", "电池", "型号", "品牌", "华为", "苹果", "颜色", "尺码", "材质", "a1", "b2", "x", "y"]


class FakeBertTokenizer:
    def __init__(self):
        vocab = SPECIALS + [f"[fill{i}]" for i in range(26)] + [":", ";"] + WORDS
        assert vocab.index(":") == 131 and vocab.index(";") == 132 and vocab.index("[unused99]") == 99
        self.vocab = {t: i for i, t in enumerate(vocab)}
        self.pad_token_id, self.unk_token_id = 0, self.vocab["[UNK]"]
        self.cls_token, self.sep_token, self.bos_token = "[CLS]", "[SEP]", "<S>"
        self.cls_token_id, self.sep_token_id, self.bos_token_id = self.vocab["[CLS]"], self.vocab["[SEP]"], self.vocab["<S>"]
        self.vocab_size = len(vocab)

    def tokenize(self, text):
        out = []
        for w in text.split():
            # ':' and ';' are their own tokens (the BERT-zh vocabulary splits punctuation), everything else is one token per word
            cur = ""
            for ch in w:
                if ch in ":;":
                    if cur:
                        out.append(cur)
                        cur = ""
                    out.append(ch)
                else:
                    cur += ch
            if cur:
                out.append(cur)
        return out

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab.get(t, self.unk_token_id) for t in tokens]

    def __call__(self, text=None, text_pair=None, max_length=None, padding=None, truncation=None, **kw):
        a = self.convert_tokens_to_ids(self.tokenize(text))
        b = self.convert_tokens_to_ids(self.tokenize(text_pair)) if text_pair is not None else None
        room = max_length - (3 if b is not None else 2)
        while len(a) + (len(b) if b is not None else 0) > room:            # longest_first, one token at a time from the end
            if b is not None and len(b) > len(a):
                b.pop()
            else:
                a.pop()
        ids = [self.cls_token_id] + a + [self.sep_token_id]
        types_ = [0] * len(ids)
        if b is not None:
            ids += b + [self.sep_token_id]
            types_ += [1] * (len(b) + 1)
        n = len(ids)
        pad = max_length - n
        return types.SimpleNamespace(data={"input_ids": ids + [0] * pad, "token_type_ids": types_ + [0] * pad, "attention_mask": [1] * n + [0] * pad})
