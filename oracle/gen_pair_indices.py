"""TEST INFRASTRUCTURE — golden vectors for the `--auxiliary_task` attribute-pair extraction, captured from the REFERENCE's own
RobertaOneTowerDataset.__getitem__ (src/data/data.py:568-612).  Build container only (needs /root/reference, read-only);
data.py imports names that transformers 5.x / this image lack (TruncationStrategy from tokenization_utils, jieba, timm), so
those modules are stubbed before the import and a stand-in tokenizer hands the dataset pre-made token ids.

    python oracle/gen_pair_indices.py      # writes tests/golden/pair_indices.json
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return type(n, (), {})


import torch, transformers  # noqa: E401,E402  (import transformers before stubbing timm)
import transformers.tokenization_utils as tu  # noqa: E402
for name in ("TruncationStrategy", "PaddingStrategy"):
    if not hasattr(tu, name):
        setattr(tu, name, getattr(transformers.utils, name, None) or getattr(__import__("transformers.tokenization_utils_base", fromlist=[name]), name))
for m in ["timm", "timm.data", "timm.data.transforms_factory", "jieba", "torch_geometric", "torch_geometric.data", "torch_geometric.nn", "PIL.ImageFile"]:
    if m not in sys.modules:
        st = _Stub(m); st.__path__ = []; sys.modules[m] = st
sys.modules["jieba"].cut = lambda s: s.split(" ")
import src.data.data as D  # noqa: E402

CLS, SEP, COLON, SEMI = 101, 102, D.COLON_ID, D.SEMICOLON_ID


class FakeTokenizer:
    """returns the ids prepared for the current sample: [CLS] src_title [SEP] src_pvs [SEP] tgt_title [SEP] tgt_pvs [SEP] pad"""
    sep_token, sep_token_id, bos_token_id = "[SEP]", SEP, CLS

    def __init__(self):
        self.next_ids = None

    def __call__(self, **kw):
        ids = self.next_ids
        return types.SimpleNamespace(data=dict(input_ids=list(ids), token_type_ids=[0] * len(ids), attention_mask=[1] * len(ids)))


def attrs(rs, keys, n, drop_colon=False):
    out = []
    for k in keys[:n]:
        out += list(k) + ([] if drop_colon and rs.rand() < 0.3 else [COLON]) + list(rs.randint(200, 206, size=rs.randint(1, 3))) + [SEMI]
    return out


def main():
    rs = np.random.RandomState(4)
    tk = FakeTokenizer()
    cases = []
    keys = [list(rs.randint(150, 160, size=rs.randint(1, 3))) for _ in range(6)]
    for c in range(40):
        n_src, n_tgt = rs.randint(0, 6), rs.randint(0, 6)
        tgt_keys = list(keys)
        if rs.rand() < 0.5 and n_tgt > 1:
            j = rs.randint(1, n_tgt)
            tgt_keys[j] = [170]                                    # a key mismatch part-way
        src_pvs, tgt_pvs = attrs(rs, keys, n_src, c % 5 == 0), attrs(rs, tgt_keys, n_tgt, c % 7 == 0)
        if c % 4 == 0 and tgt_pvs:
            tgt_pvs = tgt_pvs[:-1]                                 # truncated: the last attribute lost its ';'
        ids = [CLS] + list(rs.randint(300, 310, size=3)) + [SEP] + src_pvs + [SEP] + list(rs.randint(300, 310, size=2)) + [SEP] + tgt_pvs + [SEP] + [0] * 4
        tk.next_ids = ids
        ds = D.RobertaOneTowerDataset([(1, "a", 0, "t", "p", "b", 0, "t", "p")], tk, 8, "cls", max_seq_len_pv=12, auxiliary_task=True)
        try:
            rec = ds[0]
        except Exception as e:                                     # e.g. a first attribute without ':' (None + 1 in the reference)
            cases.append(dict(input_ids=[int(v) for v in ids], error=type(e).__name__))
            continue
        cases.append(dict(input_ids=[int(v) for v in ids], pair_indices=[[int(v) for v in r] for r in rec["pair_indices"]]))
    json.dump(dict(sep_token_id=SEP, colon_id=int(COLON), semicolon_id=int(SEMI), cases=cases),
              open(os.path.join(ROOT, "tests", "golden", "pair_indices.json"), "w"))
    print("wrote", len(cases), "cases;", sum(len(c.get("pair_indices", [])) for c in cases), "pairs;", sum("error" in c for c in cases), "errors")


if __name__ == "__main__":
    main()
