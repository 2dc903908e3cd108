"""TEST INFRASTRUCTURE — golden vectors for the data formats either side of the hot path: the REFERENCE's own dataset
classes and collate functions (src/data/data.py:37-240, 277-832, 918-989) run in the build container on seeded synthetic rows, their
records and batch tuples written to tests/golden/collates.json.  Needs /root/reference (read-only); data.py imports names this image
lacks (jieba, timm, TruncationStrategy from tokenization_utils), which are stubbed before the import exactly as in
gen_pair_indices.py.  The tokenizer is tests/fake_tokenizer.py (shared with the test that replays the rows through this repo's
datasets), so the fixture does not depend on the installed transformers version.

    python oracle/gen_collates.py      # writes tests/golden/collates.json
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return type(n, (), {})


import torch, transformers  # noqa: E401,E402
import transformers.tokenization_utils as tu  # noqa: E402
for name in ("TruncationStrategy", "PaddingStrategy"):
    if not hasattr(tu, name):
        setattr(tu, name, getattr(transformers.utils, name, None) or getattr(__import__("transformers.tokenization_utils_base", fromlist=[name]), name))
for m in ["timm", "timm.data", "timm.data.transforms_factory", "jieba", "torch_geometric", "torch_geometric.data", "torch_geometric.nn", "PIL.ImageFile"]:
    if m not in sys.modules:
        st = _Stub(m); st.__path__ = []; sys.modules[m] = st
sys.modules["jieba"].cut = lambda s: s.split(" ")
sys.modules["timm.data.transforms_factory"].create_transform = lambda **kw: (lambda img: None)
import src.data.data as D  # noqa: E402
from fake_tokenizer import WORDS, FakeBertTokenizer  # noqa: E402


def jsonable(x):
    if isinstance(x, torch.Tensor):
        return {"dtype": str(x.dtype).replace("torch.", ""), "shape": list(x.shape), "data": x.flatten().tolist()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (np.integer,)):
        return int(x)
    return x


def text_rows(rs, n):
    def text(k):
        return " ".join(rs.choice(WORDS, size=k))

    def pvs():
        return ";".join(f"{rs.choice(WORDS)}:{rs.choice(WORDS)}" for _ in range(rs.randint(0, 5)))
    rows = []
    for i in range(n):
        rows.append([str(int(rs.randint(2))), f"i{rs.randint(40)}", int(rs.randint(3)), text(rs.randint(1, 12)), pvs(),
                     f"i{rs.randint(40)}", int(rs.randint(3)), text(rs.randint(1, 12)), pvs()])
    return rows


def main():
    rs = np.random.RandomState(11)
    tk = FakeBertTokenizer()
    out = {"cases": []}

    def add(name, dataset, collate, rows, ctor):
        recs = [dataset[i] for i in range(len(dataset))]
        out["cases"].append({"name": name, "ctor": ctor, "rows": rows, "records": jsonable(recs), "batch": jsonable(collate(recs))})

    rows = text_rows(rs, 6)
    for method, pv, aux in (("cls", 12, False), ("vec_sim", 12, False), ("cls", None, False), ("cls", 16, True)):
        kw = dict(max_seq_len=8, classification_method=method, max_seq_len_pv=pv, auxiliary_task=aux)
        # the attribute-pair extraction raises on some malformed rows in the reference (pinned separately by pair_indices.json)
        use = rows
        if aux:
            use = []
            for r in rows:
                try:
                    D.RobertaOneTowerDataset([tuple(r)], tk, 8, method, max_seq_len_pv=pv, auxiliary_task=True)[0]
                    use.append(r)
                except Exception:
                    pass
        add(f"roberta_one_tower_{method}_pv{pv}_aux{int(aux)}", D.RobertaOneTowerDataset([tuple(r) for r in use], tk, 8, method, max_seq_len_pv=pv, auxiliary_task=aux),
            D.collate_one_tower, use, dict(cls="RobertaOneTowerDataset", kw=kw))
    for pv in (12, None):
        add(f"roberta_two_tower_pv{pv}", D.RobertaTwoTowerDataset([tuple(r) for r in rows], tk, 8, max_seq_len_pv=pv), D.collate_two_tower, rows,
            dict(cls="RobertaTwoTowerDataset", kw=dict(max_seq_en=8, max_seq_len_pv=pv)))

    # PKGM: entity / relation "tokenizers" are plain dicts (reference finetune_text.py builds them from the KG vocabulary files)
    ent = {f"/item/i{k}": k + 1 for k in range(40)}
    rel = {w: j + 1 for j, w in enumerate(WORDS)}
    # (an item without any attribute trips the reference's own length assertion, data.py:355: PKGM rows carry at least one)
    krows = [r for r in text_rows(rs, 16) if r[4] and r[8]][:6]
    for method in ("cls", "vec_sim"):
        add(f"pkgm_one_tower_{method}", D.PKGMOneTowerDataset([tuple(r) for r in krows], tk, ent, rel, 8, 3, method), D.collate_one_tower, krows,
            dict(cls="PKGMOneTowerDataset", kw=dict(max_seq_en=8, max_pvs=3, classification_method=method), kg=True))
    add("pkgm_two_tower", D.PKGMTwoTowerDataset([tuple(r) for r in krows], tk, ent, rel, 8, 3), D.collate_two_tower, krows,
        dict(cls="PKGMTwoTowerDataset", kw=dict(max_seq_en=8, max_pvs=3), kg=True))

    # RoBERTa + pre-extracted image embeddings: rows carry the embedding as "a,b,c" text (reference data.py:669)
    def emb():
        return ",".join(f"{v:.4f}" for v in rs.randn(5))
    irows = [[r[0], r[1], r[3], r[4], emb(), r[5], r[7], r[8], emb()] for r in rows]
    for ens in ("begin", "end"):
        for pv in (12, None):
            add(f"roberta_image_one_tower_{ens}_pv{pv}", D.RobertaImageOneTowerDataset([tuple(r) for r in irows], tk, 8, ens, max_seq_len_pv=pv),
                D.collate_multimodal, irows, dict(cls="RobertaImageOneTowerDataset", kw=dict(max_seq_len=8, ensemble=ens, max_seq_len_pv=pv)))
        add(f"roberta_image_two_tower_{ens}", D.RobertaImageTwoTowerDataset([tuple(r) for r in irows], tk, 8, ens, max_seq_len_pv=12),
            D.collate_multimodal_two_tower, irows, dict(cls="RobertaImageTwoTowerDataset", kw=dict(max_seq_len=8, ensemble=ens, max_seq_len_pv=12)))

    # image-carrying collates on hand-made records (the image tensors are the transform's output; what is pinned is the tuple
    # order, the dtypes and that samples whose image failed to load are dropped -- reference data.py:37-95)
    def img(k):
        return torch.arange(3 * 4 * 4, dtype=torch.float32).reshape(3, 4, 4) + 100 * k
    coca, image = [], []
    for i, r in enumerate(rows):
        ds = D.PairedMultimodalDataset([(r[0], r[1], r[3], r[4], "/nonexistent.jpg", r[5], r[7], r[8], "/nonexistent.jpg")], "sum", 16, False, tk, 8,
                                       max_seq_len_pv=12)
        rec = ds[0]                                               # text part from the reference's dataset; no image (open fails)
        assert "src_image" not in rec
        if i != 2:                                                # sample 2 stays without images: the collate must drop it
            rec["src_image"], rec["tgt_image"] = img(2 * i), img(2 * i + 1)
        coca.append(rec)
        irec = {"labels": int(r[0]), "src_item_id": r[1], "tgt_item_id": r[5]}
        if i != 4:
            irec["src_input"], irec["tgt_input"] = img(2 * i), img(2 * i + 1)
        image.append(irec)
    out["cases"].append({"name": "coca_pair_sum", "ctor": dict(cls="PairedMultimodalDataset", kw=dict(ensemble="sum", image_size=16, is_training=False,
                         max_seq_len=8, max_seq_len_pv=12)), "rows": rows, "records": jsonable(coca), "batch": jsonable(D.collate_coca_pair(coca)),
                         "missing_image": [2]})
    out["cases"].append({"name": "paired_image", "ctor": dict(cls="PairedImageDataset", kw={}), "rows": rows, "records": jsonable(image),
                         "batch": jsonable(D.collate_image(image)), "missing_image": [4]})
    ds = D.PairedMultimodalDataset([(r[0], r[1], r[3], r[4], "/nonexistent.jpg", r[5], r[7], r[8], "/nonexistent.jpg") for r in rows], "cross_attn", 16,
                                   False, tk, 8, max_seq_len_pv=None)
    out["cases"].append({"name": "coca_pair_cross_attn_text_only", "ctor": dict(cls="PairedMultimodalDataset", kw=dict(ensemble="cross_attn", image_size=16,
                         is_training=False, max_seq_len=8, max_seq_len_pv=None)), "rows": rows, "records": jsonable([ds[i] for i in range(len(ds))]), "batch": None})
    path = os.path.join(ROOT, "tests", "golden", "collates.json")
    json.dump(out, open(path, "w"), ensure_ascii=False)
    print("wrote", path, len(out["cases"]), "cases", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
