"""Datasets and collate functions producing the tensor layouts of the reference's src/data/data.py.

Scope: the *formats* feeding the train step are part of the hot-path boundary —
every collate_* below returns the reference's tuple order (two id lists first, then tensors / None), so the
train scripts can slice batch[2:] exactly as the reference does.  Tokenisation itself is host-side plumbing:
the tokenizer is whatever object the script passes in (transformers.BertTokenizer when a vocab.txt exists);
jieba word segmentation is used when importable and skipped otherwise; images are decoded with PIL and go through
data/transforms.py ImageTransform (the reference's timm create_transform: evaluation = bilinear Resize + CenterCrop, training =
RandomResizedCrop + flip + colour jitter), on the host or - with raw=True / --gpu_preproc - on the GPU (data/gpu_preproc.py).
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from .transforms import ImageTransform

IMG_TOKEN = "[unused99]"
IMG_TOKEN_ID = 99

try:                                    # reference data.py:1
    import jieba

    def _cut(text):
        return " ".join(jieba.cut(text))
except Exception:                       # not installed in this image: keep the raw attribute string
    def _cut(text):
        return text


def _tok(tokenizer, text, max_length, text_pair=None):
    out = tokenizer(text=text, text_pair=text_pair, max_length=max_length, padding="max_length", truncation="longest_first")
    return out.data if hasattr(out, "data") else out


def _item_text(tokenizer, title, pvs, max_seq_len, max_seq_len_pv):
    """reference data.py:533-545 / :799-806: title [SEP] segmented-pvs, and the padded length."""
    if max_seq_len is None:
        return pvs, max_seq_len_pv
    if max_seq_len_pv is None:
        return title, max_seq_len
    return " ".join((title, tokenizer.sep_token, _cut(pvs))), max_seq_len + max_seq_len_pv


# ------------------------------------------------------------------------------------------------ collates
def _t(x, dtype=torch.long):
    return torch.tensor(x, dtype=dtype)


def collate_one_tower(inputs):
    """reference data.py:172-201."""
    pos = [i["position_ids"] for i in inputs if "position_ids" in i]
    pair_indices = [_t(i["pair_indices"]) for i in inputs if "pair_indices" in i]
    return ([i["src_item_id"] for i in inputs], [i["tgt_item_id"] for i in inputs], pair_indices,
            _t([i["input_ids"] for i in inputs]), _t([i["token_type_ids"] for i in inputs]),
            _t([i["attention_mask"] for i in inputs]), _t(pos) if pos else None, _t([i["labels"] for i in inputs]))


def collate_two_tower(inputs):
    """reference data.py:204-240."""
    pos = [i["position_ids"] for i in inputs if "position_ids" in i]
    return ([i["src_item_id"] for i in inputs], [i["tgt_item_id"] for i in inputs],
            _t([i["input_ids_1"] for i in inputs]), _t([i["attention_mask_1"] for i in inputs]), _t([i["token_type_ids_1"] for i in inputs]),
            _t([i["input_ids_2"] for i in inputs]), _t([i["attention_mask_2"] for i in inputs]), _t([i["token_type_ids_2"] for i in inputs]),
            _t(pos) if pos else None, _t([i["labels"] for i in inputs]))


def collate_image(inputs):
    """reference data.py:77-95: samples whose image failed to load are dropped."""
    keep = [i for i in inputs if "src_input" in i and "tgt_input" in i]
    return ([i["src_item_id"] for i in keep], [i["tgt_item_id"] for i in keep], stack_images([i["src_input"] for i in keep]),
            stack_images([i["tgt_input"] for i in keep]), _t([i["labels"] for i in keep]))


def collate_multimodal(inputs):
    """reference data.py:98-128 (RoBERTa + image embeddings, one tower)."""
    pos = [i["position_ids"] for i in inputs if "position_ids" in i]
    return ([i["src_item_id"] for i in inputs], [i["tgt_item_id"] for i in inputs], _t([i["image_index"] for i in inputs if "image_index" in i]),
            _t([i["src_img_emb"] for i in inputs], torch.float32), _t([i["tgt_img_emb"] for i in inputs], torch.float32),
            _t([i["input_ids"] for i in inputs]), _t([i["token_type_ids"] for i in inputs]), _t([i["attention_mask"] for i in inputs]),
            _t(pos) if pos else None, _t([i["labels"] for i in inputs]))


def collate_multimodal_two_tower(inputs):
    """reference data.py:131-169 (the same position_ids tensor, or None, sits in both towers' slots)."""
    pos = [i["position_ids"] for i in inputs if "position_ids" in i]
    pos = _t(pos) if pos else None
    return ([i["src_item_id"] for i in inputs], [i["tgt_item_id"] for i in inputs],
            _t([i["input_ids_1"] for i in inputs]), _t([i["attention_mask_1"] for i in inputs]), _t([i["token_type_ids_1"] for i in inputs]), pos,
            _t([i["src_img_emb"] for i in inputs], torch.float32),
            _t([i["input_ids_2"] for i in inputs]), _t([i["attention_mask_2"] for i in inputs]), _t([i["token_type_ids_2"] for i in inputs]), pos,
            _t([i["tgt_img_emb"] for i in inputs], torch.float32), _t([i["labels"] for i in inputs]))


def collate_coca_pair(inputs):
    """reference data.py:37-74."""
    keep = [i for i in inputs if "src_image" in i and "tgt_image" in i]
    spos = [i["src_position_ids"] for i in keep if "src_position_ids" in i]
    tpos = [i["tgt_position_ids"] for i in keep if "tgt_position_ids" in i]
    return ([i["src_item_id"] for i in keep], [i["tgt_item_id"] for i in keep],
            _t([i["src_input_ids"] for i in keep]), _t([i["src_attention_mask"] for i in keep]), _t([i["src_token_type_ids"] for i in keep]),
            _t(spos) if spos else None, stack_images([i["src_image"] for i in keep]),
            _t([i["tgt_input_ids"] for i in keep]), _t([i["tgt_attention_mask"] for i in keep]), _t([i["tgt_token_type_ids"] for i in keep]),
            _t(tpos) if tpos else None, stack_images([i["tgt_image"] for i in keep]), _t([i["labels"] for i in keep]))


# ------------------------------------------------------------------------------------------------ datasets
COLON_ID, SEMICOLON_ID = 131, 132        # ":" and ";" in the BERT-zh vocabulary (reference data.py:11-12)


def _next_attribute(ids, p, colon, semicolon):
    """Scan ids[p:] up to the next ";".  Returns (found, p', colon, prev_semicolon, semicolon): the last ":" seen so far (kept
    from earlier attributes when this one has none, as the reference's running variables do), the previous and the new ";"."""
    while p < len(ids):
        if ids[p] == COLON_ID:
            colon = p
        elif ids[p] == SEMICOLON_ID:
            return True, p + 1, colon, semicolon, p
        p += 1
    return False, p, colon, None, semicolon


def attribute_pair_indices(input_ids, sep_token_id):
    """reference data.py:568-612 (`--auxiliary_task`): walk the `key:value;` attributes of the source and the target item in
    step while their keys agree; one row (src start, src end, tgt start, tgt end, values equal) per aligned attribute, positions
    in the one-tower sequence [CLS] title [SEP] attributes [SEP] title [SEP] attributes [SEP]."""
    seps = [i for i, t in enumerate(input_ids) if t == sep_token_id]
    src_off, tgt_off = seps[0] + 1, seps[2] + 1
    src, tgt = input_ids[seps[0] + 1:seps[1]], input_ids[seps[2] + 1:seps[3]]
    out = []
    sp = tp = 0
    s_colon = t_colon = None
    s_semi = t_semi = -1
    while sp < len(src) and tp < len(tgt):
        ok, sp, s_colon, s_prev, s_semi = _next_attribute(src, sp, s_colon, s_semi)
        if not ok:
            break
        ok, tp, t_colon, t_prev, t_semi = _next_attribute(tgt, tp, t_colon, t_semi)
        if not ok:
            break
        # values are sliced before the keys are compared, as in the reference: an attribute without any ':' so far raises here
        same = src[s_colon + 1:s_semi] == tgt[t_colon + 1:t_semi]
        if src[s_prev + 1:s_colon] != tgt[t_prev + 1:t_colon]:          # keys differ: stop
            break
        out.append([s_prev + 1 + src_off, s_semi + src_off, t_prev + 1 + tgt_off, t_semi + tgt_off, 1 if same else 0])
    return out


class RobertaOneTowerDataset(Dataset):
    """reference data.py:519-620 (with `auxiliary_task`, records also carry `pair_indices`, :568-612)."""

    def __init__(self, data, text_tokenizer, max_seq_len, classification_method, max_seq_len_pv=None, auxiliary_task=False):
        self.data, self.tk = data, text_tokenizer
        self.max_seq_len, self.max_seq_len_pv, self.method = max_seq_len, max_seq_len_pv, classification_method
        self.auxiliary_task = auxiliary_task

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        label, src_id, _sc, src_title, src_pvs, tgt_id, _tc, tgt_title, tgt_pvs = self.data[item]
        src_text, L = _item_text(self.tk, src_title, src_pvs, self.max_seq_len, self.max_seq_len_pv)
        tgt_text, _ = _item_text(self.tk, tgt_title, tgt_pvs, self.max_seq_len, self.max_seq_len_pv)
        if self.method == "vec_sim":
            s, t = _tok(self.tk, src_text, L), _tok(self.tk, tgt_text, L)
            rec = {"input_ids": s["input_ids"] + [self.tk.bos_token_id] + t["input_ids"][1:],
                   "token_type_ids": s["token_type_ids"] + [x + 1 for x in t["token_type_ids"]],
                   "attention_mask": s["attention_mask"] + t["attention_mask"]}
        else:
            rec = dict(_tok(self.tk, src_text, 2 * L, text_pair=tgt_text))
        rec.update(labels=int(label), src_item_id=src_id, tgt_item_id=tgt_id)
        if self.auxiliary_task:
            rec["pair_indices"] = attribute_pair_indices(rec["input_ids"], self.tk.sep_token_id)
        return rec


class RobertaTwoTowerDataset(Dataset):
    """reference data.py:786-832."""

    def __init__(self, data, text_tokenizer, max_seq_en, max_seq_len_pv=None):
        self.data, self.tk, self.max_seq_len, self.max_seq_len_pv = data, text_tokenizer, max_seq_en, max_seq_len_pv

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        label, src_id, _sc, src_title, src_pvs, tgt_id, _tc, tgt_title, tgt_pvs = self.data[item]
        pv = self.max_seq_len_pv
        src_text, L = _item_text(self.tk, src_title, src_pvs, self.max_seq_len, pv) if pv is not None else (src_title, self.max_seq_len)
        tgt_text, _ = _item_text(self.tk, tgt_title, tgt_pvs, self.max_seq_len, pv) if pv is not None else (tgt_title, self.max_seq_len)
        s, t = _tok(self.tk, src_text, L), _tok(self.tk, tgt_text, L)
        return {"input_ids_1": s["input_ids"], "token_type_ids_1": s["token_type_ids"], "attention_mask_1": s["attention_mask"],
                "input_ids_2": t["input_ids"], "token_type_ids_2": t["token_type_ids"], "attention_mask_2": t["attention_mask"],
                "labels": int(label), "src_item_id": src_id, "tgt_item_id": tgt_id}


class _PKGMBase(Dataset):
    def __init__(self, data, text_tokenizer, kg_entity_tokenizer, kg_relation_tokenizer, max_seq_en, max_pvs, classification_method="cls"):
        self.data, self.tk = data, text_tokenizer
        self.ent, self.rel = kg_entity_tokenizer, kg_relation_tokenizer
        self.max_seq_len, self.max_pvs, self.method = max_seq_en, max_pvs, classification_method

    def __len__(self):
        return len(self.data)

    def _kg_ids(self, item_id, pvs):
        """reference data.py:300-318: relation ids of the attributes, preceded by the item's entity id."""
        ids = []
        for pv in pvs.split(";"):
            try:
                r, _ = pv.split(":", maxsplit=1)
            except ValueError:
                continue
            ids.append(self.rel[r])
        if ids:
            ids.insert(0, self.ent[f"/item/{item_id}"])
        return ids[:1 + self.max_pvs]

    def _text(self, title, first_id, type_id):
        """reference pad_text_sequence (data.py:370-378): padding positions carry mask 0 AND token type 0"""
        ids = self.tk.convert_tokens_to_ids(self.tk.tokenize(title))[:self.max_seq_len - 2]
        ids = [first_id] + ids + [self.tk.sep_token_id]
        n = len(ids)
        pad = self.max_seq_len - n
        return ids + [0] * pad, [1] * n + [0] * pad, [type_id] * n + [0] * pad

    def _kg(self, ids, type_id):
        """reference pad_kg_sequence (data.py:380-389): 1 entity + max_pvs relations as ids; masks / types cover 2*max_pvs embedded
        rows, padding rows typed 0.  An item without any attribute has no entity id either and trips the reference's own length
        assertion (data.py:351-356): the same AssertionError here."""
        assert len(ids) >= 1, "PKGM sample without attributes (reference data.py:355 asserts on it too)"
        n_rel = len(ids) - 1
        ids = ids + [0] * (1 + self.max_pvs - len(ids))
        mask = [1] * (2 * n_rel) + [0] * (2 * (self.max_pvs - n_rel))
        return ids, mask, [type_id] * (2 * n_rel) + [0] * (2 * (self.max_pvs - n_rel))


class PKGMOneTowerDataset(_PKGMBase):
    """reference data.py:277-391: ids [B, 2*(S+P+1)], mask / types / positions [B, 2*(S+2P)]."""

    def __getitem__(self, item):
        label, src_id, _sc, src_title, src_pvs, tgt_id, _tc, tgt_title, tgt_pvs = self.data[item]
        st, sm, stt = self._text(src_title, self.tk.cls_token_id, 0)
        first = self.tk.bos_token_id if self.method == "vec_sim" else self.tk.sep_token_id
        tt, tm, ttt = self._text(tgt_title, first, 1)
        sk, skm, skt = self._kg(self._kg_ids(src_id, src_pvs), 0)
        tk_, tkm, tkt = self._kg(self._kg_ids(tgt_id, tgt_pvs), 1)
        rec = {"input_ids": st + sk + tt + tk_, "attention_mask": sm + skm + tm + tkm, "token_type_ids": stt + skt + ttt + tkt,
               "position_ids": list(range(2 * (self.max_seq_len + 2 * self.max_pvs))), "labels": int(label), "src_item_id": src_id,
               "tgt_item_id": tgt_id}
        assert len(rec["input_ids"]) == 2 * (self.max_seq_len + self.max_pvs + 1)
        assert len(rec["token_type_ids"]) == len(rec["position_ids"]) == len(rec["attention_mask"])
        return rec


class PKGMTwoTowerDataset(_PKGMBase):
    """reference data.py:394-516."""

    def __getitem__(self, item):
        label, src_id, _sc, src_title, src_pvs, tgt_id, _tc, tgt_title, tgt_pvs = self.data[item]
        st, sm, stt = self._text(src_title, self.tk.cls_token_id, 0)
        tt, tm, ttt = self._text(tgt_title, self.tk.cls_token_id, 0)
        sk, skm, skt = self._kg(self._kg_ids(src_id, src_pvs), 1)              # data.py:432 / :465: KG rows are typed 1 in both towers
        tk_, tkm, tkt = self._kg(self._kg_ids(tgt_id, tgt_pvs), 1)
        return {"input_ids_1": st + sk, "attention_mask_1": sm + skm, "token_type_ids_1": stt + skt,
                "input_ids_2": tt + tk_, "attention_mask_2": tm + tkm, "token_type_ids_2": ttt + tkt,
                "position_ids": list(range(self.max_seq_len + 2 * self.max_pvs)), "labels": int(label), "src_item_id": src_id,
                "tgt_item_id": tgt_id}


def _img_emb(text):
    """reference data.py:669: the row carries the pre-extracted image embedding as comma-separated floats (a JSON list is accepted too)"""
    if not isinstance(text, str):
        return [float(v) for v in text]
    return [float(v) for v in text.strip().lstrip("[").rstrip("]").split(",")]


def _image_item_text(tokenizer, title, pvs, max_seq_len, max_seq_len_pv, ensemble):
    text, L = _item_text(tokenizer, title, pvs, max_seq_len, max_seq_len_pv)
    if ensemble == "begin":                                   # data.py:650-652: [unused99] [SEP] in front of the item's text
        text = " ".join((IMG_TOKEN, tokenizer.sep_token, text))
    return text, L


class RobertaImageOneTowerDataset(Dataset):
    """reference data.py:623-679: the text pair, with `ensemble="begin"` an [unused99] [SEP] image slot in front of each item's text,
    + the two pre-extracted image embeddings; `image_index` = position of the SECOND image token (begin only)."""

    def __init__(self, data, text_tokenizer, max_seq_len, ensemble, max_seq_len_pv=None):
        self.data, self.tk, self.max_seq_len, self.max_seq_len_pv, self.ensemble = data, text_tokenizer, max_seq_len, max_seq_len_pv, ensemble

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        label, src_id, src_title, src_pvs, src_emb, tgt_id, tgt_title, tgt_pvs, tgt_emb = self.data[item]
        src_text, L = _image_item_text(self.tk, src_title, src_pvs, self.max_seq_len, self.max_seq_len_pv, self.ensemble)
        tgt_text, _ = _image_item_text(self.tk, tgt_title, tgt_pvs, self.max_seq_len, self.max_seq_len_pv, self.ensemble)
        rec = dict(_tok(self.tk, src_text, 2 * L, text_pair=tgt_text))
        rec.update(labels=int(label), src_item_id=src_id, tgt_item_id=tgt_id, src_img_emb=_img_emb(src_emb), tgt_img_emb=_img_emb(tgt_emb))
        if self.ensemble == "begin":
            first = rec["input_ids"].index(IMG_TOKEN_ID)
            rec["image_index"] = rec["input_ids"].index(IMG_TOKEN_ID, first + 1)      # ValueError when truncation lost it, as in the reference
        return rec


class RobertaImageTwoTowerDataset(Dataset):
    """reference data.py:682-753."""

    def __init__(self, data, text_tokenizer, max_seq_len, ensemble, max_seq_len_pv=None):
        self.data, self.tk, self.max_seq_len, self.max_seq_len_pv, self.ensemble = data, text_tokenizer, max_seq_len, max_seq_len_pv, ensemble

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        label, src_id, src_title, src_pvs, src_emb, tgt_id, tgt_title, tgt_pvs, tgt_emb = self.data[item]
        src_text, L = _image_item_text(self.tk, src_title, src_pvs, self.max_seq_len, self.max_seq_len_pv, self.ensemble)
        tgt_text, _ = _image_item_text(self.tk, tgt_title, tgt_pvs, self.max_seq_len, self.max_seq_len_pv, self.ensemble)
        s, t = _tok(self.tk, src_text, L), _tok(self.tk, tgt_text, L)
        return {"input_ids_1": s["input_ids"], "input_ids_2": t["input_ids"], "token_type_ids_1": s["token_type_ids"],
                "token_type_ids_2": t["token_type_ids"], "attention_mask_1": s["attention_mask"], "attention_mask_2": t["attention_mask"],
                "labels": int(label), "src_item_id": src_id, "tgt_item_id": tgt_id, "src_img_emb": _img_emb(src_emb),
                "tgt_img_emb": _img_emb(tgt_emb), "image_index": 0}


class RawImage:
    """Decoded but unprocessed frame for the GPU pipeline (data/gpu_preproc.py): uint8 [H, W, 3] + the transform's random draw."""
    __slots__ = ("u8", "params")

    def __init__(self, u8, params):
        self.u8, self.params = u8, params

    @property
    def flip(self):
        return bool(self.params.flip)


class RawImageBatch:
    """What a collate function emits for RawImage samples: the frames stay a list (their sizes differ)."""

    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)


def stack_images(items):
    """torch.stack for processed [3,S,S] tensors (the reference path), RawImageBatch for RawImage samples."""
    if items and isinstance(items[0], RawImage):
        return RawImageBatch(items)
    return torch.stack(items)


def load_image(path, transform, raw=False):
    """PIL open -> RGB -> `transform` (data/transforms.py ImageTransform = the reference's timm create_transform, data.py:838-866):
    CHW fp32.  raw=True stops after the decode and returns a RawImage carrying the transform's random parameters: crop / resize /
    flip / jitter / normalisation then run on the GPU (same arithmetic, gpu_preproc.py)."""
    from PIL import Image
    img = Image.open(path).convert("RGB")
    params = transform.draw(img.width, img.height)
    if raw:
        return RawImage(torch.from_numpy(np.array(img, dtype=np.uint8)), params)
    return transform.apply(img, params)


class PairedImageDataset(Dataset):
    """reference data.py:835-869."""

    def __init__(self, data, input_size, is_training, hflip=0.5, color_jitter=None, raw=False, seed=None):
        self.data, self.size, self.raw = data, input_size, raw
        self.transform = ImageTransform(input_size, is_training, hflip, color_jitter, seed)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        label, src_id, src_path, tgt_id, tgt_path = self.data[item]
        rec = {"labels": int(label), "src_item_id": src_id, "tgt_item_id": tgt_id}
        try:
            rec["src_input"] = load_image(src_path, self.transform, raw=self.raw)
            rec["tgt_input"] = load_image(tgt_path, self.transform, raw=self.raw)
        except Exception:
            pass                        # the collate drops samples without images (reference data.py:848-860, :84)
        return rec


class PairedMultimodalDataset(Dataset):
    """reference data.py:918-989 (CoCa pairs: text ids with explicit position ids 0..L-1 + two images)."""

    def __init__(self, data, ensemble, image_size, is_training, text_tokenizer, max_seq_len, max_seq_len_pv=None, hflip=0.5,
                 color_jitter=None, raw=False, seed=None):
        self.data, self.ensemble, self.size, self.raw = data, ensemble, image_size, raw
        self.transform = ImageTransform(image_size, is_training, hflip, color_jitter, seed)
        self.tk, self.max_seq_len, self.max_seq_len_pv = text_tokenizer, max_seq_len, max_seq_len_pv

    def __len__(self):
        return len(self.data)

    def __getitem__(self, item):
        label, src_id, src_title, src_pvs, src_path, tgt_id, tgt_title, tgt_pvs, tgt_path = self.data[item]
        pv = self.max_seq_len_pv
        src_text, L = _item_text(self.tk, src_title, src_pvs, self.max_seq_len, pv) if pv is not None else (src_title, self.max_seq_len)
        tgt_text, _ = _item_text(self.tk, tgt_title, tgt_pvs, self.max_seq_len, pv) if pv is not None else (tgt_title, self.max_seq_len)
        if self.ensemble == "sum":
            src_text, tgt_text = " ".join((self.tk.bos_token, src_text)), " ".join((self.tk.bos_token, tgt_text))
        s, t = _tok(self.tk, src_text, L), _tok(self.tk, tgt_text, L)
        rec = {"src_item_id": src_id, "tgt_item_id": tgt_id, "labels": int(label),
               "src_input_ids": s["input_ids"], "src_token_type_ids": s["token_type_ids"], "src_attention_mask": s["attention_mask"],
               "src_position_ids": list(range(len(s["input_ids"]))),
               "tgt_input_ids": t["input_ids"], "tgt_token_type_ids": t["token_type_ids"], "tgt_attention_mask": t["attention_mask"],
               "tgt_position_ids": list(range(len(t["input_ids"])))}
        try:
            rec["src_image"] = load_image(src_path, self.transform, raw=self.raw)
            rec["tgt_image"] = load_image(tgt_path, self.transform, raw=self.raw)
        except Exception:
            pass
        return rec
