"""Deterministic synthetic item pairs in the tensor layouts the reference's collate functions produce
(src/data/data.py:37-74 collate_coca_pair, :172-240 collate_one_tower / collate_two_tower, :77-95
collate_image) — no dataset, tokenizer vocabulary or images exist offline.

Text: title_len ~ U{8..48}, pv_len ~ U{16..203}; ids = [CLS] title [SEP] pvs [SEP], right-padded with 0 to
max_seq_len + max_seq_len_pv (BERT-zh vocab 21128: PAD 0, CLS 101, SEP 102; draws equal to a special id are
remapped to 1000).  Images: N(0,1) fp32 NCHW (what timm's eval transform yields after mean/std).  Seed 2345
is the reference default (finetune_multimodal.py:57).
"""
import numpy as np
import torch

PAD, CLS, SEP, IMG_TOKEN = 0, 101, 102, 99
VOCAB = 21128


def _item_tokens(rs, max_title=50, max_pv=205, full_length=False):
    L = max_title + max_pv
    if full_length:
        tl, pl = max_title - 2, max_pv - 1
    else:
        tl, pl = int(rs.randint(8, max_title - 1)), int(rs.randint(16, max_pv - 1))
    body = rs.randint(1, VOCAB, size=tl + pl)
    body[np.isin(body, (PAD, IMG_TOKEN, CLS, SEP))] = 1000
    ids = np.concatenate([[CLS], body[:tl], [SEP], body[tl:], [SEP]])
    out = np.zeros(L, dtype=np.int64)
    out[:len(ids)] = ids
    return out


def two_tower_text(rs, n_pairs, max_title=50, max_pv=205, full_length=False):
    """-> dict of [n, L] int64 arrays for both towers (token_type_ids all 0, position_ids None)."""
    a = np.stack([_item_tokens(rs, max_title, max_pv, full_length) for _ in range(n_pairs)])
    b = np.stack([_item_tokens(rs, max_title, max_pv, full_length) for _ in range(n_pairs)])
    return dict(input_ids_1=a, attention_mask_1=(a != 0).astype(np.int64), token_type_ids_1=np.zeros_like(a),
                input_ids_2=b, attention_mask_2=(b != 0).astype(np.int64), token_type_ids_2=np.zeros_like(b))


def one_tower_text(rs, n_pairs, max_title=50, max_pv=205, full_length=False):
    """[CLS] src [SEP] src_pv [SEP] tgt [SEP] tgt_pv [SEP] padded to 2*(max_title+max_pv); segment ids 0 / 1
    (what tokenizer(text, text_pair, max_length=2L, padding=max_length) yields, reference data.py:558-559)."""
    L = 2 * (max_title + max_pv)
    ids = np.zeros((n_pairs, L), dtype=np.int64)
    tt = np.zeros((n_pairs, L), dtype=np.int64)
    for i in range(n_pairs):
        a = _item_tokens(rs, max_title, max_pv, full_length)
        b = _item_tokens(rs, max_title, max_pv, full_length)
        a, b = a[a != 0], b[b != 0][1:]               # second segment has no [CLS]
        ids[i, :len(a)] = a
        ids[i, len(a):len(a) + len(b)] = b
        tt[i, len(a):len(a) + len(b)] = 1
    return dict(input_ids=ids, attention_mask=(ids != 0).astype(np.int64), token_type_ids=tt)


class SyntheticCocaPairs:
    """Batches for CoCaForItemAlignment in collate_coca_pair order (reference data.py:73-74, after the two id lists):
    input_ids_1, attention_mask_1, token_type_ids_1, position_ids_1, images_1, ..._2, labels."""

    def __init__(self, n_pairs, image_size=384, max_title=50, max_pv=205, seed=2345, full_length=False):
        rs = np.random.RandomState(seed)
        self.text = two_tower_text(rs, n_pairs, max_title, max_pv, full_length)
        self.labels = rs.randint(0, 2, size=n_pairs).astype(np.int64)
        self.n, self.image_size, self.seed = n_pairs, image_size, seed

    def batch(self, indices, device, device_images=False):
        """device_images: draw the N(0,1) images with the device's generator (fast for many large resident batches; a different
        stream than the host generator, still a function of (seed, first index) only)."""
        idx = np.asarray(indices)
        t = {k: torch.from_numpy(v[idx]).to(device) for k, v in self.text.items()}
        S = self.image_size
        if device_images and torch.device(device).type == "cuda":
            g = torch.Generator(device=device).manual_seed(self.seed * 7919 + int(idx[0]))
            im1 = torch.randn((len(idx), 3, S, S), generator=g, device=device)
            im2 = torch.randn((len(idx), 3, S, S), generator=g, device=device)
        else:
            g = torch.Generator(device="cpu").manual_seed(self.seed * 7919 + int(idx[0]))
            im1 = torch.randn((len(idx), 3, S, S), generator=g).to(device)
            im2 = torch.randn((len(idx), 3, S, S), generator=g).to(device)
        labels = torch.from_numpy(self.labels[idx]).to(device)
        # PairedMultimodalDataset emits explicit position ids 0..L-1 (reference data.py:979,984)
        pos = torch.arange(t["input_ids_1"].shape[1], device=device).unsqueeze(0).expand(len(idx), -1).contiguous()
        return (t["input_ids_1"], t["attention_mask_1"], t["token_type_ids_1"], pos, im1,
                t["input_ids_2"], t["attention_mask_2"], t["token_type_ids_2"], pos, im2, labels)
