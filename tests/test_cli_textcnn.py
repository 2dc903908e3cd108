"""BASELINE.json configs[0]: TextCNN two_tower cls/ce through the kept finetune_text.py CLI, on synthetic files, on CPU
(plumbing only: data loading, tokenisation, collate order, train loop, eval sweep, checkpoint + prediction files)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORDS = ["手机", "红色", "蓝色", "大号", "小号", "棉", "电池", "型号", "品牌", "华为", "苹果", "颜色", "尺码", "材质", "a1", "b2", "x", "y"]


def make_data(root, n_train=48, n_test=16, seed=0):
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "raw"), exist_ok=True)
    os.makedirs(os.path.join(root, "processed", "v1"), exist_ok=True)
    items = [f"i{k}" for k in range(40)]
    with open(os.path.join(root, "raw", "item_info.jsonl"), "w", encoding="utf-8") as w:
        for it in items:
            w.write(json.dumps({"item_id": it, "cate_name": "c%d" % rs.randint(3), "item_image_name": it + ".jpg"}, ensure_ascii=False) + "\n")
    json.dump({"c0": 0, "c1": 1, "c2": 2}, open(os.path.join(root, "processed", "cate2id.json"), "w"))

    def text(n):
        return " ".join(rs.choice(WORDS, size=n))

    def pvs():
        return ";".join(f"{rs.choice(WORDS)}:{rs.choice(WORDS)}" for _ in range(rs.randint(1, 4)))
    for name, n in (("finetune_train.tsv", n_train), ("finetune_test.tsv", n_test)):
        with open(os.path.join(root, "processed", "v1", name), "w", encoding="utf-8") as w:
            for _ in range(n):
                a, b = rs.choice(items, 2, replace=False)
                w.write("\t".join([str(rs.randint(2)), a, text(4), pvs(), b, text(4), pvs()]) + "\n")
    pre = os.path.join(root, "pretrained")
    os.makedirs(pre, exist_ok=True)
    # ":" / ";" sit at ids 131 / 132 as in the BERT-zh vocabulary (reference data.py:11-12 COLON_ID / SEMICOLON_ID)
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]", "<S>"] + [f"[fill{i}]" for i in range(26)] + [":", ";"] + WORDS
    assert vocab.index(":") == 131 and vocab.index(";") == 132
    open(os.path.join(pre, "vocab.txt"), "w", encoding="utf-8").write("\n".join(vocab) + "\n")
    cfg = dict(hidden_size=32, num_hidden_layers=1, num_attention_heads=1, intermediate_size=64, vocab_size=len(vocab),
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    json.dump(cfg, open(os.path.join(root, "textcnn.json"), "w"))
    return pre


def test_finetune_text_textcnn_cpu(tmp_path):
    root = str(tmp_path)
    pre = make_data(root)
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_text.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "textcnn.json"), "--model_name", "textcnn", "--data_version", "v1", "--interaction_type", "two_tower",
           "--classification_method", "cls", "--similarity_measure", "NA", "--loss_type", "ce", "--do_train", "--do_eval", "--do_pred",
           "--train_batch_size", "16", "--eval_batch_size", "8", "--num_train_epochs", "2", "--learning_rate", "1e-3", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "12", "--max_position_embeddings", "64",
           "--filter_sizes", "1,2,3,5", "--num_filters", "4"]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", PYTHONPATH=ROOT)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = os.path.join(out, "textcnn-v1-two_tower-cls-NA-ce")
    assert os.path.exists(os.path.join(d, "text_finetune_epoch-1.bin"))
    assert os.path.exists(os.path.join(d, "hyperparamter.txt"))
    w = json.load(open(os.path.join(d, "weights.json")))
    assert len(w["w"]) == 2 and len(w["w"][0]) == 2 * 4 * 4
    lines = [json.loads(l) for l in open(os.path.join(d, "deepAI_result_threshold=0.5.jsonl"))]
    assert len(lines) == 16 and set(lines[0]) == {"src_item_id", "src_item_emb", "tgt_item_id", "tgt_item_emb", "threshold"}
    assert "threshold=0.1" in r.stderr and "f1=" in r.stderr and "[Epoch-1 Step-0] loss:" in r.stderr
