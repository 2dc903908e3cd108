"""End-to-end runs of the kept CLIs on the GPU with the HIP engine (synthetic files written to a temp dir): data loading,
tokenisation, collate order, train loop with the fused AdamW, eval sweep, checkpoint + prediction files — the drop-in
surface of SURVEY §8(b)(i).  finetune_multimodal.py (CoCa sum, roberta-shaped tiny text tower + vit_base_patch16_224) and
finetune_image.py (eca_nfnet_l0 two-tower, with the GPU input pipeline and decode workers)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_cli_textcnn import ROOT, WORDS, make_data

pytestmark = pytest.mark.gpu


def _images(directory, names, size, seed=0):
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(directory, exist_ok=True)
    for i, n in enumerate(names):
        h, w = (size, size) if i % 3 else (size + 13, size - 7)            # a few frames of another size
        Image.fromarray(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(os.path.join(directory, n), quality=90)


def _run(cmd, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return r


def test_finetune_multimodal_coca_gpu(gpu, tmp_path):
    root = str(tmp_path)
    pre = make_data(root, n_train=16, n_test=8)
    _images(os.path.join(root, "raw", "item_images"), [f"i{k}.jpg" for k in range(40)], 96)
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    json.dump(cfg, open(os.path.join(root, "coca_tiny.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_multimodal.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "coca_tiny.json"), "--model_name", "coca_tiny", "--data_version", "v1", "--interaction_type", "two_tower",
           "--classification_method", "cls", "--ensemble", "sum", "--loss_type", "ce", "--do_train", "--do_eval", "--do_pred",
           "--train_batch_size", "8", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "12", "--max_position_embeddings", "64",
           "--image_size", "224", "--image_model_name", "vit_base_patch16_224", "--gpu_preproc", "--num_workers", "2", "--fp16"]
    r = _run(cmd)
    dirs = [d for d in os.listdir(out) if d.startswith("coca_tiny")]
    assert len(dirs) == 1, os.listdir(out)
    d = os.path.join(out, dirs[0])
    files = os.listdir(d)
    assert any(f.endswith("epoch-0.bin") for f in files), files
    assert "weights.json" in files and "hyperparamter.txt" in files
    pred = [f for f in files if f.endswith(".jsonl")]
    assert pred, files
    lines = [json.loads(l) for l in open(os.path.join(d, pred[0]))]
    assert len(lines) == 8 and set(lines[0]) == {"src_item_id", "src_item_emb", "tgt_item_id", "tgt_item_emb", "threshold"}
    assert "f1=" in r.stderr and "loss:" in r.stderr


@pytest.mark.parametrize("model_name,size", [("eca_nfnet_l0", 64), ("vit_base_patch16_224", 224), ("resnetv2_50", 64), ("resnetv2_50x1_bitm", 64)])
def test_finetune_image_gpu(gpu, tmp_path, model_name, size):
    root = str(tmp_path)
    rs = np.random.RandomState(1)
    items = [f"i{k}" for k in range(12)]
    with open(os.path.join(root, "item_info.jsonl"), "w", encoding="utf-8") as w:
        for it in items:
            w.write(json.dumps({"item_id": it, "item_image_name": it + ".jpg"}) + "\n")
    _images(os.path.join(root, "item_images"), [it + ".jpg" for it in items], 80)
    for name, n in (("item_train_pair.jsonl", 8), ("item_valid_pair.jsonl", 4), ("item_test_pair.jsonl", 4)):
        with open(os.path.join(root, name), "w", encoding="utf-8") as w:
            for _ in range(n):
                a, b = rs.choice(items, 2, replace=False)
                w.write(json.dumps({"src_item_id": a, "tgt_item_id": b, "item_label": str(rs.randint(2))}) + "\n")
    json.dump(dict(hidden_dropout_prob=0.1, num_labels=2), open(os.path.join(root, "img.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_image.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "img.json"), "--model_name", model_name, "--data_version", "v1", "--do_train", "--do_eval", "--do_pred",
           "--train_batch_size", "4", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--image_size", str(size), "--gpu_preproc", "--num_workers", "2"]
    r = _run(cmd)
    dirs = os.listdir(out)
    assert len(dirs) == 1, dirs
    files = os.listdir(os.path.join(out, dirs[0]))
    assert any(f.endswith("epoch-0.bin") for f in files) and "weights.json" in files, files
    assert "f1=" in r.stderr and "loss:" in r.stderr


@pytest.mark.parametrize("interaction,method,measure,loss", [("one_tower", "cls", "NA", "ce"), ("two_tower", "vec_sim", "cosine", "cosine")])
def test_finetune_text_roberta_gpu(gpu, tmp_path, interaction, method, measure, loss):
    root = str(tmp_path)
    pre = make_data(root, n_train=16, n_test=8)
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    json.dump(cfg, open(os.path.join(root, "roberta_tiny.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_text.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "roberta_tiny.json"), "--model_name", "roberta_tiny", "--data_version", "v1", "--interaction_type", interaction,
           "--classification_method", method, "--similarity_measure", measure, "--loss_type", loss, "--do_train", "--do_eval", "--do_pred",
           "--train_batch_size", "8", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "12", "--max_position_embeddings", "64", "--fp16"]
    if interaction == "two_tower":
        cmd.append("--unpad")                      # the unpadded tower path through the CLI
    r = _run(cmd)
    dirs = os.listdir(out)
    assert len(dirs) == 1, dirs
    files = os.listdir(os.path.join(out, dirs[0]))
    assert any(f.endswith("epoch-0.bin") for f in files) and "hyperparamter.txt" in files, files
    assert "f1=" in r.stderr and "loss:" in r.stderr


def test_finetune_text_auxiliary_task_gpu(gpu, tmp_path):
    """`--auxiliary_task` (reference finetune_text.py:82, text.py:66-102,1478-1480): the dataset aligns the `key:value;` attributes
    of both items, the model adds the attribute-pair cross-entropy (span means -> pair head) to the loss."""
    root = str(tmp_path)
    pre = make_data(root, n_train=16, n_test=8)
    rs = np.random.RandomState(3)
    for name, n in (("finetune_train.tsv", 16), ("finetune_test.tsv", 8)):
        with open(os.path.join(root, "processed", "v1", name), "w", encoding="utf-8") as w:
            for _ in range(n):
                a, b = rs.choice(40, 2, replace=False)
                keys = list(rs.choice(WORDS, 3, replace=False))
                pv = lambda: "".join(f"{k}:{rs.choice(WORDS[:3])};" for k in keys)
                w.write("\t".join([str(rs.randint(2)), f"i{a}", " ".join(rs.choice(WORDS, 3)), pv(), f"i{b}", " ".join(rs.choice(WORDS, 3)), pv()]) + "\n")
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    json.dump(cfg, open(os.path.join(root, "roberta_tiny.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_text.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "roberta_tiny.json"), "--model_name", "roberta_tiny", "--data_version", "v1", "--interaction_type", "one_tower",
           "--classification_method", "cls", "--similarity_measure", "NA", "--loss_type", "ce", "--do_train", "--do_eval", "--auxiliary_task",
           "--train_batch_size", "8", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "16", "--max_position_embeddings", "64", "--fp16"]
    r = _run(cmd)
    dirs = os.listdir(out)
    files = os.listdir(os.path.join(out, dirs[0]))
    assert any(f.endswith("epoch-0.bin") for f in files), files
    import torch
    sd = torch.load(os.path.join(out, dirs[0], [f for f in files if f.endswith("epoch-0.bin")][0]), map_location="cpu")
    assert tuple(sd["auxiliary_task.out_proj.weight"].shape) == (2, 256)
    assert "f1=" in r.stderr and "loss:" in r.stderr


@pytest.mark.parametrize("interaction", ["one_tower", "two_tower"])
def test_finetune_text_pkgm_gpu(gpu, tmp_path, interaction):
    """pkgm_* through finetune_text.py: entity / relation id files, KG rows spliced into the sequence (ia_kg_* kernels)."""
    root = str(tmp_path)
    pre = make_data(root, n_train=16, n_test=8)
    with open(os.path.join(root, "processed", "entity2id.txt"), "w", encoding="utf-8") as w:
        for k in range(40):
            w.write(f"/item/i{k}\t{k + 1}\n")
    with open(os.path.join(root, "processed", "relation2id.txt"), "w", encoding="utf-8") as w:
        for k, word in enumerate(WORDS):
            w.write(f"{word}\t{k + 1}\n")
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
               num_entities=41, num_relations=len(WORDS) + 1, kg_embedding_dim=128, entity_projection_bias=False)
    json.dump(cfg, open(os.path.join(root, "pkgm_tiny.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_text.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "pkgm_tiny.json"), "--model_name", "pkgm_tiny", "--data_version", "v1", "--interaction_type", interaction,
           "--classification_method", "cls", "--similarity_measure", "NA", "--loss_type", "ce", "--do_train", "--do_eval", "--do_pred",
           "--train_batch_size", "8", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_pvs", "4", "--max_position_embeddings", "64", "--fp16"]
    r = _run(cmd)
    dirs = os.listdir(out)
    assert len(dirs) == 1, dirs
    files = os.listdir(os.path.join(out, dirs[0]))
    assert any(f.endswith("epoch-0.bin") for f in files), files
    assert "f1=" in r.stderr and "loss:" in r.stderr


def test_finetune_multimodal_coca_cross_attn_gpu(gpu, tmp_path):
    """--ensemble cross_attn through the CLI: a 768-wide 2-layer text tower + vit_base_patch16_224 + one multimodal layer."""
    root = str(tmp_path)
    pre = make_data(root, n_train=8, n_test=4)
    _images(os.path.join(root, "raw", "item_images"), [f"i{k}.jpg" for k in range(40)], 64)
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=768, num_hidden_layers=2, num_attention_heads=12, intermediate_size=1024, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
               num_hidden_layers_multimodal=1, num_attention_heads_multimodal=12, feedforward_multiplication_multimodal=2)
    json.dump(cfg, open(os.path.join(root, "coca_x.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_multimodal.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "coca_x.json"), "--model_name", "coca_x", "--data_version", "v1", "--interaction_type", "two_tower",
           "--classification_method", "cls", "--ensemble", "cross_attn", "--loss_type", "ce", "--do_train", "--do_eval",
           "--train_batch_size", "4", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "12", "--max_position_embeddings", "64",
           "--image_size", "224", "--image_model_name", "vit_base_patch16_224", "--fp16"]
    r = _run(cmd)
    assert "f1=" in r.stderr and "loss:" in r.stderr


@pytest.mark.parametrize("interaction,ensemble", [("one_tower", "begin"), ("two_tower", "begin"), ("one_tower", "end")])
def test_finetune_multimodal_roberta_image_gpu(gpu, tmp_path, interaction, ensemble):
    """roberta_image_* through finetune_multimodal.py: rows carry pre-extracted image embeddings (comma-separated floats, as the reference's files), spliced in as
    tokens (`begin`) or fed to the head (`end`)."""
    root = str(tmp_path)
    pre = make_data(root, n_train=16, n_test=8)
    rs = np.random.RandomState(3)
    D = 64
    for name in ("finetune_train.tsv", "finetune_test.tsv"):       # add the two embedding columns
        path = os.path.join(root, "processed", "v1", name)
        rows = [l.rstrip("\n").split("\t") for l in open(path, encoding="utf-8")]
        with open(path, "w", encoding="utf-8") as w:
            for label, a, at, ap, b, bt, bp in rows:
                # the reference's row format (data.py:669): comma-separated floats
                ea, eb = (",".join(f"{float(x):.4f}" for x in rs.standard_normal(D)) for _ in range(2))
                w.write("\t".join([label, a, at, ap, ea, b, bt, bp, eb]) + "\n")
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    json.dump(cfg, open(os.path.join(root, "ri.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, os.path.join(ROOT, "finetune_multimodal.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "ri.json"), "--model_name", "roberta_image_tiny", "--data_version", "v1", "--interaction_type", interaction,
           "--classification_method", "cls", "--ensemble", ensemble, "--loss_type", "ce", "--do_train", "--do_eval",
           "--train_batch_size", "8", "--eval_batch_size", "4", "--num_train_epochs", "2", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "12", "--max_position_embeddings", "64",
           "--image_hidden_size", str(D), "--fp16"]
    r = _run(cmd)
    assert "f1=" in r.stderr and "loss:" in r.stderr
    if (interaction, ensemble) != ("two_tower", "begin"):
        return
    # uniform model soup over the two epoch checkpoints + prediction with the averaged weights (reference model_soup_multimodal.py)
    import torch
    d = os.path.join(out, os.listdir(out)[0])
    pattern = os.path.join(d, "multimodal_finetune_epoch-{}.bin")
    soup = [sys.executable, os.path.join(ROOT, "model_soup_multimodal.py")] + cmd[2:]
    for flag in ("--do_train", "--do_eval"):
        soup.remove(flag)
    soup += ["--file_state_dict", pattern, "--epochs", "0,1", "--threshold", "0.5"]
    _run(soup)
    files = os.listdir(d)
    assert "multimodal_finetune-uniform_soup-epoch-0,1.bin" in files and "deepAI_result_uniform_soup_threshold=0.5.jsonl" in files, files
    a, b = (torch.load(pattern.format(e), map_location="cpu") for e in (0, 1))
    avg = torch.load(os.path.join(d, "multimodal_finetune-uniform_soup-epoch-0,1.bin"), map_location="cpu")
    k = "classifier.out_proj.weight"
    assert torch.allclose(avg[k], (a[k] + b[k]) / 2)
    assert len(open(os.path.join(d, "deepAI_result_uniform_soup_threshold=0.5.jsonl")).readlines()) == 8


def test_finetune_text_under_torchrun_rccl(gpu, tmp_path):
    """The CLI under torch.distributed.run with the RCCL process group live (one rank, IA_DP_FORCE_COLLECTIVES=1 makes the
    bucket all-reduces, the arena broadcast and the barriers actually run): the multi-GPU train loop on a 1-GPU box."""
    root = str(tmp_path)
    pre = make_data(root, n_train=16, n_test=8)
    vocab_size = len(open(os.path.join(pre, "vocab.txt"), encoding="utf-8").read().split("\n")) - 1
    cfg = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=vocab_size,
               max_position_embeddings=64, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    json.dump(cfg, open(os.path.join(root, "roberta_tiny.json"), "w"))
    out = os.path.join(root, "out")
    os.makedirs(out)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           "29533", os.path.join(ROOT, "finetune_text.py"), "--data_dir", root, "--output_dir", out, "--config_file",
           os.path.join(root, "roberta_tiny.json"), "--model_name", "roberta_tiny", "--data_version", "v1", "--interaction_type", "two_tower",
           "--classification_method", "cls", "--similarity_measure", "NA", "--loss_type", "ce", "--do_train", "--do_eval",
           "--train_batch_size", "8", "--eval_batch_size", "4", "--num_train_epochs", "1", "--learning_rate", "1e-4", "--log_steps", "1",
           "--pretrained_model_path", pre, "--max_seq_len", "8", "--max_seq_len_pv", "12", "--max_position_embeddings", "64", "--fp16"]
    env = dict(os.environ, PYTHONPATH=ROOT, IA_DP_FORCE_COLLECTIVES="1")
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    dirs = os.listdir(out)
    assert len(dirs) == 1 and any(f.endswith("epoch-0.bin") for f in os.listdir(os.path.join(out, dirs[0])))
    assert "f1=" in r.stderr and "loss:" in r.stderr


def test_bench_single_gpu_with_forced_collectives(gpu):
    """bench.py --gpus 1 under IA_DP_FORCE_COLLECTIVES=1 (RCCL initialised, every gradient bucket all-reduced although there is one
    rank: the exact multi-GPU code path on a 1-GPU box): exactly one JSON line on stdout, n_gpus 1."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_DP_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-pmc",
                        "--no-cpu-baseline", "--no-variants", "--pairs-per-gpu", "16"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["value"] > 0
    assert res["roofline"]["bound"] in ("mfma", "hbm") and res["roofline"]["frac"] > 0


def test_bench_two_ranks_under_the_launcher(gpu):
    """bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run --nproc-per-node 2 ... bench.py
    --gpus 2): rank / device selection, the rank-sharded synthetic batches, the bucketed gradient all-reduce, the barrier +
    max-over-ranks timing and rank 0's single JSON line.  One GPU here, so both ranks share it over gloo (IA_DP_BACKEND); on a
    multi-GPU node the same path runs over RCCL with one process per GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_DP_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IA_DP_FORCE_COLLECTIVES"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29733",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-pmc", "--no-cpu-baseline", "--no-variants",
           "--pairs-per-gpu", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 32 and res["config"]["parallelism"] == "dp2"
    assert res["value"] > 0 and res["scaling"] == "weak"
    import math
    assert math.isfinite(res["final_loss"])


def test_bench_gpus_2_launches_itself(gpu):
    """`python bench.py --gpus 2 ...` with NO launcher environment: the parent must start the two ranks itself (as child processes,
    before it touches the GPU), forward rank 0's single JSON line and return 0.  One GPU here: both ranks share it over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA_DP_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IA_DP_FORCE_COLLECTIVES", "GROUP_RANK", "ROLE_RANK",
              "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-pmc", "--no-cpu-baseline",
           "--no-variants", "--pairs-per-gpu", "16"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 32 and res["config"]["parallelism"] == "dp2"
    assert res["value"] > 0 and res["scaling"] == "weak"
