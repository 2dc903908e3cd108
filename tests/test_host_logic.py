"""Host-side logic that needs no GPU: schedules, sharding, the bucketed reducer's bookkeeping, synthetic data layouts,
output structs, the loud failure of the HIP models on CPU, and the TextCNN plumbing model against its golden vector."""
import numpy as np
import pytest
import torch

from golden_util import load_case, weights


def test_linear_schedule_matches_transformers():
    from transformers import get_linear_schedule_with_warmup
    from item_alignment_amd.train import linear_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = get_linear_schedule_with_warmup(opt, 30, 100)
    for step in range(100):
        assert abs(sch.get_last_lr()[0] - linear_schedule_with_warmup(step, 30, 100)) < 1e-12
        opt.step(); sch.step()


def test_shard_indices_partition_the_epoch():
    from item_alignment_amd.dist import shard_indices
    for world in (1, 2, 4, 8):
        shards = [shard_indices(1003, r, world, epoch_seed=7) for r in range(world)]
        assert len({len(s) for s in shards}) == 1                      # equal shard sizes (the loss is a batch mean)
        allidx = torch.cat(shards)
        assert len(set(allidx.tolist())) == len(allidx)                # disjoint
    # identical global order at any world size: rank r of world w holds positions r::w of the same permutation
    one = shard_indices(1000, 0, 1, 3)
    assert torch.equal(shard_indices(1000, 1, 4, 3), one[1::4])


def test_bucket_reducer_bookkeeping_single_process():
    from item_alignment_amd.dist import GradBucketReducer
    flat = torch.arange(1000, dtype=torch.float32)
    params, off = [], 0
    for n in (100, 300, 50, 550):
        p = torch.nn.Parameter(torch.zeros(n)); params.append((p, off, n)); off += n
    r = GradBucketReducer(flat, params, bucket_bytes=256 * 4)
    assert r.buckets[0] == (744, 1000) and r.buckets[-1][0] == 0       # buckets run from the end of the arena
    launched = []
    r._launch = lambda i: (launched.append(i), r.launched.__setitem__(i, True))
    r.grads_ready([params[3][0]])                                       # not final (another node may add to it): nothing goes out
    assert launched == []
    r.grads_ready([params[3][0]], final=True)                           # last parameter covers buckets 0,1 fully, 2 partly
    assert launched == [0, 1]
    r.grads_ready([params[2][0], params[1][0]], final=True)
    assert 2 in launched
    r.grads_ready([params[0][0]], final=True)
    assert sorted(launched) == list(range(len(r.buckets)))
    assert r.finish() == 1.0


def test_synthetic_batches_have_the_collate_layouts():
    from item_alignment_amd.data.synthetic import SyntheticCocaPairs, one_tower_text
    d = SyntheticCocaPairs(8, image_size=32, seed=2345)
    b = d.batch([0, 1, 2], "cpu")
    assert len(b) == 11 and b[3].shape == (3, 255) and b[3][0, :3].tolist() == [0, 1, 2]
    ids, mask = b[0], b[1]
    assert ids.shape == (3, 255) and ids.dtype == torch.int64 and (ids[:, 0] == 101).all()
    assert torch.equal(mask, (ids != 0).long())
    assert b[4].shape == (3, 3, 32, 32) and b[10].shape == (3,)
    o = one_tower_text(np.random.RandomState(1), 4)
    assert o["input_ids"].shape == (4, 510) and set(np.unique(o["token_type_ids"])) <= {0, 1}
    first_sep = [int(np.where(r == 102)[0][1]) for r in o["input_ids"]]
    for r, s in zip(o["token_type_ids"], first_sep):
        assert r[:s + 1].sum() == 0 and r[s + 1] == 1


def test_output_struct_access_patterns():
    from item_alignment_amd.models import SequenceClassifierOutput
    o = SequenceClassifierOutput(loss=torch.tensor(1.0), logits=torch.zeros(2, 2), probs=torch.zeros(2))
    assert o.loss is o["loss"] is o[0]
    assert o.src_embeds is None
    assert list(o.keys()) == ["loss", "logits", "probs"]


def test_position_ids_rule():
    from item_alignment_amd.models import create_position_ids_from_input_ids
    ids = torch.tensor([[5, 6, 7, 0, 0], [9, 0, 3, 4, 0]])
    assert create_position_ids_from_input_ids(ids, 0).tolist() == [[1, 2, 3, 0, 0], [1, 0, 2, 3, 0]]


def test_hip_models_fail_loudly_without_gpu():
    import item_alignment_amd.models as M
    from item_alignment_amd._lib import ItemAlignError
    from test_models_gpu import cfg_of
    case = load_case("roberta_one_tower_cls_ce")
    model = M.RobertaOneTower(cfg_of(case))
    i = case.inputs
    with pytest.raises(ItemAlignError):
        model(input_ids=i["input_ids"], attention_mask=i["attention_mask"], token_type_ids=i["token_type_ids"], labels=i["labels"])


def test_textcnn_plumbing_model_matches_reference():
    """BASELINE.json configs[0] (TextCNN two_tower, CPU): the product class against the reference golden vector."""
    import item_alignment_amd.models as M
    from test_models_gpu import cfg_of
    case = load_case("textcnn_two_tower")
    model = M.TextCNNTwoTower(cfg_of(case), {})
    missing, unexpected = model.load_state_dict(weights(case), strict=False)
    assert not unexpected and all("position_ids" in k for k in missing)
    model.eval()
    i = case.inputs
    out = model(input_ids_1=i["input_ids_1"], input_ids_2=i["input_ids_2"], labels=i["labels"])
    for k, want in case.outs.items():
        assert torch.allclose(getattr(out, k), want, atol=1e-5, rtol=1e-4), k
    out.loss.backward()
    p = dict(model.named_parameters())
    for k, want in case.grads.items():
        assert torch.allclose(p[k].grad, want, atol=1e-5, rtol=1e-4), k


def test_nfnet_oracle_structure():
    """The NF-Net restatement (timm is absent: parity unpinned) must at least reproduce the published structure of
    eca_nfnet_l0: 24.14 M parameters with the 1000-way fc (timm model card), 2304 features, ECA kernel sizes from the channel
    count, and the stage bookkeeping the reference mirrors in-tree (image.py:98-137)."""
    import torch
    from oracle import ref_models as O
    cfg = O.nfnet_cfg("eca_nfnet_l0")
    spec = O.nfnet_state_spec(cfg)
    n = sum(int(torch.tensor(s).prod()) for _, s in spec) + 2304 * 1000 + 1000
    assert abs(n / 1e6 - 24.14) < 0.01, n
    assert cfg.num_features == 2304
    plan = O.nfnet_plan(cfg)
    assert [len(s) for s in plan] == [1, 2, 6, 3]
    assert [s[0]["stride"] for s in plan] == [1, 2, 2, 2] and all(b["stride"] == 1 for s in plan for b in s[1:])
    assert [s[0]["mid_chs"] for s in plan] == [64, 128, 384, 384] and [s[0]["groups"] for s in plan] == [1, 2, 6, 6]
    assert plan[0][0]["beta"] == 1.0 and abs(plan[1][1]["beta"] - 1 / (1 + 0.04) ** 0.5) < 1e-12
    assert abs(plan[2][5]["beta"] - 1 / (1 + 5 * 0.04) ** 0.5) < 1e-12
    assert [O.eca_kernel_size(c) for c in (256, 512, 1536)] == [5, 5, 5]
    x = torch.randn(1, 3, 64, 64)
    from oracle.weights import seeded_state_dict
    y = O.nfnet_forward_features(seeded_state_dict(spec, 1), "img_encoder", cfg, x)
    assert tuple(y.shape) == (1, 2304, 2, 2) and torch.isfinite(y).all()


def test_nfnet_module_matches_timm_names():
    from item_alignment_amd.models import create_model
    from oracle import ref_models as O
    net = create_model("eca_nfnet_l0")
    want = {k[len("img_encoder."):]: s for k, s in O.nfnet_state_spec(O.nfnet_cfg("eca_nfnet_l0"))}
    have = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert set(want) | {"head.fc.weight", "head.fc.bias"} == set(have)
    assert all(have[k] == tuple(s) for k, s in want.items())
    assert sum(p.numel() for p in net.parameters()) == 24143924


def test_resnetv2_structure_and_timm_names():
    """The ResNetV2 restatement (timm is absent: parity unpinned) reproduces the published structure of resnetv2_50: 25.55 M
    parameters with the 1000-way fc (timm model table), 2048 features, the first block of every stage projecting and carrying
    the stride; the HIP module has timm's state_dict keys (BatchNorm buffers included)."""
    import torch
    from item_alignment_amd.models import create_model
    from oracle import ref_models as O
    cfg = O.resnetv2_cfg("resnetv2_50")
    spec = O.resnetv2_state_spec(cfg)
    n = sum(int(torch.tensor(s).prod()) for _, s in spec) + 2048 * 1000 + 1000
    assert abs(n / 1e6 - 25.55) < 0.01, n
    plan = O.resnetv2_plan(cfg)
    assert [len(s) for s in plan] == [3, 4, 6, 3]
    assert [s[0]["stride"] for s in plan] == [1, 2, 2, 2] and all(b["stride"] == 1 and not b["downsample"] for s in plan for b in s[1:])
    assert [s[0]["mid_chs"] for s in plan] == [64, 128, 256, 512] and all(s[0]["downsample"] for s in plan)
    net = create_model("resnetv2_50")
    want = {k[len("img_encoder."):]: s for k, s in spec}
    have = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    extra = set(have) - set(want)
    assert set(want) <= set(have)
    assert all(k.startswith("head.fc") or k.endswith(("running_mean", "running_var", "num_batches_tracked")) for k in extra), extra
    assert all(have[k] == tuple(s) for k, s in want.items())
    assert sum(p.numel() for p in net.parameters()) == n
    x = torch.randn(2, 3, 64, 64)
    from oracle.weights import seeded_state_dict
    y = O.resnetv2_forward_features(seeded_state_dict(spec, 1), "img_encoder", cfg, x, True, O.resnetv2_running_stats(cfg))
    assert tuple(y.shape) == (2, 2048, 2, 2) and torch.isfinite(y).all()


def test_strided_convolution_dispatch():
    """Which strided 3x3 convolutions leave the patch-matrix path (host logic of csrc/conv.hip, no GPU needed): 64 channels per group in
    and out, and one 64-channel input shared by 64 n output channels (the NF-Net stem's conv4); everything else -- the 3-channel conv1, the
    ResNets' ungrouped 128 / 256 / 512-channel convolutions -- stays on ia_conv_nhwc_*.  eca_nfnet_l0's five strided convolutions split 4 : 1."""
    import os
    import subprocess
    import sys
    from item_alignment_amd import _lib
    from item_alignment_amd.models import create_model
    from item_alignment_amd.models.nfnet import ScaledStdConv2d
    lib = _lib.load()
    yes = [(64, 64, 1), (128, 128, 2), (384, 384, 6), (64, 128, 1), (64, 256, 1)]
    no = [(3, 16, 1), (8, 16, 1), (128, 128, 1), (256, 256, 1), (96, 96, 1), (64, 96, 1), (128, 256, 2), (64, 64, 2)]
    assert all(lib.ia_conv3x3_s2_supported(*a) == 1 and lib.ia_conv3x3_s2_dgrad_supported(*a) == 1 for a in yes)
    assert all(lib.ia_conv3x3_s2_supported(*a) == 0 and lib.ia_conv3x3_s2_dgrad_supported(*a) == 0 for a in no)
    assert lib.ia_conv3x3_s2_padded_workspace_bytes(2, 40, 40, 128, 128, 2) > 0 and lib.ia_conv3x3_s2_padded_workspace_bytes(2, 40, 40, 128, 128, 1) == 0
    net = create_model("eca_nfnet_l0")
    strided = [m for m in net.modules() if isinstance(m, ScaledStdConv2d) and m.kernel_size == 3 and m.stride == 2]
    assert len(strided) == 5 and sum(m.strided_direct for m in strided) == 4 and not net.stem.conv1.strided_direct
    # the switches (read when asked, so a fresh process): everything off / the sliced stem form off
    code = ("from item_alignment_amd import _lib; l = _lib.load(); "
            "print(l.ia_conv3x3_s2_supported(64, 64, 1), l.ia_conv3x3_s2_dgrad_supported(64, 64, 1), l.ia_conv3x3_s2_dgrad_supported(64, 128, 1))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env, want in (({"IA_CONV_S2_DIRECT": "0"}, "0 0 0"), ({"IA_CONV_S2_DGRAD": "1"}, "1 1 0"), ({"IA_CONV_S2_DGRAD": "0"}, "1 0 0"), ({}, "1 1 1")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env={**os.environ, **env})
        assert r.stdout.split("\n")[0].strip() == want, (env, r.stdout, r.stderr[-300:])


def test_bit_towers_have_timm_names_and_shapes():
    """`create_model("resnetv2_50x1_bitm")` (a BiT name: finetune_image.py:23 lists resnetv2_50x3_bitm_in21k) builds the GroupNorm +
    StdConv2d tower with timm's state_dict keys and shapes -- no BatchNorm buffers -- and the head of the named variant; the x3 width
    scales the stem and every stage (the model is built on the meta device: 217 M parameters are not allocated here)."""
    import torch
    from item_alignment_amd.models import create_model
    from item_alignment_amd.models.resnetv2 import GroupNormAct, StdConv2d
    from oracle import ref_models as O
    cfg = O.resnetv2_cfg("resnetv2_50x1_bitm")
    spec = O.resnetv2_state_spec(cfg)
    net = create_model("resnetv2_50x1_bitm")
    want = {k[len("img_encoder."):]: tuple(s) for k, s in spec}
    have = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert set(want) <= set(have) and all(have[k] == s for k, s in want.items())
    assert set(have) - set(want) == {"head.fc.weight", "head.fc.bias"} and have["head.fc.weight"] == (1000, 2048, 1, 1)
    assert isinstance(net.stem.conv, StdConv2d) and net.stem.conv.std_eps == 1e-8 and net.stem.fixed
    assert isinstance(net.norm, GroupNormAct) and net.norm.num_groups == 32 and net.norm.eps == 1e-5
    assert all(isinstance(m, (StdConv2d, torch.nn.Conv2d)) for m in net.modules() if hasattr(m, "kernel_size") and not isinstance(m, torch.nn.MaxPool2d))
    with torch.device("meta"):
        big = create_model("resnetv2_50x3_bitm_in21k")
    assert big.num_features == 6144 and big.stem.conv.out_channels == 192 and tuple(big.head.fc.weight.shape) == (21843, 6144, 1, 1)
    assert sum(p.numel() for p in big.parameters()) == 345399315          # timm lists 217.32 M with 1000 classes (345.40 M with 21 843)


def test_resize_tables_reproduce_pillow_bit_exact():
    """The host-built resampling tables (data/gpu_preproc.py: Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc) and
    the 8.22 fixed-point two-pass arithmetic the GPU kernels implement reproduce PIL Image.resize(..., BICUBIC / BILINEAR) bit for
    bit (numpy emulation of csrc/image.hip here; the kernels themselves are checked in test_kernels_gpu.py)."""
    import numpy as np
    from PIL import Image
    from item_alignment_amd.data.gpu_preproc import PRECISION_BITS, precompute_coeffs

    def one_pass(x, n_out, axis, filt):
        b, k, _ = precompute_coeffs(x.shape[axis], n_out, filt)
        x = np.moveaxis(x.astype(np.int64), axis, 0)
        out = np.zeros((n_out,) + x.shape[1:], dtype=np.int64)
        for i in range(n_out):
            lo, n = b[i]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[i, :n].astype(np.int64), x[lo:lo + n], axes=(0, 0))
            out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
        return np.moveaxis(out, 0, axis)
    rs = np.random.RandomState(0)
    for H, W, S in [(800, 800, 384), (333, 517, 384), (90, 70, 224), (801, 640, 800)]:
        img = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
        for filt, pil in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
            ref = np.asarray(Image.fromarray(img).resize((S, S), pil))
            got = one_pass(one_pass(img, S, 1, filt), S, 0, filt)          # Pillow: horizontal pass first
            assert np.array_equal(ref, got.astype(np.uint8)), (H, W, S, filt)


def test_image_transform_matches_the_timm_restatement():
    """data/transforms.py ImageTransform (host execution) against oracle/timm_transform.py, the restatement of the reference's
    timm create_transform(input_size, is_training, hflip, color_jitter) [third party, parity unpinned] (data.py:838-841):
    evaluation = Resize(floor(S / 0.875), bilinear) + CenterCrop, training = RandomResizedCrop + flip + ColorJitter.  The random
    draw of the product follows timm's / torchvision's procedures (same random.Random stream -> same crop box), and given the same
    draw both produce the same tensor bit for bit."""
    import random
    import numpy as np
    from PIL import Image
    from item_alignment_amd.data.transforms import ImageTransform, center_crop_geometry
    from oracle import timm_transform as TT
    rs = np.random.RandomState(3)
    for (h, w), S in [((800, 800), 384), ((333, 517), 224), ((97, 64), 64), ((500, 1400), 384)]:
        img = Image.fromarray(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8))
        assert center_crop_geometry(w, h, S) == TT.eval_geometry(w, h, S)
        ev = ImageTransform(S, False)
        assert torch.equal(ev(img), TT.eval_transform(img, S))
        tr = ImageTransform(S, True, hflip=0.5, color_jitter=0.4, seed=11)
        ref_rng = random.Random(11)
        for _ in range(4):
            p = tr.draw(w, h)
            assert p.box == TT.random_resized_crop_params(ref_rng, w, h)          # timm get_params on the same stream
            # the product's remaining draws (flip, jitter order and factors) advance the same stream
            flip = ref_rng.random() < 0.5
            order = [0, 1, 2, 3]; ref_rng.shuffle(order)
            factors = [ref_rng.uniform(0.6, 1.4) for _ in range(3)]
            assert (p.flip, p.jitter) == (flip, (tuple(order), *factors))
            want = TT.train_transform(img, S, TT.TrainParams(p.box, p.flip, p.jitter))
            assert torch.equal(tr.apply(img, p), want)
    # no colour jitter unless asked for (the reference's CLI default is None), no flip with hflip = 0
    p = ImageTransform(64, True, hflip=0.0, color_jitter=None, seed=1).draw(100, 80)
    assert p.jitter is None and p.flip is False and p.train


def test_raw_image_mode_collates_without_resizing(tmp_path):
    """--gpu_preproc host side: datasets hand decoded uint8 frames (any size) + the flip decision to the collate functions,
    which keep them as a RawImageBatch (no stacking of unequal sizes); the default path still yields [B,3,S,S] fp32."""
    import numpy as np
    import torch
    from PIL import Image
    from item_alignment_amd.data.datasets import PairedImageDataset, RawImage, RawImageBatch, collate_image
    rs = np.random.RandomState(0)
    paths = []
    for i, (h, w) in enumerate([(40, 50), (33, 71), (64, 64), (20, 90)]):
        p = tmp_path / f"{i}.png"
        Image.fromarray(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(p)
        paths.append(str(p))
    data = [(1, "a", paths[0], "b", paths[1]), (0, "c", paths[2], "d", paths[3]), (1, "e", paths[0], "f", str(tmp_path / "missing.png"))]
    raw = PairedImageDataset(data, 32, True, hflip=0.5, raw=True, seed=4)
    src_ids, tgt_ids, a, b, labels = collate_image([raw[i] for i in range(3)])
    assert src_ids == ["a", "c"] and labels.tolist() == [1, 0]                 # the sample with a missing image is dropped
    assert isinstance(a, RawImageBatch) and isinstance(b, RawImageBatch) and len(a) == 2
    assert isinstance(a.items[0], RawImage) and a.items[0].u8.dtype == torch.uint8 and tuple(a.items[0].u8.shape) == (40, 50, 3)
    assert tuple(b.items[1].u8.shape) == (20, 90, 3)
    pr = a.items[0].params
    assert pr.train and len(pr.box) == 4 and pr.box[2] <= 40 and pr.box[3] <= 50 and pr.jitter is None
    std = PairedImageDataset(data, 32, False)
    _, _, a2, b2, _ = collate_image([std[i] for i in range(2)])
    assert tuple(a2.shape) == (2, 3, 32, 32) and a2.dtype == torch.float32


def test_attribute_pair_indices_match_reference_vectors():
    """`--auxiliary_task` attribute alignment (data/datasets.py attribute_pair_indices) against vectors captured from the
    reference's RobertaOneTowerDataset.__getitem__ (oracle/gen_pair_indices.py; data.py:568-612): aligned keys, a key mismatch
    part-way, truncated last attributes, attributes without ':' (stale-colon behaviour and the TypeError of a leading one)."""
    import json
    import os
    from item_alignment_amd.data import datasets as DS
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pair_indices.json")))
    assert (fx["colon_id"], fx["semicolon_id"]) == (DS.COLON_ID, DS.SEMICOLON_ID)
    n_pairs = 0
    for c in fx["cases"]:
        if "error" in c:
            with pytest.raises(Exception) as e:
                DS.attribute_pair_indices(c["input_ids"], fx["sep_token_id"])
            assert type(e.value).__name__ == c["error"]
        else:
            assert DS.attribute_pair_indices(c["input_ids"], fx["sep_token_id"]) == c["pair_indices"]
            n_pairs += len(c["pair_indices"])
    assert n_pairs > 20


def test_image_transform_draws_follow_the_global_generator():
    """Without an explicit seed the augmentation parameters come from the global `random` module, which DataLoader reseeds per worker
    and per epoch (a private generator inside the dataset object was pickled to every worker in the same state, so
    all workers drew the same crops and every epoch repeated them)."""
    import pickle
    import random
    from item_alignment_amd.data.transforms import ImageTransform
    tr = ImageTransform(64, True, hflip=0.5, color_jitter=0.4)
    clone = pickle.loads(pickle.dumps(tr))            # what a forkserver worker receives
    random.seed(1001)
    a = [clone.draw(300, 200) for _ in range(4)]
    random.seed(1002)                                  # another worker / another epoch
    b = [clone.draw(300, 200) for _ in range(4)]
    random.seed(1001)
    c = [tr.draw(300, 200) for _ in range(4)]
    assert a == c and a != b
    seeded = ImageTransform(64, True, hflip=0.5, color_jitter=0.4, seed=5)
    random.seed(1)
    d = seeded.draw(300, 200)
    random.seed(2)
    assert ImageTransform(64, True, hflip=0.5, color_jitter=0.4, seed=5).draw(300, 200) == d


def test_bench_self_launch_builds_a_child_launcher_command(monkeypatch):
    """bench.py --gpus N started bare must hand the SAME arguments to `python -m torch.distributed.run --nproc-per-node N` as a child
    process (never an exec) on a loopback rendezvous, and return the child's exit code."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeChild:
        pid = 1

        def wait(self, timeout=None):
            return 7

    def fake_popen(cmd, env=None, start_new_session=False, **kw):
        seen["cmd"], seen["env"], seen["session"] = cmd, env, start_new_session
        return FakeChild()
    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["session"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # main() takes that route before importing the engine or touching torch.cuda
    monkeypatch.setattr(bench, "self_launch", lambda n: 5)
    import pytest
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5


def test_bench_self_launch_takes_the_child_group_down_on_sigterm(tmp_path):
    """The launcher child of `bench.py --gpus N` lives in its own session, so a SIGTERM to the parent (a driver timeout) must reach it
    explicitly: the parent kills the child's process group and leaves with 128 + SIGTERM.  A sleeping stand-in plays the launcher."""
    import os
    import signal
    import subprocess
    import sys
    import textwrap
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pidfile = tmp_path / "child.pid"
    prog = textwrap.dedent(f"""
        import importlib.util, os, subprocess, sys
        spec = importlib.util.spec_from_file_location("bench_under_test", {os.path.join(root, "bench.py")!r})
        bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
        real = subprocess.Popen
        def fake(cmd, env=None, start_new_session=False, **kw):
            c = real([sys.executable, "-c", "import time; time.sleep(600)"], start_new_session=start_new_session)
            open({str(pidfile)!r}, "w").write(str(c.pid))
            return c
        bench.subprocess.Popen = fake
        sys.argv = ["bench.py", "--gpus", "2"]
        raise SystemExit(bench.self_launch(2))
    """)
    parent = subprocess.Popen([sys.executable, "-c", prog])
    try:
        for _ in range(600):
            if pidfile.exists() and pidfile.read_text():
                break
            time.sleep(0.1)
        child_pid = int(pidfile.read_text())
        time.sleep(0.3)                                   # let the parent reach child.wait() with its handlers installed
        parent.send_signal(signal.SIGTERM)
        assert parent.wait(timeout=30) == 128 + signal.SIGTERM
        for _ in range(100):
            try:
                os.kill(child_pid, 0)
            except ProcessLookupError:
                break
            # a killed child of the (now dead) parent may linger as a zombie of init for a moment
            try:
                if open(f"/proc/{child_pid}/stat").read().split(")")[1].split()[0] == "Z":
                    break
            except FileNotFoundError:
                break
            time.sleep(0.1)
        else:
            raise AssertionError("the launcher child survived the parent's SIGTERM")
    finally:
        if parent.poll() is None:
            parent.kill()
        try:
            os.kill(int(pidfile.read_text()), signal.SIGKILL)
        except Exception:
            pass


def test_load_tokenizer_never_grows_the_vocabulary(tmp_path):
    """cli_common.load_tokenizer registers the image / BOS placeholders as special tokens only when the checkpoint's vocabulary already
    holds them: a vocabulary without '<S>' must keep its size (an appended id would index past the embedding table) and resolve the
    word to [UNK], as the reference's `tokenizer.bos_token = BOS_TOKEN` does (finetune_text.py:186-189)."""
    from types import SimpleNamespace
    import pytest
    pytest.importorskip("transformers")
    from item_alignment_amd.cli_common import load_tokenizer
    base = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]", "手", "机"]
    for extra, has_bos in ((["<S>"], True), ([], False)):
        d = tmp_path / f"vocab_{int(has_bos)}"
        d.mkdir()
        (d / "vocab.txt").write_text("\n".join(base + extra) + "\n", encoding="utf-8")
        tk = load_tokenizer(SimpleNamespace(pretrained_model_path=str(d), do_lower_case=True))
        assert len(tk) == tk.vocab_size == len(base) + len(extra)
        ids = tk.convert_tokens_to_ids(tk.tokenize("[unused99] <S> 手"))
        assert max(ids) < tk.vocab_size
        assert ids[0] == 99
        assert (tk.vocab["<S>"] in ids) if has_bos else (tk.unk_token_id in ids)


def test_direct_convolution_swizzle_keys_are_the_best_of_their_family():
    """The XOR keys of the direct 3x3 convolution's LDS tiles (csrc/conv.hip: dconv::akey / bkey) under the b128 service-group model of
    MI355X_MICROARCH.md (tools/abl/swizzle_search.py): the filter bank is conflict-free, the input tile at most 2-way (a fragment that
    wraps a 30-pixel tile row), and no (shift, mask) candidate of the searched family does better -- what DESIGN.md 4.6 claims."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("swizzle_search", os.path.join(root, "tools", "abl", "swizzle_search.py"))
    ss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ss)
    shipped_a = {64: lambda P: P & 7, 32: lambda P: (P >> 1) & 3, 16: lambda P: 0}
    for ci, key in shipped_a.items():
        cost = ss.a_cost(ci, key)
        assert cost == 2, (ci, cost)
        best = min(ss.a_cost(ci, lambda P, s=s, m=m: (P >> s) & m) for s in range(5) for m in (0, 1, 3, 7) if m < ci // 8)
        assert best == cost, (ci, best, cost)
        assert ss.a_cost(ci, lambda P: 0) >= cost
    for co, shift in {64: 3, 32: 2, 16: 1}.items():
        assert ss.b_cost(co, lambda n, s=shift: (n >> s) & 3) == 1, co
        assert ss.b_cost(co, lambda n: 0) == 2, co
    # the kernel's own key functions say the same thing as the table above
    src = open(os.path.join(root, "item_alignment_amd", "csrc", "conv.hip")).read()
    assert "return CI == 64 ? (P & 7) : CI == 32 ? ((P >> 1) & 3) : 0;" in src
    assert "return (row >> (CO == 64 ? 3 : CO == 32 ? 2 : 1)) & 3;" in src


def test_unsupported_image_tower_is_a_usage_error_at_argparse_time():
    """`finetune_image.py --model_name resnetv2_50d_evos` (any name goes to timm in the reference, finetune_image.py:191,215) has no HIP
    tower: the run ends in argparse with the supported set named, not with an error in the middle of model construction; `create_model`
    itself raises a ValueError with the same text.  The BiT names of the reference's help text (finetune_image.py:23) ARE built."""
    import os
    import subprocess
    import sys
    from item_alignment_amd.models.image import create_model, supported_image_encoders
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = supported_image_encoders()
    assert "vit_base_patch16_384" in names and "eca_nfnet_l0" in names and "resnetv2_50" in names
    assert "resnetv2_50x3_bitm_in21k" in names and "resnetv2_50x1_bitm" in names and "resnetv2_152x4_bitm" in names
    with pytest.raises(ValueError, match="resnetv2_50d_evos.*EvoNorm.*supported: .*eca_nfnet_l0"):
        create_model("resnetv2_50d_evos")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "finetune_image.py"), "--data_dir", "/nonexistent", "--output_dir", "/nonexistent",
                        "--data_version", "v0", "--model_name", "resnetv2_50d_evos"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "has no HIP tower" in r.stderr and "resnetv2_50" in r.stderr and "usage:" in r.stderr


def test_fused_backward_exchange_layout_is_conflict_free():
    """The dS exchange tile of the one-kernel attention backward (csrc/attention.hip, bwdf: [32 keys][32 queries] bf16, dense 64-byte rows,
    8-byte chunk c of row r at chunk position c ^ xkey(r)) under the LDS bank model of the guide (64 banks x 4 bytes, a b64 access served
    in passes of 32 lanes): every pass of the part-A writes (32 key rows x one chunk) and of the part-C transpose reads (8 rows x 32
    bytes) touches 64 distinct banks, both sides address the same bytes for the same (key, query), and the layout rounds 4-5 used
    (72-byte rows, no XOR) fails the read test -- the two-way conflicts behind 26 % of that kernel's LDS-active cycles."""
    def xkey(r):
        return (((r >> 2) & 1) << 2) | ((r >> 3) & 3)

    def addr_new(row, chunk):
        return row * 64 + ((chunk ^ xkey(row)) << 3)

    def addr_old(row, chunk):
        return row * 72 + chunk * 8

    def banks(addrs):                                   # an 8-byte access covers two banks
        return [b for a in addrs for b in ((a >> 2) & 63, ((a >> 2) + 1) & 63)]

    def write_passes(addr):                             # lane (lk, hh) writes chunk hh + 4 hf + 2 s of row lk
        for hf in (0, 1):
            for s in (0, 1):
                for hh in (0, 1):
                    yield [addr(lk, hh + 4 * hf + 2 * s) for lk in range(32)]

    def read_passes(addr):                              # 16-lane group g4: rows 4 g4 + (p >> 2) [+ 16], chunk 4 qh + (p & 3)
        for qh in (0, 1):
            for second in (0, 16):
                for half in (0, 1):
                    yield [addr(4 * g4 + (p >> 2) + second, 4 * qh + (p & 3)) for g4 in (2 * half, 2 * half + 1) for p in range(16)]

    for p in write_passes(addr_new):
        assert len(set(banks(p))) == 64
    for p in read_passes(addr_new):
        assert len(set(banks(p))) == 64
    assert all(len(set(banks(p))) == 64 for p in write_passes(addr_old))
    assert any(len(set(banks(p))) < 64 for p in read_passes(addr_old))
    # a bijection of the tile's 256 chunks onto 2 KiB
    assert sorted(addr_new(r, c) for r in range(32) for c in range(8)) == list(range(0, 2048, 8))
    # the kernel's own address forms: part A's (row base | chunk hh) ^ (j << 4) and part C's second read (first ^ 16) + 16 rows
    for lk in range(32):
        for hh in (0, 1):
            base = lk * 64 + ((hh ^ xkey(lk)) << 3)
            for j in range(4):
                assert base ^ (j << 4) == addr_new(lk, hh + 2 * j)
    for row in range(16):
        for chunk in range(8):
            assert (addr_new(row, chunk) ^ 16) + 16 * 64 == addr_new(row + 16, chunk)
