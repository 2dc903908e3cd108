"""Layer-engine parity on the GPU: ia_layer_fwd / ia_layer_bwd (one RoBERTa post-LN layer, one ViT pre-LN
block) against the oracle's fp32 restatement of the same layer evaluated with torch autograd on the same
bf16-rounded weights and inputs.  Tolerances: activations 2e-2 of max; gradients: cosine >= 0.995 and
2e-2..5e-2 of max (bf16 activations)."""
import ctypes as C
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()


def cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()


def make_layer(H, I, dev, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=0.05: (torch.randn(*s, generator=g) * sc).to(dev)
    return dict(w_qkv=r(3 * H, H), b_qkv=r(3 * H, sc=0.02), w_o=r(H, H), b_o=r(H, sc=0.02), ln1_g=1 + r(H, sc=0.1), ln1_b=r(H, sc=0.02),
                w_fc1=r(I, H), b_fc1=r(I, sc=0.02), w_fc2=r(H, I), b_fc2=r(H, sc=0.02), ln2_g=1 + r(H, sc=0.1), ln2_b=r(H, sc=0.02))


def ref_layer(P, x, mask, nh, pre_ln, eps):
    import torch.nn.functional as F
    B, L, H = x.shape
    def attn(h):
        qkv = F.linear(h, P["w_qkv"], P["b_qkv"]).view(B, L, 3, nh, 64)
        q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
        s = q @ k.transpose(-1, -2) * 0.125
        if mask is not None:
            s = s + (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
        return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, H)
    if not pre_ln:
        a = F.layer_norm(F.linear(attn(x), P["w_o"], P["b_o"]) + x, (H,), P["ln1_g"], P["ln1_b"], eps)
        h = F.linear(F.gelu(F.linear(a, P["w_fc1"], P["b_fc1"])), P["w_fc2"], P["b_fc2"])
        return F.layer_norm(h + a, (H,), P["ln2_g"], P["ln2_b"], eps)
    x2 = x + F.linear(attn(F.layer_norm(x, (H,), P["ln1_g"], P["ln1_b"], eps)), P["w_o"], P["b_o"])
    return x2 + F.linear(F.gelu(F.linear(F.layer_norm(x2, (H,), P["ln2_g"], P["ln2_b"], eps), P["w_fc1"], P["b_fc1"])), P["w_fc2"], P["b_fc2"])


@pytest.mark.parametrize("B,L,nh,pre_ln,masked", [(3, 40, 2, False, True), (2, 255, 4, False, True), (2, 65, 2, True, False),
                                                   (2, 577, 12, True, False), (2, 510, 16, False, True)])
def test_layer_fwd_bwd(gpu, B, L, nh, pre_ln, masked):
    from item_alignment_amd import _lib
    from item_alignment_amd._lib import LayerCfg, LayerGrads, LayerWeights
    lib = _lib.load()
    H, I = nh * 64, nh * 256
    M = B * L
    P32 = make_layer(H, I, gpu, 3)
    bf = lambda t: t.to(torch.bfloat16)
    mats = ("w_qkv", "w_o", "w_fc1", "w_fc2")
    Pb = {k: bf(v) for k, v in P32.items() if k in mats}
    Pref = {k: (Pb[k].float() if k in mats else v).clone().requires_grad_(True) for k, v in P32.items()}
    x = bf(torch.randn(B, L, H, generator=torch.Generator().manual_seed(5)).to(gpu))
    dy = bf(torch.randn(B, L, H, generator=torch.Generator().manual_seed(6)).to(gpu))
    mask = None
    if masked:
        lens = torch.tensor([L - 2 - 3 * i for i in range(B)])
        mask = (torch.arange(L)[None] < lens[:, None]).to(torch.uint8).to(gpu)
    eps = 1e-6 if pre_ln else 1e-12
    xr = x.float().requires_grad_(True)
    yr = ref_layer(Pref, xr, mask, nh, pre_ln, eps)
    yr.backward(dy.float())

    cfg = LayerCfg(B=B, L=L, H=H, I=I, nh=nh, pre_ln=int(pre_ln), eps=eps, hidden_drop=0.0, attn_drop=0.0, seed=1, layer_id=0)
    w, g = LayerWeights(), LayerGrads()
    G = {k: torch.zeros_like(v) for k, v in P32.items()}
    for k in P32:
        setattr(w, k, (Pb[k] if k in mats else P32[k]).data_ptr())
        setattr(g, k, G[k].data_ptr())
    stash = torch.empty(lib.ia_layer_stash_bytes(C.byref(cfg)), device=gpu, dtype=torch.uint8)
    y = torch.empty(M, H, device=gpu, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ia_layer_fwd(C.byref(cfg), C.byref(w), x.data_ptr(), _lib.ptr(mask), y.data_ptr(), stash.data_ptr(), st), "fwd")
    valid = mask.bool().view(-1) if mask is not None else torch.ones(M, dtype=torch.bool, device=gpu)
    assert rel(y[valid], yr.detach().view(M, H)[valid]) < 2e-2
    # the forward-only layer (evaluation / prediction): the same output bit for bit, from a transient scratch
    y_inf = torch.empty_like(y)
    isz = lib.ia_layer_infer_scratch_bytes(C.byref(cfg))
    assert 0 < isz < stash.numel()
    iscr = torch.empty(isz, device=gpu, dtype=torch.uint8)
    _lib.check(lib.ia_layer_fwd_infer(C.byref(cfg), C.byref(w), x.data_ptr(), _lib.ptr(mask), y_inf.data_ptr(), iscr.data_ptr(), isz, st), "fwd_infer")
    assert torch.equal(y_inf[valid], y[valid])
    assert lib.ia_layer_fwd_infer(C.byref(cfg), C.byref(w), x.data_ptr(), _lib.ptr(mask), y_inf.data_ptr(), iscr.data_ptr(), isz - 1, st) == -3
    scratch = torch.empty(lib.ia_layer_bwd_scratch_bytes(C.byref(cfg)), device=gpu, dtype=torch.uint8)
    dx = dy.clone().view(M, H)
    _lib.check(lib.ia_layer_bwd(C.byref(cfg), C.byref(w), C.byref(g), x.data_ptr(), _lib.ptr(mask), y.data_ptr(), stash.data_ptr(), dx.data_ptr(),
                                dx.data_ptr(), scratch.data_ptr(), scratch.numel(), st), "bwd")
    torch.cuda.synchronize()
    assert torch.isfinite(dx.float()).all()
    assert cos(dx, xr.grad.view(M, H)) > 0.995 and rel(dx, xr.grad.view(M, H)) < 5e-2
    if not pre_ln:
        # split-residual form (ia_layer_bwd2): the output gradient arrives in two parts, the input gradient leaves in two parts whose
        # sum is the input gradient of the single-tensor form; parameter gradients accumulate a second, equal contribution
        G2 = {k: torch.zeros_like(v) for k, v in P32.items()}
        g2 = LayerGrads()
        for k in P32:
            setattr(g2, k, G2[k].data_ptr())
        part = bf(torch.randn(M, H, generator=torch.Generator().manual_seed(7)).to(gpu) * 0.5)
        da, db = (dy.view(M, H).float() - part.float()).to(torch.bfloat16), part.clone()
        dxa, dxb = da.clone(), db.clone()                      # in place: dx over dy, dx2 over dy2
        _lib.check(lib.ia_layer_bwd2(C.byref(cfg), C.byref(w), C.byref(g2), x.data_ptr(), _lib.ptr(mask), y.data_ptr(), stash.data_ptr(),
                                     dxa.data_ptr(), dxb.data_ptr(), dxa.data_ptr(), dxb.data_ptr(), scratch.data_ptr(), scratch.numel(), st), "bwd2")
        torch.cuda.synchronize()
        both = dxa.float() + dxb.float()
        assert cos(both, xr.grad.view(M, H)) > 0.995 and rel(both, xr.grad.view(M, H)) < 5e-2
        assert rel(both, dx) < 3e-2                            # bf16 rounding of the split (da + db) and of the two-term sums
        for k in P32:
            assert cos(G2[k], Pref[k].grad) > 0.995, (k, cos(G2[k], Pref[k].grad))
    if pre_ln:
        # the fc2 bias gradient hand-over of a ViT stack: the block emits the column sums of its dx (= the block below's dy) out of its
        # last LayerNorm backward, and a block told that its dy was summed that way skips the column-sum pass (b_fc2 untouched)
        G3 = {k: torch.zeros_like(v) for k, v in P32.items()}
        g3 = LayerGrads()
        for k in P32:
            setattr(g3, k, G3[k].data_ptr())
        below = torch.full((H,), 0.25, device=gpu, dtype=torch.float32)
        cfg3 = LayerCfg(B=B, L=L, H=H, I=I, nh=nh, pre_ln=1, eps=eps, hidden_drop=0.0, attn_drop=0.0, seed=1, layer_id=0,
                        dx_colsum_out=below.data_ptr(), dy_colsum_done=1)
        dx3 = dy.clone().view(M, H)
        _lib.check(lib.ia_layer_bwd(C.byref(cfg3), C.byref(w), C.byref(g3), x.data_ptr(), _lib.ptr(mask), y.data_ptr(), stash.data_ptr(),
                                    dx3.data_ptr(), dx3.data_ptr(), scratch.data_ptr(), scratch.numel(), st), "bwd (column-sum hand-over)")
        torch.cuda.synchronize()
        assert torch.equal(dx3, dx)
        want_cs = 0.25 + dx.float().sum(0)
        assert (below - want_cs).abs().max().item() <= 2e-3 * (1.0 + want_cs.abs().max().item())
        assert G3["b_fc2"].abs().max().item() == 0.0 and torch.equal(G3["w_fc2"], G["w_fc2"])
    for k in P32:
        want = Pref[k].grad
        assert torch.isfinite(G[k]).all(), k
        assert cos(G[k], want) > 0.995, (k, cos(G[k], want))
        assert rel(G[k], want) < 5e-2, (k, rel(G[k], want))
    if masked and not pre_ln:
        # ia_layer_cfg::masked_rows_dead (round 6): with the output gradient zero at every masked position -- what an encoder whose heads
        # read no masked position hands down -- the backward that skips those rows (attention query blocks, LayerNorm rows) returns the
        # same input gradient and the same parameter gradients, bit for bit, and zero rows at the masked positions
        dyz = (dy.view(M, H) * valid[:, None].to(dy.dtype)).contiguous()
        res = []
        for flag in (0, 1):
            cf = LayerCfg(B=B, L=L, H=H, I=I, nh=nh, pre_ln=0, eps=eps, hidden_drop=0.0, attn_drop=0.0, seed=1, layer_id=0, masked_rows_dead=flag)
            Gf = {k: torch.zeros_like(v) for k, v in P32.items()}
            gf = LayerGrads()
            for k in P32:
                setattr(gf, k, Gf[k].data_ptr())
            dxf = dyz.clone()
            _lib.check(lib.ia_layer_bwd(C.byref(cf), C.byref(w), C.byref(gf), x.data_ptr(), _lib.ptr(mask), y.data_ptr(), stash.data_ptr(),
                                        dxf.data_ptr(), dxf.data_ptr(), scratch.data_ptr(), scratch.numel(), st), f"bwd (masked_rows_dead={flag})")
            torch.cuda.synchronize()
            res.append((dxf, Gf))
        assert torch.equal(res[0][0], res[1][0])
        assert res[1][0][~valid].float().abs().max().item() == 0.0
        for k in P32:
            assert torch.equal(res[0][1][k], res[1][1][k]), k
