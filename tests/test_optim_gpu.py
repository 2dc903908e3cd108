"""The optimiser boundary: a maintainer who keeps the reference's own train loop -- grouped parameters,
`AdamW(...)`, `get_linear_schedule_with_warmup`, `optimizer.zero_grad() / loss.backward() / optimizer.step() / scheduler.step()`
(reference finetune_multimodal.py:296-315, 371-468) -- gets the fused arena AdamW through `item_alignment_amd.optim.AdamW`, and a
maintainer who keeps `torch.optim.AdamW` itself still trains the weights the GEMMs read (the bf16 shadow follows the masters)."""
import pytest
import torch

from golden_util import load_case
from test_models_gpu import build, g

pytestmark = pytest.mark.gpu


def two_tower_args(case):
    return dict(input_ids_1=g(case, "input_ids_1"), attention_mask_1=g(case, "attention_mask_1"), token_type_ids_1=g(case, "token_type_ids_1"),
                input_ids_2=g(case, "input_ids_2"), attention_mask_2=g(case, "attention_mask_2"), token_type_ids_2=g(case, "token_type_ids_2"),
                labels=g(case, "labels"))


def linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps):
    """transformers.get_linear_schedule_with_warmup, verbatim semantics (a LambdaLR over param_groups[i]['lr'])."""
    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


def grouped(model, weight_decay):
    no_decay = ["bias", "LayerNorm.weight"]
    return [{"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
            {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]


def fresh(case):
    import item_alignment_amd.models as M
    from test_models_gpu import cfg_of
    from golden_util import weights
    model = M.RobertaTwoTower(cfg_of(case))
    model.load_state_dict(weights(case), strict=False)
    return model            # still on the CPU, no arena: where the reference builds its optimizer (:309 against :342)


@pytest.mark.parametrize("build_before_cuda", [True, False])
def test_reference_loop_body_with_the_arena_optimizer(gpu, build_before_cuda):
    """The reference's literal loop shape against arena.adamw_step + the same schedule evaluated by hand: identical weights (both sides
    run the same kernels on the same inputs, dropout off), bit for bit, after five steps across the warm-up knee."""
    from item_alignment_amd.optim import AdamW
    from item_alignment_amd.models import functional as Fn
    case = load_case("roberta_two_tower_ce")
    args = two_tower_args(case)
    lr, wd, total, warm, steps = 1e-4, 0.01, 8, 2, 6

    # ---- side A: the reference loop, unchanged but for the AdamW import
    model = fresh(case)
    if not build_before_cuda:
        model.cuda()
        model.ensure_arena()
    optimizer = AdamW(grouped(model, wd), lr=lr, eps=1e-8, betas=(0.9, 0.98))
    scheduler = linear_schedule_with_warmup(optimizer, warm, total)
    if build_before_cuda:
        model.cuda()
        for state in optimizer.state.values():              # finetune_multimodal.py:343-347 (empty at this point, as in the reference)
            for k, v in state.items():
                if torch.is_tensor(v):
                    state[k] = v.cuda()
    model.eval()                                            # dropout off so that the two sides see the same arithmetic
    losses_a = []
    for step in range(steps):
        optimizer.zero_grad()
        Fn.set_step_seed(step)
        output = model(**args)
        loss = output.loss
        loss.backward()
        optimizer.step()
        scheduler.step()
        losses_a.append(loss.item())
    a = {n: p.detach().clone() for n, p in model.named_parameters()}
    shadow_a = model.param_arena.shadow.clone()
    assert model.param_arena.stale_refreshes == 0           # the arena optimizer rewrites the shadow itself
    sd = optimizer.state_dict()
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == steps

    # ---- side B: this repo's own loop (arena.adamw_step with the schedule evaluated by hand)
    ref = fresh(case).cuda().eval()
    arena = ref.param_arena
    losses_b = []
    for step in range(steps):
        arena.zero_grad()
        Fn.set_step_seed(step)
        loss = ref(**args).loss
        loss.backward()
        f = step / max(1, warm) if step < warm else max(0.0, (total - step) / max(1, total - warm))
        arena.adamw_step(lr * f, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)
        losses_b.append(loss.item())
    torch.cuda.synchronize()
    # Bit for bit -- except what descends from the three embedding tables: their gradient rows are accumulated with fp32 atomic adds
    # in arrival order (embed_ln_bwd_kernel; torch's own embedding backward on a GPU does the same), so two runs of the SAME code differ
    # there by ~1e-7 relative, and everything downstream of the tables inherits last-bit differences from the second step on.
    assert losses_a[0] == losses_b[0]
    assert max(abs(x - y) for x, y in zip(losses_a, losses_b)) < 1e-5, (losses_a, losses_b)
    assert losses_a[-1] < losses_a[0], losses_a
    for n, p in ref.named_parameters():
        assert torch.allclose(a[n], p.detach(), rtol=0, atol=2e-6), (n, (a[n] - p.detach()).abs().max().item())
    assert (shadow_a.float() - arena.shadow.float()).abs().max().item() <= 2.0 ** -8 * arena.shadow.float().abs().max().item()


def test_optimizer_state_dict_round_trip(gpu):
    """optimizer.state_dict() / load_state_dict() carry the moments and the step count (torch's layout): a resumed run continues
    exactly where the first one would have."""
    from item_alignment_amd.optim import AdamW
    case = load_case("roberta_two_tower_ce")
    args = two_tower_args(case)

    def run(model, opt, n):
        for _ in range(n):
            opt.zero_grad()
            model(**args).loss.backward()
            opt.step()

    m1 = fresh(case).cuda().eval()
    o1 = AdamW(grouped(m1, 0.01), lr=1e-4, betas=(0.9, 0.98))
    run(m1, o1, 2)
    sd_opt = {k: (v if k != "state" else {i: {kk: vv.clone() if torch.is_tensor(vv) else vv for kk, vv in st.items()} for i, st in v.items()})
              for k, v in o1.state_dict().items()}
    sd_model = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    run(m1, o1, 2)
    m2 = fresh(case).cuda().eval()
    m2.ensure_arena()
    m2.load_state_dict(sd_model)
    o2 = AdamW(grouped(m2, 0.01), lr=1e-4, betas=(0.9, 0.98))
    o2.load_state_dict(sd_opt)
    run(m2, o2, 2)
    torch.cuda.synchronize()
    # The three embedding tables' gradients are fp32 atomic adds in arrival order (embed_ln_bwd_kernel): an element whose contributions
    # cancel to ~1e-5 of their size moves by ~1e-2 relative from run to run, which Adam's normalised update turns into ~1e-6 of weight
    # at lr 1e-4 -- at the common bar this test failed about one full-suite run in five (round 6).  Everything else is bit-stable up
    # to what those tables feed forward and keeps 2e-6.
    for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        atol = 2e-5 if n.endswith("_embeddings.weight") else 2e-6
        assert torch.allclose(p.detach(), q.detach(), rtol=0, atol=atol), (n, (p.detach() - q.detach()).abs().max().item())


def test_optimizer_loads_a_torch_adamw_checkpoint_saved_before_its_first_step(gpu):
    """The reference's checkpoints hold `torch.optim.AdamW.state_dict()`: groups without `step_count`, and -- saved before the first
    step -- no per-parameter state at all (round-5 advisor: KeyError).  Loading one and stepping works, from step 0."""
    from item_alignment_amd.optim import AdamW
    case = load_case("roberta_two_tower_ce")
    args = two_tower_args(case)
    model = fresh(case).cuda().eval()
    model.ensure_arena()
    foreign = torch.optim.AdamW(grouped(model, 0.01), lr=1e-4, betas=(0.9, 0.98), eps=1e-8)
    opt = AdamW(grouped(model, 0.01), lr=3e-4, betas=(0.9, 0.999))
    opt.load_state_dict(foreign.state_dict())
    assert all(g["step_count"] == 0 and g["lr"] == 1e-4 and tuple(g["betas"]) == (0.9, 0.98) for g in opt.param_groups)
    opt.zero_grad()
    model(**args).loss.backward()
    opt.step()
    assert all(g["step_count"] == 1 for g in opt.param_groups)


@pytest.mark.parametrize("set_to_none", [False, True])
def test_foreign_torch_adamw_no_longer_trains_on_a_stale_shadow(gpu, set_to_none):
    """A maintainer who keeps `torch.optim.AdamW(model.parameters())`: torch updates the fp32 masters (views of the arena) in place;
    the GEMMs read the bf16 shadow.  Before the staleness guard the shadow kept the initial weights for ever -- the loss never moved
    and nothing said so.  Now every forward compares the parameters' version counters with those at the last refresh and re-casts the
    shadow when torch has written them: the loss goes down and the shadow equals bf16(master) at every forward."""
    case = load_case("roberta_two_tower_ce")
    args = two_tower_args(case)
    model = fresh(case).cuda().eval()
    arena = model.param_arena
    optimizer = torch.optim.AdamW(grouped(model, 0.01), lr=1e-4, betas=(0.9, 0.98), eps=1e-8)
    losses = []
    for step in range(6):
        # set_to_none=True (torch's default) drops the p.grad views: the next forward clears the gradient arena and re-points them
        optimizer.zero_grad(set_to_none=set_to_none)
        out = model(**args)
        assert torch.equal(arena.shadow.float(), arena.master.to(torch.bfloat16).float()), step
        out.loss.backward()
        optimizer.step()
        losses.append(out.loss.item())
    assert arena.stale_refreshes == 5, arena.stale_refreshes        # one per forward that followed a torch step
    assert losses[-1] < losses[0] - 0.01, losses
    # the same six steps through the arena optimizer give the same trajectory within fp32 update noise (same gradients, same rule)
    from item_alignment_amd.optim import AdamW
    ref = fresh(case).cuda().eval()
    opt = AdamW(grouped(ref, 0.01), lr=1e-4, betas=(0.9, 0.98), eps=1e-8)
    ref_losses = []
    for step in range(6):
        opt.zero_grad()
        out = ref(**args)
        out.loss.backward()
        opt.step()
        ref_losses.append(out.loss.item())
    assert max(abs(x - y) for x, y in zip(losses, ref_losses)) < 2e-2 * max(1.0, abs(ref_losses[0])), (losses, ref_losses)
