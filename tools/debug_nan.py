import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_util import load_case
import test_models_gpu as T
from item_alignment_amd.models import functional as Fn

case = load_case("roberta_two_tower_ce")
model = T.build(case, "RobertaTwoTower")
g = lambda k: case.inputs[k].cuda()
names = {id(p): n for n, p in model.named_parameters()}
def hook(params):
    torch.cuda.synchronize()
    for p in params:
        print("  ready", names[id(p)], "finite" if torch.isfinite(p.grad).all() else "NAN", float(p.grad.abs().max()))
Fn.register_grad_ready_hook(hook)
orig = Fn.EncoderStackFn.backward
def bw(ctx, *grads):
    for i, gr in enumerate(grads):
        if gr is not None:
            print("encoder bwd grad", i, gr.shape, gr.dtype, torch.isfinite(gr.float()).all().item(), float(gr.float().abs().max()))
    return orig(ctx, *grads)
Fn.EncoderStackFn.backward = staticmethod(bw)
out = model(input_ids_1=g("input_ids_1"), attention_mask_1=g("attention_mask_1"), token_type_ids_1=g("token_type_ids_1"),
            input_ids_2=g("input_ids_2"), attention_mask_2=g("attention_mask_2"), token_type_ids_2=g("token_type_ids_2"), labels=g("labels"))
model.param_arena.zero_grad()
out.loss.backward()
