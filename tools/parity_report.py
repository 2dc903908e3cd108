"""Prints the relative error (vs the reference golden vectors) of every output / gradient of every fixture."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_util import load_case, weights
import test_models_gpu as T
import item_alignment_amd.models as M

def g(case, k):
    v = case.inputs.get(k); return None if v is None else v.cuda()

for name in ["roberta_one_tower_cls_ce", "roberta_two_tower_ce"]:
    case = load_case(name)
    if "one_tower" in name:
        model = T.build(case, "RobertaOneTower")
        out = model(input_ids=g(case, "input_ids"), attention_mask=g(case, "attention_mask"), token_type_ids=g(case, "token_type_ids"),
                    labels=g(case, "labels"), output_hidden_states=True)
        m = g(case, "attention_mask").bool().cpu()
        for k, idx in (("hidden0", 0), ("hidden1", 1), ("hidden_last", -1)):
            print(name, k, T.rel(out.hidden_states[idx].float().cpu()[m], case.extra[k][m]))
    else:
        model = T.build(case, "RobertaTwoTower")
        out = model(input_ids_1=g(case, "input_ids_1"), attention_mask_1=g(case, "attention_mask_1"), token_type_ids_1=g(case, "token_type_ids_1"),
                    input_ids_2=g(case, "input_ids_2"), attention_mask_2=g(case, "attention_mask_2"), token_type_ids_2=g(case, "token_type_ids_2"),
                    labels=g(case, "labels"))
    for k, want in case.outs.items():
        print(name, k, T.rel(getattr(out, k).detach(), want), getattr(out, k).detach().flatten()[:4].tolist(), want.flatten()[:4].tolist())
    model.param_arena.zero_grad(); out.loss.backward(); torch.cuda.synchronize()
    P = dict(model.named_parameters())
    for k, want in case.grads.items():
        print(name, "grad", k, T.rel(P[k].grad, want))
