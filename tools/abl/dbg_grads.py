"""One forward + backward of the bench model; lists parameters whose gradient is not finite and the per-tower gradient norms."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from item_alignment_amd.data.synthetic import SyntheticCocaPairs
from item_alignment_amd.models import functional as Fn
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = bench.roberta_large_config()
if len(sys.argv) > 2 and sys.argv[2] == "nodrop":
    cfg.hidden_dropout_prob = 0.0; cfg.attention_probs_dropout_prob = 0.0
dev = torch.device("cuda:0")
model = bench.build_model(cfg).to(dev).train()
arena = model.param_arena
data = SyntheticCocaPairs(50000, image_size=cfg.image_size, seed=2345)
b = data.batch(list(range(pairs)), dev, device_images=True)
Fn.set_step_seed(12345)
arena.zero_grad()
out = model(*b[:10], labels=b[10])
out.loss.backward()
torch.cuda.synchronize()
print("loss", float(out.loss))
bad = []
for n, p in zip(arena.names, arena.params):
    g = p.grad
    if not torch.isfinite(g).all():
        bad.append((n, int((~torch.isfinite(g)).sum()), g.numel()))
print(f"BWD={os.environ.get('IA_ATTN_BWD')} pairs={pairs}: {len(bad)} params with non-finite grads")
for x in bad[:12] + bad[-12:]:
    print("   ", x)
tot = {}
for n, p in zip(arena.names, arena.params):
    k = n.split(".")[0]
    tot[k] = tot.get(k, 0.0) + float(torch.nan_to_num(p.grad).double().pow(2).sum())
print({k: v ** 0.5 for k, v in tot.items()})
