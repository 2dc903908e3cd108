"""eca_nfnet_l0 two-tower train step (config C3 shapes: 800x800 images) on one MI355X: images/s and model TFLOP/s.
usage: python tools/nfnet_bench.py [pairs] [size]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import item_alignment_amd.models as M

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 800
cfg = SimpleNamespace(num_labels=2, hidden_dropout_prob=0.1, loss_type="ce", loss_margin=0.0, classification_method="cls", hidden_size=2304)
enc = M.create_model("eca_nfnet_l0")
model = M.NFNetTwoTower(cfg, enc).cuda().train()
arena = model.param_arena
g = torch.Generator().manual_seed(0)
im1, im2 = torch.randn((pairs, 3, S, S), generator=g).cuda(), torch.randn((pairs, 3, S, S), generator=g).cuda()
labels = torch.randint(0, 2, (pairs,), generator=g).cuda()


def step():
    arena.zero_grad()
    out = model(im1, im2, labels)
    out.loss.backward()
    arena.adamw_step(1e-5)
    return out.loss


for _ in range(2):
    l = step()
torch.cuda.synchronize()
t0 = time.time()
n = 5
for _ in range(n):
    l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
flops = 3 * 2 * 54.1e9 * (S / 800) ** 2 * 2 * pairs          # 54.1 GMAC / image forward at 800^2, train = 3x
print(f"eca_nfnet_l0 two-tower {S}x{S}, {pairs} pairs/step: {dt*1e3:.1f} ms/step, {pairs/dt:.2f} pairs/s, {flops/dt/1e12:.1f} TFLOP/s, loss {l.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
