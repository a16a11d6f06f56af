import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg
P, synth = pkg(), pkg("synth")
dev = torch.device("cuda:0")
m = P.QuadtreeCNN(12).to(dev).train()
x = torch.randn(8, 3, 224, 224, device=dev); f = torch.randn(8, 47, device=dev); y = torch.randint(0, 12, (8,), device=dev)
loss = torch.nn.functional.cross_entropy(m(x, f), y); loss.backward()
ptrs = sorted((p.grad.data_ptr(), p.grad.numel() * 4, n) for n, p in m.named_parameters() if p.grad is not None)
lo, hi = ptrs[0][0], ptrs[-1][0] + ptrs[-1][1]
print("grad span MB", (hi - lo) / 1e6, "sum MB", sum(s for _, s, _ in ptrs) / 1e6)
print("storage sizes", {p.grad.untyped_storage().nbytes() for _, p in m.named_parameters() if p.grad is not None})
