"""Developer aid: cost of running backward in its four phases (HEAD / LAYER4 / LAYER32 / LAYER1) without any collective."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg
P, synth = pkg(), pkg("synth")
dev = torch.device("cuda:0")
def run(phased):
    m = P.QuadtreeCNN(12, max_batch=256)
    m.load_state_dict(synth.synth_state_dict(m)); m = m.to(dev).train()
    if phased:
        m._grad_sync = lambda *a: None
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    x = torch.randn(256, 3, 224, 224, device=dev); f = torch.randn(256, 47, device=dev); y = torch.randint(0, 12, (256,), device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x, f), y).backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e3
for tp in (False, True, False, True):
    print("phased", tp, round(run(tp), 3), "ms/step", flush=True)
