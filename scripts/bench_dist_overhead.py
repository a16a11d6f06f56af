"""Developer aid: does merely initialising torch.distributed (RCCL) slow the single-GPU step?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg
P, synth = pkg(), pkg("synth")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
def run():
    m = P.QuadtreeCNN(12, max_batch=256)
    m.load_state_dict(synth.synth_state_dict(m)); m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    x = torch.randn(256, 3, 224, 224, device=dev); f = torch.randn(256, 47, device=dev); y = torch.randint(0, 12, (256,), device=dev)
    def step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x, f), y).backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e3
print("before init", round(run(), 3), flush=True)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
mode = sys.argv[1] if len(sys.argv) > 1 else "devid"
if mode == "devid":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
print("after init (no collective yet)", round(run(), 3), flush=True)
t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
print("after first collective", round(run(), 3), flush=True)
dist.destroy_process_group()
print("after destroy", round(run(), 3), flush=True)
