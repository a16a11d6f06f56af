import os, sys, time, torch, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg, PKG
P, synth = pkg(), pkg("synth")
dp = importlib.import_module(PKG + ".dp")
import torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
what = sys.argv[1]
if what != "none":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = P.QuadtreeCNN(12, max_batch=256)
m.load_state_dict(synth.synth_state_dict(m)); m = m.to(dev).train()
if what == "attach":
    dp.attach_data_parallel(m)
elif what == "bcast":
    dp.broadcast_state(m, 0)
elif what == "reducer":
    m._grad_sync = dp.GradBucketReducer()
elif what == "lambda":
    m._grad_sync = lambda *a: None
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
x = torch.randn(256, 3, 224, 224, device=dev); f = torch.randn(256, 47, device=dev); y = torch.randint(0, 12, (256,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(x, f), y).backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
if what == "barrier": dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print(what, round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms/step", flush=True)
