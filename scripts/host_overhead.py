"""Developer aid: host time to ENQUEUE one training step (no synchronisation) next to its GPU time."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd"
P = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda:0"); B = 256
m = P.QuadtreeCNN(12, compute_dtype=torch.bfloat16, max_batch=B); m.load_state_dict(synth.synth_state_dict(m)); m = m.to(dev).train()
opt = P.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-4, model=m)
x = torch.randn(B, 3, 224, 224, device=dev); f = torch.randn(B, 47, device=dev); y = torch.randint(0, 12, (B,), device=dev)
def step():
    opt.zero_grad(set_to_none=True); loss = torch.nn.functional.cross_entropy(m(x, f), y); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step (host), complete {1e3*(t2-t0)/20:.2f} ms/step")
