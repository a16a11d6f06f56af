"""Developer aid (GPU box): does replaying the eval forward / the train step from a HIP graph shorten it?
Captures `model(images, feats)` (eval) with torch.cuda.CUDAGraph and times eager vs replay."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("QTCNN_RESNET18_WEIGHTS", "none")
from _util import pkg
q = pkg("quadtree")
synth = pkg("synth")
dev = torch.device("cuda:0")
B = 256
model = q.QuadtreeCNN(num_classes=12, pretrained=False).to(dev)
model.eval()
g = torch.Generator().manual_seed(0)
images = torch.randn(B, 3, 224, 224, generator=g).to(dev)
feats = torch.randn(B, 47, generator=g).to(dev)

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

with torch.no_grad():
    eager = timeit(lambda: model(images, feats))
    print(f"eager eval forward: {eager:.3f} ms", flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): model(images, feats)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr):
            out = model(images, feats)
        rep = timeit(gr.replay)
        print(f"graph replay:       {rep:.3f} ms", flush=True)
        ref = model(images, feats)
        gr.replay(); torch.cuda.synchronize()
        print("max |graph - eager| =", float((out - ref).abs().max()))
    except Exception as e:
        print("capture failed:", repr(e)[:400])
