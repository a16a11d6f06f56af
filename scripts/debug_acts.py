"""Developer aid: compare plan-internal activations / gradients with the CPU oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg, rel_err
import oracle.quadtree_oracle as o
dt = torch.float32
P, synth = pkg(), pkg("synth")
dev = torch.device("cuda:0")
B = 4
m = P.QuadtreeCNN(12, dropout_rate=0.0, compute_dtype=dt)
m.load_state_dict(synth.synth_state_dict(m))
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
x = synth.synth_images(B, salt=1); f = synth.synth_pose_features(B, salt=1); y = synth.synth_labels(B, 12, salt=1)
logits = m(x.to(dev), f.to(dev)); loss = torch.nn.functional.cross_entropy(logits, y.to(dev)); loss.backward(); torch.cuda.synchronize()
# oracle with retained taps
keys = o.trainable_keys(sd0, False)
sd = o.unique_params(sd0, keys)
taps = {}
_bb = o._basic_block
def bb(sd_, prefix, x_, stride, train):
    out = _bb(sd_, prefix, x_, stride, train)
    taps["blk:" + prefix] = out
    return out
o._basic_block = bb
ref = o.quadtree_forward(sd, x, f, train=True, dropout_p=0.0, taps=taps)
for t in taps.values(): t.retain_grad()
torch.nn.functional.cross_entropy(ref, y).backward()
eng = m._engine
def nchw(t, C, H): return t.float().cpu().view(B, H, H, C).permute(0, 3, 1, 2)
blocks = [(f"blk:base_cnn.layer{L}.{b}", (L-1)*2+b, 64 << (L-1), 56 >> (L-1)) for L in range(1,5) for b in range(2)]
for name, blk, C, H in blocks:
    a = nchw(eng.buffer(f"block{blk}.out", (B*H*H, C)), C, H)
    print(name, "out err", rel_err(a, taps[name].detach()))
    g = nchw(eng.buffer(f"block{blk}.gout", (B*H*H, C)), C, H)
    gref = taps[name].grad * (taps[name].detach() > 0)
    print(name, "gout err", rel_err(g, gref), float(gref.abs().max()))
name, blk, C, H = "blk:base_cnn.layer4.0", 6, 512, 7
g = nchw(eng.buffer(f"block{blk}.gout", (B*H*H, C)), C, H)
gref = taps[name].grad * (taps[name].detach() > 0)
d = (g - gref).abs()
print("err by image", d.amax((1,2,3)))
print("err by row h", d.amax((0,1,3)))
print("err by col w", d.amax((0,1,2)))
ch = d.amax((0,2,3)); print("bad channels", (ch > 1e-6).sum().item(), (ch > 1e-6).nonzero().flatten()[:40])
nz = (d > 1e-6).nonzero(); print("num bad", nz.shape[0], "of", d.numel()); print(nz[:20])
