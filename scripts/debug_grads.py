"""Developer aid: per-parameter gradient error against the golden fixtures (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg, rel_err, summary
case = sys.argv[1] if len(sys.argv) > 1 else "qs_quadtree_train"
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.bfloat16
g = np.load(os.path.join(ROOT, "tests/golden/train_b4.npz"))
P, synth = pkg(), pkg("synth")
dev = torch.device("cuda:0")
m = P.QuadtreeCNN(12, dropout_rate=0.0, freeze_backbone=(case == "rn_fusion_train"), compute_dtype=dt)
m.load_state_dict(synth.synth_state_dict(m)); m = m.to(dev).train()
x = synth.synth_images(4, salt=1).to(dev); f = synth.synth_pose_features(4, salt=1).to(dev); y = synth.synth_labels(4, 12, salt=1).to(dev)
logits = m(x, f); loss = torch.nn.functional.cross_entropy(logits, y); loss.backward(); torch.cuda.synchronize()
print("logits err", rel_err(logits.detach().cpu(), g[f"{case}/logits"]), "loss", loss.item(), float(g[f"{case}/loss"]))
params = dict(m.named_parameters())
for n in reversed(list(g[f"{case}/grad_names"])):
    gr = params[n].grad
    if gr is None: print(f"{n:50s} MISSING"); continue
    s = summary(gr); pre = f"{case}/grad/{n}"
    gs = g[f"{pre}/sample"]; sc = max(float(np.abs(gs).max()), 1e-30)
    e = float(np.abs(s["sample"] - gs).max()) / sc
    a = abs(s["abssum"] - float(g[f"{pre}/abssum"])) / max(float(g[f"{pre}/abssum"]), 1e-30)
    print(f"{n:50s} sample_err {e:9.2e} abssum_err {a:9.2e}")
