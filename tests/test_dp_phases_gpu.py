"""GPU: the data-parallel form of backward (four qt_plan_backward phase calls, each followed by the hand-over of one
gradient bucket to the reducer: <pkg>/engine.py::backward, <pkg>/dp.py) must produce the gradients of the one-call
backward, and the buckets must tile the flat gradient buffer exactly once, in the order the phases finish them.
Runs with a recording stand-in for the reducer (world size 1: no collective needed to check the phase split); the
collective itself is covered on CPU by tests/test_dp_gloo.py.
"""
import pytest
import torch
import torch.nn.functional as F

from _util import pkg, rel_err

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class Recorder:
    """stands where dp.GradBucketReducer stands: callable(bucket, phase, side_fence)"""

    def __init__(self):
        self.calls = []
        self.joined = 0
        self.stream = None

    def __call__(self, bucket, phase, side_fence=None):
        if bucket is None:
            self.joined += 1
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=bucket.device)
        # what the real reducer does around its all-reduce: order the bucket behind both producer streams
        self.stream.wait_stream(torch.cuda.current_stream())
        side_fence(self.stream)
        with torch.cuda.stream(self.stream):
            bucket.mul_(1.0)  # a kernel on the communication stream that touches the whole bucket
        self.calls.append((phase, bucket.data_ptr(), bucket.numel()))


def _model(kind, dt):
    P, synth = pkg(), pkg("synth")
    if kind == "quadtree":
        m = P.QuadtreeCNN(12, dropout_rate=0.0, compute_dtype=dt)
    elif kind == "attention":
        m = P.AttentionHierarchicalCNN(12, dropout_rate=0.0, compute_dtype=dt)
    else:
        m = P.CnnLstm(12, sequence_length=3, dropout_rate=0.0, compute_dtype=dt)
    m.load_state_dict(synth.synth_state_dict(m))
    return m


@pytest.mark.parametrize("kind", ["quadtree", "attention", "cnn_lstm"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_phase_split_backward_equals_one_call_backward(kind, dt):
    dev = _dev()
    synth = pkg("synth")
    B = 6
    x, f = synth.synth_images(B, salt=91).to(dev), synth.synth_pose_features(B, salt=91).to(dev)
    if kind == "cnn_lstm":
        x, f = x.view(2, 3, 3, 224, 224), f.view(2, 3, 47)
    y = synth.synth_labels(x.shape[0], 12, salt=91).to(dev)
    m = _model(kind, dt).to(dev).train()

    def grads():
        for p in m.parameters():
            p.grad = None
        F.cross_entropy(m(x, f), y).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    one = grads()
    rec = Recorder()
    m._grad_sync = rec
    m._engine.grad_sync = rec
    split = grads()
    assert sorted(one) == sorted(split)
    for n in one:
        # same kernels, same inputs: only the order of float atomics in the generic weight-gradient kernel may differ
        assert rel_err(split[n].cpu(), one[n].cpu()) <= 2e-5, n
    assert rec.joined == 1
    phases = [c[0] for c in rec.calls]
    assert phases == sorted(phases) and phases[0] == 1 and len(set(phases)) == len(phases)
    # the buckets are consecutive slices of one flat buffer and cover every gradient exactly once
    total = sum(((g.numel() + 3) // 4) * 4 for g in split.values())
    assert sum(c[2] for c in rec.calls) == total
    for (_, p0, n0), (_, p1, _n1) in zip(rec.calls, rec.calls[1:]):
        assert p1 == p0 + 4 * n0
    if kind == "quadtree":
        assert phases == [1, 2, 4, 8]
        assert rec.calls[-1][2] * 4 < 1 << 20   # only the small layer1 + stem bucket is exposed after the last kernel
    if kind == "cnn_lstm":
        assert phases == [1]                      # frozen backbone: one bucket (pose MLP, LSTM, classifier)
