"""Deterministic synthetic weights and inputs (torch-RNG independent).

The reference loads ImageNet ResNet-18 weights by URL
(/root/reference/Quadtree_from scratch/models.py:221, resnet/models.py:12,76);
that file is not available offline, so parity fixtures, tests and the benchmark
fill every tensor of a `state_dict` with a rule that depends only on the
tensor's state_dict key, its shape and the flat element index.  The same rule
regenerates identical tensors in the golden-vector generator (which runs the
reference) and on the GPU box (which cannot see the reference), so the 100+ MB
of weights never have to be committed.
"""
import zlib

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def hash_uniform(tag, n, salt=0):
    """n float64 values in [0, 1), a pure function of (tag, salt, index)."""
    seed = np.uint64(zlib.crc32(tag.encode()) + (int(salt) << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + _splitmix64(np.array([seed], dtype=np.uint64))[0]
        bits = _splitmix64(idx)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def hash_normal(tag, n, salt=0):
    """Approximately N(0,1): Box-Muller on two hashed uniforms."""
    u1 = hash_uniform(tag + "/u1", n, salt)
    u2 = hash_uniform(tag + "/u2", n, salt)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def _fill(key, shape, lo, hi, salt):
    n = int(np.prod(shape)) if len(shape) else 1
    v = lo + (hi - lo) * hash_uniform(key, n, salt)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def synth_tensor(key, ref, salt=0):
    """Value for one state_dict entry `key` shaped/typed like `ref`."""
    shape = tuple(ref.shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == "running_mean":
        return _fill(key, shape, -0.3, 0.3, salt)
    if leaf == "running_var":
        return _fill(key, shape, 0.6, 1.4, salt)
    if leaf == "weight" and len(shape) == 1:  # BatchNorm gamma
        return _fill(key, shape, 0.6, 1.4, salt)
    if leaf == "bias":
        return _fill(key, shape, -0.2, 0.2, salt)
    if leaf == "weight" and len(shape) == 4:  # conv, Kaiming-uniform (ReLU gain)
        a = float(np.sqrt(6.0 / (shape[1] * shape[2] * shape[3])))
        return _fill(key, shape, -a, a, salt)
    if leaf == "weight" and len(shape) == 5:  # conv3d, Kaiming-uniform (ReLU gain)
        a = float(np.sqrt(6.0 / (shape[1] * shape[2] * shape[3] * shape[4])))
        return _fill(key, shape, -a, a, salt)
    if leaf.startswith("bias_") and len(shape) == 1:  # nn.LSTM bias_ih_l{k} / bias_hh_l{k}
        return _fill(key, shape, -0.2, 0.2, salt)
    if leaf.startswith("weight_") and len(shape) == 2:  # nn.LSTM weight_ih_l{k} / weight_hh_l{k}
        a = float(np.sqrt(3.0 / shape[1]))
        return _fill(key, shape, -a, a, salt)
    if leaf == "weight" and len(shape) == 2:  # linear
        a = float(np.sqrt(3.0 / shape[1]))
        return _fill(key, shape, -a, a, salt)
    raise ValueError(f"no synthetic rule for {key} {shape}")


def synth_state_dict(model, salt=0):
    """A full state_dict for `model`; aliased keys (same storage) get the value
    of their first-seen key so that load_state_dict is order independent."""
    out, seen = {}, {}
    for key, ref in model.state_dict().items():
        ident = (ref.data_ptr(), tuple(ref.shape)) if ref.numel() else (key, ())
        if ident not in seen:
            seen[ident] = synth_tensor(key, ref, salt)
        out[key] = seen[ident]
    return out


def synth_images(batch, salt=0, size=224):
    """[B,3,size,size] f32 NCHW, roughly the N(0,1) of a normalised image
    (/root/reference/Quadtree_from scratch/dataloader.py:39-43)."""
    v = hash_normal("image", batch * 3 * size * size, salt)
    return torch.from_numpy(v.astype(np.float32)).reshape(batch, 3, size, size)


def synth_pose_features(batch, salt=0, realistic=True):
    """[B,47] f32.  realistic=True mimics the un-standardised MediaPipe vector
    (/root/reference/experiment/1_prepare_still_image_dataset.py:101-113):
    33 visibilities in [0,1], 10 angles in [0,180], 3 distances in [0,5],
    1 positive ratio.  realistic=False gives N(0,1) (the benchmark input)."""
    if not realistic:
        v = hash_normal("pose", batch * 47, salt)
        return torch.from_numpy(v.astype(np.float32)).reshape(batch, 47)
    u = hash_uniform("pose", batch * 47, salt).reshape(batch, 47)
    v = np.empty_like(u)
    v[:, :33] = u[:, :33]
    v[:, 33:43] = 180.0 * u[:, 33:43]
    v[:, 43:46] = 5.0 * u[:, 43:46]
    v[:, 46] = np.exp(2.0 * u[:, 46] - 1.0)
    return torch.from_numpy(v.astype(np.float32))


def synth_labels(batch, num_classes=12, salt=0):
    u = hash_uniform("label", batch, salt)
    return torch.from_numpy(np.minimum((u * num_classes).astype(np.int64), num_classes - 1))
