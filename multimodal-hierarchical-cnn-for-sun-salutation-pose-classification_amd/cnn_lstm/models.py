"""Drop-in for /root/reference/cnn+lstm/models.py (`from models import get_model`).

CnnLstm (:14-89) and get_model (:145-153) with the reference's signatures; the per-frame ResNet-18 runs on the
MFMA conv kernels, the LSTM recurrence in csrc/lstm.hip.  Ji3DCNN (:93-142, the '3d_cnn' option: Conv3d stream +
LSTM on the pose vectors) runs its 3x3x3 convolutions as three 2-D MFMA launches each (<pkg>/video3d.py).
"""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.basename(_PKG_DIR)
if os.path.dirname(_PKG_DIR) not in sys.path:
    sys.path.insert(0, os.path.dirname(_PKG_DIR))
_impl = importlib.import_module(_PKG + ".quadtree")
QtError = importlib.import_module(_PKG + "._lib").QtError
FusedAdam = importlib.import_module(_PKG + ".optim").FusedAdam  # optional replacement of optim.Adam(...)


class CnnLstm(_impl.CnnLstm):
    def __init__(self, num_classes, sequence_length=4, numerical_feature_dim=47, dropout_rate=0.5, lstm_hidden_size=256,
                 **kw):
        super().__init__(num_classes, sequence_length, numerical_feature_dim, dropout_rate, lstm_hidden_size, **kw)


class Ji3DCNN(importlib.import_module(_PKG + ".video3d").Ji3DCNN):
    """reference cnn+lstm/models.py:93-142: three Conv3d blocks + LSTM(47 -> 64) + classifier (<pkg>/video3d.py)"""

    def __init__(self, num_classes, sequence_length=4, numerical_feature_dim=47, dropout_rate=0.5, **kw):
        super().__init__(num_classes, sequence_length, numerical_feature_dim, dropout_rate, **kw)


def get_model(model_name, num_classes, device, seq_len=4, num_features=47):
    if model_name == 'cnn_lstm':
        model = CnnLstm(num_classes, sequence_length=seq_len, numerical_feature_dim=num_features)
    elif model_name == '3d_cnn':
        model = Ji3DCNN(num_classes, sequence_length=seq_len, numerical_feature_dim=num_features)
    else:
        raise ValueError(f"Unknown model name: {model_name}")
    return model.to(device)
