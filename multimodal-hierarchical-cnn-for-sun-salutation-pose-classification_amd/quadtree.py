"""nn.Module surface of the reference models, executed by the HIP plan.

Drop-in counterparts (same constructor arguments, attribute tree, state_dict
keys, forward signature, train()/eval() semantics) of
  QuadtreeCNN        /root/reference/Quadtree_from scratch/models.py:214-305
                     /root/reference/resnet/models.py:70-180   (`mode`, frozen backbone)
  StandardResNetCNN  /root/reference/resnet/models.py:7-65
  AttentionHierarchicalCNN  /root/reference/Quadtree_from scratch/models.py:6-101
  CnnLstm            /root/reference/cnn+lstm/models.py:14-89
"""
import torch
import torch.nn as nn

from . import engine as _engine
from . import modules as M
from ._lib import QtError

IMAGE_HW = 224


class _PlanModel(nn.Module):
    """Shared machinery: lazily builds a PlanEngine for the current device /
    batch / dtype, binds parameters + buffers by state_dict key and routes
    forward through PlanFunction."""

    _model_kind = _engine.QT_MODEL_QUADTREE

    def _init_plan_state(self, compute_dtype=None, max_batch=None):
        self.compute_dtype = compute_dtype or _engine.default_compute_dtype()
        self._max_batch_hint = max_batch
        self._engine = None
        self._engine_key = None
        self._param_list = None
        self._param_plan_index = None
        self._grad_sync = None  # set by dp.attach_data_parallel
        self._eval_keep_for_backward = False

    # engines hold device memory and ctypes handles: keep them out of pickles / deepcopy
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_engine", "_engine_key", "_param_list", "_param_plan_index", "_grad_sync"):
            state[k] = None
        return state

    def _plan_mode(self):
        return getattr(self, "mode", "fusion")

    def invalidate_packed_weights(self):
        """Call after changing parameters in a way torch does not record (no version bump, no
        backward since the last forward): the next forward re-packs the bf16/f32 operand copies."""
        if self._engine is not None:
            self._engine.invalidate_weights()

    def _ensure_engine(self, batch, device):
        key = (str(device), self.compute_dtype)
        if self._engine is not None and self._engine_key == key and batch <= self._engine.max_batch:
            return self._engine
        cap = max(batch, self._max_batch_hint or 0)
        self._engine = None  # free the old workspace first
        self._engine = _engine.PlanEngine(self._model_kind, self._plan_mode(), self.num_classes,
                                          self.numerical_feature_dim, self.dropout_rate, cap,
                                          self.compute_dtype, device, **self._plan_extra())
        self._engine_key = key
        self._param_list = None
        self._engine.grad_sync = self._grad_sync
        return self._engine

    def _plan_name(self, name):
        """state_dict key of this module -> name of the tensor in the plan's table"""
        return name

    def _plan_extra(self):
        return {}

    def _plan_writes_grad(self, plan_name):
        """Does backward produce a gradient for this tensor?  The reference leaves `.grad` of a branch its `mode`
        never runs at None (resnet/models.py:141-180: the branch is not in the autograd graph), and of the unused
        `base_cnn.fc`; the plan skips the same branches, so their gradients must not be requested (they would be
        views of uninitialised memory)."""
        if plan_name.startswith("base_cnn.fc."):
            return False
        mode = self._plan_mode()
        if mode == "image_only" and plan_name.startswith("numerical_mlp."):
            return False
        if mode == "numerical_only" and not plan_name.startswith(("numerical_mlp.", "classifier.")):
            return False
        return True

    def _eval_needs_backbone_backward(self):
        """eval() forward under autograd with a trainable ResNet parameter: the plan must keep what backward needs"""
        if not torch.is_grad_enabled() or self._param_list is None:
            return False
        for (name, _), p, idx in zip(self.named_parameters(), self._param_list, self._param_plan_index):
            if idx >= 0 and p.requires_grad and self._plan_name(name).startswith("base_cnn."):
                return True
        return False

    def _bind(self, eng):
        tensors = {self._plan_name(n): t for n, t in self.named_parameters()}
        tensors.update({self._plan_name(n): t for n, t in self.named_buffers()})
        eng.bind(tensors)
        if self._param_list is None:
            named = list(self.named_parameters())
            self._param_list = [p for _, p in named]
            self._param_plan_index = [
                eng.index.get(self._plan_name(n), -1) if self._plan_writes_grad(self._plan_name(n)) else -1
                for n, _ in named]
        return sum(p._version for p in self._param_list)

    # ---- Grad-CAM compatibility (reference: resnet/grad_cam_analysis.py:251-259,286,306-316;
    # Quadtree_from scratch/grad_cam.py:72-83).  The reference registers a forward hook and a
    # full backward hook on `model.base_cnn.layer4`.  The fused plan never calls that module, so
    # the hooks are served from the plan's buffers with the tensors torch would have passed:
    # output [B,512,7,7] f32 NCHW, grad_output[0] = d(loss)/d(layer4 output).
    def _layer4_nchw(self, engine, batch, name):
        t = engine.buffer(name, (engine.max_batch * 49, 512))[:batch * 49]
        return t.float().view(batch, 7, 7, 512).permute(0, 3, 1, 2).contiguous()

    def _fire_layer4_forward_hooks(self, engine, batch):
        if not hasattr(self, "base_cnn"):
            return
        layer4 = self.base_cnn.layer4
        if not layer4._forward_hooks or self._plan_mode() == "numerical_only":
            return
        out = self._layer4_nchw(engine, batch, "block7.out")
        inp = self._layer4_nchw_in(engine, batch)
        for hook in list(layer4._forward_hooks.values()):
            hook(layer4, (inp,), out)

    def _layer4_nchw_in(self, engine, batch):
        t = engine.buffer("block5.out", (engine.max_batch * 196, 256))[:batch * 196]
        return t.float().view(batch, 14, 14, 256).permute(0, 3, 1, 2).contiguous()

    def _fire_layer4_backward_hooks(self, engine, batch):
        if not hasattr(self, "base_cnn"):
            return
        layer4 = self.base_cnn.layer4
        if not layer4._backward_hooks or self._plan_mode() == "numerical_only":
            return
        # layer4's output only feeds AdaptiveAvgPool2d(1,1): its gradient is dfused[:, :512] / 49
        # broadcast over the 7x7 positions
        ld = engine.buffer_ld("dfused")
        d = engine.buffer("dfused", (engine.max_batch, ld))[:batch, :512].float() / 49.0
        g = d.view(batch, 512, 1, 1).expand(batch, 512, 7, 7).contiguous()
        for hook in list(layer4._backward_hooks.values()):
            hook(layer4, (None,), (g,))

    def _check_hooks(self):
        """The plan never calls the leaf modules, so a hook registered on one of them would silently never fire.
        Only `base_cnn.layer4` (the module the reference's Grad-CAM scripts hook, resnet/grad_cam_analysis.py:258-259)
        is served from the plan's buffers; hooks on any other submodule raise here instead of staying dead."""
        served = getattr(getattr(self, "base_cnn", None), "layer4", None)
        for name, mod in self.named_modules():
            if mod is self or mod is served:
                continue
            if mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks:
                raise QtError(f"hook registered on submodule {name!r}: the fused HIP plan only serves hooks on "
                              "`base_cnn.layer4` (forward hook + full backward hook, the Grad-CAM recipe) and on the "
                              "model itself; hooks on other submodules would never fire")

    def _run(self, image_input, numerical_input):
        self._check_hooks()
        ref = image_input if image_input is not None else numerical_input
        device = ref.device
        if device.type != "cuda":
            raise QtError("this build of the model runs on an AMD GPU only: move the model and its inputs "
                          "to cuda:N (there is no CPU fallback for the product path)")
        batch = int(ref.shape[0])
        if image_input is not None:
            if image_input.dim() != 4 or tuple(image_input.shape[1:]) != (3, IMAGE_HW, IMAGE_HW):
                raise ValueError(f"image_input must be [B,3,{IMAGE_HW},{IMAGE_HW}], got {tuple(image_input.shape)}")
            image_input = image_input.contiguous().float()
        if numerical_input is not None:
            if numerical_input.dim() != 2 or numerical_input.shape[1] != self.numerical_feature_dim or \
                    numerical_input.shape[0] != batch:
                raise ValueError(f"numerical_input must be [{batch},{self.numerical_feature_dim}], "
                                 f"got {tuple(numerical_input.shape)}")
            numerical_input = numerical_input.contiguous().float()
        with torch.cuda.device(device):
            eng = self._ensure_engine(batch, device)
            version = self._bind(eng)
            need_bwd = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list)
            eng.pack_weights(version, need_bwd)
            self._eval_keep_for_backward = (not self.training) and self._eval_needs_backbone_backward()
            return _engine.PlanFunction.apply(self, image_input, numerical_input, *self._param_list)


class QuadtreeCNN(_PlanModel):
    def __init__(self, num_classes, cnn_feature_dim=512, numerical_feature_dim=47, dropout_rate=0.5,
                 mode="fusion", freeze_backbone=False, compute_dtype=None, max_batch=None, pretrained=True):
        super().__init__()
        if cnn_feature_dim != 512:
            raise ValueError("the gfx950 kernels are specialised for cnn_feature_dim=512 (the reference default)")
        self.mode = mode
        self.num_classes = num_classes
        self.numerical_feature_dim = numerical_feature_dim
        self.dropout_rate = dropout_rate

        self.base_cnn = M.ResNet18()
        M.load_pretrained_resnet18(self.base_cnn, pretrained, frozen=freeze_backbone)
        if freeze_backbone:
            for param in self.base_cnn.parameters():
                param.requires_grad = False
        b = self.base_cnn
        self.features_extractor = nn.Sequential(b.conv1, b.bn1, b.relu, b.maxpool, b.layer1, b.layer2, b.layer3)
        self.quadrant_processor = nn.Sequential(
            M.Conv2d(256, cnn_feature_dim // 4, 3, padding=1), M.ReLU(inplace=True), M.MaxPool2d(2, 2))
        self.global_processor = nn.Sequential(b.layer4, b.avgpool)
        self.image_feature_dim = 512 + (cnn_feature_dim // 4) * 3 * 3 * 4
        assert self.image_feature_dim == 5120, f"Image feature dim mismatch: Expected 5120, got {self.image_feature_dim}"
        self.numerical_mlp = nn.Sequential(
            M.Linear(numerical_feature_dim, numerical_feature_dim * 2), M.ReLU(inplace=True),
            M.Dropout(dropout_rate), M.Linear(numerical_feature_dim * 2, cnn_feature_dim // 2))
        self.numerical_output_dim = cnn_feature_dim // 2
        if mode == "fusion":
            width = self.image_feature_dim + self.numerical_output_dim
        elif mode == "image_only":
            width = self.image_feature_dim
        elif mode == "numerical_only":
            width = self.numerical_output_dim
        else:
            raise ValueError(f"Invalid mode: {mode}. Choose from 'fusion', 'image_only', 'numerical_only', "
                             "'standard_resnet_only'.")
        self.final_classifier_input_dim = width
        self.combined_feature_dim = width
        self.classifier = nn.Sequential(
            M.Linear(width, width // 2), M.ReLU(inplace=True), M.Dropout(dropout_rate),
            M.Linear(width // 2, num_classes))
        # Grad-CAM attributes of resnet/models.py:131-139
        self.gradients = None
        self.activations = None
        self._init_plan_state(compute_dtype, max_batch)

    def save_gradient_hook(self, module, grad_input, grad_output):
        self.gradients = grad_output[0]

    def save_activation_hook(self, module, input, output):
        self.activations = output

    def forward(self, image_input, numerical_input):
        # inputs of an unused branch may be uninitialised memory (torch.empty dummies,
        # /root/reference/experiment/test_on_video_cnn.py:264-271): never touch them
        if self.mode == "numerical_only":
            image_input = None
        if self.mode == "image_only":
            numerical_input = None
        return self._run(image_input, numerical_input)


class StandardResNetCNN(_PlanModel):
    _model_kind = _engine.QT_MODEL_STANDARD_RESNET

    def __init__(self, num_classes, dropout_rate=0.5, compute_dtype=None, max_batch=None, pretrained=True):
        super().__init__()
        self.num_classes = num_classes
        self.numerical_feature_dim = 47
        self.dropout_rate = dropout_rate
        self.base_cnn = M.ResNet18()
        M.load_pretrained_resnet18(self.base_cnn, pretrained, frozen=True)
        for param in self.base_cnn.parameters():
            param.requires_grad = False
        b = self.base_cnn
        self.features_extractor = nn.Sequential(b.conv1, b.bn1, b.relu, b.maxpool, b.layer1, b.layer2, b.layer3,
                                                b.layer4)
        self.avgpool = b.avgpool
        self.classifier = nn.Sequential(M.Linear(512, 256), M.ReLU(inplace=True), M.Dropout(dropout_rate),
                                        M.Linear(256, num_classes))
        self.gradients = None
        self.activations = None
        self._init_plan_state(compute_dtype, max_batch)

    def save_gradient_hook(self, module, grad_input, grad_output):
        self.gradients = grad_output[0]

    def save_activation_hook(self, module, input, output):
        self.activations = output

    def forward(self, image_input, numerical_input=None):  # numerical_input is ignored (resnet/models.py:56)
        return self._run(image_input, None)


class AttentionHierarchicalCNN(_PlanModel):
    """/root/reference/Quadtree_from scratch/models.py:6-101: features to layer2 (28x28x128), global branch
    layer3 + layer4 + avgpool, a conv+ReLU+mean head on the 4 quadrants (14x14) and on the 16 sub-quadrants
    (7x7), an attention gate (64 -> 32 -> 1, softmax over the 16) that blends the sub-quadrant vectors, a
    one-layer numerical MLP and a 1216 -> 1024 -> C classifier.  As in the reference the ResNet is a local of
    __init__ (:11): there is no `base_cnn` attribute and the state_dict has 134 keys under
    features_extractor.{0,1,4,5}, global_processor.{0,1}, quadrant_processor.0, sub_quadrant_processor.0,
    attention_gate.{0,2}, numerical_mlp.0, classifier.{0,3}."""
    _model_kind = _engine.QT_MODEL_ATTENTION
    _PLAN_PREFIX = (("features_extractor.0.", "base_cnn.conv1."), ("features_extractor.1.", "base_cnn.bn1."),
                    ("features_extractor.4.", "base_cnn.layer1."), ("features_extractor.5.", "base_cnn.layer2."),
                    ("global_processor.0.", "base_cnn.layer3."), ("global_processor.1.", "base_cnn.layer4."))

    def __init__(self, num_classes, numerical_feature_dim=47, dropout_rate=0.5, compute_dtype=None, max_batch=None,
                 pretrained=True):
        super().__init__()
        self.num_classes = num_classes
        self.numerical_feature_dim = numerical_feature_dim
        self.dropout_rate = dropout_rate
        base_cnn = M.ResNet18()
        M.load_pretrained_resnet18(base_cnn, pretrained)
        b = base_cnn
        self.features_extractor = nn.Sequential(b.conv1, b.bn1, b.relu, b.maxpool, b.layer1, b.layer2)
        base_feature_channels = 128
        self.global_processor = nn.Sequential(b.layer3, b.layer4, b.avgpool)
        self.quadrant_processor = nn.Sequential(
            M.Conv2d(base_feature_channels, 128, 3, padding=1), M.ReLU(inplace=True), M.AdaptiveAvgPool2d((1, 1)))
        self.sub_quadrant_processor = nn.Sequential(
            M.Conv2d(base_feature_channels, 64, 3, padding=1), M.ReLU(inplace=True), M.AdaptiveAvgPool2d((1, 1)))
        self.attention_gate = nn.Sequential(M.Linear(64, 32), M.ReLU(), M.Linear(32, 1))
        total_image_feature_dim = 512 + 4 * 128 + 64
        self.numerical_mlp = nn.Sequential(M.Linear(numerical_feature_dim, 128), M.ReLU(inplace=True),
                                           M.Dropout(dropout_rate))
        combined_feature_dim = total_image_feature_dim + 128
        self.classifier = nn.Sequential(M.Linear(combined_feature_dim, 1024), M.ReLU(inplace=True),
                                        M.Dropout(dropout_rate), M.Linear(1024, num_classes))
        self._init_plan_state(compute_dtype, max_batch)

    def _plan_name(self, name):
        for mine, plan in self._PLAN_PREFIX:
            if name.startswith(mine):
                return plan + name[len(mine):]
        return name

    def forward(self, image_input, numerical_input):
        return self._run(image_input, numerical_input)


class CnnLstm(_PlanModel):
    """/root/reference/cnn+lstm/models.py:14-89: a frozen ResNet-18 (children()[:-1], i.e. up to avgpool) on every
    frame, Linear(47,128)-ReLU-Linear(128,128) on every pose vector, the two concatenated per time step into a
    2-layer LSTM (640 -> hidden, batch_first, dropout between layers), the last step's output into
    Linear(hidden,128)-ReLU-Dropout-Linear(128,C).  forward(image_sequence [B,T,3,224,224],
    numerical_sequence [B,T,47]) -> logits [B,C].  state_dict: cnn_backbone.{0,1,4,5,6,7}.*, numerical_mlp.{0,2}.*,
    lstm.{weight,bias}_{ih,hh}_l{0,1}, classifier.{0,3}.*."""
    _model_kind = _engine.QT_MODEL_CNN_LSTM
    _PLAN_PREFIX = (("cnn_backbone.0.", "base_cnn.conv1."), ("cnn_backbone.1.", "base_cnn.bn1."),
                    ("cnn_backbone.4.", "base_cnn.layer1."), ("cnn_backbone.5.", "base_cnn.layer2."),
                    ("cnn_backbone.6.", "base_cnn.layer3."), ("cnn_backbone.7.", "base_cnn.layer4."))

    def __init__(self, num_classes, sequence_length=4, numerical_feature_dim=47, dropout_rate=0.5, lstm_hidden_size=256,
                 compute_dtype=None, max_batch=None, pretrained=True):
        super().__init__()
        if lstm_hidden_size not in (256, 64):
            raise ValueError("the gfx950 LSTM kernel is instantiated for lstm_hidden_size 256 (reference default) and 64")
        self.sequence_length = sequence_length
        self.num_classes = num_classes
        self.numerical_feature_dim = numerical_feature_dim
        self.dropout_rate = dropout_rate
        self.lstm_hidden_size = lstm_hidden_size
        resnet = M.ResNet18()
        M.load_pretrained_resnet18(resnet, pretrained, frozen=True)
        self.cnn_backbone = nn.Sequential(*list(resnet.children())[:-1])
        for param in self.cnn_backbone.parameters():
            param.requires_grad = False
        self.numerical_mlp = nn.Sequential(M.Linear(numerical_feature_dim, 128), M.ReLU(), M.Linear(128, 128))
        self.lstm = M.LSTM(512 + 128, lstm_hidden_size, num_layers=2, batch_first=True, dropout=dropout_rate)
        self.classifier = nn.Sequential(M.Linear(lstm_hidden_size, 128), M.ReLU(), M.Dropout(dropout_rate),
                                        M.Linear(128, num_classes))
        self._seq_len_bound = None
        self._init_plan_state(compute_dtype, max_batch)

    def _plan_name(self, name):
        for mine, plan in self._PLAN_PREFIX:
            if name.startswith(mine):
                return plan + name[len(mine):]
        return name

    def _plan_extra(self):
        return {"seq_len": self._seq_len_bound, "lstm_hidden": self.lstm_hidden_size}

    def forward(self, image_sequence, numerical_sequence):
        if image_sequence.dim() != 5 or numerical_sequence.dim() != 3:
            raise ValueError("expected image_sequence [B,T,3,224,224] and numerical_sequence [B,T,F], got "
                             f"{tuple(image_sequence.shape)} and {tuple(numerical_sequence.shape)}")
        batch_size, seq_len = int(image_sequence.shape[0]), int(image_sequence.shape[1])
        if tuple(numerical_sequence.shape[:2]) != (batch_size, seq_len):
            raise ValueError("image_sequence and numerical_sequence disagree on [B,T]")
        if self._seq_len_bound != seq_len:  # the plan is laid out for whole sequences of one length
            self._seq_len_bound = seq_len
            self._engine = None
        if self._max_batch_hint and self._max_batch_hint % seq_len:
            self._max_batch_hint = (self._max_batch_hint // seq_len + 1) * seq_len
        frames = image_sequence.reshape(batch_size * seq_len, *image_sequence.shape[2:])
        poses = numerical_sequence.reshape(batch_size * seq_len, numerical_sequence.shape[2])
        return self._run(frames, poses)
