// Native executor ("plan") for the QuadtreeCNN / StandardResNetCNN hot path.
//
// One object describes the whole network for a fixed maximum batch: the tensor
// table (names = the reference's state_dict keys), the workspace
// layout in HBM and the launch sequence.  The host language makes three calls per
// training step (pack weights, forward, backward); every kernel is enqueued from
// here on the caller's stream, no synchronisation, no allocation.
//
// Graph restated (not copied) from
//   /root/reference/Quadtree_from scratch/models.py:216-305  (QuadtreeCNN)
//   /root/reference/resnet/models.py:7-65,70-180              (StandardResNetCNN, modes)
//   /root/reference/Quadtree_from scratch/models.py:6-101    (AttentionHierarchicalCNN: split after layer2,
//                                                             quadrant + sub-quadrant heads, attention gate)
//   /root/reference/cnn+lstm/models.py:14-89                 (CnnLstm: frozen per-frame ResNet-18, pose MLP,
//                                                             2-layer LSTM over the sequence, classifier)
// and torchvision's ResNet-18 BasicBlock wiring.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "qt_common.h"

// QTCNN_S2_DGRAD_MERGED (default 1): the data gradient of a 3x3 stride-2 conv is one 2x2-tap launch over the gradient
// map (qt_conv_desc.dst_merge) instead of four parity-class gathers (0: same-box A/B; identical sums up to f32 order)
static bool merged_s2_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("QTCNN_S2_DGRAD_MERGED");
    v = e ? (atoi(e) != 0) : 1;
  }
  return v != 0;
}

// QTCNN_DS_SLOT (default 1): where the generic tile runs the merged stride-2 data gradient (28x28 / 14x14 gradient maps), the
// block's 1x1 / stride-2 downsample rides in the same launch as a fifth tap slot (qt_conv_desc.dst_merge_extra) instead of
// a launch of its own whose output is then re-read as a residual (0: same-box A/B)
static bool ds_slot_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("QTCNN_DS_SLOT");
    v = e ? (atoi(e) != 0) : 1;
  }
  return v != 0;
}

namespace {

constexpr int kImg = 224;

struct TensorInfo {
  std::string name;
  int kind;  // 0 parameter f32, 1 buffer f32, 2 int64 counter
  int ndim;
  int shape[4];
};

struct BnL {
  int gamma, beta, rmean, rvar, nbt, C;
  size_t mean, invstd, scale, shift, coef;
};

struct ConvL {
  int w, bias;  // tensor indices (bias -1 if none)
  int cin, cout, k, stride, pad, hin, hout;
  int bn;  // -1 if none
  int regions = 1;  // images the conv sees per input image (quadrant heads: 4 or 16 regions of the shared map)
  size_t w_fwd, w_dgrad, dw;
  size_t y, gy;
  // stride-2 data gradient as four parity-class gathers (qt_pack_dgrad_s2)
  long long cls_off[4];
  int cls_kh[4], cls_kw[4];
  bool merged_dgrad = false;   // 3x3 stride 2: one 2x2-tap launch for all four parity classes
  bool merged5 = false;        // ... whose operand has a fifth tap slot per row: the block's downsample (slot_conv) rides along
  int slot_conv = -1;          // merged5: the downsample conv;  on the downsample: the conv1 whose operand holds its slot
};

struct Block {
  int conv1, conv2, ds;
  size_t a1, out, gout, gtmp;
  size_t a1_bits, out_bits;   // ReLU masks of a1 / out, one bit per element (written by the training forward)
};

struct LinL {
  int w, b, in, out;
  size_t w_fwd, w_dgrad;
};

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

struct qt_plan {
  qt_plan_desc d;
  int esz;
  std::vector<TensorInfo> tensors;
  std::vector<BnL> bns;
  std::vector<ConvL> convs;  // 0 stem, 1..19 blocks, 20 quadrant conv (if present)
  std::vector<Block> blocks;
  int quad_conv = -1, sub_conv = -1;
  bool has_image = true, has_numerical = false, standard = false, attention = false;
  LinL mlp0{}, mlp1{}, cls0{}, cls3{}, att0{}, att2{};
  // attention head (AttentionHierarchicalCNN): sub-quadrant vectors [B][16][64] f32 and what the gate's backward needs
  size_t vsub = 0, dvsub = 0, att_act = 0, att_alpha = 0, att_ds = 0, att_dpre = 0, gbase_tmp2 = 0;
  // CnnLstm: nn.LSTM layers (all f32); `batch` counts frames, sequences = frames / seq_len
  bool lstm = false;
  int seq_len = 0, lstm_h = 0;
  struct LstmL {
    int w_ih, w_hh, b_ih, b_hh, in;
    size_t whh_t, wih_t, xproj, gates, cell, hprev, hout, dgates;
  } lstm_l[2];
  size_t lstm_x0 = 0;  // f32 copy of the fused features (bf16 build)
  size_t lstm_x1 = 0, lstm_dx1 = 0, lstm_dlast = 0, lstm_hid = 0, lstm_dhid = 0, lstm_dz = 0;
  int fused_ld = 0, img_cols = 0, mlp_col0 = 0, hidden_dim = 0;
  // workspace offsets
  size_t ws_bytes = 0;
  size_t xpad = 0, p0 = 0, argmax = 0, g_p0 = 0, ymax = 0;
  size_t q = 0, dq = 0, fused = 0, dfused = 0, h1 = 0, dh1 = 0, hidden = 0, dhidden = 0;
  size_t stats = 0, ones = 0, zeros = 0, gbase_tmp = 0;
  size_t stats_bn2 = 0, stats_ds = 0, stats_bn1 = 0;
  size_t dw_begin = 0, dw_end = 0;
  size_t lin_ws = 0, lin_ws_bytes = 0;          // split-K partials of classifier.0 (forward / backward-input)
  size_t wgrad_part = 0, wgrad_part_bytes = 0;  // partial filters of the streaming weight-gradient kernel
  // The packed operand copies of every conv / linear weight are current (forward copies; data-gradient copies) for the
  // master tensors at these addresses: lets qt_plan_adam_step leave FROZEN weights alone (resnet/ variant, CnnLstm: 11 M
  // backbone weights that used to be re-packed every step).  Any qt_plan_pack_weights call re-establishes it.
  bool packed_fwd = false, packed_bwd = false;
  unsigned long long packed_sig = 0;
  bool dw_dirty = true;  // weight-gradient scratch holds sums of an earlier backward
  int bwd_rows_bn2 = 0;  // carried from the layer4 phase to the rest-of-backbone phase  // BatchNorm-backward partials emitted by dgrad epilogues
  // optional per-launch timing of the MFMA kernels (bench.py roofline): HIP events on
  // the launch stream around every igemm / wgrad launch while enabled
  struct Timed { hipEvent_t a, b; double flops; double bytes; int kind; };
  double last_profile_bytes[3] = {0, 0, 0};   // algorithmic HBM bytes per kind of the last profile (each operand once)
  bool profiling = false;
  std::vector<Timed> timed;
  // weight gradients run on a plan-owned side stream, concurrently with the BatchNorm /
  // data-gradient chain of the next layer (they only share read-only inputs); joined at the
  // end of every backward phase.  QTCNN_SIDE_STREAM=0 keeps everything on the caller's stream.
  bool use_side = true;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // state of the last forward
  int last_batch = 0, last_training = 0;
  unsigned long long last_seed = 0;

  int add_tensor(const std::string& name, int kind, std::initializer_list<int> shape) {
    TensorInfo t;
    t.name = name;
    t.kind = kind;
    t.ndim = (int)shape.size();
    int i = 0;
    for (int s : shape) t.shape[i++] = s;
    for (; i < 4; ++i) t.shape[i] = 1;
    tensors.push_back(t);
    return (int)tensors.size() - 1;
  }
  int add_bn(const std::string& prefix, int C) {
    BnL b;
    b.C = C;
    b.gamma = add_tensor(prefix + ".weight", 0, {C});
    b.beta = add_tensor(prefix + ".bias", 0, {C});
    b.rmean = add_tensor(prefix + ".running_mean", 1, {C});
    b.rvar = add_tensor(prefix + ".running_var", 1, {C});
    b.nbt = add_tensor(prefix + ".num_batches_tracked", 2, {});
    bns.push_back(b);
    return (int)bns.size() - 1;
  }
  int add_conv(const std::string& wname, int cin, int cout, int k, int stride, int pad, int hin, int bn, bool bias) {
    ConvL c;
    c.w = add_tensor(wname + ".weight", 0, {cout, cin, k, k});
    c.bias = bias ? add_tensor(wname + ".bias", 0, {cout}) : -1;
    c.cin = cin; c.cout = cout; c.k = k; c.stride = stride; c.pad = pad;
    c.hin = hin; c.hout = (hin + 2 * pad - k) / stride + 1;
    c.bn = bn;
    convs.push_back(c);
    return (int)convs.size() - 1;
  }
  LinL add_linear(const std::string& name, int in, int out) {
    LinL l;
    l.w = add_tensor(name + ".weight", 0, {out, in});
    l.b = add_tensor(name + ".bias", 0, {out});
    l.in = in; l.out = out;
    l.w_fwd = l.w_dgrad = 0;
    return l;
  }
};

namespace {

struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes);
    return o;
  }
};

void build_graph(qt_plan* p) {
  const qt_plan_desc& d = p->d;
  p->standard = d.model == QT_MODEL_STANDARD_RESNET;
  p->attention = d.model == QT_MODEL_ATTENTION;
  p->lstm = d.model == QT_MODEL_CNN_LSTM;
  if (p->lstm) p->standard = true;  // the per-frame branch is the ResNet-18 up to avgpool, no quadrant head
  p->seq_len = d.seq_len;
  p->lstm_h = d.lstm_hidden;
  p->has_image = p->standard || p->attention || d.mode != QT_MODE_NUMERICAL_ONLY;
  p->has_numerical = p->attention || p->lstm || (!p->standard && d.mode != QT_MODE_IMAGE_ONLY);
  // ---- backbone (always present in the tensor table: state_dict parity) ----
  {
    const int w = p->add_tensor("base_cnn.conv1.weight", 0, {64, 3, 7, 7});
    const int bn = p->add_bn("base_cnn.bn1", 64);
    ConvL c;
    c.w = w; c.bias = -1; c.cin = 3; c.cout = 64; c.k = 7; c.stride = 2; c.pad = 3; c.hin = kImg; c.hout = 112; c.bn = bn;
    p->convs.push_back(c);
  }
  int h = 56, cin = 64;
  for (int L = 1; L <= 4; ++L) {
    const int cout = 64 << (L - 1);
    for (int b = 0; b < 2; ++b) {
      const std::string pre = "base_cnn.layer" + std::to_string(L) + "." + std::to_string(b);
      const int stride = (b == 0 && L > 1) ? 2 : 1;
      Block blk;
      // tensor order follows torchvision's BasicBlock: conv1, bn1, conv2, bn2, downsample
      const int w1 = p->add_tensor(pre + ".conv1.weight", 0, {cout, cin, 3, 3});
      const int bn1 = p->add_bn(pre + ".bn1", cout);
      const int w2 = p->add_tensor(pre + ".conv2.weight", 0, {cout, cout, 3, 3});
      const int bn2 = p->add_bn(pre + ".bn2", cout);
      ConvL c1;
      c1.w = w1; c1.bias = -1; c1.cin = cin; c1.cout = cout; c1.k = 3; c1.stride = stride; c1.pad = 1;
      c1.hin = h; c1.hout = (h + 2 - 3) / stride + 1; c1.bn = bn1;
      p->convs.push_back(c1);
      blk.conv1 = (int)p->convs.size() - 1;
      ConvL c2;
      c2.w = w2; c2.bias = -1; c2.cin = cout; c2.cout = cout; c2.k = 3; c2.stride = 1; c2.pad = 1;
      c2.hin = c1.hout; c2.hout = c1.hout; c2.bn = bn2;
      p->convs.push_back(c2);
      blk.conv2 = (int)p->convs.size() - 1;
      blk.ds = -1;
      if (stride != 1 || cin != cout) {
        const int wd = p->add_tensor(pre + ".downsample.0.weight", 0, {cout, cin, 1, 1});
        const int bnd = p->add_bn(pre + ".downsample.1", cout);
        ConvL cd;
        cd.w = wd; cd.bias = -1; cd.cin = cin; cd.cout = cout; cd.k = 1; cd.stride = stride; cd.pad = 0;
        cd.hin = h; cd.hout = c1.hout; cd.bn = bnd;
        p->convs.push_back(cd);
        blk.ds = (int)p->convs.size() - 1;
        if (stride == 2) {   // candidates for the shared data-gradient launch (decided in layout_workspace)
          p->convs[blk.conv1].slot_conv = blk.ds;
          p->convs[blk.ds].slot_conv = blk.conv1;
        }
      }
      p->blocks.push_back(blk);
      h = c1.hout;
      cin = cout;
    }
  }
  if (!p->attention && !p->lstm) {  // (those models drop / never register the ResNet's fc: models.py:11, cnn+lstm/models.py:23)
    p->add_tensor("base_cnn.fc.weight", 0, {1000, 512});  // present in the state_dict, never used
    p->add_tensor("base_cnn.fc.bias", 0, {1000});
  }
  if (p->attention) {
    // models.py:21-30: both heads read layer2's 28x28x128 map, as 4 regions of 14x14 and 16 regions of 7x7
    p->quad_conv = p->add_conv("quadrant_processor.0", 128, 128, 3, 1, 1, 14, -1, true);
    p->convs[p->quad_conv].regions = 4;
    p->sub_conv = p->add_conv("sub_quadrant_processor.0", 128, 64, 3, 1, 1, 7, -1, true);
    p->convs[p->sub_conv].regions = 16;
    p->att0 = p->add_linear("attention_gate.0", 64, 32);
    p->att2 = p->add_linear("attention_gate.2", 32, 1);
    p->mlp0 = p->add_linear("numerical_mlp.0", d.numerical_dim, 128);
    p->img_cols = 512 + 4 * 128 + 64;  // global + four quadrant vectors + the attended sub-quadrant vector (:41)
    p->fused_ld = p->img_cols + 128;
    p->mlp_col0 = p->img_cols;
  } else if (!p->standard) {
    p->quad_conv = p->add_conv("quadrant_processor.0", 256, 128, 3, 1, 1, 7, -1, true);
    p->convs[p->quad_conv].regions = 4;
    p->mlp0 = p->add_linear("numerical_mlp.0", d.numerical_dim, d.numerical_dim * 2);
    p->mlp1 = p->add_linear("numerical_mlp.3", d.numerical_dim * 2, 256);
    p->img_cols = 512 + 4 * 1152;
    p->fused_ld = (p->has_image ? p->img_cols : 0) + (p->has_numerical ? 256 : 0);
    p->mlp_col0 = p->has_image ? p->img_cols : 0;
  } else if (p->lstm) {
    // cnn+lstm/models.py:31-56: MLP 47 -> 128 -> 128, LSTM(640 -> H, 2 layers), classifier H -> 128 -> C
    const int H = p->lstm_h;
    p->mlp0 = p->add_linear("numerical_mlp.0", d.numerical_dim, 128);
    p->mlp1 = p->add_linear("numerical_mlp.2", 128, 128);
    p->img_cols = 512;
    p->fused_ld = 512 + 128;
    p->mlp_col0 = 512;
    for (int l = 0; l < 2; ++l) {
      qt_plan::LstmL& L = p->lstm_l[l];
      const std::string sfx = "_l" + std::to_string(l);
      L.in = l == 0 ? p->fused_ld : H;
      L.w_ih = p->add_tensor("lstm.weight_ih" + sfx, 0, {4 * H, L.in});
      L.w_hh = p->add_tensor("lstm.weight_hh" + sfx, 0, {4 * H, H});
      L.b_ih = p->add_tensor("lstm.bias_ih" + sfx, 0, {4 * H});
      L.b_hh = p->add_tensor("lstm.bias_hh" + sfx, 0, {4 * H});
    }
    p->hidden_dim = 128;
    p->cls0 = p->add_linear("classifier.0", H, 128);
    p->cls3 = p->add_linear("classifier.3", 128, d.num_classes);
    return;
  } else {
    p->img_cols = 512;
    p->fused_ld = 512;
    p->mlp_col0 = 0;
  }
  p->hidden_dim = p->standard ? 256 : (p->attention ? 1024 : p->fused_ld / 2);
  p->cls0 = p->add_linear("classifier.0", p->fused_ld, p->hidden_dim);
  p->cls3 = p->add_linear("classifier.3", p->hidden_dim, d.num_classes);
}

void layout_workspace(qt_plan* p) {
  Bump ws;
  const size_t B = (size_t)p->d.batch;
  const size_t es = (size_t)p->esz;
  // constants
  p->ones = ws.take(2048 * 4);
  p->zeros = ws.take(2048 * 4);
  // statistics scratch: the stem has the most partial rows
  {
    const int rows = qt_stats_capacity_rows(qt_cdiv((long long)B * 112 * 112, 128));
    size_t bytes = (size_t)rows * 2 * 64 * 4;
    const size_t bwd = (size_t)qt_stats_capacity_rows(2048) * 2 * 512 * 4;
    p->stats = ws.take(bytes > bwd ? bytes : bwd);
    // dgrad-epilogue partials: one row per 128-pixel tile (+ the fold rows); layer1 is the largest
    // (... or, larger, the merged stride-2 data gradient of layer2.0 on the patch-resident kernel: one row per 196-pixel
    // tile of the 28x28 gradient map, wave row and parity class = 32 rows per image)
    const size_t ep0 = (size_t)qt_stats_capacity_rows(qt_cdiv((long long)B * 56 * 56, 128) + 8) * 2 * 64 * 4;
    const size_t ep1 = (size_t)qt_stats_capacity_rows((int)(32 * B + 8)) * 2 * 64 * 4;
    const size_t ep = ep0 > ep1 ? ep0 : ep1;
    p->stats_bn2 = ws.take(ep);
    p->stats_ds = ws.take(ep);
    p->stats_bn1 = ws.take(ep);
  }
  for (auto& b : p->bns) {
    b.mean = ws.take(b.C * 4);
    b.invstd = ws.take(b.C * 4);
    b.scale = ws.take(b.C * 4);
    b.shift = ws.take(b.C * 4);
    b.coef = ws.take(3 * b.C * 4);
  }
  // packed weights + KRSC gradient scratch
  for (size_t i = 0; i < p->convs.size(); ++i) {
    ConvL& c = p->convs[i];
    const size_t n = (i == 0) ? (size_t)64 * 8 * 32 : (size_t)c.cout * c.cin * c.k * c.k;
    if (i > 0 && c.stride == 2)
      qt_pack_dgrad_s2(p->d.dtype, reinterpret_cast<const float*>(8), nullptr, c.cout, c.cin, c.k, c.cls_off, c.cls_kh,
                       c.cls_kw, nullptr);
    c.w_fwd = ws.take(n * es);
    // 3x3 stride 2: the data-gradient operand holds all four parity classes with 2 x 2 tap slots each (16/9 of the
    // filter, the unused slots zero: qt_pack_dgrad_s2_merged)
    c.merged_dgrad = i > 0 && c.stride == 2 && c.k == 3 && merged_s2_enabled();
    // (conv_pt.hip serves the 7x7 gradient map with its four-tap instantiation: no fifth slot there)
    c.merged5 = c.merged_dgrad && c.slot_conv >= 0 && c.hout >= 14 && ds_slot_enabled();
    if (c.k == 3 && c.slot_conv >= 0 && !c.merged5) {   // no shared launch: the downsample keeps its own data gradient
      p->convs[c.slot_conv].slot_conv = -1;
      c.slot_conv = -1;
    }
    c.w_dgrad = ws.take((c.merged5 ? (size_t)20 * c.cout * c.cin : c.merged_dgrad ? (size_t)16 * c.cout * c.cin : n) * es);
  }
  // f32 [O][kh][kw][I] weight-gradient scratch, contiguous, zeroed by ONE memset per backward.  Only the convolutions whose
  // weight gradient is ACCUMULATED into it get a slice: the stem, the region heads (quadrant / sub-quadrant convs) and, in the
  // f32 build, every 3x3.  The streaming kernels (bf16: 3x3 stride 1, the stride-2 pair of a transition block) write .grad
  // in OIHW themselves and the generic 1x1 path zeroes .grad itself: round 3 zeroed 45 MB here per step, 11 MB of it needed.
  p->dw_begin = ws.off;
  for (size_t i = 0; i < p->convs.size(); ++i) {
    ConvL& c = p->convs[i];
    const size_t n = (i == 0) ? (size_t)64 * 8 * 32 : (size_t)c.cout * c.cin * c.k * c.k;
    bool scratch = i == 0;
    if (!scratch && c.k != 1) {   // (the descriptor Exec::conv_desc / quad_desc / region_desc hands to wgrad())
      qt_conv_desc d;
      memset(&d, 0, sizeof(d));
      const int S = c.regions == 16 ? 4 : (c.regions == 4 ? 2 : 1);
      d.dtype = p->d.dtype; d.mode = QT_CONV_FWD; d.batch = (int)B;
      d.kh = d.kw = c.k; d.stride = c.stride; d.pad = c.pad;
      d.in_h = d.in_w = c.hin; d.out_h = d.out_w = c.hout; d.k_per_tap = c.cin; d.n_out = c.cout;
      d.quad = S == 1 ? 0 : S;
      d.src_pix_stride = c.cin; d.src_row_stride = S * c.hin * c.cin;
      d.src_img_stride = (long long)S * c.hin * S * c.hin * c.cin;
      scratch = qt_conv2d_wgrad_workspace_bytes(&d) == 0;
    }
    c.dw = scratch ? ws.take(n * 4) : 0;
  }
  p->dw_end = ws.off;
  // one partial filter per range of positions (at most 256 workgroups x 64 x 9 x 64 f32, whatever the
  // batch): the streaming weight-gradient launches run one after another on the side stream and share it
  p->wgrad_part_bytes = (size_t)256 * 64 * 9 * 64 * 4;
  p->wgrad_part = ws.take(p->wgrad_part_bytes);
  {
    const int mb = B < 256 ? (int)B : 256;
    const size_t f = qt_linear_workspace_bytes(mb, p->cls0.out, p->cls0.in), g = qt_linear_workspace_bytes(mb, p->cls0.in, p->cls0.out);
    p->lin_ws_bytes = f > g ? f : g;
    p->lin_ws = ws.take(p->lin_ws_bytes > 0 ? p->lin_ws_bytes : 16);
  }
  for (LinL* l : {&p->cls0}) {
    l->w_fwd = ws.take((size_t)l->in * l->out * es);
    l->w_dgrad = ws.take((size_t)l->in * l->out * es);
  }
  if (p->has_image) {
    p->xpad = ws.take(B * QT_STEM_PAD_H * QT_STEM_PAD_W * 4 * es);
    p->p0 = ws.take(B * 56 * 56 * 64 * es);
    p->g_p0 = ws.take(B * 56 * 56 * 64 * es);
    p->argmax = ws.take(B * 56 * 56 * 64);
    p->ymax = ws.take(B * 56 * 56 * 64 * es);  // raw conv1 output at the pooling argmax (bn1 backward sums)
    for (size_t i = 0; i < p->convs.size(); ++i) {
      ConvL& c = p->convs[i];
      const size_t n = B * c.regions * c.hout * c.hout * c.cout;
      c.y = ws.take(n * es);
      c.gy = ws.take(n * es);
    }
    for (auto& blk : p->blocks) {
      const ConvL& c2 = p->convs[blk.conv2];
      const size_t n = B * c2.hout * c2.hout * c2.cout;
      blk.a1 = ws.take(n * es);
      blk.out = ws.take(n * es);
      blk.gout = ws.take(n * es);
      blk.a1_bits = ws.take(n / 8);
      blk.out_bits = ws.take(n / 8);
      const ConvL& c1 = p->convs[blk.conv1];
      blk.gtmp = blk.ds >= 0 ? ws.take(B * c1.hin * c1.hin * c1.cin * es) : 0;
    }
    p->gbase_tmp = ws.take(B * 28 * 28 * 128 * es);  // (>= layer3's 14x14x256 map)
    if (p->attention) {
      p->gbase_tmp2 = ws.take(B * 28 * 28 * 128 * es);
      p->vsub = ws.take(B * 16 * 64 * 4);
      p->dvsub = ws.take(B * 16 * 64 * 4);
      p->att_act = ws.take(B * 16 * 32 * 4);
      p->att_dpre = ws.take(B * 16 * 32 * 4);
      p->att_alpha = ws.take(B * 16 * 4);
      p->att_ds = ws.take(B * 16 * 4);
    }
  }
  if (!p->standard) {
    p->q = p->convs[p->quad_conv].y;
    p->dq = p->convs[p->quad_conv].gy;
    p->h1 = ws.take(B * p->mlp0.out * 4);
    p->dh1 = ws.take(B * p->mlp0.out * 4);
  }
  if (p->lstm) {
    const size_t H = (size_t)p->lstm_h;
    p->h1 = ws.take(B * p->mlp0.out * 4);
    p->dh1 = ws.take(B * p->mlp0.out * 4);
    for (int l = 0; l < 2; ++l) {
      qt_plan::LstmL& L = p->lstm_l[l];
      L.whh_t = ws.take(H * 4 * H * 4);
      L.wih_t = ws.take((size_t)L.in * 4 * H * 4);
      L.xproj = ws.take(B * 4 * H * 4);
      L.gates = ws.take(B * 4 * H * 4);
      L.dgates = ws.take(B * 4 * H * 4);
      L.cell = ws.take(B * H * 4);
      L.hprev = ws.take(B * H * 4);
      L.hout = ws.take(B * H * 4);
    }
    p->lstm_x0 = ws.take(B * p->fused_ld * 4);
    p->lstm_x1 = ws.take(B * H * 4);
    p->lstm_dx1 = ws.take(B * H * 4);
    p->lstm_dlast = ws.take(B * H * 4);
    p->lstm_hid = ws.take(B * 128 * 4);
    p->lstm_dhid = ws.take(B * 128 * 4);
    p->lstm_dz = ws.take(B * 128 * 4);
  }
  p->fused = ws.take(B * p->fused_ld * es);
  p->dfused = ws.take(B * p->fused_ld * es);
  p->hidden = ws.take(B * p->hidden_dim * es);
  p->dhidden = ws.take(B * p->hidden_dim * es);
  p->ws_bytes = ws.off;
}

// ------------------------------------------------------------------------------
struct Exec {
  qt_plan* p;
  unsigned char* ws;
  void* const* T;  // tensor pointers
  void* stream;
  int B;
  int dt;
  int status = QT_OK;
  size_t stats_off = (size_t)-1;  // BatchNorm partial-sum scratch in use ((size_t)-1: p->stats)

  // ---- plan-owned side stream: independent branches run next to the caller's stream ----
  void* wstream = nullptr;  // == stream when the side stream is off
  bool forked = false;
  void hip(hipError_t e, const char* what) {
    if (e != hipSuccess && status == QT_OK) {
      qt_set_error("%s: %s", what, hipGetErrorString(e));
      status = QT_ERR_LAUNCH;
    }
  }
  void setup_side() {
    wstream = stream;
    if (!p->use_side) return;
    if (!p->side) {
      hip(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking), "hipStreamCreate");
      hip(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming), "hipEventCreate");
      hip(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming), "hipEventCreate");
    }
    if (ok()) wstream = p->side;
  }
  // everything enqueued on `stream` so far happens before later side-stream work
  void fork() {
    if (!wstream || wstream == stream || !ok()) return;
    hip(hipEventRecord(p->ev_fork, static_cast<hipStream_t>(stream)), "hipEventRecord");
    hip(hipStreamWaitEvent(p->side, p->ev_fork, 0), "hipStreamWaitEvent");
    forked = true;
  }
  // side-stream work happens before anything enqueued on `stream` afterwards
  void join() {
    if (wstream == stream || !wstream || !forked || !ok()) return;
    hip(hipEventRecord(p->ev_join, p->side), "hipEventRecord");
    hip(hipStreamWaitEvent(static_cast<hipStream_t>(stream), p->ev_join, 0), "hipStreamWaitEvent");
    forked = false;
  }

  // run `body` with every launch going to the side stream (after fork()), with its own stats scratch
  template <typename F> void on_side(size_t scratch, F&& body) {
    void* main_stream = stream;
    const size_t main_stats = stats_off;
    stream = wstream ? wstream : stream;
    stats_off = scratch;
    body();
    stream = main_stream;
    stats_off = main_stats;
  }

  template <typename X = void> X* at(size_t off) const { return reinterpret_cast<X*>(ws + off); }
  float* tf(int idx) const { return idx < 0 ? nullptr : static_cast<float*>(T[idx]); }
  bool ok() const { return status == QT_OK; }
  void run(int st) {
    if (status == QT_OK && st != QT_OK) status = st;
  }

  qt_conv_desc conv_desc(const ConvL& c, int mode) const {
    qt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt;
    d.mode = mode;
    d.batch = B;
    d.kh = d.kw = c.k;
    d.stride = c.stride;
    d.pad = c.pad;
    if (mode == QT_CONV_FWD) {
      d.in_h = d.in_w = c.hin; d.out_h = d.out_w = c.hout;
      d.k_per_tap = c.cin; d.n_out = c.cout;
      d.src_pix_stride = c.cin; d.src_row_stride = c.hin * c.cin; d.src_img_stride = (long long)c.hin * c.hin * c.cin;
    } else {
      d.in_h = d.in_w = c.hout; d.out_h = d.out_w = c.hin;
      d.k_per_tap = c.cout; d.n_out = c.cin;
      d.src_pix_stride = c.cout; d.src_row_stride = c.hout * c.cout; d.src_img_stride = (long long)c.hout * c.hout * c.cout;
    }
    return d;
  }
  // bf16: a 32-element tap is half a K-step, so the forward runs 8 row taps (the
  // 8th has zero weights); the weight gradient always uses the 7 real taps.
  int stem_taps() const { return dt == QT_BF16 ? 8 : 7; }
  qt_conv_desc stem_desc(bool for_wgrad = false) const {
    qt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt; d.mode = QT_CONV_FWD; d.batch = B;
    d.in_h = QT_STEM_PAD_H; d.in_w = QT_STEM_PAD_W; d.out_h = d.out_w = 112;
    d.k_per_tap = 32; d.n_out = 64; d.kh = for_wgrad ? 7 : stem_taps(); d.kw = 1; d.stride = 2; d.pad = 0;
    d.src_pix_stride = 4; d.src_row_stride = QT_STEM_PAD_W * 4;
    d.src_img_stride = (long long)QT_STEM_PAD_H * QT_STEM_PAD_W * 4;
    return d;
  }
  qt_conv_desc quad_desc(int mode) const {
    const ConvL& c = p->convs[p->quad_conv];
    qt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt; d.mode = mode; d.batch = B; d.kh = d.kw = 3; d.stride = 1; d.pad = 1; d.quad = 1;
    if (mode == QT_CONV_FWD) {
      d.in_h = d.in_w = 7; d.out_h = d.out_w = 7; d.k_per_tap = c.cin; d.n_out = c.cout;
      d.src_pix_stride = c.cin; d.src_row_stride = 14 * c.cin; d.src_img_stride = 14ll * 14 * c.cin;
    } else {
      d.in_h = d.in_w = 7; d.out_h = d.out_w = 14; d.k_per_tap = c.cout; d.n_out = c.cin;
      d.src_pix_stride = c.cout; d.src_row_stride = 7 * c.cout; d.src_img_stride = 49ll * c.cout;
    }
    return d;
  }
  // a head conv applied to each of the S x S regions of a shared map, zero halo at the seams
  // (AttentionHierarchicalCNN, models.py:62-78): FWD reads the un-split map, DGRAD scatters back onto it
  qt_conv_desc region_desc(const ConvL& c, int S, int mode) const {
    qt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt; d.mode = mode; d.batch = B; d.kh = d.kw = c.k; d.stride = 1; d.pad = c.pad; d.quad = S;
    d.in_h = d.in_w = c.hin;
    if (mode == QT_CONV_FWD) {
      d.out_h = d.out_w = c.hin; d.k_per_tap = c.cin; d.n_out = c.cout;
      d.src_pix_stride = c.cin; d.src_row_stride = S * c.hin * c.cin;
      d.src_img_stride = (long long)S * c.hin * S * c.hin * c.cin;
    } else {
      d.out_h = d.out_w = S * c.hin; d.k_per_tap = c.cout; d.n_out = c.cin;
      d.src_pix_stride = c.cout; d.src_row_stride = c.hin * c.cout; d.src_img_stride = (long long)c.hin * c.hin * c.cout;
    }
    return d;
  }
  // y [rows][out] = x [rows][in] W^T (W [out][in]) as a 1x1 convolution on 1x1 images, f32 MFMA whatever the plan's dtype
  qt_conv_desc dense_f32(int rows, int in, int out) const {
    qt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = QT_F32; d.mode = QT_CONV_FWD; d.batch = rows; d.in_h = d.in_w = d.out_h = d.out_w = 1;
    d.kh = d.kw = 1; d.stride = 1; d.pad = 0;
    d.k_per_tap = in; d.n_out = out; d.src_pix_stride = in; d.src_row_stride = in; d.src_img_stride = in;
    return d;
  }
  qt_conv_desc linear_desc(int in, int out, int mode) const {
    qt_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt; d.mode = mode; d.batch = B; d.in_h = d.in_w = d.out_h = d.out_w = 1;
    d.kh = d.kw = 1; d.stride = 1; d.pad = 0;
    if (mode == QT_CONV_FWD) {
      d.k_per_tap = in; d.n_out = out; d.src_pix_stride = in; d.src_row_stride = in; d.src_img_stride = in;
    } else {
      d.k_per_tap = out; d.n_out = in; d.src_pix_stride = out; d.src_row_stride = out; d.src_img_stride = out;
    }
    return d;
  }

  struct BnLink {  // a BatchNorm that consumes the gradient a dgrad launch writes
    const void* y = nullptr;
    const float* mean = nullptr;
    const float* invstd = nullptr;
    float* partial = nullptr;
  };
  void igemm(const qt_conv_desc& d, const void* src, const void* w, void* dst, const float* scale, const float* shift,
             const void* res, const void* mask, float* stats, int relu, int kind = -1, const BnLink* links = nullptr,
             int nlinks = 0, const unsigned char* mask_bits = nullptr, const void* extra_src = nullptr) {
    if (!ok()) return;
    qt_conv_desc dd = d;
    dd.relu = relu;
    qt_conv_io io = {src, w, dst, scale, shift, res, mask, stats};
    io.relu_mask_bits = mask_bits;
    io.extra_src = extra_src;
    for (int k = 0; k < nlinks && k < 2; ++k) {
      io.bwd_bn[k].y = links[k].y; io.bwd_bn[k].mean = links[k].mean;
      io.bwd_bn[k].invstd = links[k].invstd; io.bwd_bn[k].partial = links[k].partial;
    }
    const int slot = begin_timed(conv_flops(d), kind >= 0 ? kind : (d.mode == QT_CONV_FWD ? 0 : 1), nullptr, conv_bytes(dd, io));
    run(qt_conv2d_igemm(&dd, &io, stream));
    end_timed(slot);
  }

  // classifier.0 and its backward-input product: split-K kernel for small batches in bf16 (every CU
  // streams a slice of the 29 MB weight matrix once); other shapes and the f32 build run as an
  // implicit GEMM on 1x1 images.  d describes the product like a 1x1 convolution.
  void linear(const qt_conv_desc& d, const void* x, const void* w, void* y, const float* bias, int relu) {
    if (!ok()) return;
    if (dt == QT_BF16 && B <= 256) {
      const int slot = begin_timed(conv_flops(d), d.mode == QT_CONV_FWD ? 0 : 1, nullptr,
                                   2.0 * ((double)B * d.k_per_tap + (double)d.k_per_tap * d.n_out + (double)B * d.n_out));
      const int st = qt_linear_bf16(x, w, bias, relu, y, B, d.n_out, d.k_per_tap, at(p->lin_ws), p->lin_ws_bytes, stream);
      end_timed(slot);
      if (st == QT_OK) return;
      if (st != QT_ERR_UNSUPPORTED) {
        run(st);
        return;
      }
    }
    igemm(d, x, w, y, nullptr, bias, nullptr, nullptr, nullptr, relu);
  }

  // algorithmic FLOPs (2*MAC of the convolution as the reference computes it: the
  // 7x7x3 stem counts 147 taps, a stride-2 dgrad counts the forward conv's MACs)
  double conv_flops(const qt_conv_desc& d) const {
    const double imgs = (double)d.batch * qt_quad_regions(d.quad);
    const double fwd_pixels = d.mode == QT_CONV_FWD ? (double)d.out_h * d.out_w : (double)d.in_h * d.in_w;
    const bool stem = d.k_per_tap == 32 && d.kw == 1 && d.stride == 2 && d.n_out == 64;
    // (merged parity classes: 4 C outputs x 4 tap slots stand for the C outputs x 9 taps of the stride-2 conv)
    // (... plus, with the fifth slot, the C x C outputs x 1 tap of the 1x1 downsample)
    const double k = stem ? 147.0 : (d.dst_merge ? (d.dst_merge_extra ? 10.0 : 9.0) / 4.0 : (double)d.kh * d.kw) * d.k_per_tap;
    return 2.0 * imgs * fwd_pixels * k * d.n_out;
  }
  // algorithmic HBM bytes of a conv launch: source map, weights, destination and every per-pixel epilogue operand, each once
  double conv_bytes(const qt_conv_desc& d, const qt_conv_io& io) const {
    const double es = d.dtype == QT_F32 ? 4.0 : 2.0;
    const double imgs = (double)d.batch * (d.mode == QT_CONV_FWD ? 1.0 : 1.0);
    const bool stem = d.k_per_tap == 32 && d.kw == 1 && d.stride == 2 && d.n_out == 64;
    const double regions = qt_quad_regions(d.quad);
    const double src = stem ? imgs * d.in_h * d.in_w * 4.0 * es
                            : imgs * regions * (double)d.in_h * d.in_w * d.k_per_tap * es;
    const double rows = d.dst_merge ? imgs * (double)d.dst_h * d.dst_w : imgs * (d.mode == QT_CONV_FWD ? regions : 1.0) * d.out_h * d.out_w;
    const double ncol = d.dst_merge ? d.dst_merge : d.n_out;
    double per_pixel = 1.0;   // dst
    if (io.residual) per_pixel += d.dst_merge_res0 ? 0.25 : 1.0;
    if (io.relu_mask) per_pixel += 1.0;
    if (io.relu_mask_bits) per_pixel += 1.0 / (8.0 * es);
    if (io.bwd_bn[0].y) per_pixel += 1.0;
    if (io.bwd_bn[1].y) per_pixel += 1.0;
    const double wgt = ((double)d.kh * d.kw + (d.dst_merge_extra ? 1.0 : 0.0)) * d.k_per_tap * d.n_out * es;
    return src * (d.dst_merge_extra ? 2.0 : 1.0) + wgt + rows * ncol * es * per_pixel;
  }
  int begin_timed(double flops, int kind, void* on = nullptr, double bytes = 0.0) {
    if (!p->profiling) return -1;
    qt_plan::Timed t;
    if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) return -1;
    t.flops = flops;
    t.bytes = bytes;
    t.kind = kind;
    (void)hipEventRecord(t.a, static_cast<hipStream_t>(on ? on : stream));
    p->timed.push_back(t);
    return (int)p->timed.size() - 1;
  }
  void end_timed(int slot, void* on = nullptr) {
    if (slot >= 0) (void)hipEventRecord(p->timed[slot].b, static_cast<hipStream_t>(on ? on : stream));
  }

  long long rows_of(const ConvL& c) const { return (long long)B * c.hout * c.hout; }

  // conv (+ train-mode statistics -> scale/shift of its BatchNorm)
  // training: 1 = batch statistics, 2 = eval statistics but the raw conv output is kept for backward, 0 = nothing here
  void conv_bn_stats(const ConvL& c, const qt_conv_desc& d, const void* src, int training) {
    BnL& bn = p->bns[c.bn];
    if (training == 2) {
      igemm(d, src, at(c.w_fwd), at(c.y), nullptr, nullptr, nullptr, nullptr, nullptr, 0);
      return;
    }
    if (training == 1) {
      float* part = at<float>(stats_off == (size_t)-1 ? p->stats : stats_off);
      igemm(d, src, at(c.w_fwd), at(c.y), nullptr, nullptr, nullptr, nullptr, part, 0);
      if (!ok()) return;
      const int rows = qt_conv2d_stats_rows(&d);
      run(qt_bn_finalize(part, rows, bn.C, rows_of(c), tf(bn.gamma), tf(bn.beta), tf(bn.rmean),
                         tf(bn.rvar), static_cast<long long*>(T[bn.nbt]), p->d.bn_momentum, p->d.bn_eps,
                         at<float>(bn.mean), at<float>(bn.invstd), at<float>(bn.scale), at<float>(bn.shift), stream));
    }
    // eval: scale / shift of every BatchNorm were set by eval_affines() at the start of the forward
  }
  // conv1 (3x3 / stride 2) and the downsample (1x1 / stride 2) of a transition block from ONE staged input patch
  // (csrc/conv_s2.hip); false: the problem is not covered (small batches, QTCNN_S2_CONV=0) -> the two generic launches.
  // training as conv_bn_stats: 0 = eval epilogues (conv1 -> bn1 -> ReLU into blk.a1, downsample -> its BatchNorm into
  // cd.y), 1 = raw outputs + batch statistics of both, 2 = raw outputs
  bool transition_pair(const Block& blk, const void* x, int training) {
    if (!ok() || blk.ds < 0) return false;
    const ConvL& c1 = p->convs[blk.conv1];
    const ConvL& cd = p->convs[blk.ds];
    if (c1.k != 3 || c1.stride != 2 || c1.pad != 1 || cd.k != 1 || cd.stride != 2 || cd.pad != 0) return false;
    qt_conv_s2_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dt; d.batch = B; d.in_h = d.in_w = c1.hin; d.c_in = c1.cin; d.c_out = c1.cout;
    if (!qt_conv_s2_pair_supported(&d)) return false;
    BnL& b1 = p->bns[c1.bn];
    BnL& bd = p->bns[cd.bn];
    qt_conv_s2_io io;
    memset(&io, 0, sizeof(io));
    io.src = x; io.w_conv = at(c1.w_fwd); io.w_down = at(cd.w_fwd);
    io.y_conv = training == 0 ? at(blk.a1) : at(c1.y);
    io.y_down = at(cd.y);
    if (training == 0) {
      d.relu_conv = 1;
      io.scale_conv = at<float>(b1.scale); io.shift_conv = at<float>(b1.shift);
      io.scale_down = at<float>(bd.scale); io.shift_down = at<float>(bd.shift);
    } else if (training == 1) {
      io.stats_conv = at<float>(p->stats);
      io.stats_down = at<float>(p->stats_ds);
    }
    const double es = dt == QT_F32 ? 4.0 : 2.0;
    const int slot = begin_timed(conv_flops(conv_desc(c1, QT_CONV_FWD)) + conv_flops(conv_desc(cd, QT_CONV_FWD)), 0, nullptr,
                                 es * ((double)B * c1.hin * c1.hin * c1.cin + 2.0 * B * c1.hout * c1.hout * c1.cout +
                                       10.0 * c1.cin * c1.cout));
    run(qt_conv_s2_pair(&d, &io, stream));
    end_timed(slot);
    if (training == 1 && ok()) {
      const int rows = qt_conv_s2_pair_stats_rows(&d);
      run(qt_bn_finalize(at<float>(p->stats), rows, b1.C, rows_of(c1), tf(b1.gamma), tf(b1.beta), tf(b1.rmean), tf(b1.rvar),
                         static_cast<long long*>(T[b1.nbt]), p->d.bn_momentum, p->d.bn_eps, at<float>(b1.mean),
                         at<float>(b1.invstd), at<float>(b1.scale), at<float>(b1.shift), stream));
      run(qt_bn_finalize(at<float>(p->stats_ds), rows, bd.C, rows_of(cd), tf(bd.gamma), tf(bd.beta), tf(bd.rmean), tf(bd.rvar),
                         static_cast<long long*>(T[bd.nbt]), p->d.bn_momentum, p->d.bn_eps, at<float>(bd.mean),
                         at<float>(bd.invstd), at<float>(bd.scale), at<float>(bd.shift), stream));
    }
    return true;
  }
  // eval mode: running statistics -> scale / shift of all BatchNorms in one launch
  void eval_affines(bool for_backward) {
    std::vector<qt_bn_eval_item> items;
    for (const BnL& bn : p->bns) {
      qt_bn_eval_item q;
      q.gamma = tf(bn.gamma); q.beta = tf(bn.beta); q.running_mean = tf(bn.rmean); q.running_var = tf(bn.rvar);
      q.scale = at<float>(bn.scale); q.shift = at<float>(bn.shift); q.C = bn.C;
      q.mean = for_backward ? at<float>(bn.mean) : nullptr;
      q.invstd = for_backward ? at<float>(bn.invstd) : nullptr;
      items.push_back(q);
    }
    for (size_t j = 0; j < items.size() && ok(); j += 32)
      run(qt_bn_eval_affine_batched(items.data() + j, (int)std::min<size_t>(32, items.size() - j), p->d.bn_eps, stream));
  }
};

// qt_pack_item.stride2_dgrad of a conv's data-gradient operand, and where it lives: a downsample whose data gradient rides
// in its block's merged launch writes the fifth tap slot of conv1's operand instead of an operand of its own
int dgrad_layout(const ConvL& c) {
  if (c.stride != 2) return 0;
  if (c.k == 1) return c.slot_conv >= 0 ? 4 : 1;
  return c.merged5 ? 3 : (c.merged_dgrad ? 2 : 1);
}
size_t dgrad_operand(const qt_plan* p, const ConvL& c) {
  return (c.k == 1 && c.slot_conv >= 0) ? p->convs[c.slot_conv].w_dgrad : c.w_dgrad;
}

unsigned long long weight_sig(const qt_plan* p, void* const* T) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](const void* ptr) { h = (h ^ (unsigned long long)(uintptr_t)ptr) * 1099511628211ull; };
  for (const ConvL& c : p->convs) mix(T[c.w]);
  mix(T[p->cls0.w]);
  return h;
}

int pack_weights(qt_plan* p, void* workspace, void* const* T, int for_backward, void* stream) {
  Exec e{p, static_cast<unsigned char*>(workspace), T, stream, p->d.batch, p->d.dtype};
  // one launch for every conv / linear operand (plus the 9 K-element stem filter)
  std::vector<qt_pack_item> items;
  auto add = [&](const float* w, void* fwd, void* dgrad, int O, int I, int k, int s2) {
    qt_pack_item q;
    q.w_oihw = w; q.w_fwd = fwd; q.w_dgrad = dgrad; q.O = O; q.I = I; q.k = k; q.stride2_dgrad = s2;
    items.push_back(q);
  };
  if (p->has_image) {
    for (size_t i = 0; i < p->convs.size(); ++i) {
      const ConvL& c = p->convs[i];
      if (i == 0)
        e.run(qt_pack_stem_weight(e.dt, e.tf(c.w), e.at(c.w_fwd), e.stem_taps(), stream));
      else
        add(e.tf(c.w), e.at(c.w_fwd), for_backward ? e.at(dgrad_operand(p, c)) : nullptr, c.cout, c.cin, c.k, dgrad_layout(c));
    }
  }
  if (!p->lstm)  // (CnnLstm's classifier is a thin f32 product: no packed copy)
    add(e.tf(p->cls0.w), e.at(p->cls0.w_fwd), for_backward ? e.at(p->cls0.w_dgrad) : nullptr, p->cls0.out, p->cls0.in, 1,
        false);
  for (size_t j = 0; j < items.size() && e.ok(); j += 32)
    e.run(qt_pack_weights_batched(e.dt, items.data() + j, (int)std::min<size_t>(32, items.size() - j), stream));
  p->packed_fwd = e.ok();
  p->packed_bwd = e.ok() && for_backward != 0;
  p->packed_sig = weight_sig(p, T);
  return e.status;
}

// Optimizer step + operand re-packing in one pass over the parameters: the conv /
// linear weights that have packed copies are updated INSIDE the packing kernel, everything else by the
// plain multi-tensor kernel; tensors without a gradient are only re-packed.
int adam_step(qt_plan* p, void* workspace, void* const* T, float* const* G, float* const* M1, float* const* M2,
              const qt_adam_desc* adam, int for_backward, void* stream) {
  Exec e{p, static_cast<unsigned char*>(workspace), T, stream, p->d.batch, p->d.dtype};
  std::vector<bool> fused(p->tensors.size(), false);
  std::vector<qt_pack_item> upd_items, pack_items;
  std::vector<qt_adam_item> upd_state, plain;
  auto numel_of = [&](int idx) {
    long long n = 1;
    for (int d = 0; d < p->tensors[idx].ndim; ++d) n *= p->tensors[idx].shape[d];
    return n;
  };
  auto add = [&](int widx, void* fwd, void* dgrad, int O, int I, int k, int s2) -> int {
    qt_pack_item q;
    q.w_oihw = e.tf(widx); q.w_fwd = fwd; q.w_dgrad = dgrad; q.O = O; q.I = I; q.k = k; q.stride2_dgrad = s2;
    if (G[widx]) {
      QT_CHECK_ARG(M1[widx] && M2[widx], "qt_plan_adam_step: %s has a gradient but no optimizer state",
                   p->tensors[widx].name.c_str());
      qt_adam_item u;
      u.param = e.tf(widx); u.grad = G[widx]; u.exp_avg = M1[widx]; u.exp_avg_sq = M2[widx]; u.numel = numel_of(widx);
      upd_items.push_back(q);
      upd_state.push_back(u);
    } else {
      pack_items.push_back(q);
    }
    fused[widx] = true;
    return (int)QT_OK;
  };
  if (p->has_image)
    for (size_t i = 1; i < p->convs.size(); ++i) {
      const ConvL& c = p->convs[i];
      if (int st = add(c.w, e.at(c.w_fwd), for_backward ? e.at(dgrad_operand(p, c)) : nullptr, c.cout, c.cin, c.k, dgrad_layout(c))) return st;
    }
  if (!p->lstm)
    if (int st = add(p->cls0.w, e.at(p->cls0.w_fwd), for_backward ? e.at(p->cls0.w_dgrad) : nullptr, p->cls0.out, p->cls0.in, 1, false))
      return st;
  for (size_t i = 0; i < p->tensors.size(); ++i) {
    if (p->tensors[i].kind != 0 || fused[i] || !G[i]) continue;
    QT_CHECK_ARG(T[i] && M1[i] && M2[i], "qt_plan_adam_step: %s has a gradient but no parameter / optimizer state",
                 p->tensors[i].name.c_str());
    qt_adam_item u;
    u.param = e.tf((int)i); u.grad = G[i]; u.exp_avg = M1[i]; u.exp_avg_sq = M2[i]; u.numel = numel_of((int)i);
    plain.push_back(u);
  }
  if (!plain.empty()) e.run(qt_adam_multi(plain.data(), (int)plain.size(), adam, stream));
  if (p->has_image)  // conv1's filter was updated by the plain kernel above
    e.run(qt_pack_stem_weight(e.dt, e.tf(p->convs[0].w), e.at(p->convs[0].w_fwd), e.stem_taps(), stream));
  for (size_t j = 0; j < upd_items.size() && e.ok(); j += 32) {
    const int cnt = (int)std::min<size_t>(32, upd_items.size() - j);
    e.run(qt_adam_pack_weights_batched(e.dt, upd_items.data() + j, upd_state.data() + j, adam, cnt, stream));
  }
  // weights without a gradient did not change: their copies are re-packed only if they are not known to be current
  const unsigned long long sig = weight_sig(p, T);
  const bool frozen_current = p->packed_fwd && (!for_backward || p->packed_bwd) && p->packed_sig == sig;
  if (!frozen_current)
    for (size_t j = 0; j < pack_items.size() && e.ok(); j += 32)
      e.run(qt_pack_weights_batched(e.dt, pack_items.data() + j, (int)std::min<size_t>(32, pack_items.size() - j), stream));
  p->packed_fwd = e.ok();
  p->packed_bwd = e.ok() && (for_backward != 0);
  p->packed_sig = sig;
  return e.status;
}

int forward(qt_plan* p, void* workspace, void* const* T, const float* image, const float* numerical, float* logits,
            int batch, int training, unsigned long long seed, void* stream) {
  Exec e{p, static_cast<unsigned char*>(workspace), T, stream, batch, p->d.dtype};
  e.setup_side();
  const int dt = e.dt;
  const bool tr = training == 1;   // batch statistics, running-stat update, dropout
  const bool unf = training != 0;  // unfused: raw conv outputs, pooling argmax ... are kept for qt_plan_backward
  hipStream_t hs = static_cast<hipStream_t>(stream);
  // The quadrant head (needs layer3's output) and the numerical MLP (needs nothing) are independent
  // of layer4: they run on the side stream and are joined before the classifier.
  auto quad_branch = [&]() {
    const ConvL& cq = p->convs[p->quad_conv];
    e.igemm(e.quad_desc(QT_CONV_FWD), e.at(p->blocks[5].out), e.at(cq.w_fwd), e.at(p->q), nullptr, e.tf(cq.bias),
            nullptr, nullptr, nullptr, 1);
    e.run(qt_quad_pool(dt, e.at(p->q), e.at(p->fused), batch, p->fused_ld, 512, e.stream));
  };
  auto mlp_branch = [&]() {
    qt_gemm_small_desc g;
    memset(&g, 0, sizeof(g));
    g.M = batch; g.N = p->mlp0.out; g.K = p->mlp0.in;
    g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
    g.a_row_stride = p->mlp0.in; g.a_k_stride = 1; g.b_row_stride = p->mlp0.in; g.b_k_stride = 1;
    g.c_row_stride = p->mlp0.out; g.relu = 1;
    e.run(qt_gemm_small(&g, numerical, e.tf(p->mlp0.w), e.tf(p->mlp0.b), e.at(p->h1), e.stream));
    if (tr && p->d.dropout_p > 0.f && !p->lstm)  // (CnnLstm's MLP has no dropout, cnn+lstm/models.py:32-36)
      e.run(qt_dropout(QT_F32, e.at(p->h1), batch, p->mlp0.out, p->mlp0.out, seed, p->d.dropout_p, e.stream));
    g.N = p->mlp1.out; g.K = p->mlp1.in;
    g.a_row_stride = p->mlp1.in; g.b_row_stride = p->mlp1.in; g.c_dtype = dt; g.c_row_stride = p->fused_ld; g.relu = 0;
    e.run(qt_gemm_small(&g, e.at(p->h1), e.tf(p->mlp1.w), e.tf(p->mlp1.b),
                        e.at<unsigned char>(p->fused) + (size_t)p->mlp_col0 * p->esz, e.stream));
  };
  // AttentionHierarchicalCNN (models.py:57-97): quadrant and sub-quadrant heads on layer2's map, the attention
  // gate over the 16 sub-quadrant vectors and the one-layer numerical MLP; all of it only needs layer2's output.
  auto attn_branch = [&]() {
    const ConvL& cq = p->convs[p->quad_conv];
    const ConvL& cs = p->convs[p->sub_conv];
    const void* base = e.at(p->blocks[3].out);
    e.igemm(e.region_desc(cq, 2, QT_CONV_FWD), base, e.at(cq.w_fwd), e.at(cq.y), nullptr, e.tf(cq.bias), nullptr, nullptr,
            nullptr, 1);
    e.run(qt_region_avgpool(dt, e.at(cq.y), e.at(p->fused), dt, batch, 2, 196, 128, p->fused_ld, 512, e.stream));
    e.igemm(e.region_desc(cs, 4, QT_CONV_FWD), base, e.at(cs.w_fwd), e.at(cs.y), nullptr, e.tf(cs.bias), nullptr, nullptr,
            nullptr, 1);
    e.run(qt_region_avgpool(dt, e.at(cs.y), e.at(p->vsub), QT_F32, batch, 4, 49, 64, 16 * 64, 0, e.stream));
    e.run(qt_attention_gate(dt, e.at<float>(p->vsub), e.tf(p->att0.w), e.tf(p->att0.b), e.tf(p->att2.w), e.tf(p->att2.b),
                            e.at<float>(p->att_act), e.at<float>(p->att_alpha), e.at(p->fused), batch, p->fused_ld,
                            512 + 4 * 128, e.stream));
    qt_gemm_small_desc g;
    memset(&g, 0, sizeof(g));
    g.M = batch; g.N = p->mlp0.out; g.K = p->mlp0.in;
    g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = dt;
    g.a_row_stride = p->mlp0.in; g.a_k_stride = 1; g.b_row_stride = p->mlp0.in; g.b_k_stride = 1;
    g.c_row_stride = p->fused_ld; g.relu = 1;
    unsigned char* z = e.at<unsigned char>(p->fused) + (size_t)p->mlp_col0 * p->esz;
    e.run(qt_gemm_small(&g, numerical, e.tf(p->mlp0.w), e.tf(p->mlp0.b), z, e.stream));
    if (tr && p->d.dropout_p > 0.f) e.run(qt_dropout(dt, z, batch, p->mlp0.out, p->fused_ld, seed, p->d.dropout_p, e.stream));
  };
  // The numerical MLP needs nothing of the image branch: its thin kernels go to the side stream FIRST, next to the
  // stem's byte-moving kernels.  (Until round 3 they were forked behind layer3 and their workgroups trickled onto CUs
  // that layer4's whole-CU convolutions released: 58 + 90 us on the side queue, layer4's convs 67 -> 88-100 us.)
  const bool mlp_early = p->has_image && p->has_numerical && !p->attention && !p->lstm && !p->standard;
  if (mlp_early) {
    e.fork();
    e.on_side(p->stats_ds, [&] { mlp_branch(); });
  }
  if (p->has_image) {
    if (!tr) e.eval_affines(unf);
    // ---- stem: pack -> conv7x7/2 (7 row taps x 32) -> BN -> ReLU -> maxpool ----
    const ConvL& c0 = p->convs[0];
    const BnL& bn0 = p->bns[c0.bn];
    const qt_conv_desc sd = e.stem_desc();
    // eval (bf16): the f32 NCHW image -> packing, conv1, folded bn1, ReLU and the max pool in ONE kernel; neither the packed
    // copy of the input nor the conv1 map is materialised
    int fused = QT_ERR_UNSUPPORTED;
    if (!unf) {
      const int slot = e.begin_timed(e.conv_flops(sd), 0, nullptr,
                                     (double)batch * (3.0 * 224 * 224 * 4 + p->esz * 56.0 * 56 * 64));
      fused = qt_stem_conv_pool_nchw(dt, image, e.at(c0.w_fwd), e.stem_taps(), e.at<float>(bn0.scale),
                                     e.at<float>(bn0.shift), e.at(p->p0), batch, stream);
      e.end_timed(slot);
      if (fused != QT_ERR_UNSUPPORTED) e.run(fused);
    }
    if (fused == QT_ERR_UNSUPPORTED) e.run(qt_pack_stem_input(dt, image, e.at(p->xpad), batch, stream));
    if (fused == QT_ERR_UNSUPPORTED) e.conv_bn_stats(c0, sd, e.at(p->xpad), training);
    if (unf) {
      e.run(qt_stem_pool(dt, e.at(c0.y), e.at<float>(bn0.scale), e.at<float>(bn0.shift), e.at(p->p0),
                         e.at<unsigned char>(p->argmax), e.at(p->ymax), batch, stream));
    } else if (fused == QT_ERR_UNSUPPORTED) {
      // conv1 + folded bn1 + ReLU + max pool on the packed input (f32 build: two kernels)
      const int slot = e.begin_timed(e.conv_flops(sd), 0, nullptr,
                                     (double)batch * p->esz * ((double)QT_STEM_PAD_H * QT_STEM_PAD_W * 4 + 56.0 * 56 * 64));
      fused = qt_stem_conv_pool(dt, e.at(p->xpad), e.at(c0.w_fwd), e.stem_taps(), e.at<float>(bn0.scale),
                                e.at<float>(bn0.shift), e.at(p->p0), batch, stream);
      e.end_timed(slot);
      if (fused == QT_ERR_UNSUPPORTED) {
        e.igemm(sd, e.at(p->xpad), e.at(c0.w_fwd), e.at(c0.y), e.at<float>(bn0.scale), e.at<float>(bn0.shift), nullptr,
                nullptr, nullptr, 1);
        e.run(qt_stem_pool(dt, e.at(c0.y), e.at<float>(p->ones), e.at<float>(p->zeros), e.at(p->p0), nullptr, nullptr,
                           batch, stream));
      } else {
        e.run(fused);
      }
    }
    // ---- residual stages ----
    size_t x = p->p0;
    for (const Block& blk : p->blocks) {
      const ConvL& c1 = p->convs[blk.conv1];
      const ConvL& c2 = p->convs[blk.conv2];
      const BnL& b1 = p->bns[c1.bn];
      const BnL& b2 = p->bns[c2.bn];
      const qt_conv_desc d1 = e.conv_desc(c1, QT_CONV_FWD), d2 = e.conv_desc(c2, QT_CONV_FWD);
      const long long M = e.rows_of(c2);
      // a transition block: conv1 and the downsample in one launch where the fused kernel covers the problem
      const bool pair = blk.ds >= 0 && e.transition_pair(blk, e.at(x), training);
      if (blk.ds >= 0 && !pair) {
        const ConvL& cd = p->convs[blk.ds];
        const BnL& bd = p->bns[cd.bn];
        const qt_conv_desc dd = e.conv_desc(cd, QT_CONV_FWD);
        e.fork();
        e.on_side(p->stats_ds, [&] {
          e.conv_bn_stats(cd, dd, e.at(x), training);
          if (!unf)
            e.igemm(dd, e.at(x), e.at(cd.w_fwd), e.at(cd.y), e.at<float>(bd.scale), e.at<float>(bd.shift), nullptr,
                    nullptr, nullptr, 0);
        });
      }
      if (!pair) e.conv_bn_stats(c1, d1, e.at(x), training);
      if (unf) {
        e.run(qt_bn_act_mask(dt, e.at(c1.y), e.at<float>(b1.scale), e.at<float>(b1.shift), nullptr, nullptr, nullptr, 1,
                             e.at(blk.a1), e.at<unsigned char>(blk.a1_bits), M, c1.cout, stream));
      } else if (!pair) {
        e.igemm(d1, e.at(x), e.at(c1.w_fwd), e.at(blk.a1), e.at<float>(b1.scale), e.at<float>(b1.shift), nullptr,
                nullptr, nullptr, 1);
      }
      if (blk.ds >= 0) {
        // the 1x1 downsample branch only needs the block input: it runs on the side stream next
        // to conv1 -> BN -> conv2 and is joined before the residual add
        const ConvL& cd = p->convs[blk.ds];
        const BnL& bd = p->bns[cd.bn];
        const qt_conv_desc dd = e.conv_desc(cd, QT_CONV_FWD);
        // (fork happened before conv1, see below)
        e.conv_bn_stats(c2, d2, e.at(blk.a1), training);
        if (!pair) e.join();   // (the fused pair ran on this stream: a head branch forked earlier keeps running beside layer4)
        if (unf) {
          e.run(qt_bn_act_mask(dt, e.at(c2.y), e.at<float>(b2.scale), e.at<float>(b2.shift), e.at(cd.y),
                               e.at<float>(bd.scale), e.at<float>(bd.shift), 1, e.at(blk.out),
                               e.at<unsigned char>(blk.out_bits), M, c2.cout, stream));
        } else {
          e.igemm(d2, e.at(blk.a1), e.at(c2.w_fwd), e.at(blk.out), e.at<float>(b2.scale), e.at<float>(b2.shift),
                  e.at(cd.y), nullptr, nullptr, 1);
        }
        (void)dd;
      } else {
        e.conv_bn_stats(c2, d2, e.at(blk.a1), training);
        if (unf) {
          e.run(qt_bn_act_mask(dt, e.at(c2.y), e.at<float>(b2.scale), e.at<float>(b2.shift), e.at(x), nullptr, nullptr, 1,
                               e.at(blk.out), e.at<unsigned char>(blk.out_bits), M, c2.cout, stream));
        } else {
          e.igemm(d2, e.at(blk.a1), e.at(c2.w_fwd), e.at(blk.out), e.at<float>(b2.scale), e.at<float>(b2.shift),
                  e.at(x), nullptr, nullptr, 1);
        }
      }
      x = blk.out;
      if (&blk == &p->blocks[p->attention ? 3 : 5] && !p->standard) {  // the heads' input is done: start the side branches
        e.fork();
        e.on_side(p->stats_ds, [&] {
          if (p->attention) {
            attn_branch();
          } else {
            quad_branch();
            if (p->has_numerical && !mlp_early) mlp_branch();
          }
        });
      }
    }
    // ---- global branch: avgpool(layer4) -> fused[:, 0:512] ----
    e.run(qt_avgpool(dt, e.at(p->blocks[7].out), e.at(p->fused), batch, 49, 512, p->fused_ld, 0, stream));
  }
  if (p->has_numerical && (!p->has_image || p->lstm)) mlp_branch();  // numerical_only / CnnLstm: nothing to overlap with
  e.join();  // quadrant + MLP columns of the fused matrix are complete
  if (p->lstm) {
    // ---- cnn+lstm/models.py:76-89: [frames][640] -> 2-layer LSTM over seq_len steps -> last step -> classifier ----
    const int T = p->seq_len, S = batch / T, H = p->lstm_h;
    qt_gemm_small_desc g;
    for (int l = 0; l < 2 && e.ok(); ++l) {
      const qt_plan::LstmL& L = p->lstm_l[l];
      // x W_ih^T + b_ih for all frames at once on the f32 MFMA path (W_ih [4H][in] is already the operand layout)
      if (l == 0) e.run(qt_cast_f32(dt, e.at(p->fused), e.at<float>(p->lstm_x0), (long long)batch * L.in, stream));
      e.igemm(e.dense_f32(batch, L.in, 4 * H), l == 0 ? e.at(p->lstm_x0) : e.at(p->lstm_x1), e.tf(L.w_ih), e.at(L.xproj),
              nullptr, e.tf(L.b_ih), nullptr, nullptr, nullptr, 0);
      e.run(qt_transpose_f32(e.tf(L.w_hh), e.at<float>(L.whh_t), 4 * H, H, stream));
      e.run(qt_lstm_forward(e.at<float>(L.xproj), e.at<float>(L.whh_t), e.tf(L.b_hh), e.at<float>(L.gates),
                            e.at<float>(L.cell), e.at<float>(L.hprev), e.at<float>(L.hout), S, T, H, stream));
      if (l == 0) {  // nn.LSTM's inter-layer dropout acts on layer 0's outputs only as layer 1's input
        e.hip(hipMemcpyAsync(e.at(p->lstm_x1), e.at(L.hout), (size_t)batch * H * 4, hipMemcpyDeviceToDevice, hs), "hipMemcpyAsync");
        if (tr && p->d.dropout_p > 0.f)
          e.run(qt_dropout(QT_F32, e.at(p->lstm_x1), batch, H, H, seed ^ 0x3C3C3C3CC3C3C3C3ull, p->d.dropout_p, stream));
      }
    }
    const float* last = e.at<float>(p->lstm_l[1].hout) + (size_t)(T - 1) * H;  // h_{T-1} of every sequence
    memset(&g, 0, sizeof(g));
    g.M = S; g.N = 128; g.K = H;
    g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
    g.a_row_stride = (long long)T * H; g.a_k_stride = 1; g.b_row_stride = H; g.b_k_stride = 1; g.c_row_stride = 128; g.relu = 1;
    e.run(qt_gemm_small(&g, last, e.tf(p->cls0.w), e.tf(p->cls0.b), e.at(p->lstm_hid), stream));
    if (tr && p->d.dropout_p > 0.f)
      e.run(qt_dropout(QT_F32, e.at(p->lstm_hid), S, 128, 128, seed ^ 0xA5A5A5A55A5A5A5Aull, p->d.dropout_p, stream));
    memset(&g, 0, sizeof(g));
    g.M = S; g.N = p->cls3.out; g.K = 128;
    g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
    g.a_row_stride = 128; g.a_k_stride = 1; g.b_row_stride = 128; g.b_k_stride = 1; g.c_row_stride = p->cls3.out;
    e.run(qt_gemm_small(&g, e.at(p->lstm_hid), e.tf(p->cls3.w), e.tf(p->cls3.b), logits, stream));
    p->last_batch = batch;
    p->last_training = training;
    p->last_seed = seed;
    return e.status;
  }
  // ---- classifier: Linear -> ReLU -> Dropout -> Linear ----
  e.linear(e.linear_desc(p->cls0.in, p->cls0.out, QT_CONV_FWD), e.at(p->fused), e.at(p->cls0.w_fwd), e.at(p->hidden),
           e.tf(p->cls0.b), 1);
  if (tr && p->d.dropout_p > 0.f)
    e.run(qt_dropout(dt, e.at(p->hidden), batch, p->hidden_dim, p->hidden_dim, seed ^ 0xA5A5A5A55A5A5A5Aull,
                     p->d.dropout_p, stream));
  {
    qt_gemm_small_desc g;
    memset(&g, 0, sizeof(g));
    g.M = batch; g.N = p->cls3.out; g.K = p->cls3.in;
    g.a_dtype = dt; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
    g.a_row_stride = p->cls3.in; g.a_k_stride = 1; g.b_row_stride = p->cls3.in; g.b_k_stride = 1;
    g.c_row_stride = p->cls3.out;
    e.run(qt_gemm_small(&g, e.at(p->hidden), e.tf(p->cls3.w), e.tf(p->cls3.b), logits, stream));
  }
  (void)hs;
  p->last_batch = batch;
  p->last_training = training;
  p->last_seed = seed;
  return e.status;
}

// zero-fill helper (hipMemsetAsync is capture-safe)
int zero(void* ptr, size_t bytes, void* stream) {
  hipError_t err = hipMemsetAsync(ptr, 0, bytes, static_cast<hipStream_t>(stream));
  if (err != hipSuccess) {
    qt_set_error("hipMemsetAsync: %s", hipGetErrorString(err));
    return QT_ERR_LAUNCH;
  }
  return QT_OK;
}

struct Bwd : Exec {
  float* const* G;  // gradient pointers (same indexing as T), NULL = not wanted
  float* gf(int idx) const { return idx < 0 ? nullptr : G[idx]; }
  bool evalbn = false;  // the forward ran BatchNorm on running statistics (training == 2): no batch-mean terms
  long long bn_count(long long M) const { return evalbn ? 0 : M; }

  // BatchNorm backward for conv c given g (in gy or external): dy -> c.gy
  void bn_backward(const ConvL& c, const void* g, void* g_out, float* pre_partial = nullptr, int pre_rows = 0) {
    if (!ok()) return;
    const BnL& bn = p->bns[c.bn];
    const long long M = rows_of(c);
    float* part = pre_partial ? pre_partial : at<float>(p->stats);
    int rows = pre_rows;
    if (!pre_partial) {
      run(qt_bn_bwd_reduce(dt, g, nullptr, at(c.y), at<float>(bn.mean), at<float>(bn.invstd), part, M, bn.C, stream));
      if (!ok()) return;
      rows = qt_bn_bwd_partial_rows(M, bn.C);
    }
    run(qt_bn_bwd_finalize(part, rows, bn.C, bn_count(M), tf(bn.gamma), at<float>(bn.invstd), gf(bn.gamma), gf(bn.beta), 0,
                           at<float>(bn.coef), stream));
    run(qt_bn_bwd_apply(dt, g, nullptr, at(c.y), at<float>(bn.mean), at<float>(bn.invstd), at<float>(bn.coef),
                        at(c.gy), g_out, M, bn.C, stream));
  }
  // data gradient of conv c: dst = conv_transpose(c.gy) (+resid) (* (mask > 0)).  A stride-2
  // conv is run as four stride-1 gathers, one per parity class of the destination pixel, so no
  // MFMA work is spent on taps that cannot reach a pixel (a dense gather would waste 3/4).
  // returns the number of partial rows each link received
  // sparse: (1x1 stride 2 as `c`) leave the three parity classes the conv does not reach unwritten instead of zeroing
  // the map; (3x3 stride 2 as `c`) `resid` is such a map: only class (0,0) adds it.  Saves a fill of the block input's
  // size and three quarters of the residual reads per transition.
  // mask / mask_bits: the ReLU mask as a tensor or as one bit per element (qt_conv_io.relu_mask_bits), at most one of them
  int dgrad(const ConvL& c, void* dst, const void* resid, const void* mask, const BnLink* links = nullptr,
            int nlinks = 0, bool sparse = false, const unsigned char* mask_bits = nullptr) {
    if (c.stride == 1) {
      const qt_conv_desc d = conv_desc(c, QT_CONV_DGRAD);
      igemm(d, at(c.gy), at(c.w_dgrad), dst, nullptr, nullptr, resid, mask, nullptr, 0, -1, links, nlinks, mask_bits);
      return qt_conv2d_stats_rows(&d);
    }
    if (c.merged_dgrad) {  // all four parity classes in one 2x2-tap launch over the gradient map
      qt_conv_desc d;
      memset(&d, 0, sizeof(d));
      d.dtype = dt; d.mode = QT_CONV_FWD; d.batch = B;
      d.in_h = d.in_w = c.hout; d.out_h = d.out_w = c.hin / 2;
      d.k_per_tap = c.cout; d.n_out = 4 * c.cin;
      d.kh = d.kw = 2; d.stride = 1; d.pad = 0;
      d.src_pix_stride = c.cout; d.src_row_stride = c.hout * c.cout; d.src_img_stride = (long long)c.hout * c.hout * c.cout;
      d.dst_sub = 2; d.dst_h = d.dst_w = c.hin; d.dst_merge = c.cin; d.dst_merge_res0 = sparse ? 1 : 0;
      d.dst_merge_extra = c.merged5 ? 1 : 0;   // (+ the downsample's gradient map through the fifth tap slot)
      igemm(d, at(c.gy), at(c.w_dgrad), dst, nullptr, nullptr, resid, mask, nullptr, 0, 1, links, nlinks, mask_bits,
            c.merged5 ? at(p->convs[c.slot_conv].gy) : nullptr);
      return qt_conv2d_stats_rows(&d);
    }
    bool empty_class = false;
    for (int cls = 0; cls < 4; ++cls) empty_class |= c.cls_kh[cls] * c.cls_kw[cls] == 0;
    if (empty_class) {  // 1x1 stride 2: three of four pixels receive nothing from the conv
      if (resid || mask || mask_bits || nlinks) {
        status = QT_ERR_UNSUPPORTED;
        qt_set_error("dgrad: 1x1 stride-2 with residual/mask/links is not wired");
        return 0;
      }
      if (!sparse) run(zero(dst, (size_t)B * c.hin * c.hin * c.cin * p->esz, stream));
    }
    int rows = 0;
    for (int cls = 0; cls < 4 && ok(); ++cls) {
      if (c.cls_kh[cls] * c.cls_kw[cls] == 0) continue;
      qt_conv_desc d;
      memset(&d, 0, sizeof(d));
      d.dtype = dt; d.mode = QT_CONV_FWD; d.batch = B;
      d.in_h = d.in_w = c.hout; d.out_h = d.out_w = c.hin / 2;
      d.k_per_tap = c.cout; d.n_out = c.cin;
      d.kh = c.cls_kh[cls]; d.kw = c.cls_kw[cls]; d.stride = 1; d.pad = 0;
      d.src_pix_stride = c.cout; d.src_row_stride = c.hout * c.cout; d.src_img_stride = (long long)c.hout * c.hout * c.cout;
      d.dst_sub = 2; d.dst_h = d.dst_w = c.hin; d.dst_off_h = cls >> 1; d.dst_off_w = cls & 1;
      BnLink l2[2];
      for (int k = 0; k < nlinks && k < 2; ++k) {
        l2[k] = links[k];
        l2[k].partial = links[k].partial + (size_t)rows * 2 * c.cin;  // each class appends its tile rows
      }
      igemm(d, at(c.gy), at<unsigned char>(c.w_dgrad) + (size_t)c.cls_off[cls] * p->esz, dst, nullptr, nullptr,
            sparse && !empty_class && cls != 0 ? nullptr : resid, mask, nullptr, 0, 1, l2, nlinks, mask_bits);
      rows += qt_conv2d_stats_rows(&d);
    }
    return rows;
  }
  double wgrad_bytes(const ConvL& c, const qt_conv_desc& f) const {
    const double es = dt == QT_F32 ? 4.0 : 2.0, imgs = (double)f.batch * qt_quad_regions(f.quad);
    return es * imgs * ((double)f.in_h * f.in_w * f.k_per_tap + (double)f.out_h * f.out_w * f.n_out) +
           4.0 * f.kh * f.kw * f.k_per_tap * f.n_out;
  }
  // weight gradient of conv c: dy = c.gy, x = src
  void wgrad(const ConvL& c, const qt_conv_desc& fwd_desc, const void* src, bool stem) {
    if (!ok() || !gf(c.w)) return;
    fork();
    void* ws_ = wstream;
    const size_t n = stem ? (size_t)64 * 7 * 32 : (size_t)c.cout * c.cin * c.k * c.k;
    if (!stem && qt_conv2d_wgrad_workspace_bytes(&fwd_desc) > 0) {
      // streaming kernels (3x3 stride 1; the stride-2 pair of a transition block): the fixed-order sum of the partial
      // filters WRITES .grad in OIHW -- no scratch, no zero fill, no atomics
      const int slot = begin_timed(conv_flops(fwd_desc), 2, ws_, wgrad_bytes(c, fwd_desc));
      run(qt_conv2d_wgrad_oihw(&fwd_desc, at(c.gy), src, gf(c.w), at(p->wgrad_part), p->wgrad_part_bytes, ws_));
      end_timed(slot, ws_);
      return;
    }
    if (!stem && c.k == 1) {  // [O][1][I] is already OIHW
      run(zero(gf(c.w), n * 4, ws_));
      const int slot = begin_timed(conv_flops(fwd_desc), 2, ws_, wgrad_bytes(c, fwd_desc));
      run(qt_conv2d_wgrad(&fwd_desc, at(c.gy), src, gf(c.w), ws_));
      end_timed(slot, ws_);
      return;
    }
    if (!stem && c.dw == 0) {
      // (the streaming kernels refused a shape they accepted when the plan was laid out, e.g. a QTCNN_* switch changed since)
      status = QT_ERR_UNSUPPORTED;
      qt_set_error("wgrad: no accumulation scratch was laid out for this convolution");
      return;
    }
    const int slot = begin_timed(conv_flops(fwd_desc), 2, ws_, wgrad_bytes(c, fwd_desc));  // c.dw was zeroed at the start of this backward
    run(qt_conv2d_wgrad_ws(&fwd_desc, at(c.gy), src, at<float>(c.dw), at(p->wgrad_part), p->wgrad_part_bytes, ws_));
    end_timed(slot, ws_);
    if (stem)
      run(qt_unpack_stem_wgrad(at<float>(c.dw), gf(c.w), 0, ws_));
    else
      run(qt_unpack_conv_wgrad(at<float>(c.dw), gf(c.w), c.cout, c.cin, c.k, c.k, 0, ws_));
  }
};

int backward(qt_plan* p, void* workspace, void* const* T, float* const* G, const float* numerical,
             const float* dlogits, int phases, void* stream) {
  Bwd e;
  e.p = p; e.ws = static_cast<unsigned char*>(workspace); e.T = T; e.stream = stream; e.B = p->last_batch;
  e.dt = p->d.dtype; e.G = G;
  e.setup_side();
  if ((phases & QT_BWD_HEAD) || p->dw_dirty) {  // once per backward, before any weight-gradient launch
    e.run(zero(e.at(p->dw_begin), p->dw_end - p->dw_begin, stream));
    p->dw_dirty = false;
  }
  if (phases & QT_BWD_LAYER1) p->dw_dirty = true;
  const int dt = e.dt;
  const int B = e.B;
  const bool tr = p->last_training == 1;
  e.evalbn = p->last_training == 2;
  const float drop_mul = (tr && p->d.dropout_p > 0.f) ? 1.f / (1.f - p->d.dropout_p) : 1.f;
  bool backbone_grads = false;
  if (p->has_image)
    for (size_t i = 0; i < p->convs.size(); ++i) {
      if ((int)i == p->quad_conv || (int)i == p->sub_conv) continue;
      if (G[p->convs[i].w]) backbone_grads = true;
    }
  if (backbone_grads && p->last_training == 0) {
    qt_set_error("qt_plan_backward: the last forward ran fused eval kernels (training = 0) and kept nothing for a backbone "
                 "backward; run it with training = 2 (eval statistics, tensors kept)");
    return QT_ERR_UNSUPPORTED;
  }
  if (p->lstm) {
    if (backbone_grads) {  // the reference freezes cnn_backbone (cnn+lstm/models.py:26-27)
      qt_set_error("qt_plan_backward: CnnLstm keeps its ResNet-18 frozen; gradients through the LSTM into the backbone are not implemented");
      return QT_ERR_UNSUPPORTED;
    }
    if (!(phases & QT_BWD_HEAD)) return QT_OK;
    // ---- classifier -> last step -> LSTM layer 1 -> (dropout) -> LSTM layer 0 -> pose MLP; everything f32 and thin ----
    const int T = p->seq_len, S = B / T, H = p->lstm_h;
    qt_gemm_small_desc g;
    auto gemm = [&](int M, int N, int K, const void* A, int adt, long long ars, long long aks, const void* Bm, int bdt,
                    long long brs, long long bks, void* C, long long crs) {
      memset(&g, 0, sizeof(g));
      g.M = M; g.N = N; g.K = K; g.a_dtype = adt; g.b_dtype = bdt; g.c_dtype = QT_F32;
      g.a_row_stride = ars; g.a_k_stride = aks; g.b_row_stride = brs; g.b_k_stride = bks; g.c_row_stride = crs;
      if (C) e.run(qt_gemm_small(&g, A, Bm, nullptr, C, stream));
    };
    const float* last = e.at<float>(p->lstm_l[1].hout) + (size_t)(T - 1) * H;
    float* dhid = e.at<float>(p->lstm_dhid);
    gemm(S, 128, p->cls3.out, dlogits, QT_F32, p->cls3.out, 1, e.tf(p->cls3.w), QT_F32, 1, 128, dhid, 128);
    e.run(qt_relu_mask_scale(QT_F32, dhid, e.at(p->lstm_hid), (long long)S * 128, drop_mul, stream));
    if (e.gf(p->cls3.b)) e.run(qt_col_sum(QT_F32, dlogits, S, p->cls3.out, p->cls3.out, e.gf(p->cls3.b), 0, stream));
    gemm(p->cls3.out, 128, S, dlogits, QT_F32, 1, p->cls3.out, e.at(p->lstm_hid), QT_F32, 1, 128, e.gf(p->cls3.w), 128);
    if (e.gf(p->cls0.b)) e.run(qt_col_sum(QT_F32, dhid, S, 128, 128, e.gf(p->cls0.b), 0, stream));
    gemm(128, H, S, dhid, QT_F32, 1, 128, last, QT_F32, 1, (long long)T * H, e.gf(p->cls0.w), H);
    gemm(S, H, 128, dhid, QT_F32, 128, 1, e.tf(p->cls0.w), QT_F32, 1, H, e.at(p->lstm_dlast), H);
    for (int l = 1; l >= 0 && e.ok(); --l) {
      const qt_plan::LstmL& L = p->lstm_l[l];
      e.run(qt_lstm_backward(l == 1 ? nullptr : e.at<float>(p->lstm_dx1), l == 1 ? e.at<float>(p->lstm_dlast) : nullptr,
                             e.at<float>(L.gates), e.at<float>(L.cell), e.tf(L.w_hh), e.at<float>(L.dgates), S, T, H, stream));
      const float* dG = e.at<float>(L.dgates);
      // dW_ih = dgates^T x, dW_hh = dgates^T h_prev: weight-gradient kernel (f32 MFMA), contraction over the frames
      auto wgrad_dense = [&](const void* X, int in, float* dw) {
        if (!dw || !e.ok()) return;
        const qt_conv_desc wd = e.dense_f32(B, in, 4 * H);
        e.run(zero(dw, (size_t)4 * H * in * 4, stream));
        e.run(qt_conv2d_wgrad(&wd, dG, X, dw, stream));
      };
      wgrad_dense(l == 0 ? e.at(p->lstm_x0) : e.at(p->lstm_x1), L.in, e.gf(L.w_ih));
      wgrad_dense(e.at(L.hprev), H, e.gf(L.w_hh));
      e.run(qt_transpose_f32(e.tf(L.w_ih), e.at<float>(L.wih_t), 4 * H, L.in, stream));  // [in][4H]: operand of dx
      if (e.gf(L.b_ih)) e.run(qt_col_sum(QT_F32, dG, B, 4 * H, 4 * H, e.gf(L.b_ih), 0, stream));
      if (e.gf(L.b_hh)) e.run(qt_col_sum(QT_F32, dG, B, 4 * H, 4 * H, e.gf(L.b_hh), 0, stream));
      if (l == 1) {
        e.igemm(e.dense_f32(B, 4 * H, H), dG, e.at(L.wih_t), e.at(p->lstm_dx1), nullptr, nullptr, nullptr, nullptr, nullptr, 0);
        if (tr && p->d.dropout_p > 0.f)
          e.run(qt_scale_by_nonzero(e.at<float>(p->lstm_dx1), e.at<float>(p->lstm_x1), (long long)B * H, drop_mul, stream));
      } else {  // only the pose-MLP columns of the fused features have trainable producers
        e.igemm(e.dense_f32(B, 4 * H, 128), dG, e.at<float>(L.wih_t) + (size_t)p->mlp_col0 * 4 * H, e.at(p->lstm_dz), nullptr,
                nullptr, nullptr, nullptr, nullptr, 0);
      }
    }
    const float* dz = e.at<float>(p->lstm_dz);
    if (e.gf(p->mlp1.b)) e.run(qt_col_sum(QT_F32, dz, B, 128, 128, e.gf(p->mlp1.b), 0, stream));
    gemm(128, 128, B, dz, QT_F32, 1, 128, e.at(p->h1), QT_F32, 1, 128, e.gf(p->mlp1.w), 128);
    gemm(B, 128, 128, dz, QT_F32, 128, 1, e.tf(p->mlp1.w), QT_F32, 1, 128, e.at(p->dh1), 128);
    e.run(qt_relu_mask_scale(QT_F32, e.at(p->dh1), e.at(p->h1), (long long)B * 128, 1.f, stream));
    if (e.gf(p->mlp0.b)) e.run(qt_col_sum(QT_F32, e.at(p->dh1), B, 128, 128, e.gf(p->mlp0.b), 0, stream));
    gemm(128, p->mlp0.in, B, e.at(p->dh1), QT_F32, 1, 128, numerical, QT_F32, 1, p->mlp0.in, e.gf(p->mlp0.w), p->mlp0.in);
    return e.status;
  }

  if (phases & QT_BWD_HEAD) {
    qt_gemm_small_desc g;
    // The chain the backbone waits for is  dlogits -> dhidden -> dfused -> (pool backward); bias sums and
    // weight gradients hang off it and go to the side stream as soon as their inputs exist.
    // ---- classifier.3: d(loss)/d(hidden), then ReLU + dropout backward ----
    memset(&g, 0, sizeof(g));
    g.M = B; g.N = p->cls3.in; g.K = p->cls3.out;
    g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = dt;
    g.a_row_stride = p->cls3.out; g.a_k_stride = 1; g.b_row_stride = 1; g.b_k_stride = p->cls3.in;
    g.c_row_stride = p->cls3.in;
    e.run(qt_gemm_small(&g, dlogits, e.tf(p->cls3.w), nullptr, e.at(p->dhidden), stream));
    e.run(qt_relu_mask_scale(dt, e.at(p->dhidden), e.at(p->hidden), (long long)B * p->hidden_dim, drop_mul, stream));
    const qt_conv_desc lf = e.linear_desc(p->cls0.in, p->cls0.out, QT_CONV_FWD);
    {
      e.fork();
      void* ss = e.wstream ? e.wstream : stream;
      void* ts = ss;
      if (e.gf(p->cls3.b)) e.run(qt_col_sum(QT_F32, dlogits, B, p->cls3.out, p->cls3.out, e.gf(p->cls3.b), 0, ts));
      if (e.gf(p->cls3.w)) {
        memset(&g, 0, sizeof(g));
        g.M = p->cls3.out; g.N = p->cls3.in; g.K = B;
        g.a_dtype = QT_F32; g.b_dtype = dt; g.c_dtype = QT_F32;
        g.a_row_stride = 1; g.a_k_stride = p->cls3.out; g.b_row_stride = 1; g.b_k_stride = p->cls3.in;
        g.c_row_stride = p->cls3.in;
        e.run(qt_gemm_small(&g, dlogits, e.at(p->hidden), nullptr, e.gf(p->cls3.w), ts));
      }
      // ---- classifier.0: bias and weight gradients ----
      if (e.gf(p->cls0.b)) e.run(qt_col_sum(dt, e.at(p->dhidden), B, p->cls0.out, p->cls0.out, e.gf(p->cls0.b), 0, ts));
      if (e.gf(p->cls0.w)) {   // written, not accumulated: no 58 MB zero fill, no float atomics (one row range per tile)
        const int slot = e.begin_timed(e.conv_flops(lf), 2, ss);
        e.run(qt_linear_wgrad(dt, e.at(p->dhidden), e.at(p->fused), e.gf(p->cls0.w), B, p->cls0.out, p->cls0.in, ss));
        e.end_timed(slot, ss);
      }
    }
    const bool need_dfused = p->has_numerical || (p->has_image && (!p->standard || backbone_grads));
    if (need_dfused)
      e.linear(e.linear_desc(p->cls0.in, p->cls0.out, QT_CONV_DGRAD), e.at(p->dhidden), e.at(p->cls0.w_dgrad), e.at(p->dfused),
               nullptr, 0);
    // ---- numerical MLP: seven small dependent kernels that only need dfused; they run on the side
    // stream (behind classifier.0's weight gradient) while the main stream enters the backbone ----
    if (p->attention) {
      // numerical_mlp = Linear -> ReLU -> Dropout (models.py:43-46), output inside the fused matrix
      e.fork();
      void* ms = e.wstream ? e.wstream : stream;
      e.run(qt_relu_mask_cols(dt, e.at(p->dfused), e.at(p->fused), e.at<float>(p->dh1), B, p->mlp0.out, p->fused_ld,
                              p->mlp_col0, drop_mul, ms));
      if (e.gf(p->mlp0.b)) e.run(qt_col_sum(QT_F32, e.at(p->dh1), B, p->mlp0.out, p->mlp0.out, e.gf(p->mlp0.b), 0, ms));
      if (e.gf(p->mlp0.w)) {
        memset(&g, 0, sizeof(g));
        g.M = p->mlp0.out; g.N = p->mlp0.in; g.K = B;
        g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
        g.a_row_stride = 1; g.a_k_stride = p->mlp0.out; g.b_row_stride = 1; g.b_k_stride = p->mlp0.in;
        g.c_row_stride = p->mlp0.in;
        e.run(qt_gemm_small(&g, e.at(p->dh1), numerical, nullptr, e.gf(p->mlp0.w), ms));
      }
    } else if (p->has_numerical) {
      e.fork();
      void* ms = e.wstream ? e.wstream : stream;
      const unsigned char* dz = e.at<unsigned char>(p->dfused) + (size_t)p->mlp_col0 * p->esz;
      if (e.gf(p->mlp1.b)) e.run(qt_col_sum(dt, dz, B, p->mlp1.out, p->fused_ld, e.gf(p->mlp1.b), 0, ms));
      if (e.gf(p->mlp1.w)) {
        memset(&g, 0, sizeof(g));
        g.M = p->mlp1.out; g.N = p->mlp1.in; g.K = B;
        g.a_dtype = dt; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
        g.a_row_stride = 1; g.a_k_stride = p->fused_ld; g.b_row_stride = 1; g.b_k_stride = p->mlp1.in;
        g.c_row_stride = p->mlp1.in;
        e.run(qt_gemm_small(&g, dz, e.at(p->h1), nullptr, e.gf(p->mlp1.w), ms));
      }
      memset(&g, 0, sizeof(g));
      g.M = B; g.N = p->mlp1.in; g.K = p->mlp1.out;
      g.a_dtype = dt; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
      g.a_row_stride = p->fused_ld; g.a_k_stride = 1; g.b_row_stride = 1; g.b_k_stride = p->mlp1.in;
      g.c_row_stride = p->mlp1.in;
      e.run(qt_gemm_small(&g, dz, e.tf(p->mlp1.w), nullptr, e.at(p->dh1), ms));
      e.run(qt_relu_mask_scale(QT_F32, e.at(p->dh1), e.at(p->h1), (long long)B * p->mlp0.out, drop_mul, ms));
      if (e.gf(p->mlp0.b)) e.run(qt_col_sum(QT_F32, e.at(p->dh1), B, p->mlp0.out, p->mlp0.out, e.gf(p->mlp0.b), 0, ms));
      if (e.gf(p->mlp0.w)) {
        memset(&g, 0, sizeof(g));
        g.M = p->mlp0.out; g.N = p->mlp0.in; g.K = B;
        g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
        g.a_row_stride = 1; g.a_k_stride = p->mlp0.out; g.b_row_stride = 1; g.b_k_stride = p->mlp0.in;
        g.c_row_stride = p->mlp0.in;
        e.run(qt_gemm_small(&g, e.at(p->dh1), numerical, nullptr, e.gf(p->mlp0.w), ms));
      }
    }
    if (p->attention) {
      // ---- quadrant vectors: mean-pool backward (+ReLU mask) -> conv bias / weight gradients ----
      const ConvL& cq = p->convs[p->quad_conv];
      const ConvL& cs = p->convs[p->sub_conv];
      e.run(qt_region_avgpool_bwd(dt, e.at(p->dfused), dt, e.at(cq.y), e.at(cq.gy), B, 2, 196, 128, p->fused_ld, 512, stream));
      if (e.gf(cq.bias)) {
        e.fork();
        e.run(qt_col_sum(dt, e.at(cq.gy), (long long)B * 4 * 196, 128, 128, e.gf(cq.bias), 0, e.wstream ? e.wstream : stream));
      }
      e.wgrad(cq, e.region_desc(cq, 2, QT_CONV_FWD), e.at(p->blocks[3].out), false);
      // ---- attention gate (models.py:81-89), then the sub-quadrant vectors ----
      e.run(qt_attention_gate_bwd(dt, e.at(p->dfused), e.at<float>(p->vsub), e.at<float>(p->att_act),
                                  e.at<float>(p->att_alpha), e.tf(p->att0.w), e.tf(p->att2.w), e.at<float>(p->att_ds),
                                  e.at<float>(p->att_dpre), e.at<float>(p->dvsub), B, p->fused_ld, 512 + 4 * 128, stream));
      {
        e.fork();
        void* as = e.wstream ? e.wstream : stream;
        const int rows = B * 16;
        if (e.gf(p->att0.w)) {  // [32][64] = dpre^T v
          memset(&g, 0, sizeof(g));
          g.M = 32; g.N = 64; g.K = rows;
          g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
          g.a_row_stride = 1; g.a_k_stride = 32; g.b_row_stride = 1; g.b_k_stride = 64; g.c_row_stride = 64;
          e.run(qt_gemm_small(&g, e.at(p->att_dpre), e.at(p->vsub), nullptr, e.gf(p->att0.w), as));
        }
        if (e.gf(p->att0.b)) e.run(qt_col_sum(QT_F32, e.at(p->att_dpre), rows, 32, 32, e.gf(p->att0.b), 0, as));
        if (e.gf(p->att2.w)) {  // [1][32] = ds^T act
          memset(&g, 0, sizeof(g));
          g.M = 1; g.N = 32; g.K = rows;
          g.a_dtype = QT_F32; g.b_dtype = QT_F32; g.c_dtype = QT_F32;
          g.a_row_stride = 0; g.a_k_stride = 1; g.b_row_stride = 1; g.b_k_stride = 32; g.c_row_stride = 32;
          e.run(qt_gemm_small(&g, e.at(p->att_ds), e.at(p->att_act), nullptr, e.gf(p->att2.w), as));
        }
        if (e.gf(p->att2.b)) e.run(qt_col_sum(QT_F32, e.at(p->att_ds), rows, 1, 1, e.gf(p->att2.b), 0, as));
      }
      e.run(qt_region_avgpool_bwd(dt, e.at(p->dvsub), QT_F32, e.at(cs.y), e.at(cs.gy), B, 4, 49, 64, 16 * 64, 0, stream));
      if (e.gf(cs.bias)) {
        e.fork();
        e.run(qt_col_sum(dt, e.at(cs.gy), (long long)B * 16 * 49, 64, 64, e.gf(cs.bias), 0, e.wstream ? e.wstream : stream));
      }
      e.wgrad(cs, e.region_desc(cs, 4, QT_CONV_FWD), e.at(p->blocks[3].out), false);
    }
    // ---- quadrant head (weights are trainable in every variant) ----
    if (p->has_image && !p->standard && !p->attention) {
      const ConvL& cq = p->convs[p->quad_conv];
      e.run(qt_quad_pool_bwd(dt, e.at(p->dfused), e.at(p->q), e.at(p->dq), B, p->fused_ld, 512, stream));
      if (e.gf(cq.bias)) {
        e.fork();
        e.run(qt_col_sum(dt, e.at(p->dq), (long long)B * 196, 128, 128, e.gf(cq.bias), 0, e.wstream ? e.wstream : stream));
      }
      e.wgrad(cq, e.quad_desc(QT_CONV_FWD), e.at(p->blocks[5].out), false);
    }
  }

  // (no join after a partial phase: the caller orders its consumer behind the side stream with
  //  qt_plan_side_fence, so the main chain never stalls on the weight-gradient stream)

  // The backbone can be run in three calls -- QT_BWD_LAYER4 (blocks 7, 6: 8.4 M of the 11.2 M backbone
  // parameters), QT_BWD_LAYER32 (blocks 5..2: 2.6 M), QT_BWD_LAYER1 (blocks 1, 0 and the stem: 0.16 M) -- so
  // that a data-parallel caller reduces each bucket while the next phase runs and only 0.6 MB of gradients
  // is left to reduce after the last kernel.
  if ((phases & QT_BWD_BACKBONE) && backbone_grads) {
    const bool do_l4 = (phases & QT_BWD_LAYER4) != 0, do_l32 = (phases & QT_BWD_LAYER32) != 0;
    const bool do_rest = (phases & QT_BWD_LAYER1) != 0;  // the last phase: layer1 and the stem
    int rows_bn2 = 0;  // partial rows of bn2 / downsample-BN of the block being entered (0 = none yet)
    if (do_l4) {
      // gradient of layer4's output through avgpool (+ ReLU mask of the block output)
      e.run(qt_avgpool_bwd(dt, e.at(p->dfused), e.at(p->blocks[7].out), e.at(p->blocks[7].gout), B, 49, 512,
                           p->fused_ld, 0, stream));
    } else {
      rows_bn2 = p->bwd_rows_bn2;
    }
    const int bi_hi = do_l4 ? 7 : (do_l32 ? 5 : 1), bi_lo = do_rest ? 0 : (do_l32 ? 2 : 6);
    // ReLU masks of the data-gradient epilogues as one bit per element (written by the training forward's qt_bn_act_mask
    // launches): 1/16 of the bytes of the bf16 activation they replace as an operand.  QTCNN_MASK_BITS=0: the activations.
    static const bool mask_bits = !(getenv("QTCNN_MASK_BITS") && atoi(getenv("QTCNN_MASK_BITS")) == 0);
    for (int bi = bi_hi; bi >= bi_lo; --bi) {
      const Block& blk = p->blocks[bi];
      const ConvL& c1 = p->convs[blk.conv1];
      const ConvL& c2 = p->convs[blk.conv2];
      const size_t x = bi == 0 ? p->p0 : p->blocks[bi - 1].out;
      const qt_conv_desc f1 = e.conv_desc(c1, QT_CONV_FWD), f2 = e.conv_desc(c2, QT_CONV_FWD);
      // bn2 / conv2: the partial sums were emitted by the dgrad that wrote blk.gout
      e.bn_backward(c2, e.at(blk.gout), nullptr, rows_bn2 ? e.at<float>(p->stats_bn2) : nullptr, rows_bn2);
      e.wgrad(c2, f2, e.at(blk.a1), false);
      {
        const BnL& b1 = p->bns[c1.bn];
        Exec::BnLink l = {e.at(c1.y), e.at<float>(b1.mean), e.at<float>(b1.invstd), e.at<float>(p->stats_bn1)};
        const int r1 = mask_bits ? e.dgrad(c2, e.at(c1.gy), nullptr, nullptr, &l, 1, false, e.at<unsigned char>(blk.a1_bits))
                                 : e.dgrad(c2, e.at(c1.gy), nullptr, e.at(blk.a1), &l, 1);
        // bn1 / conv1
        e.bn_backward(c1, e.at(c1.gy), nullptr, e.at<float>(p->stats_bn1), r1);
      }
      e.wgrad(c1, f1, e.at(x), false);
      // gradient w.r.t. the block input = conv1 dgrad + identity path (+ quadrant head for layer3's output)
      const void* resid = e.at(blk.gout);
      // the downsample's data gradient reaches one pixel in four: if nothing else reads the map densely (the region
      // heads below do), leave the rest unwritten and let only that parity class of conv1's data gradient add it
      const bool sparse_ds = blk.ds >= 0 && c1.stride == 2 && c1.k == 3 && !c1.merged5 && !(bi == 4 && p->attention) &&
                             !(bi == 6 && !p->standard && !p->attention);
      if (blk.ds >= 0) {
        const ConvL& cd = p->convs[blk.ds];
        e.bn_backward(cd, e.at(blk.gout), nullptr, rows_bn2 ? e.at<float>(p->stats_ds) : nullptr, rows_bn2);
        e.wgrad(cd, e.conv_desc(cd, QT_CONV_FWD), e.at(x), false);
        if (c1.merged5) {   // the downsample's data gradient is the fifth tap slot of conv1's launch below: no map, no residual
          resid = nullptr;
        } else {
          e.dgrad(cd, e.at(blk.gtmp), nullptr, nullptr, nullptr, 0, sparse_ds);
          resid = e.at(blk.gtmp);
        }
      }
      if (bi == 4 && p->attention) {  // layer2's output also feeds the two heads
        const ConvL& cq = p->convs[p->quad_conv];
        const ConvL& cs = p->convs[p->sub_conv];
        e.igemm(e.region_desc(cq, 2, QT_CONV_DGRAD), e.at(cq.gy), e.at(cq.w_dgrad), e.at(p->gbase_tmp), nullptr, nullptr,
                resid, nullptr, nullptr, 0);
        e.igemm(e.region_desc(cs, 4, QT_CONV_DGRAD), e.at(cs.gy), e.at(cs.w_dgrad), e.at(p->gbase_tmp2), nullptr, nullptr,
                e.at(p->gbase_tmp), nullptr, nullptr, 0);
        resid = e.at(p->gbase_tmp2);
      }
      if (bi == 6 && !p->standard && !p->attention) {
        const ConvL& cq = p->convs[p->quad_conv];
        e.igemm(e.quad_desc(QT_CONV_DGRAD), e.at(p->dq), e.at(cq.w_dgrad), e.at(p->gbase_tmp), nullptr, nullptr, resid,
                nullptr, nullptr, 0);
        resid = e.at(p->gbase_tmp);
      }
      void* gprev = bi == 0 ? e.at(p->g_p0) : e.at(p->blocks[bi - 1].gout);
      const void* mask = (bi == 0 || mask_bits) ? nullptr : e.at(x);
      const unsigned char* mbits = (bi > 0 && mask_bits) ? e.at<unsigned char>(p->blocks[bi - 1].out_bits) : nullptr;
      Exec::BnLink links[2];
      int nlinks = 0;
      if (bi > 0) {  // gprev feeds bn2 (and the downsample BN) of the previous block
        const Block& pb = p->blocks[bi - 1];
        const ConvL& pc2 = p->convs[pb.conv2];
        const BnL& pb2 = p->bns[pc2.bn];
        links[nlinks++] = {e.at(pc2.y), e.at<float>(pb2.mean), e.at<float>(pb2.invstd), e.at<float>(p->stats_bn2)};
        if (pb.ds >= 0) {
          const ConvL& pcd = p->convs[pb.ds];
          const BnL& pbd = p->bns[pcd.bn];
          links[nlinks++] = {e.at(pcd.y), e.at<float>(pbd.mean), e.at<float>(pbd.invstd), e.at<float>(p->stats_ds)};
        }
      }
      rows_bn2 = e.dgrad(c1, gprev, resid, mask, links, nlinks, sparse_ds, mbits);
      if (bi == 0) rows_bn2 = 0;
    }
    p->bwd_rows_bn2 = rows_bn2;
    if (do_rest) {
    // ---- stem ----
    const ConvL& c0 = p->convs[0];
    const BnL& bn0 = p->bns[c0.bn];
    // max-pool backward + ReLU mask + bn1 backward without materialising d(loss)/d(relu output):
    // the BatchNorm sums come from the pooled side (each pooled cell feeds exactly one conv1
    // position: 2 x 103 MB read instead of a pass over two 411 MB maps), then one kernel gathers
    // the <= 4 pooled cells of every conv1 position and writes d(loss)/d(conv1 output) directly.
    // QTCNN_STEM_FUSED=0 keeps the three-pass form (pool backward, reduce, apply) for A/B runs.
    static const bool fused = !(getenv("QTCNN_STEM_FUSED") && atoi(getenv("QTCNN_STEM_FUSED")) == 0);
    bool stem_done = false;   // conv1's weight gradient has been produced by the one-launch stem backward
    if (fused) {
      const int rows = qt_stem_bn_bwd_sums_rows(B);
      e.run(qt_stem_bn_bwd_sums(dt, e.at(p->g_p0), e.at(p->ymax), e.at<float>(bn0.scale), e.at<float>(bn0.shift),
                                e.at<float>(bn0.mean), e.at<float>(bn0.invstd), e.at<float>(p->stats), B, stream));
      e.run(qt_bn_bwd_finalize(e.at<float>(p->stats), rows, bn0.C, e.bn_count((long long)B * 112 * 112), e.tf(bn0.gamma),
                               e.at<float>(bn0.invstd), e.gf(bn0.gamma), e.gf(bn0.beta), 0, e.at<float>(bn0.coef), stream));
      // bf16: BatchNorm / ReLU / max-pool backward of conv1's output AND conv1's weight gradient in one launch on the
      // weight-gradient stream (the map d(loss)/d(conv1 output) never exists); otherwise the apply pass, then wgrad below
      if (e.gf(c0.w)) {
        e.fork();
        void* ws_ = e.wstream;
        const int slot = e.begin_timed(e.conv_flops(e.stem_desc(true)), 2, ws_, e.wgrad_bytes(c0, e.stem_desc(true)));
        const int st = qt_stem_bn_bwd_wgrad(dt, e.at(p->g_p0), e.at<unsigned char>(p->argmax), e.at(c0.y),
                                            e.at<float>(bn0.scale), e.at<float>(bn0.shift), e.at<float>(bn0.mean),
                                            e.at<float>(bn0.invstd), e.at<float>(bn0.coef), e.at(p->xpad),
                                            e.at<float>(c0.dw), B, ws_);
        e.end_timed(slot, ws_);
        if (st == QT_OK) {
          e.run(qt_unpack_stem_wgrad(e.at<float>(c0.dw), e.gf(c0.w), 0, ws_));
          stem_done = true;
        } else if (st != QT_ERR_UNSUPPORTED) {
          e.run(st);
        }
      }
      if (!stem_done)
        e.run(qt_stem_bn_bwd_apply(dt, e.at(p->g_p0), e.at<unsigned char>(p->argmax), e.at(c0.y), e.at<float>(bn0.scale),
                                   e.at<float>(bn0.shift), e.at<float>(bn0.mean), e.at<float>(bn0.invstd),
                                   e.at<float>(bn0.coef), e.at(c0.gy), B, stream));
    } else {
      e.run(qt_stem_pool_bwd(dt, e.at(p->g_p0), e.at<unsigned char>(p->argmax), e.at(c0.y), e.at<float>(bn0.scale),
                             e.at<float>(bn0.shift), e.at(c0.gy), B, stream));
      e.bn_backward(c0, e.at(c0.gy), nullptr);
    }
    if (!stem_done) e.wgrad(c0, e.stem_desc(true), e.at(p->xpad), true);
    }
  }
  // the last phase (or a head-only model) joins: afterwards the caller's stream sees every gradient
  if ((phases & QT_BWD_LAYER1) || !backbone_grads) {
    e.forked = e.forked || (p->side != nullptr && e.wstream == p->side);
    e.join();
  }
  return e.status;
}

}  // namespace

// ---------------------------------------------------------------------------------
extern "C" int qt_plan_create(const qt_plan_desc* desc, qt_plan** out) {
  QT_CHECK_ARG(desc && out, "qt_plan_create: null argument");
  QT_CHECK_ARG(desc->dtype == QT_F32 || desc->dtype == QT_BF16, "qt_plan_create: bad dtype %d", desc->dtype);
  QT_CHECK_ARG(desc->batch > 0 && desc->batch <= 4096, "qt_plan_create: batch %d out of range", desc->batch);
  QT_CHECK_ARG(desc->num_classes > 0 && desc->num_classes <= 4096, "qt_plan_create: bad num_classes");
  QT_CHECK_ARG(desc->model == QT_MODEL_QUADTREE || desc->model == QT_MODEL_STANDARD_RESNET ||
                   desc->model == QT_MODEL_ATTENTION || desc->model == QT_MODEL_CNN_LSTM, "qt_plan_create: bad model");
  QT_CHECK_ARG(desc->model != QT_MODEL_CNN_LSTM ||
                   (desc->seq_len > 0 && desc->batch % desc->seq_len == 0 && (desc->lstm_hidden == 256 || desc->lstm_hidden == 64)),
               "qt_plan_create: CnnLstm needs seq_len > 0 dividing batch (frames) and lstm_hidden 256 or 64 (got %d, %d)",
               desc->seq_len, desc->lstm_hidden);
  QT_CHECK_ARG(desc->model != QT_MODEL_QUADTREE ||
                   (desc->mode >= QT_MODE_FUSION && desc->mode <= QT_MODE_NUMERICAL_ONLY),
               "qt_plan_create: bad mode %d", desc->mode);
  QT_CHECK_ARG(desc->numerical_dim > 0 && desc->numerical_dim <= 1024, "qt_plan_create: bad numerical_dim");
  QT_CHECK_ARG(desc->dropout_p >= 0.f && desc->dropout_p < 1.f, "qt_plan_create: bad dropout_p");
  qt_plan* p = new qt_plan();
  p->d = *desc;
  p->esz = desc->dtype == QT_F32 ? 4 : 2;
  if (const char* v = getenv("QTCNN_SIDE_STREAM")) p->use_side = atoi(v) != 0;
  build_graph(p);
  layout_workspace(p);
  *out = p;
  return QT_OK;
}

extern "C" void qt_plan_destroy(qt_plan* p) {
  if (!p) return;
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  delete p;
}
extern "C" int qt_plan_num_tensors(const qt_plan* p) { return p ? (int)p->tensors.size() : QT_ERR_INVALID_ARG; }
extern "C" const char* qt_plan_tensor_name(const qt_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->tensors.size()) ? p->tensors[i].name.c_str() : nullptr;
}
extern "C" int qt_plan_tensor_kind(const qt_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->tensors.size()) ? p->tensors[i].kind : QT_ERR_INVALID_ARG;
}
extern "C" int qt_plan_tensor_shape(const qt_plan* p, int i, int* dims) {
  if (!p || !dims || i < 0 || i >= (int)p->tensors.size()) return QT_ERR_INVALID_ARG;
  for (int k = 0; k < 4; ++k) dims[k] = p->tensors[i].shape[k];
  return p->tensors[i].ndim;
}
extern "C" size_t qt_plan_workspace_bytes(const qt_plan* p) { return p ? p->ws_bytes : 0; }

// Named views into the workspace (tests / Grad-CAM style introspection): the offset of
// an activation or gradient buffer for the plan's maximum batch.
extern "C" int qt_plan_find_buffer(const qt_plan* p, const char* name, size_t* offset) {
  QT_CHECK_ARG(p && name && offset, "qt_plan_find_buffer: null argument");
  const std::string n(name);
  auto blockno = [&](const char* prefix) -> int {
    const size_t len = strlen(prefix);
    if (n.compare(0, len, prefix) != 0) return -1;
    return atoi(n.c_str() + len);
  };
  auto suffix = [&](const char* suf) {
    const size_t len = strlen(suf);
    return n.size() >= len && n.compare(n.size() - len, len, suf) == 0;
  };
  if (n == "fused") { *offset = p->fused; return QT_OK; }
  if (n == "dfused") { *offset = p->dfused; return QT_OK; }
  if (n == "hidden") { *offset = p->hidden; return QT_OK; }
  if (p->attention && n == "attention.vectors") { *offset = p->vsub; return QT_OK; }   // f32 [B][16][64]
  if (p->attention && n == "attention.weights") { *offset = p->att_alpha; return QT_OK; }  // f32 [B][16]
  if (n == "stem.pooled") { *offset = p->p0; return QT_OK; }
  if (n == "stem.gpooled") { *offset = p->g_p0; return QT_OK; }
  int b = blockno("block");
  if (b >= 0 && b < (int)p->blocks.size()) {
    const Block& blk = p->blocks[b];
    if (suffix(".out")) { *offset = blk.out; return QT_OK; }
    if (suffix(".gout")) { *offset = blk.gout; return QT_OK; }
    if (suffix(".a1")) { *offset = blk.a1; return QT_OK; }
  }
  int c = blockno("conv");
  if (c >= 0 && c < (int)p->convs.size()) {
    if (suffix(".y")) { *offset = p->convs[c].y; return QT_OK; }
    if (suffix(".gy")) { *offset = p->convs[c].gy; return QT_OK; }
  }
  qt_set_error("qt_plan_find_buffer: unknown buffer '%s'", name);
  return QT_ERR_INVALID_ARG;
}

// ---- per-launch MFMA kernel timing (kinds: 0 igemm forward, 1 igemm dgrad, 2 wgrad) ----
extern "C" int qt_plan_profile_begin(qt_plan* p) {
  QT_CHECK_ARG(p, "qt_plan_profile_begin: null plan");
  for (auto& t : p->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  p->timed.clear();
  p->profiling = true;
  return QT_OK;
}
// Stops profiling, waits for the recorded events and sums per kind:
// flops[3], ms[3], launches[3].
extern "C" int qt_plan_profile_end(qt_plan* p, double* flops, double* ms, int* launches) {
  QT_CHECK_ARG(p && flops && ms && launches, "qt_plan_profile_end: null argument");
  p->profiling = false;
  for (int k = 0; k < 3; ++k) { flops[k] = 0; ms[k] = 0; launches[k] = 0; p->last_profile_bytes[k] = 0; }
  int st = QT_OK;
  for (auto& t : p->timed) {
    float dt = 0.f;
    if (hipEventSynchronize(t.b) != hipSuccess || hipEventElapsedTime(&dt, t.a, t.b) != hipSuccess) {
      qt_set_error("qt_plan_profile_end: event query failed");
      st = QT_ERR_LAUNCH;
    } else if (t.kind >= 0 && t.kind < 3) {
      flops[t.kind] += t.flops; ms[t.kind] += dt; launches[t.kind] += 1;
      p->last_profile_bytes[t.kind] += t.bytes;
    }
    (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
  }
  p->timed.clear();
  return st;
}

// algorithmic HBM bytes (each operand of a launch once) summed per kind over the launches of the last qt_plan_profile_end
extern "C" int qt_plan_profile_bytes(const qt_plan* p, double* bytes3) {
  QT_CHECK_ARG(p && bytes3, "qt_plan_profile_bytes: null argument");
  for (int k = 0; k < 3; ++k) bytes3[k] = p->last_profile_bytes[k];
  return QT_OK;
}

// Make `waiting_stream` wait for everything enqueued so far on the plan's side stream (the
// weight gradients of the phases already issued).  No-op when the side stream is not in use.
extern "C" int qt_plan_side_fence(qt_plan* p, void* waiting_stream) {
  QT_CHECK_ARG(p, "qt_plan_side_fence: null plan");
  if (!p->side || !p->use_side) return QT_OK;
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
      hipEventRecord(ev, p->side) != hipSuccess ||
      hipStreamWaitEvent(static_cast<hipStream_t>(waiting_stream), ev, 0) != hipSuccess) {
    qt_set_error("qt_plan_side_fence: HIP error");
    return QT_ERR_LAUNCH;
  }
  (void)hipEventDestroy(ev);  // destruction is deferred until the event has completed
  return QT_OK;
}

extern "C" int qt_plan_init_workspace(qt_plan* p, void* workspace, void* stream) {
  QT_CHECK_ARG(p && workspace, "qt_plan_init_workspace: null argument");
  // ones / zeros vectors used as identity BatchNorm affine in eval mode
  std::vector<float> one(2048, 1.f);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e1 = hipMemcpyAsync(static_cast<unsigned char*>(workspace) + p->ones, one.data(), 2048 * 4,
                                 hipMemcpyHostToDevice, s);
  hipError_t e2 = hipMemsetAsync(static_cast<unsigned char*>(workspace) + p->zeros, 0, 2048 * 4, s);
  // merged stride-2 data-gradient operands: the tap slots no filter tap maps to are zero for good (the packers write
  // the nine real taps only)
  for (const ConvL& c : p->convs)
    if (c.merged_dgrad && e2 == hipSuccess)
      e2 = hipMemsetAsync(static_cast<unsigned char*>(workspace) + c.w_dgrad, 0, (size_t)(c.merged5 ? 20 : 16) * c.cout * c.cin * p->esz, s);
  hipError_t e3 = hipStreamSynchronize(s);  // `one` is a host temporary
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    qt_set_error("qt_plan_init_workspace: HIP error");
    return QT_ERR_LAUNCH;
  }
  return QT_OK;
}

extern "C" int qt_plan_pack_weights(qt_plan* p, void* workspace, void* const* tensors, int for_backward, void* stream) {
  QT_CHECK_ARG(p && workspace && tensors, "qt_plan_pack_weights: null argument");
  return pack_weights(p, workspace, tensors, for_backward, stream);
}

extern "C" int qt_plan_adam_step(qt_plan* p, void* workspace, void* const* tensors, float* const* grads,
                                 float* const* exp_avg, float* const* exp_avg_sq, const qt_adam_desc* adam, int for_backward,
                                 void* stream) {
  QT_CHECK_ARG(p && workspace && tensors && grads && exp_avg && exp_avg_sq && adam, "qt_plan_adam_step: null argument");
  return adam_step(p, workspace, tensors, grads, exp_avg, exp_avg_sq, adam, for_backward, stream);
}

extern "C" int qt_plan_forward(qt_plan* p, void* workspace, void* const* tensors, const float* image,
                               const float* numerical, float* logits, int batch, int training,
                               unsigned long long seed, void* stream) {
  QT_CHECK_ARG(p && workspace && tensors && logits, "qt_plan_forward: null argument");
  QT_CHECK_ARG(batch > 0 && batch <= p->d.batch, "qt_plan_forward: batch %d exceeds the plan's %d", batch, p->d.batch);
  QT_CHECK_ARG(!p->lstm || batch % p->seq_len == 0, "qt_plan_forward: %d frames are not whole sequences of %d", batch, p->seq_len);
  QT_CHECK_ARG(!p->has_image || image, "qt_plan_forward: image required");
  QT_CHECK_ARG(!p->has_numerical || numerical, "qt_plan_forward: numerical input required");
  QT_CHECK_ARG(((uintptr_t)workspace % 256) == 0, "qt_plan_forward: workspace must be 256-byte aligned");
  return forward(p, workspace, tensors, image, numerical, logits, batch, training, seed, stream);
}

extern "C" int qt_plan_backward(qt_plan* p, void* workspace, void* const* tensors, float* const* grads,
                                const float* numerical, const float* dlogits, int phases, void* stream) {
  QT_CHECK_ARG(p && workspace && tensors && grads && dlogits, "qt_plan_backward: null argument");
  QT_CHECK_ARG(p->last_batch > 0, "qt_plan_backward: no forward pass recorded");
  QT_CHECK_ARG(!p->has_numerical || numerical, "qt_plan_backward: numerical input required");
  return backward(p, workspace, tensors, grads, numerical, dlogits, phases, stream);
}
