// net.hip -- host side of the CAM networks: state_dict -> packed device weights, and the
// per-batch launch sequence.
//
// Reference being replaced:
//   resnet50_cam : 03b_irn/net/resnet50.py:17-108 assembled by resnet50_cam.py:12-20 with
//                  strides=(2,2,2,1); CAM.forward resnet50_cam.py:55-70
//   vgg16_cam    : net/vgg16.py:44 cfg + common_cnn.py:128-141 make_layers
//                  (conv(bias) -> ReLU -> BatchNorm(eps=1e-3)); vgg16_cam.py:24-50
//   m7_cam       : net/m7.py:41 cfg; m7_cam.py:22-47 (Grad-CAM weights as the 1x1 head)
//   construction/load: 03b_irn/step/make_cam.py:96-100
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>

namespace {

struct HostTensor {
    const float *data;
    int ndim;
    int64_t shape[4];
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};
typedef std::unordered_map<std::string, HostTensor> Dict;

struct ConvW {
    bf16_t *w = nullptr;
    float *s1 = nullptr, *b1 = nullptr, *s2 = nullptr, *b2 = nullptr;
    int Cin = 0, Cout = 0, CoutPad = 0, kh = 1, kw = 1, stride = 1, pad = 0, relu = 0, small_cin = 0;
};

// one Conv2d(1x1, bias=False) -> GroupNorm -> [Upsample] -> ReLU head of the IRNet branches
struct IrnHead {
    ConvW conv;
    int groups = 1, up = 1;
    float *gamma = nullptr, *beta = nullptr; // device, [Cout]
    int src = 0;                             // stage k (1-based) when > 0, concat buffer -src - 1 when < 0
    int dst = 0, coff = 0;                   // destination concat buffer and channel offset inside it
};
// a concat buffer: NHWC with `channels` channels at the resolution of stage `stage` divided by `stride`
struct IrnCat {
    int channels, stage, stride;
};

enum OpType { OP_CONV = 0, OP_POOL = 1, OP_GATHER = 2 };
struct Op {
    int type;
    int conv;         // index into convs (OP_CONV)
    int in, out, res; // activation buffer ids; res = -1 if none; in = -1 means the NHWC4 input
    int pk, ps, pp;   // pool kernel / stride / pad (OP_GATHER: ps = pixel stride)
    int pitch = 0;    // channels per output row when the op writes a channel range of a wider (concatenated) tensor, else 0
    int coff = 0;     // first channel of that range
    int in2 = -1;     // OP_CONV, 1x1: buffer whose pixels (ho * ps, wo * ps) supply the LAST input channels (ConvLaunch::x2), or -1
};

} // namespace

struct wsc_net {
    wsc_ctx *ctx = nullptr;
    int arch = 0, C = 0, split = 0, fmt = 0, F = 0;
    std::vector<ConvW> convs;
    std::vector<Op> ops;
    int final_buf = 0;
    ConvW head;
    float *cls_w = nullptr, *cls_b = nullptr; // classifier branch (vgg16 / m7), fp32 [Ccls][F]
    int Ccls = 0;
    int cls_max = 0; // 1: global max pooling (m7), 0: global average (vgg16)
    // IRNet EdgeDisplacement (arch >= WSC_ARCH_RESNET50_IRN): backbone stage taps + the two head branches
    std::vector<int> taps;       // op index whose output is stage k+1 (x1..x5)
    std::vector<IrnHead> heads;  // every Conv-GroupNorm-[Upsample]-ReLU head in execution order
    std::vector<IrnCat> cats;    // concat buffers; cats[0] feeds the final edge conv, cats.back() the final dp conv
    ConvW edge6, dp7b;           // final edge conv (bias) / final displacement conv
    float mean_shift[2] = {0.f, 0.f};
    std::vector<void *> allocs;
};

namespace {

int upload(wsc_net *net, const void *host, size_t bytes, void **out) {
    void *d = nullptr;
    WSC_HIP(hipMalloc(&d, bytes));
    net->allocs.push_back(d);
    WSC_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    *out = d;
    return WSC_OK;
}

int get(const Dict &d, const std::string &k, int ndim, const HostTensor **out) {
    auto it = d.find(k);
    WSC_CHECK(it != d.end(), WSC_ERR_MISSING_KEY, "state_dict: missing key '%s'", k.c_str());
    WSC_CHECK(it->second.ndim == ndim, WSC_ERR_SHAPE, "state_dict: '%s' has ndim %d, expected %d", k.c_str(),
              it->second.ndim, ndim);
    WSC_CHECK(it->second.data != nullptr, WSC_ERR_INVALID, "state_dict: '%s' has a null data pointer", k.c_str());
    *out = &it->second;
    return WSC_OK;
}
bool has(const Dict &d, const std::string &k) { return d.find(k) != d.end(); }

// Fold inference batch-norm y = (x - mean) / sqrt(var + eps) * gamma + beta into scale/shift.
int fold_bn(const Dict &d, const std::string &bn, int C, double eps_default, std::vector<float> &s,
            std::vector<float> &b) {
    const HostTensor *g, *be, *mu, *var;
    WSC_TRY(get(d, bn + ".weight", 1, &g));
    WSC_TRY(get(d, bn + ".bias", 1, &be));
    WSC_TRY(get(d, bn + ".running_mean", 1, &mu));
    WSC_TRY(get(d, bn + ".running_var", 1, &var));
    WSC_CHECK(g->shape[0] == C && be->shape[0] == C && mu->shape[0] == C && var->shape[0] == C, WSC_ERR_SHAPE,
              "state_dict: batch-norm '%s' is not %d channels", bn.c_str(), C);
    double eps = eps_default;
    auto it = d.find(bn + ".eps");
    if (it != d.end() && it->second.data) eps = it->second.data[0];
    s.resize(C);
    b.resize(C);
    for (int c = 0; c < C; ++c) {
        const double sc = (double)g->data[c] / std::sqrt((double)var->data[c] + eps);
        s[c] = (float)sc;
        b[c] = (float)((double)be->data[c] - (double)mu->data[c] * sc);
    }
    return WSC_OK;
}

// Pack OIHW fp32 weights to [CoutPad][Kw] bf16 in the kernel's K order: generic layers
// (cin / 64, kh, kw, cin % 64) -- the kh*kw taps of one 64-channel chunk are consecutive K-steps, so the
// activation lines a block gathers are re-touched within a few K-steps (L2 hits) instead of Cin/64 steps later.
int make_conv(wsc_net *net, const HostTensor *w, int stride, int pad, int relu, int small_cin,
              const std::vector<float> &s1, const std::vector<float> &b1, const std::vector<float> *s2,
              const std::vector<float> *b2, ConvW *out) {
    const int Cout = (int)w->shape[0], Cin = (int)w->shape[1], kh = (int)w->shape[2], kw = (int)w->shape[3];
    ConvW c;
    c.Cout = Cout;
    c.CoutPad = Cout <= 64 ? 64 : ((Cout + 127) / 128) * 128;
    c.kh = kh; c.kw = kw; c.stride = stride; c.pad = pad; c.relu = relu; c.small_cin = small_cin;
    int Kbase;
    if (small_cin == 0) {
        WSC_CHECK(Cin % 64 == 0, WSC_ERR_SHAPE, "conv with Cin=%d is not supported (need a multiple of 64)", Cin);
        c.Cin = Cin;
        Kbase = kh * kw * Cin;
    } else if (small_cin == 3) { // f16x3 stem on the padded input: one K-step (32 hi + 32 lo) per kernel row
        WSC_CHECK(Cin <= 4 && kw <= 7 && net->split == 2, WSC_ERR_SHAPE, "padded-stem conv needs Cin <= 4, kw <= 7, f16x3");
        c.Cin = 4;
        Kbase = kh * 32;
    } else {
        WSC_CHECK(Cin <= 4, WSC_ERR_SHAPE, "small-Cin conv needs Cin <= 4, got %d", Cin);
        WSC_CHECK(kw <= (2 << small_cin), WSC_ERR_SHAPE, "small-Cin conv: kw=%d too wide", kw);
        c.Cin = 4;
        Kbase = (((kh << small_cin) + 7) / 8) * 64;
    }
    const int planes = net->split ? 2 : 1;
    const int Kw = Kbase * planes;
    // f16x3 (split 2), generic layers: K order (cin / 32, kh, kw) and per K-step 32 hi values followed by their 32 lo values
    // (conv_igemm.hip, SPLIT 2); everything else: [hi K | lo K]
    const bool interleaved = net->split == 2 && (small_cin == 0 || small_cin == 3);
    std::vector<bf16_t> wp((size_t)c.CoutPad * Kw, 0);
    // IEEE-half modes (f16, f16x3): every output channel's weights are stored times a power of two that puts the channel's
    // largest |w| into [2^12, 2^13), and the epilogue scale s1 takes the inverse -- exact in fp32, the accumulators are fp32.
    // Half has 5 exponent bits: a checkpoint's late-layer weights (|w| ~ 1e-4 ... 1e-3, below half's smallest normal
    // 6.1e-5 for many of them) would lose their `hi` bits and all of `lo` (the split's lo = half(w - hi) is ~2^-11 |w|);
    // scaled, every weight within 2^-15 of its channel's maximum keeps the full 22-bit hi + lo pair and smaller ones an
    // absolute error of 2^-25 against a maximum of >= 2^12 (bfloat16 has fp32's exponent range).
    std::vector<float> s1v(s1);
    std::vector<float> wscale(Cout, 1.f);
    if (net->fmt == 1) {
        const size_t per = (size_t)Cin * kh * kw;
        for (int co = 0; co < Cout; ++co) {
            float mx = 0.f;
            for (size_t i = 0; i < per; ++i) mx = std::max(mx, std::fabs(w->data[(size_t)co * per + i]));
            if (!(mx > 0.f) || !std::isfinite(mx)) continue;
            int e;
            std::frexp(mx, &e); // mx = m * 2^e, m in [0.5, 1)
            // (bounded shift: a dead channel in fp32's denormal range -- weight decay leaves |w| ~ 1e-40 -- would otherwise
            // ask for 2^(13 - e) = inf, every weight * inf = inf, lo = inf - inf = NaN; with the bound such a channel keeps
            // its tiny values as half subnormals / zeros like before the packing)
            const int sh = std::max(std::min(13 - e, 100), -100);
            wscale[co] = std::ldexp(1.f, sh);
            s1v[co] = s1[co] * std::ldexp(1.f, -sh);
        }
    }
    for (int co = 0; co < Cout; ++co) {
        bf16_t *row = wp.data() + (size_t)co * Kw;
        const float wsc = wscale[co];
        auto put = [&](int k, float v) {
            v *= wsc;
            const bf16_t h = f32_to_h16(v, net->fmt);
            const int lo_at = interleaved ? 32 : Kbase;
            row[k] = h;
            if (net->split) row[lo_at + k] = f32_to_h16(v - h16_to_f32(h, net->fmt), net->fmt);
        };
        for (int ci = 0; ci < Cin; ++ci)
            for (int r = 0; r < kh; ++r)
                for (int s = 0; s < kw; ++s) {
                    const float v = w->data[(((size_t)co * Cin + ci) * kh + r) * kw + s];
                    if (small_cin == 3) {
                        put(r * 64 + s * 4 + ci, v); // kernel row r: 8 pixels x 4 channels (pixel 7, channel 3: zero weights)
                    } else if (interleaved) {
                        put((((ci >> 5) * kh + r) * kw + s) * 64 + (ci & 31), v);
                    } else if (small_cin == 0) {
                        put((((ci >> 6) * kh + r) * kw + s) * 64 + (ci & 63), v);
                    } else {
                        // kernel row r owns 2^small_cin slots of 8 = (2 pixels x 4 channels)
                        put((r << small_cin) * 8 + s * 4 + ci, v);
                    }
                }
    }
    WSC_TRY(upload(net, wp.data(), wp.size() * sizeof(bf16_t), (void **)&c.w));
    auto up_vec = [&](const std::vector<float> &v, float **dst) -> int {
        std::vector<float> p(c.CoutPad, 0.f);
        for (int i = 0; i < Cout; ++i) p[i] = v[i];
        return upload(net, p.data(), p.size() * sizeof(float), (void **)dst);
    };
    WSC_TRY(up_vec(s1v, &c.s1));
    WSC_TRY(up_vec(b1, &c.b1));
    if (s2) {
        WSC_TRY(up_vec(*s2, &c.s2));
        WSC_TRY(up_vec(*b2, &c.b2));
    }
    *out = c;
    return WSC_OK;
}

int add_conv_op(wsc_net *net, const ConvW &c, int in, int out, int res) {
    net->convs.push_back(c);
    Op op;
    op.type = OP_CONV; op.conv = (int)net->convs.size() - 1; op.in = in; op.out = out; op.res = res;
    op.pk = op.ps = op.pp = 0;
    net->ops.push_back(op);
    return WSC_OK;
}
void add_pool_op(wsc_net *net, int k, int s, int p, int in, int out) {
    Op op;
    op.type = OP_POOL; op.conv = -1; op.in = in; op.out = out; op.res = -1; op.pk = k; op.ps = s; op.pp = p;
    net->ops.push_back(op);
}

// ResNet conv (bias-free) + FixedBatchNorm.
int resnet_conv(wsc_net *net, const Dict &d, const std::string &conv, const std::string &bn, int stride, int pad,
                int relu, int small_cin, int in, int out, int res) {
    const HostTensor *w;
    WSC_TRY(get(d, conv + ".weight", 4, &w));
    std::vector<float> s, b;
    WSC_TRY(fold_bn(d, bn, (int)w->shape[0], 1e-5, s, b));
    ConvW c;
    WSC_TRY(make_conv(net, w, stride, pad, relu, small_cin, s, b, nullptr, nullptr, &c));
    return add_conv_op(net, c, in, out, res);
}

int build_resnet50_backbone(wsc_net *net, const Dict &d) {
    // stem: conv1 7x7 s2 p3 + bn1 + relu, maxpool 3x3 s2 p1          (resnet50.py:62-64, 96-99)
    // (f16x3: the padded-input form, staged like every other layer of that mode)
    WSC_TRY(resnet_conv(net, d, "resnet50.conv1", "resnet50.bn1", 2, 3, 1, /*small_cin*/ net->split == 2 ? 3 : 2, -1, 0, -1));
    add_pool_op(net, 3, 2, 1, 0, 1);
    net->taps.push_back((int)net->ops.size() - 1); // stage1 = conv1, bn1, relu, maxpool (resnet50_irn.py:15)
    int cur = 1;
    bool fuse_shortcut = true;
#ifdef WSC_AB_KNOBS
    if (const char *e = getenv("WSC_NET_SHORTCUT_FUSED")) fuse_shortcut = atoi(e) != 0; // A/B: the four-launch form of a stage's first block
#endif
    const int planes[4] = {64, 128, 256, 512};
    const int blocks[4] = {3, 4, 6, 3};
    const int strides[4] = {1, 2, 2, 1}; // resnet50_cam.py:15 strides=(2,2,2,1): [0] is the stem
    for (int L = 0; L < 4; ++L) {
        (void)planes;
        for (int bi = 0; bi < blocks[L]; ++bi) {
            const std::string pre = "resnet50.layer" + std::to_string(L + 1) + "." + std::to_string(bi);
            const int s = bi == 0 ? strides[L] : 1;
            int f[3], nf = 0;
            for (int i = 0; i < 4 && nf < 3; ++i)
                if (i != cur) f[nf++] = i;
            // Bottleneck.forward, resnet50.py:34-54; the stride sits on conv2 (resnet50.py:24)
            WSC_TRY(resnet_conv(net, d, pre + ".conv1", pre + ".bn1", 1, 0, 1, 0, cur, f[0], -1));
            bool fuse_here = false;
            if (has(d, pre + ".downsample.0.weight") && has(d, pre + ".conv3.weight") && fuse_shortcut) {
                // (only for the shapes the concatenated GEMM takes: 1x1 kernels, K1 + K2 a multiple of the 64-channel K chunk;
                // anything else keeps the separate projection conv + residual)
                const HostTensor *w3, *wd;
                WSC_TRY(get(d, pre + ".conv3.weight", 4, &w3));
                WSC_TRY(get(d, pre + ".downsample.0.weight", 4, &wd));
                fuse_here = wd->shape[0] == w3->shape[0] && w3->shape[2] == 1 && w3->shape[3] == 1 && wd->shape[2] == 1 &&
                            wd->shape[3] == 1 && w3->shape[1] % 8 == 0 && wd->shape[1] % 8 == 0 &&
                            (w3->shape[1] + wd->shape[1]) % 64 == 0;
            }
            if (fuse_here) {
                // out = relu(bn3(conv3(y2)) + bn_d(conv_d(x)))  (resnet50.py:44-52) as ONE 1x1 conv over the concatenated channels
                // [y2 | x at the block's stride]: conv2 writes its channel range of that tensor, the shortcut input is gathered
                // beside it, and the two BatchNorm scales go into the weights -- relative to sigma_c = max(|s3_c|, |sd_c|), which
                // stays in the epilogue, so that neither branch's weights leave the half range (W3 * s3/sigma | Wd * sd/sigma,
                // shift b3 + bd).  The projection's output (215 MB per plane in layer1 at 64 samples) is never written or re-read
                // as a residual, and the stage's first block has one epilogue instead of two.
                const HostTensor *w3, *wd;
                WSC_TRY(get(d, pre + ".conv3.weight", 4, &w3));
                WSC_TRY(get(d, pre + ".downsample.0.weight", 4, &wd));
                const int Co = (int)w3->shape[0], K1 = (int)w3->shape[1], K2 = (int)wd->shape[1];
                std::vector<float> s3, b3, sd, bd;
                WSC_TRY(fold_bn(d, pre + ".bn3", Co, 1e-5, s3, b3));
                WSC_TRY(fold_bn(d, pre + ".downsample.1", Co, 1e-5, sd, bd));
                std::vector<float> wc((size_t)Co * (K1 + K2)), sig(Co), sh(Co);
                for (int co = 0; co < Co; ++co) {
                    float g = std::max(std::fabs(s3[co]), std::fabs(sd[co]));
                    if (!(g > 0.f)) g = 1.f;
                    sig[co] = g;
                    sh[co] = b3[co] + bd[co];
                    for (int k = 0; k < K1; ++k) wc[(size_t)co * (K1 + K2) + k] = w3->data[(size_t)co * K1 + k] * (s3[co] / g);
                    for (int k = 0; k < K2; ++k) wc[(size_t)co * (K1 + K2) + K1 + k] = wd->data[(size_t)co * K2 + k] * (sd[co] / g);
                }
                HostTensor wt;
                wt.data = wc.data(); wt.ndim = 4; wt.shape[0] = Co; wt.shape[1] = K1 + K2; wt.shape[2] = 1; wt.shape[3] = 1;
                WSC_TRY(resnet_conv(net, d, pre + ".conv2", pre + ".bn2", s, 1, 1, 0, f[0], f[1], -1));
                ConvW c;
                WSC_TRY(make_conv(net, &wt, 1, 0, 1, 0, sig, sh, nullptr, nullptr, &c));
                if (net->split != 1) {
                    // the kernel reads the two inputs where they are: channel chunks [0, K1) from conv2's output, the rest from
                    // the block input at the block's stride (conv_igemm.hip, second A source)
                    WSC_TRY(add_conv_op(net, c, f[1], f[0], -1));
                    net->ops.back().in2 = cur;
                    net->ops.back().ps = s;
                } else {
                    // bf16x3 (three K segments per source): the concatenated tensor is materialised -- conv2 writes its channel
                    // range, the shortcut input is gathered beside it
                    net->ops.back().pitch = K1 + K2;
                    net->ops.back().coff = 0;
                    Op g;
                    g.type = OP_GATHER; g.conv = -1; g.in = cur; g.out = f[1]; g.res = -1; g.pk = 1; g.ps = s; g.pp = 0;
                    g.pitch = K1 + K2; g.coff = K1;
                    net->ops.push_back(g);
                    WSC_TRY(add_conv_op(net, c, f[1], f[0], -1));
                }
                cur = f[0];
                continue;
            }
            WSC_TRY(resnet_conv(net, d, pre + ".conv2", pre + ".bn2", s, 1, 1, 0, f[0], f[1], -1));
            int res = cur;
            if (has(d, pre + ".downsample.0.weight")) {
                WSC_TRY(resnet_conv(net, d, pre + ".downsample.0", pre + ".downsample.1", s, 0, 0, 0, cur, f[2], -1));
                res = f[2];
            }
            WSC_TRY(resnet_conv(net, d, pre + ".conv3", pre + ".bn3", 1, 0, 1, 0, f[1], f[0], res));
            cur = f[0];
        }
        net->taps.push_back((int)net->ops.size() - 1); // stage L+2 = layer L+1
    }
    net->final_buf = cur;
    return WSC_OK;
}

int build_resnet50(wsc_net *net, const Dict &d) {
    WSC_TRY(build_resnet50_backbone(net, d));
    // CAM head: F.conv2d(x, classifier.weight), resnet50_cam.py:65 (ReLU + flip-add are a separate kernel)
    const HostTensor *cw;
    WSC_TRY(get(d, "classifier.weight", 4, &cw));
    WSC_CHECK(cw->shape[0] == net->C && cw->shape[2] == 1 && cw->shape[3] == 1, WSC_ERR_SHAPE,
              "classifier.weight must be [%d][F][1][1]", net->C);
    net->F = (int)cw->shape[1];
    std::vector<float> one(net->C, 1.f), zero(net->C, 0.f);
    WSC_TRY(make_conv(net, cw, 1, 0, 0, 0, one, zero, nullptr, nullptr, &net->head));
    return WSC_OK;
}

// VGG-style stacks of common_cnn.make_layers: cfg entries >0 = conv out channels, -1 = 'M', -2 = 'D'.
int build_plain_stack(wsc_net *net, const Dict &d, const std::string &root,
                      const std::vector<std::pair<std::string, std::vector<int>>> &cfg, int *cur_io,
                      int *feat_channels, std::vector<int> *layer_taps = nullptr) {
    int cur = *cur_io;
    int in_ch = 3;
    bool first = true;
    for (const auto &layer : cfg) {
        int idx = 0;
        for (int v : layer.second) {
            if (v == -1) { // nn.MaxPool2d(2, 2), common_cnn.py:131-132
                const int out = (cur + 1) & 1;
                add_pool_op(net, 2, 2, 0, cur, out);
                cur = out;
                idx += 1;
            } else if (v == -2) { // nn.Dropout: identity in eval()
                idx += 1;
            } else {
                const std::string key = root + "." + layer.first + "." + std::to_string(idx);
                const HostTensor *w, *bias;
                WSC_TRY(get(d, key + ".weight", 4, &w));
                WSC_TRY(get(d, key + ".bias", 1, &bias));
                WSC_CHECK(w->shape[0] == v && w->shape[1] == in_ch && w->shape[2] == 3 && w->shape[3] == 3,
                          WSC_ERR_SHAPE, "'%s.weight' must be [%d][%d][3][3]", key.c_str(), v, in_ch);
                std::vector<float> one(v, 1.f), bb(bias->data, bias->data + v);
                const std::string bn = root + "." + layer.first + "." + std::to_string(idx + 2);
                ConvW c;
                if (has(d, bn + ".running_mean")) { // conv -> ReLU -> BatchNorm(eps=1e-3): common_cnn.py:138
                    std::vector<float> s2, b2;
                    WSC_TRY(fold_bn(d, bn, v, 1e-3, s2, b2));
                    WSC_TRY(make_conv(net, w, 1, 1, 1, first ? 1 : 0, one, bb, &s2, &b2, &c));
                    idx += 3;
                } else {
                    WSC_TRY(make_conv(net, w, 1, 1, 1, first ? 1 : 0, one, bb, nullptr, nullptr, &c));
                    idx += 2;
                }
                const int in = first ? -1 : cur;
                const int out = first ? 0 : ((cur + 1) & 1);
                WSC_TRY(add_conv_op(net, c, in, out, -1));
                cur = out;
                in_ch = v;
                first = false;
            }
        }
        if (layer_taps) layer_taps->push_back((int)net->ops.size() - 1); // output of this layer
    }
    *cur_io = cur;
    *feat_channels = in_ch;
    return WSC_OK;
}

// optional per-class bias of a Grad-CAM head (`gradcam_bias`, C values): the Keras drivers contract alpha with the
// PRE-BatchNorm activation; on the post-BN feature map that is a head with weights alpha / s and this bias
// (wsscam/net/common.py::pre_bn_head).
int gradcam_bias(const Dict &d, int C, std::vector<float> &bias) {
    bias.assign(C, 0.f);
    if (!has(d, "gradcam_bias")) return WSC_OK;
    const HostTensor *gb;
    WSC_TRY(get(d, "gradcam_bias", 1, &gb));
    WSC_CHECK(gb->shape[0] == C, WSC_ERR_SHAPE, "gradcam_bias must be [%d]", C);
    for (int c = 0; c < C; ++c) bias[c] = gb->data[c];
    return WSC_OK;
}

int build_vgg16(wsc_net *net, const Dict &d) {
    const std::vector<std::pair<std::string, std::vector<int>>> cfg = {
        {"layer1", {64, 64, -1}},
        {"layer2", {128, 128, -1}},
        {"layer3", {256, 256, 256, -1}},
        {"layer4", {512, 512, 512, 512, 512, 512}},
        {"layer5", {1024, -2, 1024, -2}}}; // vgg16.py:44
    int cur = 0;
    WSC_TRY(build_plain_stack(net, d, "vgg16", cfg, &cur, &net->F));
    net->final_buf = cur;
    const HostTensor *lw;
    WSC_TRY(get(d, "vgg16.classifier.0.weight", 2, &lw));
    WSC_CHECK(lw->shape[0] >= net->C && lw->shape[1] == net->F, WSC_ERR_SHAPE,
              "vgg16.classifier.0.weight must be [>=%d][%d]", net->C, net->F);
    // CAM head = the Linear weight as a 1x1 kernel: vgg16_cam.py:48 -- or, when the state dict carries
    // `gradcam_weights` (F x C), the Grad-CAM alpha of 02_cues/utilities.py:60-99 (einsum 'ijkl,lm->ijkm')
    HostTensor hw = *lw;
    hw.ndim = 4; hw.shape[0] = net->C; hw.shape[2] = 1; hw.shape[3] = 1;
    std::vector<float> one(net->C, 1.f), zero(net->C, 0.f);
    std::vector<float> wt;
    if (has(d, "gradcam_weights")) {
        const HostTensor *gw;
        WSC_TRY(get(d, "gradcam_weights", 2, &gw));
        WSC_CHECK(gw->shape[0] == net->F && gw->shape[1] == net->C, WSC_ERR_SHAPE,
                  "gradcam_weights must be [%d][%d]", net->F, net->C);
        wt.resize((size_t)net->C * net->F);
        for (int f = 0; f < net->F; ++f)
            for (int c = 0; c < net->C; ++c) wt[(size_t)c * net->F + f] = gw->data[(size_t)f * net->C + c];
        hw.data = wt.data();
        WSC_TRY(gradcam_bias(d, net->C, zero));
    }
    WSC_TRY(make_conv(net, &hw, 1, 0, 0, 0, one, zero, nullptr, nullptr, &net->head));
    net->Ccls = net->C;
    net->cls_max = 0;
    WSC_TRY(upload(net, lw->data, (size_t)net->C * net->F * sizeof(float), (void **)&net->cls_w));
    if (has(d, "vgg16.classifier.0.bias")) {
        const HostTensor *lb;
        WSC_TRY(get(d, "vgg16.classifier.0.bias", 1, &lb));
        WSC_TRY(upload(net, lb->data, (size_t)net->C * sizeof(float), (void **)&net->cls_b));
    }
    return WSC_OK;
}

int build_m7(wsc_net *net, const Dict &d) {
    const std::vector<std::pair<std::string, std::vector<int>>> cfg = {
        {"layer1", {64, 64, -1}}, {"layer2", {128, 128, -1}}, {"layer3_p1", {256, 256, 256}}}; // m7.py:41
    int cur = 0;
    WSC_TRY(build_plain_stack(net, d, "m7", cfg, &cur, &net->F));
    net->final_buf = cur;
    // Grad-CAM weights (F x C) transposed as the 1x1 head: m7_cam.py:45-46
    const HostTensor *gw;
    WSC_TRY(get(d, "gradcam_weights", 2, &gw));
    WSC_CHECK(gw->shape[0] == net->F && gw->shape[1] == net->C, WSC_ERR_SHAPE, "gradcam_weights must be [%d][%d]",
              net->F, net->C);
    std::vector<float> wt((size_t)net->C * net->F);
    for (int f = 0; f < net->F; ++f)
        for (int c = 0; c < net->C; ++c) wt[(size_t)c * net->F + f] = gw->data[(size_t)f * net->C + c];
    HostTensor hw;
    hw.data = wt.data(); hw.ndim = 4; hw.shape[0] = net->C; hw.shape[1] = net->F; hw.shape[2] = 1; hw.shape[3] = 1;
    std::vector<float> one(net->C, 1.f), zero(net->C, 0.f);
    WSC_TRY(gradcam_bias(d, net->C, zero));
    WSC_TRY(make_conv(net, &hw, 1, 0, 0, 0, one, zero, nullptr, nullptr, &net->head));
    // classifier branch: layer3_p2 (MaxPool 2x2 + Dropout) -> AdaptiveMaxPool2d(1) -> Linear + Sigmoid
    // (m7_cam.py:32-35).  max over 2x2-pooled map == global max when h, w are even.
    const HostTensor *lw;
    WSC_TRY(get(d, "m7.classifier.0.weight", 2, &lw));
    WSC_CHECK(lw->shape[0] >= net->C && lw->shape[1] == net->F, WSC_ERR_SHAPE,
              "m7.classifier.0.weight must be [>=%d][%d]", net->C, net->F);
    net->Ccls = net->C;
    net->cls_max = 1;
    WSC_TRY(upload(net, lw->data, (size_t)net->C * net->F * sizeof(float), (void **)&net->cls_w));
    if (has(d, "m7.classifier.0.bias")) {
        const HostTensor *lb;
        WSC_TRY(get(d, "m7.classifier.0.bias", 1, &lb));
        WSC_TRY(upload(net, lb->data, (size_t)net->C * sizeof(float), (void **)&net->cls_b));
    }
    return WSC_OK;
}

// ---- IRNet EdgeDisplacement heads (resnet50_irn.py:22-92, vgg16_irn.py:28-98) -----------------------------
struct HeadSpec {
    const char *name;
    int src, stride, cout, groups, up, dst, coff;
};

// Conv2d(Cin, Cout, 1, bias=False) weights `<name>.0.weight`, GroupNorm affine `<name>.1.{weight,bias}`
int add_irn_head(wsc_net *net, const Dict &d, const HeadSpec &hs, int cin_pad) {
    const HostTensor *w, *g, *b;
    const std::string nm = hs.name;
    WSC_TRY(get(d, nm + ".0.weight", 4, &w));
    WSC_TRY(get(d, nm + ".1.weight", 1, &g));
    WSC_TRY(get(d, nm + ".1.bias", 1, &b));
    WSC_CHECK(w->shape[0] == hs.cout && w->shape[2] == 1 && w->shape[3] == 1 && g->shape[0] == hs.cout &&
                  b->shape[0] == hs.cout,
              WSC_ERR_SHAPE, "'%s': expected a 1x1 conv with %d outputs + GroupNorm", hs.name, hs.cout);
    (void)cin_pad;
    IrnHead h;
    std::vector<float> one(hs.cout, 1.f), zero(hs.cout, 0.f);
    WSC_TRY(make_conv(net, w, hs.stride, 0, 0, 0, one, zero, nullptr, nullptr, &h.conv));
    h.groups = hs.groups; h.up = hs.up; h.src = hs.src; h.dst = hs.dst; h.coff = hs.coff;
    WSC_TRY(upload(net, g->data, sizeof(float) * hs.cout, (void **)&h.gamma));
    WSC_TRY(upload(net, b->data, sizeof(float) * hs.cout, (void **)&h.beta));
    net->heads.push_back(h);
    return WSC_OK;
}

int build_irn_heads(wsc_net *net, const Dict &d, const std::vector<HeadSpec> &specs, const std::vector<IrnCat> &cats,
                    const std::string &edge_final, int edge_in, const std::string &dp_final) {
    net->cats = cats;
    for (const HeadSpec &hs : specs) WSC_TRY(add_irn_head(net, d, hs, 0));
    // final edge conv = Conv2d(edge_in, 1, 1, bias=True) on the edge concat, zero-padded to cats[0].channels inputs
    const HostTensor *w6, *b6, *w7;
    WSC_TRY(get(d, edge_final + ".weight", 4, &w6));
    WSC_TRY(get(d, edge_final + ".bias", 1, &b6));
    WSC_CHECK(w6->shape[0] == 1 && w6->shape[1] == edge_in, WSC_ERR_SHAPE, "%s.weight must be [1][%d][1][1]",
              edge_final.c_str(), edge_in);
    const int cin_pad = cats[0].channels;
    std::vector<float> w6p(cin_pad, 0.f);
    for (int i = 0; i < edge_in; ++i) w6p[i] = w6->data[i];
    HostTensor t6;
    t6.data = w6p.data(); t6.ndim = 4; t6.shape[0] = 1; t6.shape[1] = cin_pad; t6.shape[2] = 1; t6.shape[3] = 1;
    std::vector<float> one1(1, 1.f), bias1(1, b6->data[0]);
    WSC_TRY(make_conv(net, &t6, 1, 0, 0, 0, one1, bias1, nullptr, nullptr, &net->edge6));
    // final displacement conv = Conv2d(256, 2, 1, bias=False); MeanShift subtracts running_mean in eval
    // (resnet50_irn.py:96-108)
    WSC_TRY(get(d, dp_final + ".weight", 4, &w7));
    WSC_CHECK(w7->shape[0] == 2 && w7->shape[1] == cats.back().channels, WSC_ERR_SHAPE, "%s.weight must be [2][%d][1][1]",
              dp_final.c_str(), cats.back().channels);
    std::vector<float> one2(2, 1.f), zero2(2, 0.f);
    WSC_TRY(make_conv(net, w7, 1, 0, 0, 0, one2, zero2, nullptr, nullptr, &net->dp7b));
    if (has(d, "mean_shift.running_mean")) {
        const HostTensor *ms;
        WSC_TRY(get(d, "mean_shift.running_mean", 1, &ms));
        WSC_CHECK(ms->shape[0] == 2, WSC_ERR_SHAPE, "mean_shift.running_mean must have 2 entries");
        net->mean_shift[0] = ms->data[0];
        net->mean_shift[1] = ms->data[1];
    }
    return WSC_OK;
}

// HeadSpec: {name, src, conv stride, out channels, GroupNorm groups, upsample, dst concat, channel offset};
// src > 0 is a backbone stage, src < 0 the concat buffer -src - 1.
// ResNet50 / VGG16 concat buffers: 0 = edge concat (5 x 32 -> 192 ch), 1 = dp3|dp4|dp5 (768, stage-3 resolution),
// 2 = dp1|dp2|dp_up3 (448), 3 = fc_dp7 hidden (256); all but 1 at the stage-2 resolution.
const std::vector<IrnCat> IRN_CATS5 = {{192, 2, 1}, {768, 3, 1}, {448, 2, 1}, {256, 2, 1}};

int build_resnet50_irn(wsc_net *net, const Dict &d) {
    WSC_TRY(build_resnet50_backbone(net, d));
    const std::vector<HeadSpec> specs = {
        {"fc_edge1", 1, 1, 32, 4, 1, 0, 0},   {"fc_edge2", 2, 1, 32, 4, 1, 0, 32},  {"fc_edge3", 3, 1, 32, 4, 2, 0, 64},
        {"fc_edge4", 4, 1, 32, 4, 4, 0, 96},  {"fc_edge5", 5, 1, 32, 4, 4, 0, 128}, {"fc_dp1", 1, 1, 64, 8, 1, 2, 0},
        {"fc_dp2", 2, 1, 128, 16, 1, 2, 64},  {"fc_dp3", 3, 1, 256, 16, 1, 1, 0},   {"fc_dp4", 4, 1, 256, 16, 2, 1, 256},
        {"fc_dp5", 5, 1, 256, 16, 2, 1, 512}, {"fc_dp6", -2, 1, 256, 16, 2, 2, 192}, {"fc_dp7", -3, 1, 256, 16, 1, 3, 0}};
    return build_irn_heads(net, d, specs, IRN_CATS5, "fc_edge6", 160, "fc_dp7.3");
}

int build_vgg16_irn(wsc_net *net, const Dict &d) {
    const std::vector<std::pair<std::string, std::vector<int>>> cfg = {
        {"layer1", {64, 64, -1}},
        {"layer2", {128, 128, -1}},
        {"layer3", {256, 256, 256, -1}},
        {"layer4", {512, 512, 512, 512, 512, 512}},
        {"layer5", {1024, -2, 1024, -2}}}; // vgg16.py:44; stage k = layer k (vgg16_irn.py:21-25)
    int cur = 0;
    WSC_TRY(build_plain_stack(net, d, "vgg16", cfg, &cur, &net->F, &net->taps));
    net->final_buf = cur;
    // ds_fac = 0.25 (vgg16_irn.py:30-98): stage1 is at 1/2 resolution and its heads use a stride-2 1x1 conv
    const std::vector<HeadSpec> specs = {
        {"fc_edge1", 1, 2, 32, 4, 1, 0, 0},   {"fc_edge2", 2, 1, 32, 4, 1, 0, 32},  {"fc_edge3", 3, 1, 32, 4, 2, 0, 64},
        {"fc_edge4", 4, 1, 32, 4, 2, 0, 96},  {"fc_edge5", 5, 1, 32, 4, 2, 0, 128}, {"fc_dp1", 1, 2, 64, 8, 1, 2, 0},
        {"fc_dp2", 2, 1, 128, 16, 1, 2, 64},  {"fc_dp3", 3, 1, 256, 16, 1, 1, 0},   {"fc_dp4", 4, 1, 256, 16, 1, 1, 256},
        {"fc_dp5", 5, 1, 256, 16, 1, 1, 512}, {"fc_dp6", -2, 1, 256, 16, 2, 2, 192}, {"fc_dp7", -3, 1, 256, 16, 1, 3, 0}};
    return build_irn_heads(net, d, specs, IRN_CATS5, "fc_edge6", 160, "fc_dp7.3");
}

// m7_irn.py:19-21,24-69,96-112: three stages (layer1 @1/2, layer2 @1/4, layer3_p1 + layer3_p2 @1/8); the edge
// branch ends at the stage-1 resolution (1/2), the displacement branch at 1/4; fc_dp4 is a head on fc_dp3's output.
int build_m7_irn(wsc_net *net, const Dict &d) {
    const std::vector<std::pair<std::string, std::vector<int>>> cfg = {
        {"layer1", {64, 64, -1}}, {"layer2", {128, 128, -1}}, {"layer3_p1", {256, 256, 256}}, {"layer3_p2", {-1, -2}}};
    int cur = 0;
    std::vector<int> layer_taps;
    WSC_TRY(build_plain_stack(net, d, "m7", cfg, &cur, &net->F, &layer_taps));
    net->final_buf = cur;
    net->taps = {layer_taps[0], layer_taps[1], layer_taps[3]};
    const std::vector<IrnCat> cats = {{128, 1, 1}, {256, 3, 1}, {448, 2, 1}, {256, 2, 1}}; // edge | dp3 | dp1|dp2|dp4 | hidden
    const std::vector<HeadSpec> specs = {
        {"fc_edge1", 1, 1, 32, 4, 1, 0, 0},  {"fc_edge2", 2, 1, 32, 4, 2, 0, 32},  {"fc_edge3", 3, 1, 32, 4, 4, 0, 64},
        {"fc_dp1", 1, 2, 64, 8, 1, 2, 0},    {"fc_dp2", 2, 1, 128, 16, 1, 2, 64},  {"fc_dp3", 3, 1, 256, 16, 1, 1, 0},
        {"fc_dp4", -2, 1, 256, 16, 2, 2, 192}, {"fc_dp5", -3, 1, 256, 16, 1, 3, 0}};
    return build_irn_heads(net, d, specs, cats, "fc_edge4", 96, "fc_dp5.3");
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Plan {
    std::vector<int> H, W, C; // per op output dims
    size_t max_act = 0;       // elements
    int hf = 0, wf = 0;       // final feature map size
};

// (SH x SW network input: square for the fixed-size configurations, the image's own size for outsize = None)
int plan_dims(const wsc_net *net, int N, int SH, int SW, Plan *pl);
int plan_dims(const wsc_net *net, int N, int S, Plan *pl) { return plan_dims(net, N, S, S, pl); }
int plan_dims(const wsc_net *net, int N, int SH, int SW, Plan *pl) {
    int bh[4] = {0, 0, 0, 0}, bw[4] = {0, 0, 0, 0}, bc[4] = {0, 0, 0, 0};
    pl->max_act = 0;
    for (const Op &op : net->ops) {
        int H = op.in < 0 ? SH : bh[op.in], W = op.in < 0 ? SW : bw[op.in], C = op.in < 0 ? 4 : bc[op.in];
        int Ho, Wo, Co;
        if (op.type == OP_CONV) {
            const ConvW &c = net->convs[op.conv];
            Ho = (H + 2 * c.pad - c.kh) / c.stride + 1;
            Wo = (W + 2 * c.pad - c.kw) / c.stride + 1;
            Co = op.pitch ? op.pitch : c.Cout;
            if (op.in2 >= 0) {
                WSC_CHECK(c.kh == 1 && c.kw == 1 && c.stride == 1 && (H - 1) * op.ps < bh[op.in2] && (W - 1) * op.ps < bw[op.in2],
                          WSC_ERR_INVALID, "internal: second input of a %d x %d layer does not cover its %d x %d output", c.kh, c.kw, H, W);
                C += bc[op.in2];
            }
            WSC_CHECK(C == c.Cin, WSC_ERR_INVALID, "internal: channel mismatch %d vs %d", C, c.Cin);
        } else if (op.type == OP_GATHER) {
            Ho = (H - 1) / op.ps + 1;
            Wo = (W - 1) / op.ps + 1;
            Co = op.pitch;
            WSC_CHECK(Ho == bh[op.out] && Wo == bw[op.out] && bc[op.out] == op.pitch && op.coff + C <= op.pitch, WSC_ERR_INVALID,
                      "internal: gather into a %d x %d x %d tensor does not fit", bh[op.out], bw[op.out], bc[op.out]);
        } else {
            Ho = (H + 2 * op.pp - op.pk) / op.ps + 1;
            Wo = (W + 2 * op.pp - op.pk) / op.ps + 1;
            Co = C;
        }
        WSC_CHECK(Ho > 0 && Wo > 0, WSC_ERR_INVALID, "input size %d x %d too small for this network", SH, SW);
        bh[op.out] = Ho; bw[op.out] = Wo; bc[op.out] = Co;
        pl->H.push_back(Ho); pl->W.push_back(Wo); pl->C.push_back(Co);
        const size_t e = (size_t)N * Ho * Wo * Co;
        if (e > pl->max_act) pl->max_act = e;
    }
    pl->hf = bh[net->final_buf];
    pl->wf = bw[net->final_buf];
    return WSC_OK;
}

// bytes of one plane of stage tap k (x_{k+1}), 256-byte aligned
size_t tap_plane_bytes(const wsc_net *net, const Plan &pl, int N, int k) {
    const int op = net->taps[k];
    return align_up((size_t)N * pl.H[op] * pl.W[op] * pl.C[op] * sizeof(bf16_t), 256);
}

// Runs the conv stack on N samples; returns pointers to the final feature map planes.  With copy_taps
// the stage outputs x1..x5 (net->taps) are copied, plane by plane (hi [, lo]), to the start of the extra
// region in stage order -- the rotating activation buffers are overwritten as the stack proceeds.
int run_backbone(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int N, int S, size_t extra_bytes,
                 const bf16_t **feat, const bf16_t **feat_lo, int *hf, int *wf, void **extra, bool copy_taps = false, int SW = 0) {
    if (SW <= 0) SW = S; // S x SW input (SW given for the non-square, native-size path)
    Plan pl;
    WSC_TRY(plan_dims(net, N, S, SW, &pl));
    const int planes = net->split ? 2 : 1;
    // f16x3 stem (small_cin == 3): the NHWC4 input carries its zero border -- rows / columns the kernel rows of the last
    // output pixel reach, an 8-pixel window per kernel row (conv_igemm.hip)
    int in_h = S, in_w = SW, in_pad = 0;
    bool fused_stem = false;
    if (!net->ops.empty() && net->ops[0].type == OP_CONV && net->ops[0].in < 0 && net->convs[net->ops[0].conv].small_cin == 3) {
        const ConvW &c0 = net->convs[net->ops[0].conv];
        in_h = (pl.H[0] - 1) * c0.stride + c0.kh;
        in_w = (pl.W[0] - 1) * c0.stride + 8;
        in_pad = c0.pad;
        // conv 7x7 / 2 / 3 (-> 64) + BN + ReLU followed by MaxPool 3 / 2 / 1 (resnet50.py:54-64): one kernel (stem_pool.hip),
        // unless the conv's own output is wanted (a tap) or the path is switched off
        bool tapped = false;
        for (int tp : net->taps) tapped = tapped || tp == 0;
        fused_stem = ctx->opt[WSC_OPT_STEM_POOL_FUSED] && net->ops.size() > 1 && net->ops[1].type == OP_POOL &&
                     net->ops[1].in == net->ops[0].out && net->ops[1].pk == 3 && net->ops[1].ps == 2 && net->ops[1].pp == 1 &&
                     c0.kh == 7 && c0.kw == 7 && c0.stride == 2 && c0.pad == 3 && c0.Cout == 64 && c0.s2 == nullptr &&
                     net->ops[0].res < 0 && net->fmt == 1 && !tapped;
        if (fused_stem) {
            stem_pool_input_dims(S, SW, &in_h, &in_w);
            in_pad = 5;
        }
    }
    const size_t in_bytes = align_up((size_t)N * in_h * in_w * 4 * sizeof(bf16_t), 256);
    const size_t act_bytes = align_up(pl.max_act * sizeof(bf16_t), 256);
    const size_t total = in_bytes * planes + act_bytes * 4 * planes + align_up(extra_bytes, 256);
    void *ws;
    WSC_TRY(wsc_ctx_workspace(ctx, total, &ws));
    char *p = (char *)ws;
    bf16_t *xin = (bf16_t *)p; p += in_bytes;
    bf16_t *xin_lo = nullptr;
    if (net->split) { xin_lo = (bf16_t *)p; p += in_bytes; }
    bf16_t *buf[4], *buf_lo[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
        buf[i] = (bf16_t *)p; p += act_bytes;
        if (net->split) { buf_lo[i] = (bf16_t *)p; p += act_bytes; }
    }
    *extra = (void *)p;

    if (in_pad > 0) WSC_TRY(launch_nchw_to_nhwc4_pad(ctx, x_dev, N, S, SW, in_h, in_w, in_pad, xin, xin_lo, net->fmt));
    else WSC_TRY(launch_nchw_to_nhwc4(ctx, x_dev, N, S, SW, xin, xin_lo, net->fmt));
    int bh[4] = {0, 0, 0, 0}, bw[4] = {0, 0, 0, 0}, bc[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < net->ops.size(); ++i) {
        const Op &op = net->ops[i];
        const int H = op.in < 0 ? S : bh[op.in], W = op.in < 0 ? SW : bw[op.in], C = op.in < 0 ? 4 : bc[op.in];
        const bf16_t *src = op.in < 0 ? xin : buf[op.in];
        const bf16_t *src_lo = op.in < 0 ? xin_lo : buf_lo[op.in];
        if (fused_stem && i == 0) continue; // computed together with the pool that follows
        if (fused_stem && i == 1) {
            const ConvW &c0 = net->convs[net->ops[0].conv];
            WSC_TRY(launch_stem_pool(ctx, xin, xin_lo, N, S, SW, c0.w, c0.kh * 64, c0.s1, c0.b1, c0.relu, buf[op.out], buf_lo[op.out]));
        } else if (op.type == OP_CONV) {
            const ConvW &c = net->convs[op.conv];
            ConvLaunch L;
            memset(&L, 0, sizeof(L));
            L.x = src; L.x_lo = src_lo; L.w = c.w;
            L.s1 = c.s1; L.b1 = c.b1; L.s2 = c.s2; L.b2 = c.b2;
            L.res = op.res >= 0 ? buf[op.res] : nullptr;
            L.res_lo = op.res >= 0 ? buf_lo[op.res] : nullptr;
            L.y = buf[op.out]; L.y_lo = buf_lo[op.out]; L.y_f32 = nullptr;
            if (op.in2 >= 0) {
                L.x2 = buf[op.in2]; L.x2_lo = buf_lo[op.in2];
                L.H2 = bh[op.in2]; L.W2 = bw[op.in2]; L.C2 = bc[op.in2]; L.stride2 = op.ps;
            }
            L.ldy = op.pitch; // (0: dense)
            if (op.pitch) {
                L.y += op.coff;
                if (L.y_lo) L.y_lo += op.coff;
            }
            L.N = N; L.H = H; L.W = W; L.Cin = c.Cin; L.Ho = pl.H[i]; L.Wo = pl.W[i];
            L.Cout = c.Cout; L.CoutPad = c.CoutPad;
            L.kh = c.kh; L.kw = c.kw; L.stride = c.stride; L.pad = c.pad; L.relu = c.relu;
            L.small_cin = c.small_cin; L.split = net->split; L.fmt = net->fmt;
            if (c.small_cin == 3 && op.in < 0) { // the padded input buffer: its own size, no padding left to apply
                L.H = in_h; L.W = in_w; L.pad = 0;
            }
            WSC_TRY(conv_igemm_launch(ctx, L));
        } else if (op.type == OP_GATHER) {
            WSC_TRY(launch_gather_strided(ctx, src, src_lo, N, H, W, C, op.ps, pl.H[i], pl.W[i], buf[op.out] + op.coff,
                                          buf_lo[op.out] ? buf_lo[op.out] + op.coff : nullptr, op.pitch));
        } else {
            WSC_TRY(launch_maxpool(ctx, src, src_lo, N, H, W, C, op.pk, op.ps, op.pp, pl.H[i], pl.W[i], buf[op.out],
                                   buf_lo[op.out], net->fmt));
        }
        bh[op.out] = pl.H[i]; bw[op.out] = pl.W[i]; bc[op.out] = pl.C[i];
        if (copy_taps)
            for (size_t k = 0; k < net->taps.size(); ++k)
                if (net->taps[k] == (int)i) {
                    char *dst = (char *)*extra;
                    for (size_t j = 0; j < k; ++j) dst += tap_plane_bytes(net, pl, N, (int)j) * planes;
                    const size_t nb = (size_t)N * pl.H[i] * pl.W[i] * pl.C[i] * sizeof(bf16_t);
                    WSC_HIP(hipMemcpyAsync(dst, buf[op.out], nb, hipMemcpyDeviceToDevice, ctx->stream));
                    if (net->split)
                        WSC_HIP(hipMemcpyAsync(dst + tap_plane_bytes(net, pl, N, (int)k), buf_lo[op.out], nb,
                                               hipMemcpyDeviceToDevice, ctx->stream));
                }
    }
    *feat = buf[net->final_buf];
    *feat_lo = buf_lo[net->final_buf];
    *hf = pl.hf;
    *wf = pl.wf;
    return WSC_OK;
}

} // namespace

// GAP / global-max + Linear + Sigmoid, defined in misc_kernels.hip (avg) -- the max flavour
// is selected by a negative hw.
extern "C" {

int wsc_net_create(wsc_ctx *ctx, int arch, const wsc_tensor_desc *weights, int n_weights, int num_classes,
                   int precision, wsc_net **out) {
    WSC_CHECK(ctx && weights && out && n_weights > 0, WSC_ERR_INVALID, "wsc_net_create: null argument");
    WSC_CHECK(num_classes > 0 && num_classes <= 64, WSC_ERR_INVALID, "num_classes=%d outside [1,64]", num_classes);
    WSC_CHECK(precision == WSC_PREC_BF16 || precision == WSC_PREC_BF16X3 || precision == WSC_PREC_F16 ||
                  precision == WSC_PREC_F16X3,
              WSC_ERR_INVALID, "unknown precision %d", precision);
    WSC_HIP(hipSetDevice(ctx->device));
    Dict d;
    for (int i = 0; i < n_weights; ++i) {
        WSC_CHECK(weights[i].name != nullptr, WSC_ERR_INVALID, "weights[%d].name is null", i);
        WSC_CHECK(weights[i].ndim >= 0 && weights[i].ndim <= 4, WSC_ERR_SHAPE, "'%s': ndim %d unsupported",
                  weights[i].name, weights[i].ndim);
        HostTensor t;
        t.data = weights[i].data;
        t.ndim = weights[i].ndim;
        for (int k = 0; k < 4; ++k) t.shape[k] = k < t.ndim ? weights[i].shape[k] : 1;
        if (t.ndim == 0) { t.ndim = 1; t.shape[0] = 1; }
        d[weights[i].name] = t;
    }
    wsc_net *net = new wsc_net();
    net->ctx = ctx;
    net->arch = arch;
    net->C = num_classes;
    net->split = precision == WSC_PREC_BF16X3 ? 1 : (precision == WSC_PREC_F16X3 ? 2 : 0);
    net->fmt = (precision == WSC_PREC_F16 || precision == WSC_PREC_F16X3) ? 1 : 0;
    int st;
    switch (arch) {
    case WSC_ARCH_RESNET50_CAM: st = build_resnet50(net, d); break;
    case WSC_ARCH_VGG16_CAM: st = build_vgg16(net, d); break;
    case WSC_ARCH_M7_CAM: st = build_m7(net, d); break;
    case WSC_ARCH_RESNET50_IRN: st = build_resnet50_irn(net, d); break;
    case WSC_ARCH_VGG16_IRN: st = build_vgg16_irn(net, d); break;
    case WSC_ARCH_M7_IRN: st = build_m7_irn(net, d); break;
    default:
        wsc_set_error("unknown arch %d", arch);
        st = WSC_ERR_INVALID;
    }
    if (st != WSC_OK) {
        wsc_net_destroy(net);
        return st;
    }
    *out = net;
    return WSC_OK;
}

void wsc_net_destroy(wsc_net *net) {
    if (!net) return;
    for (void *p : net->allocs) (void)hipFree(p);
    delete net;
}

int wsc_net_cam_size(const wsc_net *net, int S, int *h_out) {
    WSC_CHECK(net && h_out, WSC_ERR_INVALID, "wsc_net_cam_size: null argument");
    Plan pl;
    WSC_TRY(plan_dims(net, 1, S, &pl));
    *h_out = pl.hf;
    return WSC_OK;
}

int wsc_net_feat_channels(const wsc_net *net, int *f_out) {
    WSC_CHECK(net && f_out, WSC_ERR_INVALID, "wsc_net_feat_channels: null argument");
    *f_out = net->F;
    return WSC_OK;
}

int wsc_net_cam_size_hw(const wsc_net *net, int H, int W, int *h_out, int *w_out) {
    WSC_CHECK(net && h_out && w_out, WSC_ERR_INVALID, "wsc_net_cam_size_hw: null argument");
    Plan pl;
    WSC_TRY(plan_dims(net, 1, H, W, &pl));
    *h_out = pl.hf;
    *w_out = pl.wf;
    return WSC_OK;
}

// The 1x1 head (CAM classifier weights / Grad-CAM alpha) with an fp32 NHWC output: in the IEEE-half modes (f16, f16x3) with at
// most 32 classes it is the streaming kernel of cam_head.hip (an HBM stream, not a tile problem); else the tiled kernel.
static int run_head(wsc_ctx *ctx, const wsc_net *net, const ConvLaunch &L) {
    // (K % 256: each wave's quarter of K is whole trips of four 16-channel slices)
    if (L.fmt == 1 && L.split != 1 && L.Cout <= 32 && L.CoutPad >= 32 && L.Cin % 256 == 0 && L.y_f32 != nullptr &&
        ctx->opt[WSC_OPT_CAM_HEAD_STREAM] != 0) {
        // (the streaming form checks packing and alignment itself; anything it does not take goes to the tiled kernel)
        const int st = launch_cam_head(ctx, L.x, L.split == 2 ? L.x_lo : nullptr, L.N * L.Ho * L.Wo, L.Cin, L.w,
                                       L.Cin * (L.split == 2 ? 2 : 1), L.CoutPad, L.s1, L.b1, L.Cout, L.relu, L.y_f32);
        if (st != WSC_ERR_INVALID) return st;
    }
    (void)net;
    return conv_igemm_launch(ctx, L);
}

int wsc_net_forward_cam(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int B, int S, float *cam_dev,
                        float *score_dev) {
    return wsc_net_forward_cam_hw(ctx, net, x_dev, B, S, S, cam_dev, score_dev);
}

int wsc_net_forward_cam_hw(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int B, int S, int SW, float *cam_dev,
                           float *score_dev) {
    WSC_CHECK(ctx && net && x_dev && cam_dev, WSC_ERR_INVALID, "wsc_net_forward_cam: null argument");
    WSC_CHECK(B > 0 && S > 0 && SW > 0, WSC_ERR_INVALID, "wsc_net_forward_cam: B=%d input %d x %d", B, S, SW);
    WSC_CHECK(score_dev == nullptr || net->cls_w != nullptr, WSC_ERR_INVALID,
              "this architecture has no classifier branch (score_dev must be NULL)");
    WSC_HIP(hipSetDevice(ctx->device));
    const int N = 2 * B;
    Plan pl;
    WSC_TRY(plan_dims(net, N, S, SW, &pl));
    const size_t head_bytes = (size_t)N * pl.hf * pl.wf * net->C * sizeof(float);
    const bf16_t *feat, *feat_lo;
    int hf, wf;
    void *extra;
    WSC_TRY(run_backbone(ctx, net, x_dev, N, S, head_bytes, &feat, &feat_lo, &hf, &wf, &extra, false, SW));
    float *head_out = (float *)extra;
    ConvLaunch L;
    memset(&L, 0, sizeof(L));
    const ConvW &c = net->head;
    L.x = feat; L.x_lo = feat_lo; L.w = c.w; L.s1 = c.s1; L.b1 = c.b1;
    L.y_f32 = head_out;
    L.N = N; L.H = hf; L.W = wf; L.Cin = c.Cin; L.Ho = hf; L.Wo = wf; L.Cout = c.Cout; L.CoutPad = c.CoutPad;
    L.kh = 1; L.kw = 1; L.stride = 1; L.pad = 0; L.relu = 0; L.small_cin = 0; L.split = net->split; L.fmt = net->fmt;
    WSC_TRY(run_head(ctx, net, L));
    WSC_TRY(launch_flip_add(ctx, head_out, B, hf, wf, net->C, net->C, cam_dev));
    if (score_dev != nullptr)
        WSC_TRY(launch_gap_linear_sigmoid(ctx, feat, feat_lo, B, net->cls_max ? -(hf * wf) : hf * wf, net->F,
                                          net->cls_w, net->cls_b, net->Ccls, score_dev, net->fmt, 2));
    return WSC_OK;
}

int wsc_net_forward_gradcam(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int N, int S, int relu,
                            float *cams_dev, float *score_dev) {
    WSC_CHECK(ctx && net && x_dev && cams_dev, WSC_ERR_INVALID, "wsc_net_forward_gradcam: null argument");
    WSC_CHECK(N > 0 && S > 0, WSC_ERR_INVALID, "wsc_net_forward_gradcam: N=%d S=%d", N, S);
    WSC_CHECK(score_dev == nullptr || net->cls_w != nullptr, WSC_ERR_INVALID,
              "this architecture has no classifier branch (score_dev must be NULL)");
    WSC_HIP(hipSetDevice(ctx->device));
    const bf16_t *feat, *feat_lo;
    int hf, wf;
    void *extra;
    WSC_TRY(run_backbone(ctx, net, x_dev, N, S, 0, &feat, &feat_lo, &hf, &wf, &extra));
    ConvLaunch L;
    memset(&L, 0, sizeof(L));
    const ConvW &c = net->head;
    L.x = feat; L.x_lo = feat_lo; L.w = c.w; L.s1 = c.s1; L.b1 = c.b1;
    L.y_f32 = cams_dev; // fp32 NHWC [N][h][w][C]: the layout of np.einsum('ijkl,lm->ijkm')
    L.N = N; L.H = hf; L.W = wf; L.Cin = c.Cin; L.Ho = hf; L.Wo = wf; L.Cout = c.Cout; L.CoutPad = c.CoutPad;
    L.kh = 1; L.kw = 1; L.stride = 1; L.pad = 0; L.relu = relu ? 1 : 0; L.small_cin = 0; L.split = net->split;
    L.fmt = net->fmt;
    WSC_TRY(run_head(ctx, net, L));
    if (score_dev != nullptr)
        WSC_TRY(launch_gap_linear_sigmoid(ctx, feat, feat_lo, N, net->cls_max ? -(hf * wf) : hf * wf, net->F,
                                          net->cls_w, net->cls_b, net->Ccls, score_dev, net->fmt, 1));
    return WSC_OK;
}

int wsc_net_forward_features(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int N, int S, float *feat_dev) {
    WSC_CHECK(ctx && net && x_dev && feat_dev, WSC_ERR_INVALID, "wsc_net_forward_features: null argument");
    WSC_CHECK(N > 0 && S > 0, WSC_ERR_INVALID, "wsc_net_forward_features: N=%d S=%d", N, S);
    WSC_HIP(hipSetDevice(ctx->device));
    const bf16_t *feat, *feat_lo;
    int hf, wf;
    void *extra;
    WSC_TRY(run_backbone(ctx, net, x_dev, N, S, 0, &feat, &feat_lo, &hf, &wf, &extra));
    return launch_bf16_to_f32(ctx, feat, feat_lo, (size_t)N * hf * wf * net->F, feat_dev, net->fmt);
}

// EdgeDisplacement.forward (resnet50_irn.py:212-232 / vgg16_irn.py:306-321) for B images: x is the
// [orig, h-flip] pair of each image already zero-padded to the crop size S (F.pad(x, [0, S-w, 0, S-h])).
// Backbone with stage taps -> 12 Conv-GroupNorm-[Upsample]-ReLU heads into three concat buffers -> fc_edge6 /
// fc_dp7[3] -> crop to (feat_h, feat_w), sigmoid(edge[0]/2 + edge[1].flip(-1)/2), dp[0] - mean_shift.
int wsc_net_forward_edge(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int B, int S, int feat_h, int feat_w,
                         float *edge_dev, float *dp_dev) {
    WSC_CHECK(ctx && net && x_dev && edge_dev && dp_dev, WSC_ERR_INVALID, "wsc_net_forward_edge: null argument");
    WSC_CHECK(!net->heads.empty() && net->cats.size() >= 2, WSC_ERR_INVALID,
              "wsc_net_forward_edge: the network is not an IRNet EdgeDisplacement net");
    WSC_CHECK(B > 0 && S > 0 && feat_h > 0 && feat_w > 0, WSC_ERR_INVALID, "wsc_net_forward_edge: B=%d S=%d", B, S);
    WSC_HIP(hipSetDevice(ctx->device));
    const int N = 2 * B;
    const int planes = net->split ? 2 : 1;
    Plan pl;
    WSC_TRY(plan_dims(net, N, S, &pl));
    const int NS = (int)net->taps.size(), NC = (int)net->cats.size();
    std::vector<int> Hs(NS), Ws(NS), Cs(NS);
    std::vector<size_t> tap_off(NS);
    size_t off = 0;
    for (int k = 0; k < NS; ++k) {
        const int op = net->taps[k];
        Hs[k] = pl.H[op]; Ws[k] = pl.W[op]; Cs[k] = pl.C[op];
        tap_off[k] = off;
        off += tap_plane_bytes(net, pl, N, k) * planes;
    }
    auto outdim = [](int v, int stride) { return (v - 1) / stride + 1; };
    std::vector<int> cat_c(NC), cat_h(NC), cat_w(NC);
    std::vector<size_t> cat_off(NC), cat_bytes(NC);
    for (int i = 0; i < NC; ++i) {
        const IrnCat &c = net->cats[i];
        cat_c[i] = c.channels;
        cat_h[i] = outdim(Hs[c.stage - 1], c.stride);
        cat_w[i] = outdim(Ws[c.stage - 1], c.stride);
        cat_bytes[i] = align_up((size_t)N * cat_h[i] * cat_w[i] * cat_c[i] * sizeof(bf16_t), 256);
        cat_off[i] = off;
        off += cat_bytes[i] * planes;
    }
    const int He = cat_h[0], We = cat_w[0], Hd = cat_h[NC - 1], Wd = cat_w[NC - 1]; // edge / displacement maps
    WSC_CHECK(feat_h <= He && feat_w <= We && feat_h <= Hd && feat_w <= Wd, WSC_ERR_INVALID,
              "feature size %dx%d exceeds the head resolution %dx%d / %dx%d", feat_h, feat_w, He, We, Hd, Wd);
    size_t ftmp_elems = 0, part_bytes = 0;
    int max_groups = 1;
    for (const IrnHead &h : net->heads) {
        const int sh = h.src > 0 ? Hs[h.src - 1] : cat_h[-h.src - 1];
        const int sw = h.src > 0 ? Ws[h.src - 1] : cat_w[-h.src - 1];
        const int ho = outdim(sh, h.conv.stride), wo = outdim(sw, h.conv.stride);
        ftmp_elems = std::max(ftmp_elems, (size_t)N * ho * wo * h.conv.Cout);
        part_bytes = std::max(part_bytes, group_norm_partial_bytes(N, ho, wo, h.groups));
        max_groups = std::max(max_groups, h.groups);
    }
    const size_t ftmp_off = off; off += align_up(ftmp_elems * sizeof(float), 256);
    const size_t part_off = off; off += align_up(part_bytes, 256);
    const size_t stat_off = off; off += align_up(sizeof(float) * 2 * N * max_groups, 256);
    const size_t e_off = off; off += align_up(sizeof(float) * (size_t)N * He * We, 256);
    const size_t d_off = off; off += align_up(sizeof(float) * (size_t)N * Hd * Wd * 2, 256);

    const bf16_t *feat, *feat_lo;
    int hf, wf;
    void *extra;
    WSC_TRY(run_backbone(ctx, net, x_dev, N, S, off, &feat, &feat_lo, &hf, &wf, &extra, true));
    char *base = (char *)extra;
    // the padding channels of the edge concat feed zero weights but must not hold NaN bit patterns
    WSC_HIP(hipMemsetAsync(base + cat_off[0], 0, cat_bytes[0] * planes, ctx->stream));

    auto plane = [&](size_t o, size_t bytes, int which) -> bf16_t * {
        return which == 0 ? (bf16_t *)(base + o) : (net->split ? (bf16_t *)(base + o + bytes) : nullptr);
    };
    auto run_conv = [&](const ConvW &c, const bf16_t *x, const bf16_t *x_lo, int H, int W, float *y_f32) -> int {
        ConvLaunch L;
        memset(&L, 0, sizeof(L));
        L.x = x; L.x_lo = x_lo; L.w = c.w;
        L.s1 = c.s1; L.b1 = c.b1; L.s2 = nullptr; L.b2 = nullptr;
        L.y = nullptr; L.y_lo = nullptr; L.y_f32 = y_f32;
        L.N = N; L.H = H; L.W = W; L.Cin = c.Cin; L.Ho = outdim(H, c.stride); L.Wo = outdim(W, c.stride);
        L.Cout = c.Cout; L.CoutPad = c.CoutPad;
        L.kh = 1; L.kw = 1; L.stride = c.stride; L.pad = 0; L.relu = 0;
        L.small_cin = 0; L.split = net->split; L.fmt = net->fmt;
        return conv_igemm_launch(ctx, L);
    };
    float *ftmp = (float *)(base + ftmp_off);
    for (const IrnHead &h : net->heads) {
        const bf16_t *x, *x_lo;
        int sh, sw, sc;
        if (h.src > 0) {
            const int k = h.src - 1;
            const size_t pb = tap_plane_bytes(net, pl, N, k);
            x = plane(tap_off[k], pb, 0); x_lo = plane(tap_off[k], pb, 1);
            sh = Hs[k]; sw = Ws[k]; sc = Cs[k];
        } else {
            const int ci = -h.src - 1;
            x = plane(cat_off[ci], cat_bytes[ci], 0); x_lo = plane(cat_off[ci], cat_bytes[ci], 1);
            sh = cat_h[ci]; sw = cat_w[ci]; sc = cat_c[ci];
        }
        WSC_CHECK(sc == h.conv.Cin, WSC_ERR_SHAPE, "IRNet head: %d input channels, the weights expect %d", sc, h.conv.Cin);
        WSC_TRY(run_conv(h.conv, x, x_lo, sh, sw, ftmp));
        const int ho = outdim(sh, h.conv.stride), wo = outdim(sw, h.conv.stride);
        WSC_TRY(launch_group_norm_stats(ctx, ftmp, N, ho, wo, h.conv.Cout, h.groups, 1e-5f, base + part_off,
                                        base + stat_off));
        const int di = h.dst;
        WSC_TRY(launch_group_norm_apply(ctx, ftmp, base + stat_off, h.gamma, h.beta, N, ho, wo, h.conv.Cout, h.groups,
                                        h.up, 1, plane(cat_off[di], cat_bytes[di], 0), plane(cat_off[di], cat_bytes[di], 1),
                                        cat_h[di], cat_w[di], cat_c[di], h.coff, net->fmt));
    }
    float *e_out = (float *)(base + e_off), *d_out = (float *)(base + d_off);
    WSC_TRY(run_conv(net->edge6, plane(cat_off[0], cat_bytes[0], 0), plane(cat_off[0], cat_bytes[0], 1), He, We, e_out));
    WSC_TRY(run_conv(net->dp7b, plane(cat_off[NC - 1], cat_bytes[NC - 1], 0), plane(cat_off[NC - 1], cat_bytes[NC - 1], 1),
                     Hd, Wd, d_out));
    WSC_TRY(launch_edge_finish(ctx, e_out, He, We, d_out, Hd, Wd, B, feat_h, feat_w, net->mean_shift[0],
                               net->mean_shift[1], edge_dev, dp_dev));
    return WSC_OK;
}

// One convolution layer through the production kernel, NCHW fp32 in / out (layout changes and
// weight packing included): y = [relu]( conv(x, w) * scale + shift [+ residual] ).
int wsc_conv2d_nchw(wsc_ctx *ctx, const float *x_dev, int N, int Cin, int H, int W, const float *w_host, int Cout,
                    int kh, int kw, int stride, int pad, const float *scale_host, const float *shift_host,
                    const float *residual_dev, int relu, int precision, float *y_dev) {
    WSC_CHECK(ctx && x_dev && w_host && y_dev, WSC_ERR_INVALID, "wsc_conv2d_nchw: null argument");
    WSC_CHECK(Cout % 8 == 0, WSC_ERR_INVALID, "wsc_conv2d_nchw: Cout=%d must be a multiple of 8", Cout);
    WSC_HIP(hipSetDevice(ctx->device));
    wsc_net tmp;
    tmp.ctx = ctx;
    const int generic = (precision & WSC_CONV_GENERIC) ? 1 : 0;
    precision &= ~WSC_CONV_GENERIC;
    WSC_CHECK(precision >= WSC_PREC_BF16 && precision <= WSC_PREC_F16X3, WSC_ERR_INVALID, "unknown precision %d", precision);
    tmp.split = precision == WSC_PREC_BF16X3 ? 1 : (precision == WSC_PREC_F16X3 ? 2 : 0);
    tmp.fmt = (precision == WSC_PREC_F16 || precision == WSC_PREC_F16X3) ? 1 : 0;
    HostTensor wt;
    wt.data = w_host; wt.ndim = 4; wt.shape[0] = Cout; wt.shape[1] = Cin; wt.shape[2] = kh; wt.shape[3] = kw;
    std::vector<float> s1(Cout, 1.f), b1(Cout, 0.f);
    if (scale_host) s1.assign(scale_host, scale_host + Cout);
    if (shift_host) b1.assign(shift_host, shift_host + Cout);
    int small = 0;
    if (Cin <= 4) small = kw <= 4 ? 1 : 2;
    if (small == 2 && tmp.split == 2 && (stride & 1) == 0 && kw <= 7) small = 3; // the f16x3 stem form (padded input)
    ConvW c;
    int st = make_conv(&tmp, &wt, stride, pad, relu, small, s1, b1, nullptr, nullptr, &c);
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        for (void *p : tmp.allocs) (void)hipFree(p);
    };
    if (st != WSC_OK) { cleanup(); return st; }
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    const int planes = tmp.split ? 2 : 1;
    const int in_h = small == 3 ? (Ho - 1) * stride + kh : H, in_w = small == 3 ? (Wo - 1) * stride + 8 : W;
    const size_t in_e = (size_t)N * in_h * in_w * c.Cin, out_e = (size_t)N * Ho * Wo * Cout;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    void *ws;
    st = wsc_ctx_workspace(ctx, planes * (al(in_e * 2) + 2 * al(out_e * 2)), &ws);
    if (st != WSC_OK) { cleanup(); return st; }
    char *p = (char *)ws;
    bf16_t *xi = (bf16_t *)p; p += al(in_e * 2);
    bf16_t *xi_lo = nullptr; if (tmp.split) { xi_lo = (bf16_t *)p; p += al(in_e * 2); }
    bf16_t *yo = (bf16_t *)p; p += al(out_e * 2);
    bf16_t *yo_lo = nullptr; if (tmp.split) { yo_lo = (bf16_t *)p; p += al(out_e * 2); }
    bf16_t *ri = nullptr, *ri_lo = nullptr;
    if (residual_dev) { ri = (bf16_t *)p; p += al(out_e * 2); if (tmp.split) { ri_lo = (bf16_t *)p; p += al(out_e * 2); } }
    if (small == 3) st = launch_nchw_to_nhwc4_pad(ctx, x_dev, N, H, W, in_h, in_w, pad, xi, xi_lo, tmp.fmt);
    else if (small) st = launch_nchw_to_nhwc4(ctx, x_dev, N, H, W, xi, xi_lo, tmp.fmt);
    else st = launch_nchw_to_nhwc(ctx, x_dev, N, Cin, H * W, xi, xi_lo, tmp.fmt);
    if (st == WSC_OK && residual_dev) st = launch_nchw_to_nhwc(ctx, residual_dev, N, Cout, Ho * Wo, ri, ri_lo, tmp.fmt);
    if (st == WSC_OK) {
        ConvLaunch L;
        memset(&L, 0, sizeof(L));
        L.x = xi; L.x_lo = xi_lo; L.w = c.w; L.s1 = c.s1; L.b1 = c.b1; L.res = ri; L.res_lo = ri_lo;
        L.y = yo; L.y_lo = yo_lo;
        L.N = N; L.H = in_h; L.W = in_w; L.Cin = c.Cin; L.Ho = Ho; L.Wo = Wo; L.Cout = Cout; L.CoutPad = c.CoutPad;
        L.kh = kh; L.kw = kw; L.stride = stride; L.pad = small == 3 ? 0 : pad; L.relu = relu; L.small_cin = small; L.split = tmp.split;
        L.fmt = tmp.fmt;
        L.generic = generic;
        st = conv_igemm_launch(ctx, L);
    }
    if (st == WSC_OK) st = launch_nhwc_to_nchw(ctx, yo, yo_lo, N, Cout, Ho * Wo, y_dev, tmp.fmt);
    cleanup();
    return st;
}

} // extern "C"
