// step.hip -- native executor of one training step (pc_train_step): the body of the reference's inner loop
// (run_train.py:186-238: forward(train, padding=False, sparse=True) -> get_loss -> x lam_weak -> backward -> clip -> Adam) issued from
// C++ for ANY batch geometry and truncation regime.  Host code only: it computes geometry, carves tensors out of the caller's arena,
// fills the descriptors of the entry points of this library (popcorn_hip.h) and calls them.  The Python engine (popcorn_amd/engine.py)
// issues the same launches one ctypes call at a time; at the reference's real training geometry (weak_batch_size = 2 census regions of
// varying size) those ~45 calls cost more host time than a small region's kernels take.
//
// Layout rules of the arena (what makes every launch take its fast path whatever the region's width):
//   * planar fp32 activations (B, C, H, W) with rows padded to a multiple of 4 floats: every row starts on a 16-byte boundary, so the
//     conv kernels keep the aligned staged loaders and vector epilogues on the building extractor's (H + 28) x (W + 28) domain too
//     (the pad columns are never read as data);
//   * the up-sampled map of an Up block is allocated with the extent and strides of the skip tensor it is concatenated with (the
//     transposed conv fills the top-left 2h x 2w of it): the two-source conv then sees two sources of identical layout;
//   * a pooled source is handed over cropped to even extents (MaxPool2d(2) floors).
#include "common.h"
#include <chrono>

#include <string.h>
#include <new>

namespace {

enum { L_INC1 = 0, L_INC2, L_D1A, L_D1B, L_D2A, L_D2B, L_UP2A, L_UP2B, L_UP1A, L_UP1B };
enum { T_UP2 = 0, T_UP1 = 1 };
constexpr int MAXK = PC_MAX_GROUP;        // (network, stream) pairs of one grouped launch
constexpr long long PC_STEP_SIDE_PX_DEFAULT = 1ll << 40;      // pixels (B * Hp * Wp) up to which the extractor's chain runs on the side stream

// ---- tensors ------------------------------------------------------------------------------------------------------------------------
struct Ten {                               // planar fp32 view (B, C, H, W)
    float* p = nullptr;
    int B = 0, C = 0, H = 0, W = 0;
    int64_t sb = 0, sc = 0;
    int sr = 0;
    bool ok() const { return p != nullptr; }
};

inline pc_src S(const Ten& t, int mode = PC_SRC_DIRECT, int oy = 0, int ox = 0) {
    pc_src s{};
    s.ptr = t.p; s.C = t.C; s.H = t.H; s.W = t.W; s.bstride = t.sb; s.cstride = t.sc; s.rstride = t.sr;
    s.mode = mode; s.oy = oy; s.ox = ox; s.chmap[0] = 0; s.chmap[1] = 1; s.chmap[2] = 2; s.chmap[3] = 3;
    s.dtype = PC_F32; s.xstride = 1;
    return s;
}
inline pc_dst D(const Ten& t) {
    pc_dst d{};
    d.ptr = t.p; d.bstride = t.sb; d.cstride = t.sc; d.rstride = t.sr; d.dtype = PC_F32; d.xstride = 1;
    return d;
}
inline Ten chans(const Ten& t, int c0, int n) {      // t[:, c0:c0 + n]
    Ten v = t;
    v.p = t.p + (int64_t)c0 * t.sc;
    v.C = n;
    return v;
}
inline Ten crop(const Ten& t, int H, int W) {        // t[:, :, :H, :W]
    Ten v = t;
    v.H = H; v.W = W;
    return v;
}

struct Arena {
    char* base = nullptr;
    int64_t cap = 0, off = 0, peak = 0;
    bool dry = true;
    void* raw(int64_t bytes) {
        off = (off + 255) & ~(int64_t)255;
        char* p = base + off;
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
    Ten act(int B, int C, int H, int W, bool dense = false) {
        Ten t;
        t.B = B; t.C = C; t.H = H; t.W = W;
        t.sr = dense ? W : (W + 3) & ~3;
        t.sc = (int64_t)H * t.sr;
        t.sb = (int64_t)C * t.sc;
        t.p = reinterpret_cast<float*>(raw((int64_t)B * t.sb * (int64_t)sizeof(float)));
        return t;
    }
};

// ---- the executor -------------------------------------------------------------------------------------------------------------------
struct Saved {                 // activations of one (network, stream) pair that the backward pass reads
    Ten a1, a2, b1, b2, c1, c2, u2, e1, e2, u1, f1, pa2, pb2;
    void* ws_up1 = nullptr;
    void* ws_up2 = nullptr;
};

struct Step {
    pc_step_plan plan;
    pc_bn bn_nobias[2][PC_STEP_CONVS];          // trainable network: ReLU / BN factor of a layer's output (no conv bias)
    pc_bn bn_half[2][2];                        // d1b's factor for the two 8-channel halves of its output (pointers into the same BN tensors)
    pc_bn bn_half_d1a[2][2];                    // ... and d1a's (the fused backward of d1b runs per 8-channel half of its input)
    float* ones2 = nullptr;                     // device {1, 1}: weights of the partial-logit sum
    // small regions: the frozen extractor's forward chain runs on a side stream next to the trainable U-Net's (two chains of ~13 small,
    // latency-bound launches each; at B = 64 tiles -- a full chip per launch -- every multi-stream attempt lost)
    hipStream_t side = nullptr;
    hipStream_t side_for = nullptr;             // the caller's stream the side stream was measured against (pick_side_stream)
    bool side_picked = false;
    struct SideFor { hipStream_t caller; hipStream_t side; } side_cache[4] = {};     // measured pairs (caller's stream -> its side stream)
    int n_side_cache = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_dep[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int ev_next = 0;
    int64_t side_px = 0;                        // use the side stream when B * Hp * Wp <= side_px
    int64_t side_bwd_px = 0;                    // backward: the weight-gradient-only launches on the side stream up to this many pixels
    // context of the step between its phases
    bool have_fwd = false;
    int B = 0, H = 0, W = 0, pt = 0, pl = 0, Hp = 0, Wp = 0;
    Ten feats, building, Xp_u;
    uint8_t* mask = nullptr;
    const uint8_t* sel = nullptr;               // device selection flags of this step (io.sel, or unpacked from io.sel_host)
    int32_t* counts = nullptr;
    float* popcount = nullptr;
    float* popdense = nullptr;
    float* scale_map = nullptr;
    float* g_pc = nullptr;
    void* head_ws = nullptr;
    Saved sv[2];
    bool unet_ng = false, enc_ng = false, deferred_popcount = false;
    int64_t fwd_end = 0;                        // arena offset behind the forward pass's tensors
    // per call
    Arena ar;
    hipStream_t st = nullptr;
    int launches = 0;
    int err = 0;

    bool go() const { return !ar.dry && err == 0; }
    void rc(int code) {
        if (code != 0 && err == 0) err = code;
        ++launches;
    }
};

inline void pad_geometry(int H, int W, int& pt, int& pb, int& pl, int& pr) {          // add_padding(force=False), popcorn.py:231-258
    pt = pb = pl = pr = 0;
    if (H % 32 != 0) { pt = (64 - H % 64) / 2; pb = (64 - H % 64) - pt; }
    if (W % 32 != 0) { pl = (64 - W % 64) / 2; pr = (64 - W % 64) - pl; }
}

// the per-step selection flags from HOST memory: bit-packed into the kernel arguments of a one-block launch that writes them as bytes
struct SelBits { uint32_t w[PC_STEP_SEL_MAX / 32]; };
__global__ __launch_bounds__(256) void sel_unpack_kernel(const SelBits b, uint8_t* out, int n) {
    for (int i = threadIdx.x; i < n; i += 256) out[i] = (b.w[i >> 5] >> (i & 31)) & 1u;
}

struct Net {                   // one DualStreamUNet taking part in a forward pass
    const pc_step_net* n;
    bool save;                 // keep the activations for the backward pass
    bool logit_only;           // write the two partial fusion_out_conv logits instead of the feature map
};

struct Key { int e, s; };

// Forward of `nE` networks on the same padded input Xp (B, 6 stream-ordered channels, Hp, Wp), one grouped launch per layer for all
// (network, stream) pairs -- engine.py: forward_multi.  feats[e]: (B, 16, Hp, Wp) or, for logit_only networks, (B, 2, Hp, Wp).
void forward_nets(Step& X, const Net* nets, int nE, const Ten& Xp, int Hp, int Wp, Ten* feats, Saved (*saved)[2]) {
    const int B = Xp.B;
    const int H1 = Hp / 2, W1 = Wp / 2, H2 = H1 / 2, W2 = W1 / 2;
    Arena& A = X.ar;
    Key keys[MAXK];
    int K = 0;
    for (int e = 0; e < nE; ++e)
        for (int s = 0; s < 2; ++s) keys[K++] = Key{e, s};
    auto lay = [&](const Key& k) -> const pc_step_stream& { return nets[k.e].n->s[k.s]; };
    bool any_save = false;
    for (int e = 0; e < nE; ++e) any_save = any_save || nets[e].save;

    Ten a1[MAXK], a2[MAXK], pa2[MAXK], b1[MAXK], b2[MAXK], pb2[MAXK], c1[MAXK], c2[MAXK], u2[MAXK], e1[MAXK], e2[MAXK], u1[MAXK], f1[MAXK];
    void* ws1[MAXK] = {nullptr, nullptr, nullptr, nullptr};
    void* ws2[MAXK] = {nullptr, nullptr, nullptr, nullptr};

    // generic grouped conv launch over all keys
    auto conv = [&](int L, const Ten* in, const Ten* in_b, Ten* out, int Cout, int H, int W, int mode, Ten* pooled) {
        pc_src sa[MAXK], sb[MAXK];
        pc_dst dout[MAXK], dpool[MAXK];
        pc_conv_fwd_desc d[MAXK];
        bool pool_ok = pooled != nullptr;
        for (int i = 0; i < K; ++i) {
            out[i] = A.act(B, Cout, H, W);
            dout[i] = D(out[i]);
            if (pooled && !pc_conv3x3_pool_out_ok(&dout[i], H, W)) pool_ok = false;
        }
        int Cin = 0;
        for (int i = 0; i < K; ++i) {
            memset(&d[i], 0, sizeof(d[i]));
            Ten src = in[i];
            if (mode == PC_SRC_POOL2) src = crop(src, 2 * H, 2 * W);
            sa[i] = S(src, mode);
            d[i].a = &sa[i];
            Cin = src.C;
            if (in_b) {
                sb[i] = S(in_b[i]);
                d[i].b = &sb[i];
                Cin += in_b[i].C;
            }
            d[i].w = lay(keys[i]).w[L];
            d[i].bn = &lay(keys[i]).bn[L];
            d[i].out = &dout[i];
            if (pool_ok) {
                pooled[i] = A.act(B, Cout, H / 2, W / 2);
                dpool[i] = D(pooled[i]);
                d[i].pool_out = &dpool[i];
            }
        }
        if (pooled && !pool_ok)
            for (int i = 0; i < K; ++i) pooled[i] = Ten{};
        if (X.go()) X.rc(pc_conv3x3_bn_relu_fwd_group(K, d, 1, B, H, W, Cin, Cout, X.st));
    };

    // first layer: Cin differs per stream -> one launch per stream over the networks (aligned DIRECT loader on the padded input)
    {
        int c0 = 0;
        for (int s = 0; s < 2; ++s) {
            const int cin = nets[0].n->s[s].cin;
            pc_src sa[MAXK];
            pc_dst dout[MAXK];
            pc_conv_fwd_desc d[MAXK];
            const Ten xin = chans(Xp, c0, cin);
            int n = 0;
            for (int i = 0; i < K; ++i) {
                if (keys[i].s != s) continue;
                a1[i] = A.act(B, 8, Hp, Wp);
                sa[n] = S(xin);
                dout[n] = D(a1[i]);
                memset(&d[n], 0, sizeof(d[n]));
                d[n].a = &sa[n]; d[n].w = lay(keys[i]).w[L_INC1]; d[n].bn = &lay(keys[i]).bn[L_INC1]; d[n].out = &dout[n];
                ++n;
            }
            if (X.go()) X.rc(pc_conv3x3_bn_relu_fwd_group(n, d, 1, B, Hp, Wp, cin, 8, X.st));
            c0 += cin;
        }
    }
    conv(L_INC2, a1, nullptr, a2, 8, Hp, Wp, PC_SRC_DIRECT, pa2);
    if (pa2[0].ok()) conv(L_D1A, pa2, nullptr, b1, 16, H1, W1, PC_SRC_DIRECT, nullptr);
    else conv(L_D1A, a2, nullptr, b1, 16, H1, W1, PC_SRC_POOL2, nullptr);
    conv(L_D1B, b1, nullptr, b2, 16, H1, W1, PC_SRC_DIRECT, pb2);

    // composed Up blocks (conv3x3 o convT from the low-resolution map, no up-sampled tensor): exact 2x geometry; with a backward pass
    // only where pc_conv3x3_up_bwd_ok takes the level
    auto fake = [&](int C, int h, int w) {          // a tensor of the arena's layout at an aligned address (for the *_ok predicates)
        Ten t;
        t.p = reinterpret_cast<float*>(A.base); t.B = B; t.C = C; t.H = h; t.W = w;
        t.sr = (w + 3) & ~3; t.sc = (int64_t)h * t.sr; t.sb = C * t.sc;
        return t;
    };
    auto can_compose = [&](int h, int w, int hz, int wz, int Cz) {
        if (h != 2 * hz || w != 2 * wz || (h & 3)) return false;
        const Ten sk = fake(Cz, h, w), z = fake(Cz, hz, wz), g = fake(8, h, w);
        const pc_src ss = S(sk), sz = S(z), sg = S(g);
        const pc_dst dz = D(z), dg = D(g);
        if (!pc_conv3x3_up_fwd_ok(&ss, &sz, &dg, h, w, Cz, Cz)) return false;
        return !any_save || pc_conv3x3_up_bwd_ok(&sg, &sz, &dz, h, w, Cz, Cz) != 0;
    };
    const bool compose2 = can_compose(H1, W1, H2, W2, 16);
    const bool compose1 = can_compose(Hp, Wp, H1, W1, 8);
    if (compose1 || compose2) {
        // the composed operand images of the levels in use, all (network, stream) pairs: one launch (they depend on the weights only)
        pc_conv_up_fwd_desc d[2 * MAXK];
        int cs[2 * MAXK], cz[2 * MAXK];
        int n = 0;
        if (compose2)
            for (int i = 0; i < K; ++i) {
                ws2[i] = A.raw(pc_conv3x3_up_ws_bytes(16));
                memset(&d[n], 0, sizeof(d[n]));
                d[n].w = lay(keys[i]).w[L_UP2A]; d[n].wt = lay(keys[i]).wt[T_UP2]; d[n].bt = lay(keys[i]).bt[T_UP2]; d[n].ws = ws2[i];
                cs[n] = 16; cz[n] = 16; ++n;
            }
        if (compose1)
            for (int i = 0; i < K; ++i) {
                ws1[i] = A.raw(pc_conv3x3_up_ws_bytes(8));
                memset(&d[n], 0, sizeof(d[n]));
                d[n].w = lay(keys[i]).w[L_UP1A]; d[n].wt = lay(keys[i]).wt[T_UP1]; d[n].bt = lay(keys[i]).bt[T_UP1]; d[n].ws = ws1[i];
                cs[n] = 8; cz[n] = 8; ++n;
            }
        if (X.go()) X.rc(pc_conv3x3_up_compose_group(n, d, cs, cz, X.st));
    }

    // the 32 x 32 level in one launch (down2's DoubleConv + up2's transposed conv with the maps in LDS)
    bool level2 = false;
    if (pb2[0].ok() && H2 == 32 && W2 == 32) {
        pc_src sx[MAXK];
        pc_dst d1[MAXK], d2[MAXK], du[MAXK];
        pc_level2_fwd_desc d[MAXK];
        level2 = true;
        const int64_t mark = A.off;
        for (int i = 0; i < K; ++i) {
            const bool sv = nets[keys[i].e].save;
            u2[i] = compose2 ? Ten{} : A.act(B, 16, 64, 64);
            c1[i] = sv ? A.act(B, 16, 32, 32) : Ten{};
            c2[i] = (sv || compose2) ? A.act(B, 16, 32, 32) : Ten{};
            sx[i] = S(pb2[i]);
            memset(&d[i], 0, sizeof(d[i]));
            d[i].x = &sx[i];
            d[i].w1 = lay(keys[i]).w[L_D2A]; d[i].bn1 = &lay(keys[i]).bn[L_D2A];
            d[i].w2 = lay(keys[i]).w[L_D2B]; d[i].bn2 = &lay(keys[i]).bn[L_D2B];
            d[i].wt = lay(keys[i]).wt[T_UP2]; d[i].bt = lay(keys[i]).bt[T_UP2];
            if (c1[i].ok()) { d1[i] = D(c1[i]); d[i].c1 = &d1[i]; }
            if (c2[i].ok()) { d2[i] = D(c2[i]); d[i].c2 = &d2[i]; }
            if (u2[i].ok()) { du[i] = D(u2[i]); d[i].u2 = &du[i]; }
            if (!pc_level2_fwd_ok(&sx[i], u2[i].ok() ? &du[i] : nullptr)) level2 = false;
        }
        if (level2) {
            if (X.go()) X.rc(pc_level2_fwd_group(K, d, B, X.st));
        } else {
            A.off = mark;
            for (int i = 0; i < K; ++i) c1[i] = c2[i] = u2[i] = Ten{};
        }
    }
    if (!level2) {
        if (pb2[0].ok()) conv(L_D2A, pb2, nullptr, c1, 16, H2, W2, PC_SRC_DIRECT, nullptr);
        else conv(L_D2A, b2, nullptr, c1, 16, H2, W2, PC_SRC_POOL2, nullptr);
        conv(L_D2B, c1, nullptr, c2, 16, H2, W2, PC_SRC_DIRECT, nullptr);
    }

    auto convt = [&](int T, const Ten* in, Ten* out, int C, int h, int w, int Hs, int Ws) {
        // ConvTranspose2d(C, C, 2, 2): (h, w) -> (2h, 2w), written into the top-left corner of a tensor with the extent and strides of
        // the skip map (Hs, Ws) it is concatenated with; rows / columns beyond 2h / 2w are the zero padding of Up (networks.py:309-312)
        pc_src sx[MAXK];
        pc_dst dout[MAXK];
        pc_convt_fwd_desc d[MAXK];
        for (int i = 0; i < K; ++i) {
            const Ten full = A.act(B, C, Hs, Ws);
            out[i] = crop(full, 2 * h, 2 * w);
            sx[i] = S(in[i]);
            dout[i] = D(out[i]);
            d[i] = pc_convt_fwd_desc{&sx[i], lay(keys[i]).wt[T], lay(keys[i]).bt[T], &dout[i]};
        }
        if (X.go()) X.rc(pc_convt2x2_fwd_group(K, d, B, h, w, C, X.st));
    };
    auto up_conv = [&](int L, int T, const Ten* skip, const Ten* z, Ten* out, void** ws, int h, int w) {
        pc_src ss[MAXK], sz[MAXK];
        pc_dst dout[MAXK];
        pc_conv_up_fwd_desc d[MAXK];
        for (int i = 0; i < K; ++i) {
            out[i] = A.act(B, 8, h, w);
            ss[i] = S(skip[i]); sz[i] = S(z[i]); dout[i] = D(out[i]);
            d[i] = pc_conv_up_fwd_desc{&ss[i], &sz[i], lay(keys[i]).w[L], lay(keys[i]).wt[T], lay(keys[i]).bt[T], &lay(keys[i]).bn[L], &dout[i], ws[i]};
        }
        if (X.go()) X.rc(pc_conv3x3_up_fwd_group(K, d, 1 | PC_UP_PRECOMPOSED, B, h, w, skip[0].C, z[0].C, X.st));
    };

    if (compose2) {
        up_conv(L_UP2A, T_UP2, b2, c2, e1, ws2, H1, W1);
    } else {
        if (!u2[0].ok()) convt(T_UP2, c2, u2, 16, H2, W2, H1, W1);
        conv(L_UP2A, b2, u2, e1, 8, H1, W1, PC_SRC_DIRECT, nullptr);
    }
    conv(L_UP2B, e1, nullptr, e2, 8, H1, W1, PC_SRC_DIRECT, nullptr);
    if (compose1) {
        up_conv(L_UP1A, T_UP1, a2, e2, f1, ws1, Hp, Wp);
    } else {
        convt(T_UP1, e2, u1, 8, H1, W1, Hp, Wp);
        conv(L_UP1A, a2, u1, f1, 8, Hp, Wp, PC_SRC_DIRECT, nullptr);
    }
    // last layer: the two streams write their halves of the feature map, or (logit_only) their partial fusion_out_conv logit
    {
        for (int e = 0; e < nE; ++e) feats[e] = nets[e].logit_only ? A.act(B, 2, Hp, Wp) : A.act(B, 16, Hp, Wp, /*dense=*/true);
        pc_src sa[MAXK];
        pc_dst dout[MAXK];
        pc_conv_fwd_desc d[MAXK];
        for (int i = 0; i < K; ++i) {
            const Net& nt = nets[keys[i].e];
            const int f0 = lay(keys[i]).feat_c0;
            sa[i] = S(f1[i]);
            memset(&d[i], 0, sizeof(d[i]));
            d[i].a = &sa[i]; d[i].w = lay(keys[i]).w[L_UP1B]; d[i].bn = &lay(keys[i]).bn[L_UP1B];
            if (nt.logit_only) {
                dout[i] = D(chans(feats[keys[i].e], f0 / 8, 1));
                d[i].dot_w = nt.n->fusion_w + f0;
                d[i].dot_out = &dout[i];
            } else {
                dout[i] = D(chans(feats[keys[i].e], f0, 8));
                d[i].out = &dout[i];
            }
        }
        if (X.go()) X.rc(pc_conv3x3_bn_relu_fwd_group(K, d, 1, B, Hp, Wp, 8, 8, X.st));
    }
    for (int i = 0; i < K; ++i) {
        if (!saved || !nets[keys[i].e].save) continue;
        Saved& sv = saved[keys[i].e][keys[i].s];
        sv.a1 = a1[i]; sv.a2 = a2[i]; sv.b1 = b1[i]; sv.b2 = b2[i]; sv.c1 = c1[i]; sv.c2 = c2[i]; sv.u2 = u2[i]; sv.e1 = e1[i];
        sv.e2 = e2[i]; sv.u1 = u1[i]; sv.f1 = f1[i]; sv.pa2 = pa2[i]; sv.pb2 = pb2[i]; sv.ws_up1 = ws1[i]; sv.ws_up2 = ws2[i];
    }
}

// the padded, normalised, stream-ordered input of a domain: (B, 6, H + top + bottom, W + left + right), rows padded to 16 bytes
Ten ingest(Step& X, const pc_step_io& io, int top, int bottom, int left, int right) {
    const pc_step_plan& P = X.plan;
    const Ten out = X.ar.act(io.B, 6, io.H + top + bottom, io.W + left + right);
    int sel[8];
    float mean[8], stdv[8];
    int n = 0;
    for (int s = 0; s < 2; ++s)
        for (int c = 0; c < P.unet.s[s].cin; ++c) {
            const int ch = P.unet.s[s].chan[c];                    // model channel ([R,G,B,NIR,VV,VH])
            sel[n] = io.data_kind == PC_DATA_RAW ? P.band[ch] : ch;
            mean[n] = P.mean[ch]; stdv[n] = P.stdv[ch];
            ++n;
        }
    const bool norm = io.data_kind != PC_DATA_INPUT;
    const int Cin = io.data_kind == PC_DATA_RAW ? io.craw : (io.data_kind == PC_DATA_SPLIT ? 4 : 6);
    if (X.go())
        X.rc(pc_ingest_pad_strided(io.data_kind, io.data, io.data2, Cin, out.p, out.sr, io.B, n, sel, norm ? mean : nullptr, norm ? stdv : nullptr,
                                   io.H, io.W, top, bottom, left, right, X.st));
    return out;
}

void forward(Step& X, pc_step_io& io) {
    const pc_step_plan& P = X.plan;
    Arena& A = X.ar;
    const int B = io.B, H = io.H, W = io.W;
    int pt, pb, pl, pr;
    pad_geometry(H, W, pt, pb, pl, pr);
    const int Hp = H + pt + pb, Wp = W + pl + pr;
    const int p = P.extractor_pad;
    X.B = B; X.H = H; X.W = W; X.pt = pt; X.pl = pl; X.Hp = Hp; X.Wp = Wp;
    X.unet_ng = io.unet_no_grad != 0;
    X.enc_ng = io.encoder_no_grad != 0 || X.unet_ng;
    // outputs first (fixed offsets whatever follows)
    X.popcount = reinterpret_cast<float*>(A.raw((int64_t)B * 4));
    io.off_popcount = reinterpret_cast<char*>(X.popcount) - A.base;
    X.popdense = reinterpret_cast<float*>(A.raw((int64_t)B * H * W * 4));
    io.off_popdense = reinterpret_cast<char*>(X.popdense) - A.base;
    X.scale_map = reinterpret_cast<float*>(A.raw((int64_t)B * H * W * 4));
    io.off_scale = reinterpret_cast<char*>(X.scale_map) - A.base;
    X.mask = reinterpret_cast<uint8_t*>(A.raw((int64_t)B * H * W));
    io.off_mask = reinterpret_cast<char*>(X.mask) - A.base;
    X.building = A.act(B, 1, H, W, /*dense=*/true);
    io.off_building = reinterpret_cast<char*>(X.building.p) - A.base;
    X.counts = reinterpret_cast<int32_t*>(A.raw(16));
    X.sel = io.sel;
    uint8_t* sel_dev = io.sel_host ? reinterpret_cast<uint8_t*>(A.raw(H + W)) : nullptr;
    auto unpack_sel = [&]() {          // (on the stream whose chain ends in the mask kernel)
        if (!sel_dev) return;
        X.sel = sel_dev;
        if (!X.go()) return;
        SelBits bits;
        memset(&bits, 0, sizeof(bits));
        for (int i = 0; i < H + W; ++i)
            if (io.sel_host[i]) bits.w[i >> 5] |= 1u << (i & 31);
        hipLaunchKernelGGL(sel_unpack_kernel, dim3(1), dim3(256), 0, X.st, bits, sel_dev, H + W);
        X.rc((int)hipGetLastError());
    };
    X.g_pc = reinterpret_cast<float*>(A.raw((int64_t)B * 4));
    X.head_ws = A.raw(pc_head_ws_bytes(B, H, W));

    const bool same_domain = pt == p && pb == p && pl == p && pr == p;     // e.g. 100 x 100 tiles: both networks on the 128 x 128 domain
    Ten feats[2];
    Saved (*sv)[2] = nullptr;
    Saved svbuf[2][2];
    const pc_src* featsrc = nullptr;
    pc_src fs{};
    const float* am = io.admin_mask;
    if (same_domain) {
        unpack_sel();
        const Ten Xp = ingest(X, io, pt, pb, pl, pr);
        X.Xp_u = Xp;
        const Net nets[2] = {Net{&P.extractor, false, true}, Net{&P.unet, !X.unet_ng, false}};
        sv = svbuf;
        forward_nets(X, nets, 2, Xp, Hp, Wp, feats, sv);
        X.feats = feats[1];
        X.sv[0] = svbuf[1][0]; X.sv[1] = svbuf[1][1];
        fs = S(feats[0]);
        const pc_dst db = D(X.building);
        if (X.go())
            X.rc(pc_building_score_mask(&fs, X.ones2, P.extractor.fusion_b, &db, am, io.census_idx, X.sel, X.sel + H, P.occupancymodel, X.mask,
                                        X.counts, B, H, W, pt, pl, X.st));
    } else {
        // the frozen extractor on its own 14-pixel reflect-padded domain (popcorn.py:279-322)
        const bool two = X.side != nullptr && (int64_t)B * Hp * Wp <= X.side_px;
        hipStream_t main_st = X.st;
        const int64_t mark = A.off;
        if (two && X.go()) {
            // fork: the side chain starts behind everything already enqueued on the caller's stream (the batch's copies, the previous step)
            X.rc((int)hipEventRecord(X.ev_fork, main_st));
            X.rc((int)hipStreamWaitEvent(X.side, X.ev_fork, 0));
        }
        if (two) X.st = X.side;
        {
            unpack_sel();
            const Ten Xb = ingest(X, io, p, p, p, p);
            const Net nb[1] = {Net{&P.extractor, false, true}};
            forward_nets(X, nb, 1, Xb, H + 2 * p, W + 2 * p, feats, nullptr);
            fs = S(feats[0]);
            const pc_dst db = D(X.building);
            if (X.go())
                X.rc(pc_building_score_mask(&fs, X.ones2, P.extractor.fusion_b, &db, am, io.census_idx, X.sel, X.sel + H, P.occupancymodel,
                                            X.mask, X.counts, B, H, W, p, p, X.st));
        }
        if (two) {
            if (X.go()) X.rc((int)hipEventRecord(X.ev_join, X.side));
            X.st = main_st;
        } else {
            A.off = mark;      // its activations are released (stream order: everything that follows is enqueued behind the extractor's launches)
        }
        const Ten Xp = ingest(X, io, pt, pb, pl, pr);
        X.Xp_u = Xp;
        const Net nu[1] = {Net{&P.unet, !X.unet_ng, false}};
        sv = svbuf;
        forward_nets(X, nu, 1, Xp, Hp, Wp, feats, sv);
        X.feats = feats[0];
        X.sv[0] = svbuf[0][0]; X.sv[1] = svbuf[0][1];
        if (two && X.go()) X.rc((int)hipStreamWaitEvent(main_st, X.ev_join, 0));      // join: the head needs the building score and the mask
    }
    (void)featsrc;
    // sparse head + occupancy product + census sums (popcorn.py:161-190,195-228)
    const float* bld = X.building.p;
    Ten ones{};
    if (!P.occupancymodel) {
        // popcorn.py:179-181: popdensemap = relu(out), no building product
        return X.rc(PC_ENOTSUP);
    }
    (void)ones;
    const pc_src sf = S(X.feats);
    X.deferred_popcount = io.dp == 0;
    if (X.go())
        X.rc(pc_head_fwd(&sf, pt, pl, P.head_w, X.mask, bld, am, io.census_idx, X.scale_map, X.popdense, X.popcount, P.stats_dev, X.counts,
                         X.head_ws, B, H, W, PC_HEAD_FWD_PACK_BOTH | (X.deferred_popcount ? PC_HEAD_FWD_DEFER_REDUCE : 0), X.st));
    X.launches += 1;           // (pack + kernel)
    X.fwd_end = A.off;
    X.have_fwd = true;
}

// ---- backward -----------------------------------------------------------------------------------------------------------------------
struct Reduce {                // the batched second stage of all weight gradients of a backward pass (ops.py: WgradBatch)
    pc_wgrad_reduce_desc e[40];
    int n = 0;
    float* head_ptrs[8];
    int64_t slot_bytes = 0;
    // composed Up blocks: chain-rule launches behind the reduction
    pc_conv_up_bwd_desc ch8[2], ch16[2];
    pc_src ch_g[2][2], ch_z[2][2];
    pc_dst ch_gz[2][2];
    int n8 = 0, n16 = 0, nwg8 = 0, nwg16 = 0;
};

void backward(Step& X, pc_step_io& io) {
    const pc_step_plan& P = X.plan;
    Arena& A = X.ar;
    const int B = X.B, H = X.H, W = X.W, Hp = X.Hp, Wp = X.Wp;
    const int H1 = Hp / 2, W1 = Wp / 2, H2 = H1 / 2, W2 = W1 / 2;
    A.off = X.fwd_end;
    // loss forward + backward (utils/losses.py:49-76); single process: also finishes popcount / stats of the head forward
    if (X.deferred_popcount) {
        if (X.go())
            X.rc(pc_head_popcount_loss(X.head_ws, B, H, W, X.counts, io.y, P.lam4, P.scale_regularization, P.lam_weak, io.inv_B, X.popcount,
                                       P.stats_dev, P.loss_dev, X.g_pc, P.g_scale_const_dev, X.st));
    } else if (X.go()) {
        X.rc(pc_loss_fwd_bwd(X.popcount, io.y, P.stats_dev, P.lam4, P.scale_regularization, P.lam_weak, io.inv_B, B, P.loss_dev, X.g_pc,
                             P.g_scale_const_dev, X.st));
    }
    // head backward: the 8 head gradients (partials, finished by the batched reduction below) + dL/d(conv outputs of the up1b layers)
    const Ten G = A.act(B, 16, Hp, Wp, /*dense=*/true);
    {
        const pc_src sf = S(X.feats);
        const pc_dst dg = D(G);
        const bool defer = !X.unet_ng;
        if (X.go())
            X.rc(pc_head_bwd(&sf, X.pt, X.pl, P.head_w, X.mask, X.building.p, io.admin_mask, io.census_idx, X.g_pc, nullptr, nullptr,
                             P.g_scale_const_dev, P.head_dw, 0, &dg, X.unet_ng ? nullptr : &X.bn_nobias[0][L_UP1B],
                             X.unet_ng ? nullptr : &X.bn_nobias[1][L_UP1B], Hp, Wp, X.head_ws, B, H, W,
                             PC_HEAD_BWD_PACKED | (defer ? PC_HEAD_BWD_DEFER_REDUCE : 0), X.st));
        if (!defer) X.launches += 1;
    }
    const int n_unet = P.n - P.n_head;
    if (X.unet_ng || X.enc_ng) {
        // parameters without a gradient in this regime keep exact zeros in the flat buffer (the clip norm runs over all of it)
        if (X.go()) X.rc(pc_zero_fill(P.flat_g, n_unet, X.st));
    }
    if (X.unet_ng) return;

    Reduce R;
    R.slot_bytes = pc_conv3x3_wgrad_ws_bytes(32, 8);
    if (pc_convt2x2_wgrad_ws_bytes(16) > R.slot_bytes) R.slot_bytes = pc_convt2x2_wgrad_ws_bytes(16);
    auto slot = [&](int64_t bytes = 0) { return A.raw(bytes > R.slot_bytes ? bytes : R.slot_bytes); };
    auto entry = [&](const void* partial, float* dw, float* db, int nwg, int Cin, int Cout, int kind, int co_stride = 0, int col0 = 0) {
        pc_wgrad_reduce_desc& e = R.e[R.n++];
        memset(&e, 0, sizeof(e));
        e.partial = reinterpret_cast<const float*>(partial);
        e.dw = dw + col0; e.db = db; e.nwg = nwg; e.Cin = Cin; e.Cout = Cout; e.kind = kind; e.accumulate = 0; e.dw_co_stride = co_stride;
    };
    {
        // the head's partials: kind 3
        const float* pp = nullptr;
        int nwg = 0;
        pc_head_bwd_partials(X.head_ws, B, H, W, &pp, &nwg);
        pc_wgrad_reduce_desc& e = R.e[R.n++];
        memset(&e, 0, sizeof(e));
        for (int t = 0; t < 8; ++t) R.head_ptrs[t] = P.head_dw[t];
        e.partial = pp; e.dw = reinterpret_cast<float*>(R.head_ptrs); e.nwg = nwg; e.kind = 3;
    }
    const pc_step_stream* ST = P.unet.s;
    const Saved* sv = X.sv;
    const bool enc_ng = X.enc_ng;

    // data + weight gradient of an 8 -> 8 layer (or of an 8-channel column block of a wider one) in ONE launch, both streams
    struct Blk { const Ten* g; const Ten* x; const pc_bn* x_bn; Ten* out; int L; int s; int c0_add; bool with_db; const Ten* pool_act = nullptr; };
    auto bwd8 = [&](const Blk* blk, int n, int cin_total, int h, int w) {
        pc_src sg[MAXK], sx[MAXK], spa[MAXK];
        pc_dst dout[MAXK];
        pc_conv_bwd_desc d[MAXK];
        void* ws[MAXK];
        for (int i = 0; i < n; ++i) {
            sg[i] = S(*blk[i].g); sx[i] = S(*blk[i].x); dout[i] = D(*blk[i].out);
            ws[i] = slot();
            memset(&d[i], 0, sizeof(d[i]));
            d[i].g = &sg[i]; d[i].x = &sx[i]; d[i].w = ST[blk[i].s].w[blk[i].L]; d[i].x_bn = blk[i].x_bn; d[i].out = &dout[i]; d[i].ws = ws[i];
            d[i].c0_add = blk[i].c0_add;
            if (blk[i].pool_act) { spa[i] = S(*blk[i].pool_act); d[i].pool_act = &spa[i]; }     // Down block: scatter (+=) into the full-resolution gradient
        }
        int nwg = 0;
        if (X.go()) X.rc(pc_conv3x3_bwd_group(n, d, cin_total, 0, blk[0].pool_act ? 1 : 0, B, h, w, &nwg, X.st));
        for (int i = 0; i < n; ++i)
            entry(ws[i], ST[blk[i].s].dw[blk[i].L], blk[i].with_db ? ST[blk[i].s].db[blk[i].L] : nullptr, nwg, 8, blk[i].g->C, 0, cin_total * 9,
                  blk[i].c0_add * 9);
    };
    // does the fused launch take this (gradient, input block, output, pooled-from) combination?  (split-operand form, aligned fp32 tensors)
    auto bwd_ok = [&](const Ten& g, const Ten& x, const Ten& out, const Ten* pool_act, int h, int w) {
        const pc_src sg = S(g), sx = S(x);
        const pc_dst dout = D(out);
        pc_src spa;
        if (pool_act) spa = S(*pool_act);
        return pc_conv3x3_bwd_ok(&sg, &sx, &dout, pool_act ? &spa : nullptr, B, h, w) != 0;
    };
    // The weight-gradient-only launches (16-channel encoder layers, first layers) run on the side stream next to the data-gradient chain:
    // they only feed the batched reduction at the end.  side_on(): everything enqueued on the caller's stream so far (the producers of the
    // launch's operands) is ordered in front of the side stream's next launch.
    hipStream_t main_st = X.st;
    const bool two = X.side != nullptr && (int64_t)B * Hp * Wp <= X.side_bwd_px;
    bool side_used = false;
    auto side_on = [&]() {
        if (!two) return;
        if (X.go()) {
            hipEvent_t ev = X.ev_dep[X.ev_next];
            X.ev_next = (X.ev_next + 1) & 7;
            X.rc((int)hipEventRecord(ev, main_st));
            X.rc((int)hipStreamWaitEvent(X.side, ev, 0));
        }
        X.st = X.side;
        side_used = true;
    };
    auto side_off = [&]() { X.st = main_st; };
    // grouped weight gradient of layer L over both streams: x = cat[a, b]
    auto wgrad = [&](int L, const Ten* a, int mode, const Ten* b, const Ten* g, int Cout, int h, int w, int cin_total) {
        side_on();
        pc_src sa[2], sb[2], sg[2];
        pc_conv_wgrad_desc d[2];
        void* ws[2];
        int Cin = 0;
        for (int s = 0; s < 2; ++s) {
            Ten src = a[s];
            if (mode == PC_SRC_POOL2) src = crop(src, 2 * h, 2 * w);
            sa[s] = S(src, mode); sg[s] = S(g[s]);
            ws[s] = slot();
            d[s] = pc_conv_wgrad_desc{&sa[s], nullptr, &sg[s], ws[s]};
            Cin = src.C;
            if (b) { sb[s] = S(b[s]); d[s].b = &sb[s]; Cin += b[s].C; }
        }
        int nwg = 0;
        if (X.go()) X.rc(pc_conv3x3_wgrad_partial_group(2, d, B, h, w, Cin, Cout, &nwg, X.st));
        for (int s = 0; s < 2; ++s) entry(ws[s], ST[s].dw[L], ST[s].db[L], nwg, Cin, Cout, 0, cin_total ? cin_total * 9 : 0, 0);
        side_off();
    };
    struct Dg { const Ten* g; int s; int L; const Ten* act; const pc_bn* act_bn; Ten* out; int c0_add; };
    auto dgrad = [&](const Dg* q, int n, int Cin_total, int c0, int cn, int pool, int acc, int h, int w, int Cg) {
        pc_src sg[MAXK], sa[MAXK];
        pc_dst dout[MAXK];
        pc_conv_dgrad_desc d[MAXK];
        for (int i = 0; i < n; ++i) {
            sg[i] = S(*q[i].g); dout[i] = D(*q[i].out);
            memset(&d[i], 0, sizeof(d[i]));
            d[i].g = &sg[i]; d[i].w = ST[q[i].s].w[q[i].L] + 9 * q[i].c0_add; d[i].out = &dout[i];
            if (q[i].act) { sa[i] = S(*q[i].act); d[i].act = &sa[i]; d[i].act_bn = q[i].act_bn; }
        }
        if (X.go()) X.rc(pc_conv3x3_dgrad_group(n, d, Cin_total, c0, cn, pool, acc, B, h, w, Cg, X.st));
    };
    // composed Up block: dz, the composed weight gradient and the border sums from one pass over g (up_bwd.hip)
    auto up_bwd = [&](int T, int L, const Ten* g, const Ten* z, int zL, Ten* gz, void* const* fwd_ws, int h, int w, int Cz) {
        pc_conv_up_bwd_desc* d = Cz == 8 ? R.ch8 : R.ch16;
        const int lv = Cz == 8 ? 0 : 1;
        void* ws[2];
        for (int s = 0; s < 2; ++s) {
            R.ch_g[lv][s] = S(g[s]); R.ch_z[lv][s] = S(z[s]);
            if (gz) R.ch_gz[lv][s] = D(gz[s]);
            ws[s] = slot(pc_conv3x3_up_bwd_ws_bytes(B, h, Cz));
            memset(&d[s], 0, sizeof(d[s]));
            d[s].g = &R.ch_g[lv][s]; d[s].z = &R.ch_z[lv][s]; d[s].z_bn = &X.bn_nobias[s][zL]; d[s].gz = gz ? &R.ch_gz[lv][s] : nullptr;
            d[s].w = ST[s].w[L]; d[s].wt = ST[s].wt[T]; d[s].bt = ST[s].bt[T]; d[s].fwd_ws = fwd_ws[s]; d[s].ws = ws[s];
            d[s].dw = ST[s].dw[L]; d[s].dwt = ST[s].dwt[T]; d[s].dbt = ST[s].dbt[T];
        }
        int nwg = 0, part = 0;
        if (X.go()) X.rc(pc_conv3x3_up_bwd_partial_group(2, d, B, h, w, Cz, Cz, &nwg, &part, X.st));
        else { nwg = 1; part = 1; }
        for (int s = 0; s < 2; ++s) {
            pc_wgrad_reduce_desc& e = R.e[R.n++];
            memset(&e, 0, sizeof(e));
            e.partial = reinterpret_cast<const float*>(ws[s]);
            e.dw = reinterpret_cast<float*>(ws[s]) + (int64_t)nwg * part;
            e.nwg = nwg; e.Cin = part; e.kind = 2;
        }
        if (Cz == 8) { R.n8 = 2; R.nwg8 = nwg; } else { R.n16 = 2; R.nwg16 = nwg; }
    };
    // transposed conv T: weight gradient + data gradient (masked by x's producer xL) -- one launch when the fused form applies
    auto ct_bwd = [&](int T, const Ten* x, int xL, const Ten* g, Ten* out, int h, int w, int C, bool want_dx) {
        pc_src sx[2], sg[2];
        pc_dst dout[2];
        void* ws[2];
        bool fused = want_dx && (w % 16 == 0);
        for (int s = 0; s < 2; ++s) {
            sx[s] = S(x[s]); sg[s] = S(g[s]);
            if (want_dx) dout[s] = D(out[s]);
            ws[s] = slot();
        }
        int nwg = 0;
        if (fused) {
            pc_convt_bwd_desc d[2];
            for (int s = 0; s < 2; ++s) d[s] = pc_convt_bwd_desc{&sx[s], &sg[s], ST[s].wt[T], &X.bn_nobias[s][xL], &dout[s], ws[s]};
            if (X.go()) X.rc(pc_convt2x2_bwd_group(2, d, B, h, w, C, &nwg, X.st));
        } else {
            pc_convt_wgrad_desc d[2];
            for (int s = 0; s < 2; ++s) d[s] = pc_convt_wgrad_desc{&sx[s], &sg[s], ws[s]};
            if (X.go()) X.rc(pc_convt2x2_wgrad_partial_group(2, d, B, h, w, C, &nwg, X.st));
            if (want_dx) {
                pc_convt_dgrad_desc q[2];
                for (int s = 0; s < 2; ++s) q[s] = pc_convt_dgrad_desc{&sg[s], ST[s].wt[T], &sx[s], &X.bn_nobias[s][xL], &dout[s]};
                if (X.go()) X.rc(pc_convt2x2_dgrad_group(2, q, B, h, w, C, X.st));
            }
        }
        for (int s = 0; s < 2; ++s) entry(ws[s], ST[s].dwt[T], ST[s].dbt[T], nwg, C, C, 1);
    };

    Ten G_f2[2], G_f1[2], G_e2[2], G_a2[2], G_e1[2], G_b2[2], G_c2[2], G_c1[2], G_b1[2], G_a1[2], g_u1[2], g_u2[2];
    for (int s = 0; s < 2; ++s) G_f2[s] = chans(G, ST[s].feat_c0, 8);
    // up1b
    {
        Blk b[2];
        for (int s = 0; s < 2; ++s) {
            G_f1[s] = A.act(B, 8, Hp, Wp);
            b[s] = Blk{&G_f2[s], &sv[s].f1, &X.bn_nobias[s][L_UP1A], &G_f1[s], L_UP1B, s, 0, true};
        }
        bwd8(b, 2, 8, Hp, Wp);
    }
    for (int s = 0; s < 2; ++s) G_e2[s] = A.act(B, 8, H1, W1);
    const bool composed1 = sv[0].ws_up1 != nullptr, composed2 = sv[0].ws_up2 != nullptr;
    if (composed1) {
        Blk b[2];
        for (int s = 0; s < 2; ++s) {
            G_a2[s] = A.act(B, 8, Hp, Wp);
            b[s] = Blk{&G_f1[s], &sv[s].a2, &X.bn_nobias[s][L_INC2], &G_a2[s], L_UP1A, s, 0, true};
        }
        bwd8(b, 2, 16, Hp, Wp);
        Ten z[2] = {sv[0].e2, sv[1].e2};
        void* fw[2] = {sv[0].ws_up1, sv[1].ws_up1};
        up_bwd(T_UP1, L_UP1A, G_f1, z, L_UP2B, G_e2, fw, Hp, Wp, 8);
    } else {
        if (!enc_ng) {
            // both column blocks of the concat layer (networks.py:318: [skip | up]) in ONE launch over the same gradient
            Blk b[4];
            for (int s = 0; s < 2; ++s) {
                G_a2[s] = A.act(B, 8, Hp, Wp);
                g_u1[s] = A.act(B, 8, Hp, Wp);
                b[s] = Blk{&G_f1[s], &sv[s].a2, &X.bn_nobias[s][L_INC2], &G_a2[s], L_UP1A, s, 0, true};
            }
            Ten u1full[2];
            for (int s = 0; s < 2; ++s) {
                u1full[s] = crop(sv[s].u1, Hp, Wp);
                b[2 + s] = Blk{&G_f1[s], &u1full[s], nullptr, &g_u1[s], L_UP1A, s, 8, false};
            }
            bwd8(b, 4, 16, Hp, Wp);
        } else {
            Ten a[2] = {sv[0].a2, sv[1].a2}, u[2] = {crop(sv[0].u1, Hp, Wp), crop(sv[1].u1, Hp, Wp)};
            wgrad(L_UP1A, a, PC_SRC_DIRECT, u, G_f1, 8, Hp, Wp, 0);
            Dg q[2];
            for (int s = 0; s < 2; ++s) {
                g_u1[s] = A.act(B, 8, Hp, Wp);
                q[s] = Dg{&G_f1[s], s, L_UP1A, nullptr, nullptr, &g_u1[s], 0};
            }
            dgrad(q, 2, 16, 8, 8, 0, 0, Hp, Wp, 8);
        }
        Ten x[2] = {sv[0].e2, sv[1].e2};
        Ten gv[2] = {crop(g_u1[0], 2 * H1, 2 * W1), crop(g_u1[1], 2 * H1, 2 * W1)};
        ct_bwd(T_UP1, x, L_UP2B, gv, G_e2, H1, W1, 8, true);
    }
    // up2b
    {
        Blk b[2];
        for (int s = 0; s < 2; ++s) {
            G_e1[s] = A.act(B, 8, H1, W1);
            b[s] = Blk{&G_e2[s], &sv[s].e1, &X.bn_nobias[s][L_UP2A], &G_e1[s], L_UP2B, s, 0, true};
        }
        bwd8(b, 2, 8, H1, W1);
    }
    if (composed2) {
        if (!enc_ng) {
            // the two 8-channel halves of the 16-channel skip tensor: two problems per stream over the same gradient, each writing its half
            // of the data gradient and its column block of the weight gradient
            Blk b[4];
            Ten xh[2][2], oh[2][2];
            for (int s = 0; s < 2; ++s) {
                G_b2[s] = A.act(B, 16, H1, W1);
                for (int i = 0; i < 2; ++i) {
                    xh[s][i] = chans(sv[s].b2, 8 * i, 8);
                    oh[s][i] = chans(G_b2[s], 8 * i, 8);
                    b[2 * s + i] = Blk{&G_e1[s], &xh[s][i], &X.bn_half[s][i], &oh[s][i], L_UP2A, s, 8 * i, i == 0};
                }
            }
            bwd8(b, 4, 32, H1, W1);
            for (int s = 0; s < 2; ++s) G_c2[s] = A.act(B, 16, H2, W2);
        } else {
            Ten a[2] = {sv[0].b2, sv[1].b2};
            wgrad(L_UP2A, a, PC_SRC_DIRECT, nullptr, G_e1, 8, H1, W1, 32);
        }
        Ten z[2] = {sv[0].c2, sv[1].c2};
        void* fw[2] = {sv[0].ws_up2, sv[1].ws_up2};
        up_bwd(T_UP2, L_UP2A, G_e1, z, L_D2B, enc_ng ? nullptr : G_c2, fw, H1, W1, 16);
    } else {
        Ten a[2] = {sv[0].b2, sv[1].b2}, u[2] = {crop(sv[0].u2, H1, W1), crop(sv[1].u2, H1, W1)};
        wgrad(L_UP2A, a, PC_SRC_DIRECT, u, G_e1, 8, H1, W1, 0);
        if (!enc_ng) {
            Dg q[4];
            for (int s = 0; s < 2; ++s) {
                G_b2[s] = A.act(B, 16, H1, W1);
                g_u2[s] = A.act(B, 16, H1, W1);
                q[s] = Dg{&G_e1[s], s, L_UP2A, &sv[s].b2, &X.bn_nobias[s][L_D1B], &G_b2[s], 0};
                q[2 + s] = Dg{&G_e1[s], s, L_UP2A, nullptr, nullptr, &g_u2[s], 16};
            }
            dgrad(q, 4, 32, 0, 16, 0, 0, H1, W1, 8);
        } else {
            Dg q[2];
            for (int s = 0; s < 2; ++s) {
                g_u2[s] = A.act(B, 16, H1, W1);
                q[s] = Dg{&G_e1[s], s, L_UP2A, nullptr, nullptr, &g_u2[s], 0};
            }
            dgrad(q, 2, 32, 16, 16, 0, 0, H1, W1, 8);
        }
        Ten x[2] = {sv[0].c2, sv[1].c2};
        Ten gv[2] = {crop(g_u2[0], 2 * H2, 2 * W2), crop(g_u2[1], 2 * H2, 2 * W2)};
        if (!enc_ng)
            for (int s = 0; s < 2; ++s) G_c2[s] = A.act(B, 16, H2, W2);
        ct_bwd(T_UP2, x, L_D2B, gv, G_c2, H2, W2, 16, !enc_ng);
    }
    if (!enc_ng) {
        // encoder.  The 32 x 32 level: both weight gradients, the data gradient chain d2b -> d2a and the pooling scatter in one launch
        bool l2 = sv[0].pb2.ok() && H2 == 32 && W2 == 32;
        if (l2) {
            pc_src sg[2], sc[2], sx[2], sa[2];
            pc_dst dout[2];
            pc_level2_bwd_desc d[2];
            void* w1[2];
            void* w2[2];
            const int64_t mark = A.off;
            for (int s = 0; s < 2; ++s) {
                sg[s] = S(G_c2[s]); sc[s] = S(sv[s].c1); sx[s] = S(sv[s].pb2); sa[s] = S(sv[s].b2); dout[s] = D(G_b2[s]);
                w1[s] = slot(pc_level2_bwd_ws_bytes(B)); w2[s] = slot(pc_level2_bwd_ws_bytes(B));
                d[s] = pc_level2_bwd_desc{&sg[s], &sc[s], &sx[s], ST[s].w[L_D2A], ST[s].w[L_D2B], &X.bn_nobias[s][L_D2A], &sa[s],
                                          &X.bn_nobias[s][L_D1B], &dout[s], w1[s], w2[s]};
                if (!pc_level2_bwd_ok(&sg[s], &sc[s], &sx[s], &sa[s], &dout[s])) l2 = false;
            }
            if (l2) {
                int nwg = 0;
                if (X.go()) X.rc(pc_level2_bwd_group(2, d, B, &nwg, X.st));
                for (int s = 0; s < 2; ++s) {
                    entry(w1[s], ST[s].dw[L_D2A], ST[s].db[L_D2A], nwg, 16, 16, 0);
                    entry(w2[s], ST[s].dw[L_D2B], ST[s].db[L_D2B], nwg, 16, 16, 0);
                }
            } else {
                A.off = mark;
            }
        }
        if (!l2) {
            Ten c1[2] = {sv[0].c1, sv[1].c1};
            wgrad(L_D2B, c1, PC_SRC_DIRECT, nullptr, G_c2, 16, H2, W2, 0);
            Dg q[2];
            for (int s = 0; s < 2; ++s) {
                G_c1[s] = A.act(B, 16, H2, W2);
                q[s] = Dg{&G_c2[s], s, L_D2B, &sv[s].c1, &X.bn_nobias[s][L_D2A], &G_c1[s], 0};
            }
            dgrad(q, 2, 16, 0, 16, 0, 0, H2, W2, 16);
            if (sv[0].pb2.ok()) {
                Ten x[2] = {sv[0].pb2, sv[1].pb2};
                wgrad(L_D2A, x, PC_SRC_DIRECT, nullptr, G_c1, 16, H2, W2, 0);
            } else {
                Ten x[2] = {sv[0].b2, sv[1].b2};
                wgrad(L_D2A, x, PC_SRC_POOL2, nullptr, G_c1, 16, H2, W2, 0);
            }
            for (int s = 0; s < 2; ++s) q[s] = Dg{&G_c1[s], s, L_D2A, &sv[s].b2, &X.bn_nobias[s][L_D1B], &G_b2[s], 0};
            dgrad(q, 2, 16, 0, 16, 1, 1, H2, W2, 16);
        }
        for (int s = 0; s < 2; ++s) G_b1[s] = A.act(B, 16, H1, W1);
        // down1 (16 channels @ H1 x W1), round 6: data + weight gradient of each layer in ONE launch of the split-operand kernel -- d1b as the two
        // 8-channel halves of its input over the same 16-channel gradient (two problems per stream), d1a with the max-pool scatter into
        // inc2's gradient -- instead of four launches (weight gradient 16 -> 16, data gradient 16 -> 16, weight gradient 8 -> 16, pooled
        // data gradient 16 -> 8)
        bool fused_d1 = sv[0].pa2.ok() && 4 <= MAXK;
        for (int s = 0; s < 2 && fused_d1; ++s) {
            const Ten xh = chans(sv[s].b1, 8, 8), oh = chans(G_b1[s], 8, 8);
            fused_d1 = bwd_ok(G_b2[s], xh, oh, nullptr, H1, W1) && bwd_ok(G_b1[s], sv[s].pa2, G_a2[s], &sv[s].a2, H1, W1);
        }
        if (fused_d1) {
            Blk b[4];
            Ten xh[2][2], oh[2][2];
            for (int s = 0; s < 2; ++s)
                for (int i = 0; i < 2; ++i) {
                    xh[s][i] = chans(sv[s].b1, 8 * i, 8);
                    oh[s][i] = chans(G_b1[s], 8 * i, 8);
                    b[2 * s + i] = Blk{&G_b2[s], &xh[s][i], &X.bn_half_d1a[s][i], &oh[s][i], L_D1B, s, 8 * i, i == 0};
                }
            bwd8(b, 4, 16, H1, W1);
            Blk a[2];
            for (int s = 0; s < 2; ++s) a[s] = Blk{&G_b1[s], &sv[s].pa2, &X.bn_nobias[s][L_INC2], &G_a2[s], L_D1A, s, 0, true, &sv[s].a2};
            bwd8(a, 2, 8, H1, W1);
        } else {
            Ten x[2] = {sv[0].b1, sv[1].b1};
            wgrad(L_D1B, x, PC_SRC_DIRECT, nullptr, G_b2, 16, H1, W1, 0);
            Dg q[2];
            for (int s = 0; s < 2; ++s) q[s] = Dg{&G_b2[s], s, L_D1B, &sv[s].b1, &X.bn_nobias[s][L_D1A], &G_b1[s], 0};
            dgrad(q, 2, 16, 0, 16, 0, 0, H1, W1, 16);
            if (sv[0].pa2.ok()) {
                Ten xa[2] = {sv[0].pa2, sv[1].pa2};
                wgrad(L_D1A, xa, PC_SRC_DIRECT, nullptr, G_b1, 16, H1, W1, 0);
            } else {
                Ten xa[2] = {sv[0].a2, sv[1].a2};
                wgrad(L_D1A, xa, PC_SRC_POOL2, nullptr, G_b1, 16, H1, W1, 0);
            }
            for (int s = 0; s < 2; ++s) q[s] = Dg{&G_b1[s], s, L_D1A, &sv[s].a2, &X.bn_nobias[s][L_INC2], &G_a2[s], 0};
            dgrad(q, 2, 8, 0, 8, 1, 1, H1, W1, 16);
        }
        {
            Blk b[2];
            for (int s = 0; s < 2; ++s) {
                G_a1[s] = A.act(B, 8, Hp, Wp);
                b[s] = Blk{&G_a2[s], &sv[s].a1, &X.bn_nobias[s][L_INC1], &G_a1[s], L_INC2, s, 0, true};
            }
            bwd8(b, 2, 8, Hp, Wp);
        }
        // first layers: one launch per stream (2 / 4 input channels) over the padded input
        int c0 = 0;
        side_on();
        for (int s = 0; s < 2; ++s) {
            const Ten xin = chans(X.Xp_u, c0, ST[s].cin);
            const pc_src sa = S(xin), sg = S(G_a1[s]);
            void* ws = slot();
            int nwg = 0;
            if (X.go()) X.rc(pc_conv3x3_wgrad_partial(&sa, nullptr, &sg, ws, B, Hp, Wp, ST[s].cin, 8, &nwg, X.st));
            entry(ws, ST[s].dw[L_INC1], ST[s].db[L_INC1], nwg, ST[s].cin, 8, 0);
            c0 += ST[s].cin;
        }
        side_off();
    }
    if (side_used && X.go()) {         // join: the reduction reads the side stream's partials
        X.rc((int)hipEventRecord(X.ev_join, X.side));
        X.rc((int)hipStreamWaitEvent(main_st, X.ev_join, 0));
    }
    // ONE batched fixed-order reduction for all layers (+ the head), then the chain rule of the composed levels
    if (X.go()) X.rc(pc_wgrad_reduce_batch(R.n, R.e, X.st));
    if (R.n8 && R.n16) {
        if (X.go()) X.rc(pc_conv3x3_up_chain_both(R.n8, R.ch8, R.nwg8, R.n16, R.ch16, R.nwg16, 0, X.st));
    } else {
        if (R.n8 && X.go()) X.rc(pc_conv3x3_up_chain_group(R.n8, R.ch8, 0, R.nwg8, 8, 8, X.st));
        if (R.n16 && X.go()) X.rc(pc_conv3x3_up_chain_group(R.n16, R.ch16, 0, R.nwg16, 16, 16, X.st));
    }
}

void update(Step& X) {
    const pc_step_plan& P = X.plan;
    pc_adam_groups g = P.groups;
    g.active_mask = X.unet_ng ? 4 : (X.enc_ng ? 6 : 7);          // groups: 0 = encoder, 1 = decoder, 2 = head
    if (X.go())
        X.rc(pc_adam_clip_step_fused(P.flat_p, P.flat_g, P.adam_m, P.adam_v, P.n, P.n_decay, P.hyper_dev, P.weight_decay, P.beta1, P.beta2,
                                     P.eps, P.max_norm, P.norm_dev, P.step_dev, &g, X.st));
}

int run(Step& X, pc_step_io& io, int phases, bool dry) {
    X.ar.dry = dry;
    X.ar.base = dry ? reinterpret_cast<char*>((uintptr_t)1 << 40) : reinterpret_cast<char*>(io.arena);
    X.ar.cap = io.arena_bytes;
    X.ar.peak = 0;
    X.err = 0;
    X.launches = 0;
    if (phases & PC_STEP_FWD) {
        X.ar.off = 0;
        forward(X, io);
    }
    if ((phases & PC_STEP_BWD) && X.err == 0) {
        if (!X.have_fwd) return PC_EINVAL;
        backward(X, io);
    }
    if ((phases & PC_STEP_UPD) && X.err == 0) update(X);
    return X.err;
}

}  // namespace

namespace {
// ---- the side stream must not share a hardware queue with the caller's stream -----------------------------------------------------
// HIP multiplexes its streams onto a handful of hardware queues (4 by default) in creation order: a side stream that lands on the
// caller's queue runs its kernels IN ORDER with the caller's, and the fork buys nothing (the same effect serialised the copy stream of
// bench.py's host-feed legs in round 4: popcorn_amd/data/feed.py).  Which stream aliases depends on what the process created before, so
// the choice is MEASURED at the first step on a caller's stream, with the launch pattern of a small-region step: a fork, twelve 10 us
// spin kernels on either stream with a cross-stream dependency every fourth one (the backward's side_on), a join.  (Two long kernels
// alone overlap even on a shared queue; what serialises a shared queue are the barrier packets of the event waits: 1.47 instead of
// 0.57 ms per 2 x 230 x 220 step, tools/step_side_alias_probe.py.)  About 130 us when the two chains run side by side, 250+ when they
// do not; up to six fresh streams are tried, the first that overlaps is kept.
__global__ void step_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

void pick_side_stream(Step& X, hipStream_t main_st) {
    X.side_picked = true;
    X.side_for = main_st;
    if (!X.side) return;
    // a caller's stream that was measured before keeps its side stream (a trainer stepped alternately from two streams used to pay the
    // probe -- a host synchronisation + ~3 ms of spin kernels -- at every step: ADVICE round 5)
    for (int i = 0; i < X.n_side_cache; ++i)
        if (X.side_cache[i].caller == main_st) { X.side = X.side_cache[i].side; return; }
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) {
        (void)hipGetLastError();
        if (t0) (void)hipEventDestroy(t0);
        return;
    }
    const long long ticks = 10 * 100;                       // 10 us of the 100 MHz wall clock
    // the pattern on the caller's stream ALONE: the yardstick of this GPU / clock (two chains side by side take about as long, two
    // chains through one hardware queue about twice as long) -- no absolute threshold
    float base_ms = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {
        bool ok = hipEventRecord(t0, main_st) == hipSuccess;
        for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(step_spin_kernel, dim3(1), dim3(64), 0, main_st, ticks);
        ok = ok && hipEventRecord(t1, main_st) == hipSuccess && hipEventSynchronize(t1) == hipSuccess;
        float e = 1e9f;
        if (!ok || hipEventElapsedTime(&e, t0, t1) != hipSuccess) { (void)hipGetLastError(); e = 1e9f; }
        if (rep == 1) base_ms = e;
    }
    hipStream_t best = X.side;
    float best_ms = 1e9f;
    hipStream_t cand = X.side;
    hipStream_t tried[6];
    int ntried = 0;
    for (int k = 0; k < 6 && cand; ++k) {
        tried[ntried++] = cand;
        float ms = 1e9f;
        for (int rep = 0; rep < 2; ++rep) {                 // (the first round also absorbs the kernel's first-launch cost)
            bool ok = hipEventRecord(t0, main_st) == hipSuccess && hipEventRecord(X.ev_fork, main_st) == hipSuccess &&
                      hipStreamWaitEvent(cand, X.ev_fork, 0) == hipSuccess;
            for (int i = 0; i < 12; ++i) {
                hipLaunchKernelGGL(step_spin_kernel, dim3(1), dim3(64), 0, main_st, ticks);
                hipLaunchKernelGGL(step_spin_kernel, dim3(1), dim3(64), 0, cand, ticks);
                if ((i & 3) == 3 && i < 11)
                    ok = ok && hipEventRecord(X.ev_dep[i >> 2], main_st) == hipSuccess && hipStreamWaitEvent(cand, X.ev_dep[i >> 2], 0) == hipSuccess;
            }
            ok = ok && hipEventRecord(X.ev_join, cand) == hipSuccess && hipStreamWaitEvent(main_st, X.ev_join, 0) == hipSuccess &&
                 hipEventRecord(t1, main_st) == hipSuccess && hipEventSynchronize(t1) == hipSuccess;
            float e = 1e9f;
            if (!ok || hipEventElapsedTime(&e, t0, t1) != hipSuccess) { (void)hipGetLastError(); e = 1e9f; }
            if (rep == 1) ms = e;
        }
        if (ms < best_ms) { best_ms = ms; best = cand; }
        if (ms < 1.45f * base_ms) break;                    // side by side
        cand = nullptr;
        if (k + 1 < 6 && hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); cand = nullptr; }
    }
    // streams that lost are destroyed unless an earlier caller's entry of the cache uses them
    for (int i = 0; i < ntried; ++i) {
        bool cached = false;
        for (int c = 0; c < X.n_side_cache; ++c) cached = cached || X.side_cache[c].side == tried[i];
        if (tried[i] != best && !cached) (void)hipStreamDestroy(tried[i]);
    }
    X.side = best;
    if (X.n_side_cache < 4) { X.side_cache[X.n_side_cache].caller = main_st; X.side_cache[X.n_side_cache].side = best; ++X.n_side_cache; }
    if (getenv("POPCORN_CONV_DBG"))
        fprintf(stderr, "pc_train_step: side stream %p, probe pattern took %.3f ms against %.3f ms on the caller's stream alone (%d tried)\n",
                (void*)best, best_ms, base_ms, ntried);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
}

}  // namespace

// debug (tools/host_time_eager.py): host nanoseconds of the last call's sizing pass and of its launching pass
static double g_step_host_ns[2] = {0, 0};
extern "C" void pc_debug_step_host_ns(double* out) { out[0] = g_step_host_ns[0]; out[1] = g_step_host_ns[1]; }

extern "C" void* pc_step_create(const pc_step_plan* plan) {
    if (!plan) return nullptr;
    Step* X = new (std::nothrow) Step();
    if (!X) return nullptr;
    X->plan = *plan;
    for (int s = 0; s < 2; ++s) {
        for (int L = 0; L < PC_STEP_CONVS; ++L) {
            X->bn_nobias[s][L] = plan->unet.s[s].bn[L];
            X->bn_nobias[s][L].conv_bias = nullptr;
        }
        for (int i = 0; i < 2; ++i) {         // d1b's ReLU / BN factor for channels [8 i, 8 i + 8)
            pc_bn b = X->bn_nobias[s][L_D1B];
            if (b.gamma) { b.gamma += 8 * i; b.beta += 8 * i; b.mean += 8 * i; b.var += 8 * i; }
            X->bn_half[s][i] = b;
            pc_bn a = X->bn_nobias[s][L_D1A];
            if (a.gamma) { a.gamma += 8 * i; a.beta += 8 * i; a.mean += 8 * i; a.var += 8 * i; }
            X->bn_half_d1a[s][i] = a;
        }
    }
    float one[2] = {1.f, 1.f};
    if (hipMalloc(reinterpret_cast<void**>(&X->ones2), sizeof(one)) != hipSuccess ||
        hipMemcpy(X->ones2, one, sizeof(one), hipMemcpyHostToDevice) != hipSuccess) {
        delete X;
        return nullptr;
    }
    {
        const char* ev = getenv("POPCORN_STEP_SIDE_PX");          // A/B switch: 0 = never fork
        X->side_px = ev ? atoll(ev) : PC_STEP_SIDE_PX_DEFAULT;
        if (X->side_px > 0 && (hipStreamCreateWithFlags(&X->side, hipStreamNonBlocking) != hipSuccess ||
                               hipEventCreateWithFlags(&X->ev_fork, hipEventDisableTiming) != hipSuccess ||
                               hipEventCreateWithFlags(&X->ev_join, hipEventDisableTiming) != hipSuccess)) {
            (void)hipGetLastError();
            X->side = nullptr;                                    // no side stream: the sequential form
        }
        for (int i = 0; i < 8 && X->side; ++i)
            if (hipEventCreateWithFlags(&X->ev_dep[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); X->side = nullptr; }
        // backward: measured on 2 x H x W regions (tools/ab_regions.sh): -2 % .. -1 % up to 4.3 Mpx, +1.5 % from 7 Mpx on
        const char* eb = getenv("POPCORN_STEP_SIDE_BWD_PX");      // A/B switch: 0 = weight-gradient launches stay on the caller's stream
        X->side_bwd_px = eb ? atoll(eb) : 5000000ll;
    }
    return X;
}

extern "C" void pc_step_destroy(void* handle) {
    Step* X = reinterpret_cast<Step*>(handle);
    if (!X) return;
    if (X->ones2) (void)hipFree(X->ones2);
    if (X->ev_fork) (void)hipEventDestroy(X->ev_fork);
    if (X->ev_join) (void)hipEventDestroy(X->ev_join);
    for (int i = 0; i < 8; ++i)
        if (X->ev_dep[i]) (void)hipEventDestroy(X->ev_dep[i]);
    for (int i = 0; i < X->n_side_cache; ++i) {                 // side streams kept for other callers' streams
        bool dup = X->side_cache[i].side == X->side;
        for (int k = 0; k < i; ++k) dup = dup || X->side_cache[k].side == X->side_cache[i].side;
        if (!dup && X->side_cache[i].side) (void)hipStreamDestroy(X->side_cache[i].side);
    }
    if (X->side) (void)hipStreamDestroy(X->side);
    delete X;
}

extern "C" int pc_train_step(void* handle, pc_step_io* io, int phases, void* stream) {
    Step* X = reinterpret_cast<Step*>(handle);
    if (!X || !io || io->B < 1 || io->H < 1 || io->W < 1 || !(phases & 7)) return PC_EINVAL;
    if (g_pc_precision != PC_PREC_FP32) return PC_ENOTSUP;
    if ((phases & PC_STEP_FWD) && (!io->data || !io->admin_mask || !io->census_idx || (!io->sel && !io->sel_host) ||
                                   (io->sel_host && io->H + io->W > PC_STEP_SEL_MAX) || (io->data_kind == PC_DATA_SPLIT && !io->data2)))
        return PC_EINVAL;
    if ((phases & PC_STEP_BWD) && !io->y) return PC_EINVAL;
    if (phases & PC_STEP_FWD) {
        int pt, pb, pl, pr;
        pad_geometry(io->H, io->W, pt, pb, pl, pr);
        const int p = X->plan.extractor_pad;
        // reflect padding must be smaller than the input (F.pad's rule); two 2 x 2 poolings need a 4 x 4 domain
        if (pt >= io->H || pb >= io->H || pl >= io->W || pr >= io->W || p >= io->H || p >= io->W) return PC_EINVAL;
    }
    X->st = reinterpret_cast<hipStream_t>(stream);
    // An EAGER entry point: it forks onto a side stream through events of its own, and the first call on a caller's stream measures
    // which side stream runs beside it (a host synchronisation).  Under stream capture neither is legal: say so instead of failing
    // inside the probe with an opaque HIP error (captured steps go through the per-launch entry points, train.py: _capture).
    {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(X->st, &cs) != hipSuccess) { (void)hipGetLastError(); return PC_EINVAL; }
        if (cs != hipStreamCaptureStatusNone) return PC_ENOTSUP;
    }
    // dry pass: the same code path with launches off -- sizes the arena (bump allocation is deterministic)
    // (a FWD-only call of a data-parallel step is sized for its BWD / UPD calls too: they continue in the same arena)
    const auto t_a = std::chrono::steady_clock::now();
    Step probe = *X;
    int rc = run(probe, *io, (phases & PC_STEP_FWD) ? (PC_STEP_FWD | PC_STEP_BWD | PC_STEP_UPD) : phases, true);
    if (rc) return rc;
    io->arena_needed = probe.ar.peak + 256;
    if (!io->arena || io->arena_bytes < io->arena_needed || (reinterpret_cast<uintptr_t>(io->arena) & 255)) return PC_ENOMEM;
    // (the side-stream probe only once the call is going to launch: a call that returns PC_ENOMEM has not paid for it)
    if (X->side && (!X->side_picked || X->side_for != X->st)) pick_side_stream(*X, X->st);
    const auto t_b = std::chrono::steady_clock::now();
    rc = run(*X, *io, phases, false);
    io->launches = X->launches;
    g_step_host_ns[0] = std::chrono::duration<double, std::nano>(t_b - t_a).count();
    g_step_host_ns[1] = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t_b).count();
    return rc;
}
