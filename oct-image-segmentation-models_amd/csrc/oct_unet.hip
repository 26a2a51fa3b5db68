// liboct_unet_hip.so -- host plan + C ABI of the MI355X-native OCT U-Net engine.  See include/oct_unet.h.
// Graph definition follows /root/reference/oct_image_segmentation_models/models/unet.py:106-153.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/oct_unet.h"
#include "host.hpp"
#include "kernels_bwd.hpp"
#include "kernels_fwd.hpp"
#include "kernels_igemm.hpp"
#include "kernels_bx.hpp"

using namespace oct;
using namespace octh;

#ifndef OCT_SRC_HASH
#define OCT_SRC_HASH "unstamped"     // build.sh passes the SHA-256 (12 hex digits) of csrc/*.hip, csrc/*.hpp, include/*.h
#endif

namespace octh {
Options g_opt;
thread_local Profiler* t_prof = nullptr;
namespace { thread_local std::string g_err; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }

// the conv launchers live in their own translation units (tu_conv_*.hip): one explicit instantiation each
extern template int launch_igemm<3, A_NORMAL, EPI_FWD>(const IgemmArgs&, const LaunchCtx&, int*);
extern template int launch_igemm<2, A_UPF, EPI_FWD>(const IgemmArgs&, const LaunchCtx&, int*);
extern template int launch_igemm<3, A_NORMAL, EPI_MASK>(const IgemmArgs&, const LaunchCtx&, int*);
extern template int launch_igemm<3, A_NORMAL, EPI_RAW>(const IgemmArgs&, const LaunchCtx&, int*);
extern template int launch_igemm<3, A_DOWN2, EPI_MASK>(const IgemmArgs&, const LaunchCtx&, int*);

// which kernel family launch_igemm (launch_conv.hpp) hands a launch to
ConvRoute conv_route(const IgemmArgs& a, int amode, const Options& o) {
    const bool octets_ok = !(a.flags & F_TWO) || a.C0 % 8 == 0;    // staging moves 8-channel octets: one source tensor each
    if (o.mfma_mode && a.wbt && a.Mout <= 16 && a.Mout % 4 == 0 && bt_k_ok(a.Cin) && !(a.flags & F_DROP) && octets_ok &&
        (amode != A_DOWN2 || a.Cin == 8))
        return ROUTE_BT;
    if (o.mfma_mode && a.wbx && a.Mout % 32 == 0 && a.Cin % 8 == 0 && a.m_off % bx_mb(a.Mout) == 0 && octets_ok &&
        a.Cin <= (amode == A_DOWN2 ? 256 : 512))       // the kernel caches the affine rows of its K channels in LDS
        return ROUTE_BX;
    return ROUTE_F32;
}
}  // namespace octh

namespace {

enum Src { SRC_INPUT, SRC_PREV, SRC_POOL, SRC_UP, SRC_CONCAT, SRC_HEAD };

struct Layer {
    char name[32];
    int kh, kw, cin, cout, level, H, W, has_bn, src, skip_from, drop_in;
    size_t w_off, b_off, gamma_off, beta_off, mm_off, mv_off;
    // workspace (device) pointers
    void* z = nullptr; void* g = nullptr;   // activation storage type (f32 or bf16)
    float* bn = nullptr;
    float* dwp = nullptr;  // this layer's dW slabs [npb][kh*kw*cin*cout + cout]
    int dw_rows = 0;       // slabs allocated at creation: a launch never uses more (tuning options may change later)
    float* wt = nullptr;   // backward-data weights: transposed+flipped 3x3, or effective 3x3 of an up-conv (9*cin*cout)
    bf16_t* wbx_f = nullptr; bf16_t* wbx_b = nullptr;   // split weights for the bf16-pipe kernels (forward / backward-data)
    bool bt_m2_f = false, bt_m2_b = false;                // thin kernel: this layer's forward / backward-data launches use the two-pixel form
    bf16_t* wbt_f = nullptr; bf16_t* wbt_b = nullptr;   // ... for the thin bf16-pipe kernel (16-row slices; backward: cin / Cg slices)
    bool g_masked = false;   // last backward: the BN-backward transform was applied on load, `g` still holds the masked gradient g'
};

struct Plan {
    std::vector<Layer> L;
    size_t n_params = 0, n_state = 0;
    int P = 0;
    std::vector<int> enc_last;  // per level: conv index whose output is pooled + skip-concatenated
};

int check_cfg(const oct_unet_cfg* c) {
    if (!c) return fail(-1, "null cfg");
    if (c->in_ch < 1) return fail(-1, "in_ch must be >= 1");
    if (c->n_cls < 2 || c->n_cls > 8) return fail(-1, "n_cls must be in 2..8");
    if (c->pool_layers < 1 || c->pool_layers > 6) return fail(-1, "pool_layers must be in 1..6");
    if (c->conv_layers < 1) return fail(-1, "conv_layers must be >= 1");
    if (c->start_neurons < 4 || c->start_neurons > 32 || c->start_neurons % 4)
        return fail(-1, "start_neurons must be a multiple of 4 in 4..32");
    if (c->enc_k != 3 || c->dec_k != 2) return fail(-1, "only enc_kernel (3,3) / dec_kernel (2,2) are implemented");
    const int m = 1 << c->pool_layers;
    if (c->H < m || c->W < m || c->H % m || c->W % m) return fail(-1, "H and W must be multiples of 2^pool_layers");
    if (c->max_batch < 1) return fail(-1, "max_batch must be >= 1");
    if (c->dtype != 0 && c->dtype != 1) return fail(-2, "dtype must be 0 (f32) or 1 (bf16 activation storage, f32 arithmetic)");
    if (!(c->dropout_rate >= 0.f && c->dropout_rate < 1.f)) return fail(-1, "dropout_rate must be in [0,1)");
    if (c->pool_layers * (2 * c->conv_layers + 1) + c->conv_layers + 1 > ReduceAllArgs::MAXL)
        return fail(-1, "too many conv layers (pool_layers*(2*conv_layers+1)+conv_layers+1 must be <= 40)");
    // the dropped tensor is the bottleneck output: (H/2^P, W/2^P, start_neurons*2^P)
    if ((size_t)c->max_batch * (c->H >> c->pool_layers) * (c->W >> c->pool_layers) *
            (size_t)(c->start_neurons << c->pool_layers) >= (1ull << 32))
        return fail(-1, "bottleneck tensor too large for 32-bit dropout indexing");
    if ((size_t)c->max_batch * c->H * c->W * (size_t)(2 * c->start_neurons) >= (1ull << 31))
        return fail(-1, "max_batch * H * W too large for the 32-bit element indexing used inside a layer");
    return 0;
}

Plan build_plan(const oct_unet_cfg& c) {
    Plan pl; pl.P = c.pool_layers;
    const int sn = c.start_neurons, P = c.pool_layers, Lc = c.conv_layers;
    int cin = c.in_ch;
    auto add = [&](const std::string& nm, int k, int ci, int co, int lvl, int bn, int src, int skip) {
        Layer l{}; snprintf(l.name, sizeof l.name, "%s", nm.c_str());
        l.kh = l.kw = k; l.cin = ci; l.cout = co; l.level = lvl; l.H = c.H >> lvl; l.W = c.W >> lvl;
        l.has_bn = bn; l.src = src; l.skip_from = skip; l.drop_in = 0;
        l.w_off = pl.n_params; pl.n_params += (size_t)k * k * ci * co;
        l.b_off = pl.n_params; pl.n_params += co;
        if (bn) {
            l.gamma_off = pl.n_params; pl.n_params += co;
            l.beta_off = pl.n_params; pl.n_params += co;
            l.mm_off = pl.n_state; pl.n_state += co;
            l.mv_off = pl.n_state; pl.n_state += co;
        }
        pl.L.push_back(l);
    };
    for (int i = 0; i < P; ++i) {
        const int size = sn << i;
        for (int j = 0; j < Lc; ++j) {
            const int src = (i == 0 && j == 0) ? SRC_INPUT : (j == 0 ? SRC_POOL : SRC_PREV);
            add("enc" + std::to_string(i) + ".conv" + std::to_string(j), c.enc_k, cin, size, i, 1, src, -1);
            cin = size;
        }
        pl.enc_last.push_back((int)pl.L.size() - 1);
    }
    for (int j = 0; j < Lc; ++j) {
        add("mid.conv" + std::to_string(j), c.enc_k, cin, sn << P, P, 1, j == 0 ? SRC_POOL : SRC_PREV, -1);
        cin = sn << P;
    }
    for (int i = 0; i < P; ++i) {
        const int lvl = P - 1 - i, size = sn << lvl;
        add("dec" + std::to_string(i) + ".up", c.dec_k, cin, size, lvl, 1, SRC_UP, -1);
        if (i == 0 && c.dropout_rate > 0.f) pl.L.back().drop_in = 1;  // Dropout(0.5) sits on the bottleneck output
        cin = 2 * size;
        for (int j = 0; j < Lc; ++j) {
            add("dec" + std::to_string(i) + ".conv" + std::to_string(j), c.enc_k, cin, size, lvl, 1,
                j == 0 ? SRC_CONCAT : SRC_PREV, j == 0 ? pl.enc_last[lvl] : -1);
            cin = size;
        }
    }
    add("head", 1, cin, c.n_cls, 0, 0, SRC_HEAD, -1);
    return pl;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int tiles_of(int H, int W) { return cdiv(H, kTileY) * cdiv(W, kTileX); }
inline int chunk_of(int c) { return c % 16 == 0 ? 16 : (c % 8 == 0 ? 8 : 4); }   // channel chunk per thread

// ---- bf16-pipe kernel eligibility (kernels_bx.hpp): forward needs >= 32 output channels, backward-data >= 32 channels
// per launch (a concat's dX is two launches of cin/2 channels each); K channels in multiples of 8, <= 512 ----
inline bool bx_fwd_ok(const Layer& l) { return l.src != SRC_INPUT && l.has_bn && l.cin % 8 == 0 && l.cin <= 512 && l.cout % 32 == 0; }
inline int bx_bwd_cg(const Layer& l) { return l.src == SRC_CONCAT ? l.cin / 2 : l.cin; }
inline bool bx_bwd_ok(const Layer& l) { return l.src != SRC_INPUT && l.has_bn && l.cout % 8 == 0 && l.cout <= 512 && bx_bwd_cg(l) % 32 == 0; }

// thin bf16-pipe kernel (conv_bt_k): <= 16 output channels per launch, K channels exactly 8, 16 or 32
inline bool bt_fwd_ok(const Layer& l) { return l.src != SRC_INPUT && l.has_bn && l.cout <= 16 && l.cout % 4 == 0 && bt_k_ok(l.cin); }
inline bool bt_bwd_ok(const Layer& l) { const int cg = bx_bwd_cg(l); return l.src != SRC_INPUT && l.has_bn && cg <= 16 && cg % 4 == 0 && bt_k_ok(l.cout); }

// two-pixel form of the thin kernel: exactly 8 output channels per launch, 8 or 16 K channels, not the stride-2 gather
inline bool bt_m2_fwd(const Layer& l, const Options& o) { return o.bt_m2 && bt_fwd_ok(l) && l.cout == 8 && (l.cin == 8 || l.cin == 16); }
inline bool bt_m2_bwd(const Layer& l, const Options& o) { return o.bt_m2 && bt_bwd_ok(l) && l.src != SRC_UP && bx_bwd_cg(l) == 8 && (l.cout == 8 || l.cout == 16); }

// thin backward-weights on the bf16 pipe (conv_dwbt_k): the instantiated (cin, cout) pairs
inline bool dwbt_ok(const Layer& l) {
    if (l.src == SRC_INPUT || l.kh == 1 || !l.has_bn) return false;
    const int ci = l.cin, co = l.cout;
    if (l.src == SRC_UP) return (ci == 16 && co == 8) || (ci == 32 && co == 16);
    if (l.src == SRC_CONCAT && (l.cin / 2) % 8) return false;         // staging moves 8-channel octets: one source tensor each
    return (ci == 8 && co == 8) || (ci == 8 && co == 16) || (ci == 16 && co == 8) || (ci == 16 && co == 16) ||
           (ci == 16 && co == 32) || (ci == 32 && co == 16);
}

// dW plan: which kernel handles a layer, its channel chunking, pixel-tile height and pixel-block count
DwPlan dw_plan(const Layer& l, int B, int mfma_mode, int bf16, const Options& o) {
    DwPlan p{};
    if (mfma_mode && o.dwbx_enable && l.src != SRC_INPUT && l.kh != 1 && l.cin % 32 == 0 && l.cout % 32 == 0) {
        // wide layers on the bf16 pipe: one block per (32 ci, 32 co) pair and pixel slice, ~1 block per CU in total
        p.kind = 33; p.cic = 32; p.coc = 32; p.th = 4;
        p.chunks = (l.cin / 32) * (l.cout / 32);
        p.tiles = cdiv(l.H, p.th) * cdiv(l.W, kTileX);
        p.npb = std::max(1, std::min(B * p.tiles, cdiv(o.dwbx_blocks, p.chunks)));
        return p;
    }
    // fp32 mode: the three-term split of both operands makes conv_dwbt_k VALU-bound; measured per shape (B=32 256x512,
    // us, fp32-pipe kernel vs this one): 8->8 89 / 100, up-convs 98 / 113 and 55 / 66 stay on the fp32 pipe; 16->8 149 /
    // 143, 32->16 103 / 92, 8->16 39 / 35, 16->16 59 / 48, 16->32 come here.  bf16 mode (one rounding, one product): all.
    if (mfma_mode && dwbt_ok(l) && (bf16 || o.dwbt_f32_all || (l.src != SRC_UP && !(l.cin == 8 && l.cout == 8)))) {
        // thin layers on the bf16 pipe (conv_dwbt_k): one block holds all channels; 1 or 2 blocks per CU (LDS images)
        p.kind = 34; p.cic = l.cin; p.coc = l.cout; p.th = 4; p.chunks = 1;
        p.tiles = cdiv(l.H, p.th) * cdiv(l.W, kTileX);
        p.npb = std::max(1, std::min(B * p.tiles, 256 * (l.cin + l.cout >= 48 ? 1 : 2)));
        return p;
    }
    if (l.src == SRC_INPUT || l.cin % 4 || l.cout % 4 || l.kh == 1) {
        // first layer (ANY in_ch: its source is the caller's image, uint8 or f32 -- only the VALU kernel's fetch_x reads
        // that; the MFMA stagers assume an activation-typed tensor) and the 1x1 n_cls-wide head
        p.kind = 0; p.cic = (l.src == SRC_INPUT || l.cin % 4) ? 1 : chunk_of(l.cin); p.coc = l.cout % 4 ? (l.cout <= 4 ? 4 : 8) : chunk_of(l.cout); p.th = kTileY;
    } else if (l.cin >= 32 && l.cout >= 32) {
        p.kind = 32; p.cic = l.cin % 64 == 0 ? 64 : 32; p.coc = 32; p.th = p.cic == 64 ? 2 : 4;
    } else {
        p.kind = 16; p.cic = l.cin >= 16 ? 16 : 8; p.coc = 16; p.th = 8;
    }
    p.chunks = cdiv(l.cin, p.cic) * cdiv(l.cout, p.coc);
    p.tiles = cdiv(l.H, p.th) * cdiv(l.W, kTileX);
    const int total = B * p.tiles;
    // one full round of resident blocks (no half-empty tail round): the wide kernel fits 2 blocks per CU (registers),
    // for the thin one 768 blocks measured best
    int target = p.kind == 32 ? o.dw32_blocks : o.dw16_blocks;
    if (p.kind == 16 && l.src == SRC_UP && p.cic == 16 && l.cout == 8) target = target * 4 / 3;   // (dz staged 8 wide: 4 blocks per CU)
    p.npb = std::max(1, std::min(total, cdiv(target, p.chunks)));
    return p;
}

}  // namespace

struct oct_unet {
    oct_unet_cfg cfg;
    Options opt;                           // snapshot of the process-wide defaults at creation (host.hpp)
    Plan plan;
    float* params; float* grads; float* state;
    std::vector<void*> pooled, gpooled;    // per encoder level (activation storage type)
    float* stat_part = nullptr;            // BN statistic partials (fwd and bwd share it: stream-ordered)
    unsigned* fin_counters = nullptr;      // [2 * layers] arrival counters of the in-launch finalizes (kernels_fin.hpp): zero between launches
    ReduceAllArgs red{};                   // filled while backward runs; one reduce launch at the end
    float* dice_part = nullptr; double* dice_bc = nullptr; float* loss4 = nullptr;   // loss4: 8 floats (dice_finalize_k)
    float focal_w = 0.f, focal_gamma = 2.f; const float* focal_cw = nullptr;           // focal_dice_loss (0 = plain Dice)
    WtDesc* wt_descs = nullptr; int n_wt = 0; unsigned wt_total = 0;
    WbxDesc* wbx_descs = nullptr; int n_wbx_f = 0, n_wbx_b = 0; unsigned wbx_f_total = 0, wbx_b_total = 0;   // [fwd..., bwd...]
    WbtDesc* wbt_descs = nullptr; int n_wbt_f = 0, n_wbt_b = 0; unsigned wbt_f_total = 0, wbt_b_total = 0;   // [fwd..., bwd...]
    unsigned long long drop_step = 0; int drop_advance = 0;
    int last_B = 0; int last_training = 0; int have_dice = 0; int dice_final = 0;
    const void* last_x = nullptr; int last_u8 = 0;   // input of the last forward (first layer's dW re-reads it)
    hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;
    hipEvent_t tail_event = nullptr;       // recorded in backward once the decoder + bottleneck gradients are final
    // backward-weights kernels run on an internal low-priority side stream beside the backward-data chain (they are off
    // the critical path: dz_L -> dW_L feeds nothing until the slab reduce): fork / join with events, created with the handle
    hipStream_t side = nullptr; std::vector<hipEvent_t> fork_ev; hipEvent_t join_ev = nullptr, prep_ev = nullptr;
    Profiler prof;
};

namespace {

// Carve the workspace; with base == nullptr only sizes are computed.  Returns total bytes.
size_t carve(const oct_unet_cfg& c, Plan& pl, oct_unet* h, char* base, const Options& o) {
    size_t off = 0;
    auto take = [&](size_t bytes) -> char* { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    const size_t B = (size_t)c.max_batch, esz = c.dtype ? 2 : 4;   // bytes per stored activation element
    size_t stat_max = 0, dw_max = 0;
    for (auto& l : pl.L) {
        const size_t n = B * l.H * l.W * l.cout;
        void* z = l.has_bn ? (void*)take(n * esz) : nullptr;   // the head writes straight to the caller's buffers
        void* g = c.training && l.has_bn ? (void*)take(n * esz) : nullptr;
        float* bn = l.has_bn ? (float*)take((size_t)BN_ARRAYS * l.cout * 4) : nullptr;
        float* wt = (c.training && l.has_bn && l.src != SRC_INPUT) ? (float*)take((size_t)9 * l.cin * l.cout * 4) : nullptr;
        const int ns = c.dtype ? 1 : 3;
        bf16_t* wf = bx_fwd_ok(l) ? (bf16_t*)take(wbx_bytes(l.kh, l.cin, l.cout, bx_mb(l.cout), ns)) : nullptr;
        bf16_t* wb = (c.training && bx_bwd_ok(l)) ? (bf16_t*)take(wbx_bytes(3, l.cout, l.cin, bx_mb(bx_bwd_cg(l)), ns)) : nullptr;
        const bool m2f = bt_m2_fwd(l, o), m2b = bt_m2_bwd(l, o);
        bf16_t* tf = bt_fwd_ok(l) ? (bf16_t*)take(wbt_bytes(l.kh, l.cin, ns, m2f)) : nullptr;
        bf16_t* tb = (c.training && bt_bwd_ok(l)) ? (bf16_t*)take(wbt_bytes(3, l.cout, ns, m2b) * (l.cin / bx_bwd_cg(l))) : nullptr;
        if (base) { l.bt_m2_f = m2f; l.bt_m2_b = m2b; }
        if (base) { l.z = z; l.g = g; l.bn = bn; l.wt = wt; l.wbx_f = wf; l.wbx_b = wb; l.wbt_f = tf; l.wbt_b = tb; }
        // statistic partial rows: one per pixel tile; the MFMA kernels may use tiles as small as 2 x 32 pixels
        stat_max = std::max(stat_max, B * cdiv(l.H, 2) * cdiv(l.W, kTileX) * 2 * (size_t)std::max(l.cout, l.cin));
        if (c.training) {
            const size_t wsz = (size_t)l.kh * l.kw * l.cin * l.cout + l.cout;
            const size_t rows = l.src == SRC_HEAD ? (size_t)(2048 + c.max_batch)
                                                  : (size_t)std::max(dw_plan(l, c.max_batch, 0, 0, o).npb, std::max(dw_plan(l, c.max_batch, 1, 1, o).npb, dw_plan(l, c.max_batch, 1, 0, o).npb));
            float* dwp = (float*)take(rows * wsz * 4);
            if (base) { l.dwp = dwp; l.dw_rows = (int)rows; }
            dw_max = 0;
        }
    }
    for (int i = 0; i < pl.P; ++i) {
        const size_t n = B * (c.H >> (i + 1)) * (c.W >> (i + 1)) * ((size_t)c.start_neurons << i);
        void* p = (void*)take(n * esz);
        void* gp = c.training ? (void*)take(n * esz) : nullptr;
        if (h) { h->pooled.push_back(p); h->gpooled.push_back(gp); }
    }
    const int nblk_head = cdiv(c.H * c.W, kBlock);
    stat_max = std::max(stat_max, B * nblk_head * 2 * (size_t)c.start_neurons);
    float* sp = (float*)take(stat_max * 4);
    float* dp = (float*)take(B * nblk_head * 64 * 4);
    double* bc = (double*)take((B * 8 * 2 + 2) * 8);
    float* l4 = (float*)take(8 * 4);
    unsigned* fc = (unsigned*)take(2 * pl.L.size() * sizeof(unsigned));
    if (h) h->fin_counters = fc;
    WtDesc* wd = c.training ? (WtDesc*)take(pl.L.size() * sizeof(WtDesc)) : nullptr;
    WbxDesc* xd = (WbxDesc*)take(2 * pl.L.size() * sizeof(WbxDesc));
    WbtDesc* td = (WbtDesc*)take(3 * pl.L.size() * sizeof(WbtDesc));
    if (h) { h->wt_descs = wd; h->wbx_descs = xd; h->wbt_descs = td; }
    if (h) { h->stat_part = sp; h->dice_part = dp; h->dice_bc = bc; h->loss4 = l4; }
    return off;
}

DropCfg make_drop(const oct_unet* h) {
    DropCfg d;
    d.seed = h->cfg.seed; d.step = h->drop_step;
    const double r = h->cfg.dropout_rate;
    d.thresh = (unsigned)std::min(4294967295.0, std::floor(r * 4294967296.0));
    d.scale = (float)(1.0 / (1.0 - r));
    return d;
}

// statistics of block li finalized by the last block of the launch that emits them (kernels_fin.hpp): thin channel counts
// only -- the reduction is done by ONE block
inline bool fin_ok(const oct_unet* h, const Layer& l) { return h->opt.fuse_bn_finalize && l.has_bn && l.cout <= 32 && l.cout % 2 == 0; }
FinDesc fin_desc(const oct_unet* h, int li, int bwd, int B) {
    const Layer& l = h->plan.L[li];
    FinDesc f{};
    f.counter = h->fin_counters + 2 * li + (bwd ? 1 : 0); f.bwd = bwd; f.count = (double)B * l.H * l.W;
    f.bn = l.bn; f.gamma = h->params + l.gamma_off; f.beta = h->params + l.beta_off;
    f.mm = h->state + l.mm_off; f.mv = h->state + l.mv_off;
    f.eps = h->cfg.bn_eps; f.momentum = h->cfg.bn_momentum; f.unbiased = h->cfg.bn_unbiased_moving_var;
    if (bwd) { f.dgamma = h->grads + l.gamma_off; f.dbeta = h->grads + l.beta_off; }
    return f;
}

// ---------------------------------------------------------------------------------------------------------------
// forward launches
// ---------------------------------------------------------------------------------------------------------------
template <int KH, int FLAGS>
int launch_conv_fwd_co(const ConvFwdArgs& a, int B, hipStream_t s, const char* layer, double flops, double bytes) {
    const int co_t = chunk_of(a.Cout);
    dim3 grid(a.tiles, a.Cout / co_t, B), block(kBlock);
    char nm[64]; snprintf(nm, sizeof nm, "conv_fwd_k<%d,%d,%d,%s>", KH, co_t, FLAGS, AT_NAME(a.act_bf16));
    ProfScope ps(s, nm, layer, flops, bytes);
    switch (co_t) {
        case 16: AT_DISPATCH(a.act_bf16, conv_fwd_k<KH, 16, FLAGS, AT><<<grid, block, 0, s>>>(a)); break;
        case 8: AT_DISPATCH(a.act_bf16, conv_fwd_k<KH, 8, FLAGS, AT><<<grid, block, 0, s>>>(a)); break;
        default: AT_DISPATCH(a.act_bf16, conv_fwd_k<KH, 4, FLAGS, AT><<<grid, block, 0, s>>>(a)); break;
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// source description of layer li's input (shared by forward conv and dW)
struct SrcDesc { const void* x0; const float* ab0; int C0; const void* x1; const float* ab1; int C1; int flags; };

SrcDesc src_of(const oct_unet* h, int li, const void* x_in, int x_is_u8) {
    const Layer& l = h->plan.L[li];
    SrcDesc d{}; d.x1 = nullptr; d.ab1 = nullptr; d.C1 = 0;
    switch (l.src) {
        case SRC_INPUT: d.x0 = x_in; d.ab0 = nullptr; d.C0 = l.cin; d.flags = x_is_u8 ? F_U8 : 0; break;
        case SRC_PREV: case SRC_HEAD: { const Layer& p = h->plan.L[li - 1]; d.x0 = p.z; d.ab0 = p.bn; d.C0 = p.cout; d.flags = F_AFF; break; }
        case SRC_POOL: d.x0 = h->pooled[l.level - 1]; d.ab0 = nullptr; d.C0 = l.cin; d.flags = 0; break;
        case SRC_UP: { const Layer& p = h->plan.L[li - 1]; d.x0 = p.z; d.ab0 = p.bn; d.C0 = p.cout; d.flags = F_AFF | F_UP; break; }
        case SRC_CONCAT: {
            const Layer& p = h->plan.L[li - 1]; const Layer& k = h->plan.L[l.skip_from];
            d.x0 = p.z; d.ab0 = p.bn; d.C0 = p.cout; d.x1 = k.z; d.ab1 = k.bn; d.C1 = k.cout; d.flags = F_AFF | F_TWO; break;
        }
    }
    return d;
}

// algorithmic bytes of a conv's logical input, read once: low-res tensor for an up-conv, both
// halves of a concat, the pooled tensor after a pool, 1 B/px for a u8 image
double in_bytes(const Layer& l, int B, int x_is_u8, int es = 4) {
    const double px = (double)B * l.H * l.W;
    if (l.src == SRC_INPUT) return px * l.cin * (x_is_u8 ? 1 : 4);
    if (l.src == SRC_UP) return px / 4 * l.cin * es;
    return px * l.cin * es;
}

int conv_forward(oct_unet* h, int li, const void* x_in, int x_is_u8, int B, int training, hipStream_t s) {
    Layer& l = h->plan.L[li];
    const SrcDesc sd = src_of(h, li, x_in, x_is_u8);
    ConvFwdArgs a{};
    a.x0 = sd.x0; a.ab0 = sd.ab0; a.C0 = sd.C0; a.x1 = nullptr; a.ab1 = nullptr; a.C1 = 0;   // VALU kernel: first layer only
    a.w = h->params + l.w_off; a.bias = h->params + l.b_off; a.z = l.z;
    a.part = (training && l.has_bn) ? h->stat_part : nullptr;
    a.H = l.H; a.W = l.W; a.Cin = l.cin; a.Cout = l.cout;
    a.tiles_x = cdiv(l.W, kTileX); a.tiles = tiles_of(l.H, l.W);
    a.drop = make_drop(h); a.act_bf16 = h->cfg.dtype;
    const bool drop = training && l.drop_in;
    const double px = (double)B * l.H * l.W, fl = 2.0 * l.kh * l.kw * l.cin * l.cout * px;
    const int es = h->cfg.dtype ? 2 : 4;
    const double by = in_bytes(l, B, x_is_u8, es) + px * l.cout * es;   // logical input once + output once
    int rc;
    int stat_rows = B * a.tiles;
    bool fin_in_launch = false;
    if (l.src != SRC_INPUT && l.cin % 4 == 0) {   // MFMA path (every conv except the 1-channel first layer)
        IgemmArgs g{};
        g.x0 = sd.x0; g.ab0 = sd.ab0; g.C0 = sd.C0; g.x1 = sd.x1; g.ab1 = sd.ab1; g.C1 = sd.C1;
        g.flags = sd.flags | (drop ? F_DROP : 0);
        g.Cin = l.cin; g.w = a.w; g.w_ld = l.cout; g.m_off = 0; g.bias = a.bias; g.out = l.z; g.Mout = l.cout;
        g.Ho = l.H; g.Wo = l.W; g.Hi = l.src == SRC_UP ? l.H / 2 : l.H; g.Wi = l.src == SRC_UP ? l.W / 2 : l.W;
        g.part = a.part; g.drop = a.drop; g.act_bf16 = h->cfg.dtype;
        g.wbx = l.wbx_f; g.wbx_M = l.cout; g.wbt = l.wbt_f; g.bt_m2 = l.bt_m2_f;
        const LaunchCtx lc{&h->opt, B, s, l.name, fl, by};
        if (training && fin_ok(h, l) && conv_route(g, l.src == SRC_UP ? A_UPF : A_NORMAL, h->opt) == ROUTE_BT) { g.fin = fin_desc(h, li, 0, B); fin_in_launch = true; }
        rc = l.src == SRC_UP ? launch_igemm<2, A_UPF, EPI_FWD>(g, lc, &stat_rows)
                             : launch_igemm<3, A_NORMAL, EPI_FWD>(g, lc, &stat_rows);
    } else if (l.src == SRC_INPUT && l.cin == 1 && l.cout == 8 && l.kh == 3) {   // the real first layer: persistent streaming kernel
        const int tx = cdiv(l.W, 128), tiles = tx * cdiv(l.H, 8), total = B * tiles;
        const int grid = std::min(total, 2048);      // <= B*ceil(H/2)*ceil(W/32) statistic rows guaranteed by carve()
        const int bf = h->cfg.dtype;
        if (training && fin_ok(h, l)) { a.fin = fin_desc(h, li, 0, B); fin_in_launch = true; }
        ProfScope ps(s, bf ? "conv_first_fwd_k<unsigned short>" : "conv_first_fwd_k<float>", l.name, fl, by);
        AT_DISPATCH(bf, conv_first_fwd_k<AT><<<grid, kBlock, 0, s>>>(a, x_is_u8, tx, tiles, total, a.w, a.bias));
        HIP_OK(hipGetLastError());
        stat_rows = grid; rc = 0;
    } else if (l.src == SRC_INPUT) {   // other first layers (odd channel counts): uint8 /255 table on load -> VALU direct conv
        rc = x_is_u8 ? launch_conv_fwd_co<3, F_U8>(a, B, s, l.name, fl, by) : launch_conv_fwd_co<3, 0>(a, B, s, l.name, fl, by);
    } else {
        return fail(-3, "conv_forward: channel count not a multiple of 4");
    }
    if (rc) return rc;
    if (l.has_bn && training && !fin_in_launch && !(h->opt.timing_skip & 1 && h->drop_step > 2)) {
        ProfScope ps(s, "bn_fwd_finalize_k", l.name, 0, (double)stat_rows * 2 * l.cout * 4);
        BnFinArgs f{};
        f.part = h->stat_part; f.nblk = stat_rows; f.C = l.cout; f.count = (double)B * l.H * l.W;
        f.gamma = h->params + l.gamma_off; f.beta = h->params + l.beta_off; f.bn = l.bn;
        f.mm = h->state + l.mm_off; f.mv = h->state + l.mv_off;
        f.eps = h->cfg.bn_eps; f.momentum = h->cfg.bn_momentum; f.unbiased = h->cfg.bn_unbiased_moving_var;
        bn_fwd_finalize_k<<<l.cout, kBlock, 0, s>>>(f);
        HIP_OK(hipGetLastError());
    }
    return 0;
}

template <int C>
int launch_head_fwd(const HeadFwdArgs& a, int cin, int B, hipStream_t s) {
    dim3 grid(a.nblk, B), block(kBlock);
    const double px = (double)B * a.HW;
    char nm[64]; snprintf(nm, sizeof nm, "head_fwd_k<%d,%d,%s>", C, cin, AT_NAME(a.act_bf16));
    ProfScope ps(s, nm, "head", 2.0 * cin * C * px, px * (cin * 4 + (a.probs ? C * 4 : 0) + (a.argmax ? 1 : 0) + (a.labels ? 1 : 0)));
    switch (cin) {
        case 4: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 4, AT><<<grid, block, 0, s>>>(a)); break;
        case 8: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 8, AT><<<grid, block, 0, s>>>(a)); break;
        case 16: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 16, AT><<<grid, block, 0, s>>>(a)); break;
        case 12: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 12, AT><<<grid, block, 0, s>>>(a)); break;
        case 20: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 20, AT><<<grid, block, 0, s>>>(a)); break;
        case 24: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 24, AT><<<grid, block, 0, s>>>(a)); break;
        case 28: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 28, AT><<<grid, block, 0, s>>>(a)); break;
        case 32: AT_DISPATCH(a.act_bf16, head_fwd_k<C, 32, AT><<<grid, block, 0, s>>>(a)); break;
        default: return fail(-3, "head: unsupported start_neurons");
    }
    HIP_OK(hipGetLastError());
    return 0;
}

template <int C>
int launch_head_bwd(const HeadBwdArgs& a, int cin, int B, hipStream_t s) {
    dim3 grid(a.nblk, B), block(kBlock);
    const double px = (double)B * a.HW;
    char nm[64]; snprintf(nm, sizeof nm, "head_bwd_k<%d,%d,%s>", C, cin, AT_NAME(a.act_bf16));
    ProfScope ps(s, nm, "head", 6.0 * cin * C * px, px * (cin * 4 * 2 + 1));
    switch (cin) {
        case 4: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 4, AT><<<grid, block, 0, s>>>(a)); break;
        case 8: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 8, AT><<<grid, block, 0, s>>>(a)); break;
        case 16: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 16, AT><<<grid, block, 0, s>>>(a)); break;
        case 12: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 12, AT><<<grid, block, 0, s>>>(a)); break;
        case 20: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 20, AT><<<grid, block, 0, s>>>(a)); break;
        case 24: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 24, AT><<<grid, block, 0, s>>>(a)); break;
        case 28: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 28, AT><<<grid, block, 0, s>>>(a)); break;
        case 32: AT_DISPATCH(a.act_bf16, head_bwd_k<C, 32, AT><<<grid, block, 0, s>>>(a)); break;
        default: return fail(-3, "head: unsupported start_neurons");
    }
    HIP_OK(hipGetLastError());
    return 0;
}

#define DISPATCH_C(fn, ncls, ...)                       \
    ([&]() -> int {                                     \
        switch (ncls) {                                 \
            case 2: return fn<2>(__VA_ARGS__);          \
            case 3: return fn<3>(__VA_ARGS__);          \
            case 4: return fn<4>(__VA_ARGS__);          \
            case 5: return fn<5>(__VA_ARGS__);          \
            case 6: return fn<6>(__VA_ARGS__);          \
            case 7: return fn<7>(__VA_ARGS__);          \
            case 8: return fn<8>(__VA_ARGS__);          \
            default: return fail(-3, "bad n_cls");      \
        }                                               \
    })()

// blocks per image of the head kernels: ~2048 blocks in total, each walking several 256-pixel chunks
int head_nblk(int HW, int B) { return std::max(1, std::min(cdiv(HW, kBlock), cdiv(2048, B))); }

int dice_n(int C) { return 5 * C <= 16 ? 16 : (5 * C <= 32 ? 32 : 64); }

int forward_impl(oct_unet* h, const void* x, int x_is_u8, int B, int training, const oct_unet_io* io, hipStream_t s) {
    Plan& pl = h->plan;
    const int nl = (int)pl.L.size();
    // this step's weights, split / rounded into bf16 MFMA operand order (one launch per kernel family).  In a training
    // step they run on the side stream under the first layer (which does not use them).
    const bool fside = training && h->opt.dw_side_stream && h->side && !(t_prof && t_prof->on);
    hipStream_t ps_ = fside ? h->side : s;
    if (fside) { HIP_OK(hipEventRecord(h->fork_ev[0], s)); HIP_OK(hipStreamWaitEvent(h->side, h->fork_ev[0], 0)); }
    if (h->opt.mfma_mode && h->n_wbx_f) {
        ProfScope ps(ps_, "prep_wbx_k", "all", 0, (double)h->wbx_f_total * 8 * (4 + 2 * (h->cfg.dtype ? 1 : 3)));
        prep_wbx_k<<<std::min<unsigned>((h->wbx_f_total + kBlock - 1) / kBlock, 2048u), kBlock, 0, ps_>>>(h->wbx_descs, h->n_wbx_f, h->wbx_f_total);
        HIP_OK(hipGetLastError());
    }
    if (h->opt.mfma_mode && h->n_wbt_f) {
        ProfScope ps(ps_, "prep_wbt_k", "all", 0, (double)h->wbt_f_total * 8 * (4 + 2 * (h->cfg.dtype ? 1 : 3)));
        prep_wbt_k<<<std::min<unsigned>((h->wbt_f_total + kBlock - 1) / kBlock, 2048u), kBlock, 0, ps_>>>(h->wbt_descs, h->n_wbt_f, h->wbt_f_total);
        HIP_OK(hipGetLastError());
    }
    if (fside) HIP_OK(hipEventRecord(h->prep_ev, h->side));
    if (!training) {  // (a, b) of every block from the moving statistics: one launch
        ProfScope ps(s, "bn_infer_all_k", "all", 0, (double)pl.n_state * 4 * 3);
        BnInferAll ia{};
        for (auto& l : pl.L)
            if (l.has_bn) {
                auto& e = ia.L[ia.n++];
                e.gamma = h->params + l.gamma_off; e.beta = h->params + l.beta_off;
                e.mm = h->state + l.mm_off; e.mv = h->state + l.mv_off; e.bn = l.bn; e.C = l.cout;
            }
        bn_infer_all_k<<<ia.n, 128, 0, s>>>(ia, h->cfg.bn_eps);
        HIP_OK(hipGetLastError());
    }
    for (int li = 0; li < nl - 1; ++li) {
        Layer& l = pl.L[li];
        if (l.src == SRC_POOL) {  // pool the previous block's output (BN+ReLU applied on load)
            const Layer& p = pl.L[li - 1];
            const size_t n = (size_t)B * (p.H / 2) * (p.W / 2) * (p.cout / 4);
            const int grid = (int)std::min<size_t>((n + kBlock - 1) / kBlock, 8192);
            const int bf = h->cfg.dtype;
            ProfScope ps(s, bf ? "pool_fwd_k<unsigned short>" : "pool_fwd_k<float>", p.name, 0, (double)n * (bf ? 8 : 16) * 5);   // read 4 px, write 1
            AT_DISPATCH(bf, pool_fwd_k<AT><<<grid, kBlock, 0, s>>>((const AT*)p.z, p.bn, (AT*)h->pooled[l.level - 1], B, p.H, p.W, p.cout));
            HIP_OK(hipGetLastError());
        }
        if (fside && li == 1) HIP_OK(hipStreamWaitEvent(s, h->prep_ev, 0));     // first layer that reads the split weights
        const int rc = conv_forward(h, li, x, x_is_u8, B, training, s);
        if (rc) return rc;
    }
    if (fside && nl - 1 <= 1) HIP_OK(hipStreamWaitEvent(s, h->prep_ev, 0));
    // head
    const Layer& hd = pl.L[nl - 1]; const Layer& last = pl.L[nl - 2];
    HeadFwdArgs a{};
    a.z = last.z; a.ab = last.bn; a.w = h->params + hd.w_off; a.bias = h->params + hd.b_off;
    a.probs = io ? io->probs : nullptr; a.argmax = io ? io->argmax : nullptr; a.labels = io ? io->labels : nullptr;
    a.dice_part = h->dice_part; a.HW = hd.H * hd.W; a.nblk = head_nblk(a.HW, B); a.act_bf16 = h->cfg.dtype;
    a.focal_on = h->focal_w > 0.f; a.focal_gamma = h->focal_gamma; a.focal_cw = h->focal_cw; a.focal_clip_mod = h->opt.focal_clip_mod;
    const int rc = DISPATCH_C(launch_head_fwd, h->cfg.n_cls, a, hd.cin, B, s);
    if (rc) return rc;
    h->last_B = B; h->last_training = training; h->have_dice = a.labels != nullptr;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// backward launches
// ---------------------------------------------------------------------------------------------------------------
template <int KH>
int launch_dw(const ConvBwdWArgs& a, int ci_t, int co_t, hipStream_t s, const char* layer, double flops, double bytes) {
    dim3 grid(a.npb, cdiv(a.Cin, ci_t), cdiv(a.Cout, co_t)), block(kBlock);
    char nm[64]; snprintf(nm, sizeof nm, "conv_bwd_w_k<%d,%d,%d,%s>", KH, ci_t, co_t, AT_NAME(a.act_bf16));
    ProfScope ps(s, nm, layer, flops, bytes);
#define DW_CASE(CI, CO) if (ci_t == CI && co_t == CO) { AT_DISPATCH(a.act_bf16, conv_bwd_w_k<KH, CI, CO, AT><<<grid, block, 0, s>>>(a)); HIP_OK(hipGetLastError()); return 0; }
    DW_CASE(1, 4) DW_CASE(1, 8) DW_CASE(1, 16)
#undef DW_CASE
    return fail(-3, "dW: unsupported channel chunking");
}

// index of the first bottleneck conv: layers [0, idx) are the encoder
int first_mid_layer(const Plan& pl) { return pl.enc_last.empty() ? 0 : pl.enc_last.back() + 1; }

// register a layer's slabs for the single end-of-backward reduce launch
void queue_reduce(oct_unet* h, const Layer& l, int npb) {
    ReduceAllArgs& R = h->red;
    ReduceAllArgs::Entry& e = R.L[R.n];
    e.part = l.dwp; e.dw = h->grads + l.w_off; e.db = h->grads + l.b_off; e.npb = npb;
    e.wsize = (unsigned)((size_t)l.kh * l.kw * l.cin * l.cout); e.stride = e.wsize + l.cout;
    e.jw = e.stride < 16384 ? 16 : 64;
    e.blk_start = R.n ? R.L[R.n - 1].blk_start + cdiv((int)R.L[R.n - 1].stride, R.L[R.n - 1].jw) : 0;
    ++R.n;
}

int flush_reduce(oct_unet* h, hipStream_t s) {
    ReduceAllArgs& R = h->red;
    if (!R.n) return 0;
    const ReduceAllArgs::Entry& last = R.L[R.n - 1];
    const unsigned blocks = last.blk_start + cdiv((int)last.stride, last.jw);
    double bytes = 0;
    for (int i = 0; i < R.n; ++i) bytes += (double)R.L[i].npb * R.L[i].stride * 4;
    ProfScope ps(s, "reduce_all_k", "all", 0, bytes);
    reduce_all_k<<<blocks, kBlock, 0, s>>>(R);
    HIP_OK(hipGetLastError());
    R.n = 0;
    return 0;
}

// the real first layer (1 -> 8 channels, 3x3, image input): streaming backward-weights kernel; it can apply the layer's
// BN-backward transform itself (nothing else reads that dz)
inline bool first_dw_streams(const Layer& l) {
    return l.src == SRC_INPUT && l.cin == 1 && l.cout == 8 && l.kh == 3 && l.has_bn && !l.drop_in;
}

// backward-weights of block li.  fused_apply: `dz` is the masked gradient g' and the kernel applies the BN-backward
// transform of the block on load (every kernel but the generic VALU one of odd first layers can)
int conv_backward_w(oct_unet* h, int li, const void* x_in, int x_is_u8, const void* dz, int B, hipStream_t s, bool fused_apply) {
    const Layer& l = h->plan.L[li];
    const Options& o = h->opt;
    const SrcDesc sd = src_of(h, li, x_in, x_is_u8);
    DwPlan p = dw_plan(l, B, o.mfma_mode, h->cfg.dtype, o);
    p.npb = std::min(p.npb, l.dw_rows);
    ConvBwdWArgs a{};
    a.x0 = sd.x0; a.ab0 = sd.ab0; a.C0 = sd.C0; a.x1 = sd.x1; a.ab1 = sd.ab1; a.C1 = sd.C1;
    a.flags = sd.flags | (l.drop_in ? F_DROP : 0);
    a.dz = dz; a.part = l.dwp;
    a.B = B; a.H = l.H; a.W = l.W; a.Cin = l.cin; a.Cout = l.cout;
    a.tiles_x = cdiv(l.W, kTileX); a.tiles = p.tiles; a.total_tiles = B * a.tiles; a.npb = p.npb;
    a.drop = make_drop(h); a.act_bf16 = h->cfg.dtype;
    if (fused_apply) { a.zf = l.z; a.bnf = l.bn; }
    const double px = (double)B * l.H * l.W, fl = 2.0 * l.kh * l.kw * l.cin * l.cout * px;
    const int es = h->cfg.dtype ? 2 : 4;
    const double by = in_bytes(l, B, x_is_u8, es) + px * l.cout * es * (fused_apply ? 2 : 1);   // conv input once + dz once (fused: g' and z)
    const bool up = l.src == SRC_UP;
    const LaunchCtx lc{&o, B, s, l.name, fl, by};
    int rc = 0;
    if (p.kind == 0 && l.cin == 1 && l.cout == 8 && l.kh == 3 && !(a.flags & (F_AFF | F_DROP | F_TWO | F_UP))) {
        // the real first layer: streaming reduction kernel (kernels_bwd.hpp)
        const int tx = cdiv(l.W, 128), tiles = tx * cdiv(l.H, 8), total = B * tiles;
        const int grid = std::min(total, a.npb);
        a.npb = grid;
        const int bf = a.act_bf16;
        ProfScope ps(s, bf ? "conv_dw_first_k<unsigned short>" : "conv_dw_first_k<float>", l.name, fl, by);
        if (fused_apply) AT_DISPATCH(bf, (conv_dw_first_k<AT, true><<<grid, kBlock, 0, s>>>(a, tx, tiles, total)));
        else AT_DISPATCH(bf, (conv_dw_first_k<AT, false><<<grid, kBlock, 0, s>>>(a, tx, tiles, total)));
        HIP_OK(hipGetLastError());
    } else if (p.kind == 0) {
        if (fused_apply) return fail(-3, "conv_backward_w: the generic first-layer kernel does not apply the BN-backward transform");
        rc = launch_dw<3>(a, p.cic, p.coc, s, l.name, fl, by);   // other 1-channel / odd-channel first layers
    } else if (p.kind == 33 || p.kind == 34) {
        rc = launch_dw_bf16pipe(a, p, l.kh, up, lc);
    } else {
        rc = launch_dw_f32pipe(a, p, l.kh, up, lc);
    }
    if (rc) return rc;
    queue_reduce(h, l, a.npb);
    return 0;
}

// finalize (and, unless the consumers apply it on load, apply) BN backward for block li: its g buffer holds masked
// gradients, stat_part the partials
// `done` (optional): an event that must complete when the block's dz inputs are final.  It is bound to the LAST kernel
// launched here as that dispatch's own completion signal (hipExtLaunchKernelGGL's stopEvent) instead of being recorded
// behind it: a recorded event is a marker packet of its own in the stream, and the next backward-data launch waits
// ~6 us for the command processor to retire it -- on every block that forks a backward-weights kernel.  *bound = false
// when nothing was launched (the caller then records the event the ordinary way).
int bn_backward(oct_unet* h, int li, int nblk, int B, hipStream_t s, bool finalize_only, bool finalized_in_launch,
                hipEvent_t done = nullptr, bool* bound = nullptr) {
    const Layer& l = h->plan.L[li];
    if (bound) *bound = false;
    if (!finalized_in_launch && !(h->opt.timing_skip & 2 && h->drop_step > 2)) {
        BnBwdFinArgs f{};
        f.part = h->stat_part; f.nblk = nblk; f.C = l.cout; f.count = (double)B * l.H * l.W;
        f.bn = l.bn; f.gamma = h->params + l.gamma_off; f.dgamma = h->grads + l.gamma_off; f.dbeta = h->grads + l.beta_off;
        ProfScope ps(s, "bn_bwd_finalize_k", l.name, 0, (double)nblk * 2 * l.cout * 4);
        if (done && finalize_only) {
            hipExtLaunchKernelGGL(bn_bwd_finalize_k, dim3(l.cout), dim3(kBlock), 0, s, nullptr, done, 0, f);
            if (bound) *bound = true;
        } else {
            bn_bwd_finalize_k<<<l.cout, kBlock, 0, s>>>(f);
        }
        HIP_OK(hipGetLastError());
    }
    if (finalize_only) return 0;      // the consumers apply the transform themselves
    const size_t n4 = (size_t)B * l.H * l.W * l.cout / 4;
    const int grid = (int)std::min<size_t>((n4 + kBlock - 1) / kBlock, 8192);
    const int bf = h->cfg.dtype;
    if (bf && l.cout % 8 == 0) {           // 16-byte accesses in bf16 mode
        const size_t n8 = n4 / 2;
        const int grid8 = (int)std::min<size_t>((n8 + kBlock - 1) / kBlock, 8192);
        ProfScope ps(s, "bn_bwd_apply8_bf16_k", l.name, 0, (double)n4 * 8 * 3);
        if (done) {
            hipExtLaunchKernelGGL(bn_bwd_apply8_bf16_k, dim3(grid8), dim3(kBlock), 0, s, nullptr, done, 0,
                                  (bf16_t*)l.g, (const bf16_t*)l.z, (const float*)l.bn, n8, l.cout);
            if (bound) *bound = true;
        } else {
            bn_bwd_apply8_bf16_k<<<grid8, kBlock, 0, s>>>((bf16_t*)l.g, (const bf16_t*)l.z, l.bn, n8, l.cout);
        }
        HIP_OK(hipGetLastError());
        return 0;
    }
    ProfScope ps(s, bf ? "bn_bwd_apply_k<unsigned short>" : "bn_bwd_apply_k<float>", l.name, 0, (double)n4 * (bf ? 8 : 16) * 3);
    if (done) {
        AT_DISPATCH(bf, hipExtLaunchKernelGGL(bn_bwd_apply_k<AT>, dim3(grid), dim3(kBlock), 0, s, nullptr, done, 0,
                                              (AT*)l.g, (const AT*)l.z, (const float*)l.bn, n4, l.cout));
        if (bound) *bound = true;
    } else {
        AT_DISPATCH(bf, bn_bwd_apply_k<AT><<<grid, kBlock, 0, s>>>((AT*)l.g, (const AT*)l.z, l.bn, n4, l.cout));
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int backward_impl(oct_unet* h, const void* x_in, int x_is_u8, const unsigned char* labels, int macro, float loss_scale, hipStream_t s) {
    Plan& pl = h->plan;
    const int nl = (int)pl.L.size(), B = h->last_B;
    const Layer& hd = pl.L[nl - 1]; const Layer& last = pl.L[nl - 2];
    // head: dlogits, masked gradient of the last block + its statistics
    HeadBwdArgs hb{};
    hb.z = last.z; hb.bn = last.bn; hb.w = h->params + hd.w_off; hb.bias = h->params + hd.b_off;
    hb.labels = labels; hb.bc = h->dice_bc; hb.g = last.g; hb.part = h->stat_part; hb.wpart = hd.dwp;
    hb.HW = hd.H * hd.W; hb.nblk = head_nblk(hb.HW, B); hb.B = B; hb.macro = macro; hb.loss_scale = loss_scale; hb.act_bf16 = h->cfg.dtype;
    hb.focal_w = h->focal_w; hb.focal_gamma = h->focal_gamma; hb.focal_cw = h->focal_cw; hb.focal_clip_mod = h->opt.focal_clip_mod; hb.inv_count = 1.f / ((float)B * hb.HW);
    // backward-data weights of every block for this step's parameters: transposed / effective fp32 kernels, then their
    // bf16 operand layouts.  Nothing needs them before the first backward-data launch, so they run on the side stream
    // under the head backward and the last block's BN backward.
    const bool side_ok = h->opt.dw_side_stream && h->side && !(t_prof && t_prof->on);
    hipStream_t ps_ = side_ok ? h->side : s;
    if (side_ok) { HIP_OK(hipEventRecord(h->fork_ev[0], s)); HIP_OK(hipStreamWaitEvent(h->side, h->fork_ev[0], 0)); }
    {
        ProfScope ps(ps_, "prep_wt_k", "all", 0, (double)h->wt_total * 8);
        prep_wt_k<<<std::min<unsigned>((h->wt_total + kBlock - 1) / kBlock, 2048u), kBlock, 0, ps_>>>(h->wt_descs, h->n_wt, h->wt_total);
        HIP_OK(hipGetLastError());
    }
    if (h->opt.mfma_mode && h->n_wbx_b) {
        ProfScope ps(ps_, "prep_wbx_k", "all", 0, (double)h->wbx_b_total * 8 * (4 + 2 * (h->cfg.dtype ? 1 : 3)));
        prep_wbx_k<<<std::min<unsigned>((h->wbx_b_total + kBlock - 1) / kBlock, 2048u), kBlock, 0, ps_>>>(h->wbx_descs + h->n_wbx_f, h->n_wbx_b, h->wbx_b_total);
        HIP_OK(hipGetLastError());
    }
    if (h->opt.mfma_mode && h->n_wbt_b) {
        ProfScope ps(ps_, "prep_wbt_k", "all", 0, (double)h->wbt_b_total * 8 * (4 + 2 * (h->cfg.dtype ? 1 : 3)));
        prep_wbt_k<<<std::min<unsigned>((h->wbt_b_total + kBlock - 1) / kBlock, 2048u), kBlock, 0, ps_>>>(h->wbt_descs + h->n_wbt_f, h->n_wbt_b, h->wbt_b_total);
        HIP_OK(hipGetLastError());
    }
    if (side_ok) HIP_OK(hipEventRecord(h->prep_ev, h->side));
    bool prep_pending = side_ok;
    bool pending_fin = false;                 // the pending statistics were finalized by the launch that emitted them
    if (fin_ok(h, last)) { hb.fin = fin_desc(h, nl - 2, 1, B); pending_fin = true; }
    int rc = DISPATCH_C(launch_head_bwd, h->cfg.n_cls, hb, hd.cin, B, s);
    if (rc) return rc;
    int pending_nblk = B * hb.nblk;           // number of stat partial rows waiting for block (li-1)
    bool forked = false;
    struct PendingDw { int li; bool fuse; };
    std::vector<PendingDw> pend;          // forked backward-weights launches not yet issued
    unsigned n_forks = 0;
    h->red.n = 0;
    queue_reduce(h, hd, pending_nblk);   // head kernel/bias gradient rows written by head_bwd_k

    for (int li = nl - 2; li >= 0; --li) {
        Layer& l = pl.L[li];
        const Options& o = h->opt;
        // backward-data launches of this block: dz x transposed / effective weights through the MFMA implicit-GEMM kernels
        auto dx_args = [&](void* gout, int Cg, int ci_off, const Layer* prod, bool up) {
            IgemmArgs g{};
            g.x0 = l.g; g.C0 = l.cout; g.flags = 0; g.Cin = l.cout;
            g.w = l.wt; g.w_ld = l.cin; g.m_off = ci_off; g.out = gout; g.Mout = Cg;
            g.Hi = l.H; g.Wi = l.W; g.Ho = up ? l.H / 2 : l.H; g.Wo = up ? l.W / 2 : l.W;
            g.part = prod ? h->stat_part : nullptr; g.zin = prod ? prod->z : nullptr; g.bnin = prod ? prod->bn : nullptr;
            g.drop_out = (up && l.drop_in) ? 1 : 0; g.drop = make_drop(h); g.act_bf16 = h->cfg.dtype;
            g.wbx = l.wbx_b; g.wbx_M = l.cin;
            g.wbt = l.wbt_b ? l.wbt_b + (size_t)(ci_off / Cg) * (wbt_bytes(3, l.cout, h->cfg.dtype ? 1 : 3, l.bt_m2_b) / 2) : nullptr;
            g.bt_m2 = l.bt_m2_b;
            return g;
        };
        // g buffer of block li is complete (+ partials in stat_part).  Its BN-backward transform dz = ga g' + gb z + gd is
        // applied by the consumers of dz while they stage it -- the backward-weights kernel and the backward-data launches --
        // whenever all of them can (the bf16-pipe conv kernels and every MFMA backward-weights kernel); otherwise by the
        // stand-alone pass, in place.  First layer: its dz has ONE consumer, the streaming backward-weights kernel.
        const int dwkind = dw_plan(l, B, o.mfma_mode, h->cfg.dtype, o).kind;
        bool fuse;
        if (l.src == SRC_INPUT) {
            fuse = o.fuse_first_apply && li == 0 && first_dw_streams(l) && dwkind == 0 &&
                   !(src_of(h, li, x_in, x_is_u8).flags & (F_AFF | F_DROP | F_TWO | F_UP));
        } else {
            const int cg = bx_bwd_cg(l);
            const bool up = l.src == SRC_UP;
            const ConvRoute r0 = conv_route(dx_args(nullptr, cg, 0, nullptr, up), up ? A_DOWN2 : A_NORMAL, o);
            // (three bf16-pipe instantiations stay out: the stride-2 gather beyond the coefficient rows its LDS holds; the
            //  thin kernel at 32 K channels, whose staging registers for g' AND z no longer fit; and the thin kernel at 16 K
            //  channels with 16 output channels, where the second raw register set spills under the 256-register budget of two
            //  blocks per CU: 87 us against 45 + 33 for the separate pass at B = 32, 128 x 256)
            fuse = o.fuse_bn_apply && dwkind != 0 && r0 != ROUTE_F32 && !(up && r0 == ROUTE_BX && l.cout > kBxGbDown2MaxC) &&
                   !(r0 == ROUTE_BT && l.cout == 32) && !(r0 == ROUTE_BT && l.cout == 16 && cg == 16 && !o.fuse_bn_apply16) &&
                   (l.src != SRC_CONCAT || conv_route(dx_args(nullptr, cg, cg, nullptr, false), A_NORMAL, o) != ROUTE_F32);
        }
        l.g_masked = fuse;
        // 3x3 layers with 8 output channels on the thin kernel (the full-resolution convs): their backward-data launches
        // reduce the backward-weights too (conv_bt_k FDW) -- g', z and the producer's z are read once for both
        bool fdw = false;
        if (fuse && o.fuse_dw_thin && l.kh == 3 && l.cout == 8 && !l.drop_in && (l.src == SRC_PREV || l.src == SRC_CONCAT) &&
            bx_bwd_cg(l) == 8 && l.dw_rows >= std::min(B * cdiv(l.W, 32) * cdiv(l.H, 8), 512) &&      // (one slab per block of that launch)
            conv_route(dx_args(nullptr, 8, 0, nullptr, false), A_NORMAL, o) == ROUTE_BT &&
            (l.src != SRC_CONCAT || conv_route(dx_args(nullptr, 8, 8, nullptr, false), A_NORMAL, o) == ROUTE_BT))
            fdw = true;
        // (the per-launch profiler wants serial launches; the input layer has no backward-data launch to run beside, so its
        //  backward-weights kernel stays on the caller's stream: no fork, no join wait in front of the slab reduce)
        const bool fork = side_ok && !fdw && l.src != SRC_INPUT;
        // Forked backward-weights launches go to the side stream in GROUPS (dw_fork_group blocks per fork): every event the
        // side stream waits for costs the caller's stream ~6 us in front of its next launch (the dispatch that carries the
        // completion signal has to be retired by the command processor first), and a block's dz, z and record stay valid
        // until the end of the backward pass, so its backward-weights kernel may start a block or two late.
        const bool tail_here = li == first_mid_layer(pl);      // every gradient from the first bottleneck conv on is queued here
        if (fork) pend.push_back({li, fuse});
        const bool flush = !pend.empty() && (!fork || (int)pend.size() >= o.dw_fork_group || tail_here);
        hipEvent_t fe = flush ? h->fork_ev[1 + n_forks++ % (h->fork_ev.size() - 1)] : nullptr;
        bool fe_bound = false;
        rc = bn_backward(h, li, pending_nblk, B, s, fuse, pending_fin, (flush && o.fork_on_launch) ? fe : nullptr, &fe_bound);
        if (rc) return rc;
        pending_fin = false;
        if (flush) {
            if (!fe_bound) HIP_OK(hipEventRecord(fe, s));        // dz of every pending block is final here
            HIP_OK(hipStreamWaitEvent(h->side, fe, 0));
            for (const PendingDw& p : pend) {
                rc = conv_backward_w(h, p.li, x_in, x_is_u8, pl.L[p.li].g, B, h->side, p.fuse);
                if (rc) return rc;
            }
            pend.clear();
            forked = true;
        }
        if (!fdw && !fork) {
            rc = conv_backward_w(h, li, x_in, x_is_u8, l.g, B, s, fuse);
            if (rc) return rc;
        }
        if (tail_here && (forked || h->tail_event)) {
            // every parameter gradient at offsets >= L[li].w_off (bottleneck, decoder, head: Keras creation order) is
            // final once the queued slabs are summed: the DP launcher all-reduces that segment on a side stream while
            // the encoder backward runs.  The sum runs on the handle's side stream, behind the backward-weights
            // kernels that write those slabs (the slabs written by launches of the caller's stream -- head, fused dX+dW --
            // are older than the fork event that stream last waited for): the caller's stream neither waits for the side
            // stream here nor carries the reduce, and the end-of-backward reduce is left with the encoder's slabs.
            hipStream_t rs = forked ? h->side : s;
            rc = flush_reduce(h, rs);
            if (rc) return rc;
            if (h->tail_event) HIP_OK(hipEventRecord(h->tail_event, rs));
        }
        if (l.src == SRC_INPUT) break;
        if (prep_pending) { HIP_OK(hipStreamWaitEvent(s, h->prep_ev, 0)); prep_pending = false; }   // first backward-data launch
        int rows = 0;
        // (xsrc: with fdw, the layer whose output is this launch's slice of the conv input -- `prod` when the launch masks)
        auto dx = [&](void* gout, int Cg, int ci_off, const Layer* prod, bool up, const Layer* xsrc = nullptr, bool first = true) -> int {
            IgemmArgs g = dx_args(gout, Cg, ci_off, prod, up);
            if (fuse) { g.gb_z = l.z; g.gb_bn = l.bn; }
            const double px = (double)B * l.H * l.W, pxg = (double)B * g.Ho * g.Wo;
            double fl = 2.0 * l.kh * l.kw * Cg * l.cout * px;                // algorithmic flops of the original conv's dX
            const int es = h->cfg.dtype ? 2 : 4;
            double by = px * l.cout * es * (fuse ? 2 : 1) + pxg * Cg * es * (prod ? 2 : 1);
            if (fdw) {
                g.dw_x = xsrc->z; g.dw_ab = xsrc->bn; g.dw_part = l.dwp; g.dw_Cin = l.cin; g.dw_ci_off = ci_off; g.dw_bias = first ? 1 : 0;
                fl *= 2; if (!prod) by += pxg * Cg * es;                      // + the dW flops; + the X read where no mask reads it
            }
            const LaunchCtx lc{&o, B, s, l.name, fl, by};
            if (prod && fin_ok(h, *prod) && conv_route(g, up ? A_DOWN2 : A_NORMAL, o) == ROUTE_BT) {
                g.fin = fin_desc(h, (int)(prod - pl.L.data()), 1, B); pending_fin = true;
            }
            if (up) return launch_igemm<3, A_DOWN2, EPI_MASK>(g, lc, &rows);
            return prod ? launch_igemm<3, A_NORMAL, EPI_MASK>(g, lc, &rows)
                        : launch_igemm<3, A_NORMAL, EPI_RAW>(g, lc, &rows);
        };
        switch (l.src) {
            case SRC_PREV: {
                Layer& p = pl.L[li - 1];
                rc = dx(p.g, p.cout, 0, &p, false, &p, true);
                if (rc) return rc;
                pending_nblk = rows;
                if (fdw) queue_reduce(h, l, rows);
                break;
            }
            case SRC_POOL: {  // gradient wrt the pooled tensor (raw), then route through the pool into block li-1
                Layer& p = pl.L[li - 1];
                rc = dx(h->gpooled[l.level - 1], l.cin, 0, nullptr, false);
                if (rc) return rc;
                PoolBwdArgs pb{};
                pb.gp = h->gpooled[l.level - 1]; pb.z = p.z; pb.bn = p.bn; pb.g = p.g; pb.part = h->stat_part;
                pb.act_bf16 = h->cfg.dtype; pb.H = p.H; pb.W = p.W; pb.C = p.cout; pb.tiles_x = cdiv(p.W / 2, kTileX); pb.tiles = tiles_of(p.H / 2, p.W / 2);
                const int bf = h->cfg.dtype, c4 = p.cout / 4;
                const double pbytes = (double)B * p.H * p.W * p.cout * (bf ? 2 : 4) * 3.25;
                if (c4 >= 1 && c4 <= 64 && (c4 & (c4 - 1)) == 0) {     // flat, channel-contiguous mapping
                    const bool v8 = bf && p.cout % 8 == 0;                               // bf16 storage: 8 channels (16 bytes) per thread
                    const size_t items = (size_t)B * (p.H / 2) * (p.W / 2) * (v8 ? c4 / 2 : c4);
                    const size_t cap = (size_t)B * cdiv(p.H, 2) * cdiv(p.W, kTileX);     // statistic rows carve() guarantees
                    const int grid = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(cdiv((int)items, kBlock), 2048), cap));
                    if (fin_ok(h, p)) { pb.fin = fin_desc(h, li - 1, 1, B); pending_fin = true; }
                    ProfScope ps(s, bf ? (v8 ? "pool_bwd_flat_k<unsigned short,8>" : "pool_bwd_flat_k<unsigned short>") : "pool_bwd_flat_k<float>", p.name, 0, pbytes);
                    if (v8) pool_bwd_flat_k<bf16_t, 8><<<grid, kBlock, 0, s>>>(pb, B);
                    else AT_DISPATCH(bf, pool_bwd_flat_k<AT><<<grid, kBlock, 0, s>>>(pb, B));
                    HIP_OK(hipGetLastError());
                    pending_nblk = grid;
                    break;
                }
                const int c_t = 4;
                dim3 grid(pb.tiles, p.cout / c_t, B);
                ProfScope ps(s, bf ? "pool_bwd_k<4,unsigned short>" : "pool_bwd_k<4,float>", p.name, 0, pbytes);
                AT_DISPATCH(bf, pool_bwd_k<4, AT><<<grid, kBlock, 0, s>>>(pb));
                HIP_OK(hipGetLastError());
                pending_nblk = B * pb.tiles;
                break;
            }
            case SRC_UP: {
                Layer& p = pl.L[li - 1];
                rc = dx(p.g, p.cout, 0, &p, true);
                if (rc) return rc;
                pending_nblk = rows;
                break;
            }
            case SRC_CONCAT: {
                Layer& p = pl.L[li - 1]; Layer& k = pl.L[l.skip_from];
                // skip half first (raw, merged later by pool_bwd of that encoder level) ...
                rc = dx(k.g, k.cout, p.cout, nullptr, false, &k, false);
                if (rc) return rc;
                const int rows_skip = rows;
                // ... then the up-path half, whose statistics must be the ones pending for block li-1
                rc = dx(p.g, p.cout, 0, &p, false, &p, true);
                if (rc) return rc;
                pending_nblk = rows;
                if (fdw) {
                    if (rows != rows_skip) return fail(-3, "fused backward-weights: the two halves ran different grids");
                    queue_reduce(h, l, rows);
                }
                break;
            }
            default: return fail(-3, "backward: bad src");
        }
        if (rc) return rc;
    }
    if (forked) { HIP_OK(hipEventRecord(h->join_ev, h->side)); HIP_OK(hipStreamWaitEvent(s, h->join_ev, 0)); }
    return flush_reduce(h, s);
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* oct_last_error(void) { return g_err.c_str(); }
const char* oct_version(void) { return "oct_unet_hip 0.3 (gfx950) src:" OCT_SRC_HASH; }

void oct_unet_cfg_default(oct_unet_cfg* c) {
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->in_ch = 1; c->n_cls = 3; c->H = 256; c->W = 512; c->max_batch = 1;
    c->start_neurons = 8; c->pool_layers = 4; c->conv_layers = 2; c->enc_k = 3; c->dec_k = 2;
    c->dtype = 0; c->training = 0; c->bn_eps = 1e-3f; c->bn_momentum = 0.99f; c->dropout_rate = 0.5f;
    c->bn_unbiased_moving_var = 1; c->seed = 0x0C7ull;
}

int oct_unet_cfg_check(const oct_unet_cfg* c) { return check_cfg(c); }

size_t oct_unet_param_count(const oct_unet_cfg* c) { return check_cfg(c) ? 0 : build_plan(*c).n_params; }
size_t oct_unet_state_count(const oct_unet_cfg* c) { return check_cfg(c) ? 0 : build_plan(*c).n_state; }
int oct_unet_layer_count(const oct_unet_cfg* c) { return check_cfg(c) ? -1 : (int)build_plan(*c).L.size(); }

size_t oct_unet_workspace_bytes(const oct_unet_cfg* c) {
    if (check_cfg(c)) return 0;
    Plan pl = build_plan(*c);
    return carve(*c, pl, nullptr, nullptr, g_opt);
}

int oct_unet_layer_info(const oct_unet_cfg* c, int index, oct_layer_info* out) {
    if (int rc = check_cfg(c)) return rc;
    if (!out) return fail(-1, "null out");
    Plan pl = build_plan(*c);
    if (index < 0 || index >= (int)pl.L.size()) return fail(-1, "layer index out of range");
    const Layer& l = pl.L[index];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", l.name);
    out->kh = l.kh; out->kw = l.kw; out->cin = l.cin; out->cout = l.cout; out->has_bn = l.has_bn;
    out->out_h = l.H; out->out_w = l.W;
    out->kernel_off = l.w_off; out->bias_off = l.b_off; out->gamma_off = l.gamma_off; out->beta_off = l.beta_off;
    out->moving_mean_off = l.mm_off; out->moving_var_off = l.mv_off;
    return 0;
}

int oct_unet_create(const oct_unet_cfg* c, float* params, float* grads, float* state, void* ws, size_t ws_bytes,
                    oct_unet** out) {
    if (int rc = check_cfg(c)) return rc;
    if (!out || !params || !state || !ws) return fail(-1, "null pointer argument");
    if (c->training && !grads) return fail(-1, "training handle needs a grads buffer");
    oct_unet* h = new oct_unet();
    h->cfg = *c; h->plan = build_plan(*c); h->opt = g_opt;
    h->params = params; h->grads = grads; h->state = state;
    const size_t need = carve(*c, h->plan, nullptr, nullptr, h->opt);
    if (ws_bytes < need) { delete h; return fail(-4, "workspace too small: need " + std::to_string(need) + " bytes"); }
    if (((uintptr_t)ws & 255) || ((uintptr_t)params & 15) || ((uintptr_t)state & 15) || (grads && ((uintptr_t)grads & 15))) {
        delete h; return fail(-1, "buffers must be aligned (workspace 256 B, params/grads/state 16 B)");
    }
    carve(*c, h->plan, h, (char*)ws, h->opt);
    if (c->training) {
        std::vector<WtDesc> d;
        unsigned off = 0;
        for (auto& l : h->plan.L) {
            if (!l.wt) continue;
            WtDesc w{}; w.w = params + l.w_off; w.wt = l.wt; w.kh = l.kh; w.cin = l.cin; w.cout = l.cout;
            w.mode = l.kh == 3 ? 0 : 1; w.start = off; w.count = 9u * l.cin * l.cout;
            off += w.count; d.push_back(w);
        }
        h->n_wt = (int)d.size(); h->wt_total = off;
        hipError_t e = hipMemcpy(h->wt_descs, d.data(), d.size() * sizeof(WtDesc), hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete h; return fail(-5, std::string("hipMemcpy(wt_descs): ") + hipGetErrorString(e)); }
    }
    {   // split-weight descriptors: forward entries first, then backward-data entries
        std::vector<WbxDesc> d;
        const int ns = c->dtype ? 1 : 3;
        auto items = [](int KH, int Kc, int M, int MB) { return (unsigned)(((Kc + 15) / 16) * ((M + MB - 1) / MB) * KH * KH * MB * 2); };
        unsigned off = 0;
        for (auto& l : h->plan.L) {
            if (!l.wbx_f) continue;
            WbxDesc w{}; w.src = params + l.w_off; w.dst = l.wbx_f; w.KH = l.kh; w.Kc = l.cin; w.M = l.cout; w.ld = l.cout;
            w.MB = bx_mb(l.cout); w.NS = ns; w.start = off; w.count = items(w.KH, w.Kc, w.M, w.MB);
            off += w.count; d.push_back(w);
        }
        h->n_wbx_f = (int)d.size(); h->wbx_f_total = off;
        off = 0;
        for (auto& l : h->plan.L) {
            if (!l.wbx_b) continue;
            WbxDesc w{}; w.src = l.wt; w.dst = l.wbx_b; w.KH = 3; w.Kc = l.cout; w.M = l.cin; w.ld = l.cin;
            w.MB = bx_mb(bx_bwd_cg(l)); w.NS = ns; w.start = off; w.count = items(3, w.Kc, w.M, w.MB);
            off += w.count; d.push_back(w);
        }
        h->n_wbx_b = (int)d.size() - h->n_wbx_f; h->wbx_b_total = off;
        if (!d.empty()) {
            hipError_t e = hipMemcpy(h->wbx_descs, d.data(), d.size() * sizeof(WbxDesc), hipMemcpyHostToDevice);
            if (e != hipSuccess) { delete h; return fail(-5, std::string("hipMemcpy(wbx_descs): ") + hipGetErrorString(e)); }
        }
    }
    {   // thin-kernel weight slices: forward entries first, then backward-data entries (one per 16-row slice)
        std::vector<WbtDesc> d;
        const int ns = c->dtype ? 1 : 3;
        unsigned off = 0;
        for (auto& l : h->plan.L) {
            if (!l.wbt_f) continue;
            WbtDesc w{}; w.src = params + l.w_off; w.dst = l.wbt_f; w.KH = l.kh; w.Kc = l.cin; w.M = l.cout; w.ld = l.cout;
            w.m_off = 0; w.CT = l.cin; w.NS = ns; w.m2 = l.bt_m2_f; w.start = off; w.count = (unsigned)wbt_groups(l.kh, l.cin, l.bt_m2_f) * 64;
            off += w.count; d.push_back(w);
        }
        h->n_wbt_f = (int)d.size(); h->wbt_f_total = off;
        off = 0;
        for (auto& l : h->plan.L) {
            if (!l.wbt_b) continue;
            const int cg = bx_bwd_cg(l);
            for (int sl = 0; sl < l.cin / cg; ++sl) {
                WbtDesc w{}; w.src = l.wt; w.dst = l.wbt_b + (size_t)sl * (wbt_bytes(3, l.cout, ns, l.bt_m2_b) / 2); w.KH = 3; w.Kc = l.cout;
                w.M = l.cin; w.ld = l.cin; w.m_off = sl * cg; w.CT = l.cout; w.NS = ns; w.m2 = l.bt_m2_b; w.start = off;
                w.count = (unsigned)wbt_groups(3, l.cout, l.bt_m2_b) * 64;
                off += w.count; d.push_back(w);
            }
        }
        h->n_wbt_b = (int)d.size() - h->n_wbt_f; h->wbt_b_total = off;
        if (!d.empty()) {
            hipError_t e = hipMemcpy(h->wbt_descs, d.data(), d.size() * sizeof(WbtDesc), hipMemcpyHostToDevice);
            if (e != hipSuccess) { delete h; return fail(-5, std::string("hipMemcpy(wbt_descs): ") + hipGetErrorString(e)); }
        }
    }
    {
        hipError_t e0 = hipMemset(h->fin_counters, 0, 2 * h->plan.L.size() * sizeof(unsigned));
        if (e0 != hipSuccess) { delete h; return fail(-5, std::string("hipMemset(fin_counters): ") + hipGetErrorString(e0)); }
    }
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = (float)((double)i / 255.0);
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_u8_lut), lut, sizeof lut);
    if (e != hipSuccess) { delete h; return fail(-5, std::string("hipMemcpyToSymbol: ") + hipGetErrorString(e)); }
    if (c->training) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);         // lo = numerically largest = least urgent
        if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo) != hipSuccess) h->side = nullptr;
        if (h->side) {
            h->fork_ev.resize(9);
            // the events order work between two streams of ONE device: no system-scope fence (an L2 write-back in front of
            // the next launch, which reads these layers' tensors from L2) unless asked for
            const unsigned evf = hipEventDisableTiming | (h->opt.event_sysfence ? 0u : (unsigned)hipEventDisableSystemFence);
            bool ok = hipEventCreateWithFlags(&h->join_ev, evf) == hipSuccess &&
                      hipEventCreateWithFlags(&h->prep_ev, evf) == hipSuccess;
            for (auto& e : h->fork_ev) ok = ok && hipEventCreateWithFlags(&e, evf) == hipSuccess;
            if (!ok) { (void)hipStreamDestroy(h->side); h->side = nullptr; }
        }
    }
    *out = h;
    return 0;
}

void oct_unet_destroy(oct_unet* h) {
    if (!h) return;
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        for (auto e : h->fork_ev) if (e) (void)hipEventDestroy(e);
        if (h->join_ev) (void)hipEventDestroy(h->join_ev);
        if (h->prep_ev) (void)hipEventDestroy(h->prep_ev);
        (void)hipStreamDestroy(h->side);
    }
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    delete h;
}

int oct_unet_forward(oct_unet* h, const void* x, int x_is_u8, int B, int training, const oct_unet_io* io, oct_stream_t stream) {
    if (!h || !x) return fail(-1, "null handle or input");
    if (B < 1 || B > h->cfg.max_batch) return fail(-1, "B out of range (1..max_batch)");
    if (training && !h->cfg.training) return fail(-1, "handle was created without training workspaces");
    if (training) {  // every training forward draws a fresh dropout mask; backward replays the same one
        if (h->drop_advance) ++h->drop_step;
        h->drop_advance = 1;
    }
    t_prof = &h->prof;
    const int rc = forward_impl(h, x, x_is_u8, B, training, io, (hipStream_t)stream);
    if (rc) return rc;
    h->last_x = x; h->last_u8 = x_is_u8; h->dice_final = 0;
    return 0;
}

int oct_unet_loss_dice(oct_unet* h, float smooth, float* out4, oct_stream_t stream) {
    if (!h) return fail(-1, "null handle");
    if (!h->have_dice) return fail(-1, "loss_dice needs a preceding forward with io.labels");
    DiceFinArgs a{};
    a.part = h->dice_part; a.B = h->last_B; a.C = h->cfg.n_cls; a.nblk = head_nblk(h->cfg.H * h->cfg.W, h->last_B);
    a.N = dice_n(a.C); a.smooth = smooth; a.out4 = h->loss4; a.out4_user = out4; a.bc = h->dice_bc;
    a.n_user = 4; a.inv_count = 1.0 / ((double)h->last_B * h->cfg.H * h->cfg.W); a.focal_w = h->focal_w;
    dice_finalize_k<<<1, kBlock, 0, (hipStream_t)stream>>>(a);
    HIP_OK(hipGetLastError());
    h->dice_final = 1;
    return 0;
}

int oct_unet_set_focal_dice(oct_unet* h, float focal_loss_weight, float gamma, const float* class_weight_dev) {
    if (!h) return fail(-1, "null handle");
    if (!(focal_loss_weight >= 0.f && focal_loss_weight <= 1.f) || !(gamma >= 0.f)) return fail(-1, "focal_dice: weight must be in [0,1], gamma >= 0");
    h->focal_w = focal_loss_weight; h->focal_gamma = gamma; h->focal_cw = class_weight_dev;
    return 0;
}

int oct_unet_loss_focal_dice(oct_unet* h, float smooth, float* out8, oct_stream_t stream) {
    if (!h) return fail(-1, "null handle");
    if (!h->have_dice) return fail(-1, "loss_focal_dice needs a preceding forward with io.labels");
    DiceFinArgs a{};
    a.part = h->dice_part; a.B = h->last_B; a.C = h->cfg.n_cls; a.nblk = head_nblk(h->cfg.H * h->cfg.W, h->last_B);
    a.N = dice_n(a.C); a.smooth = smooth; a.out4 = h->loss4; a.out4_user = out8; a.bc = h->dice_bc;
    a.n_user = 8; a.inv_count = 1.0 / ((double)h->last_B * h->cfg.H * h->cfg.W); a.focal_w = h->focal_w;
    dice_finalize_k<<<1, kBlock, 0, (hipStream_t)stream>>>(a);
    HIP_OK(hipGetLastError());
    h->dice_final = 1;
    return 0;
}

int oct_unet_backward(oct_unet* h, const unsigned char* labels, int macro, float loss_scale, oct_stream_t stream) {
    if (!h || !labels) return fail(-1, "null handle or labels");
    if (!h->last_training || !h->have_dice || !h->dice_final)
        return fail(-1, "backward needs a training forward with io.labels followed by oct_unet_loss_dice");
    t_prof = &h->prof;
    const int rc = backward_impl(h, h->last_x, h->last_u8, labels, macro, loss_scale, (hipStream_t)stream);
    if (rc && h->side && h->join_ev) {
        // an error part-way: work may have been forked to the internal side stream -- join it to the caller's stream before
        // returning, so that the caller's stream order covers everything this call launched
        const std::string msg = oct_last_error();
        if (hipEventRecord(h->join_ev, h->side) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)stream, h->join_ev, 0);
        fail(rc, msg);
    }
    return rc;
}

int oct_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                  long step, oct_stream_t stream) {
    if (!p || !g || !m || !v || step < 1) return fail(-1, "adam: bad arguments");
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)b2, (double)step)) / (1.0 - std::pow((double)b1, (double)step));
    const int grid = (int)std::min<size_t>((n + kBlock - 1) / kBlock, 4096);
    adam_k<<<grid, kBlock, 0, (hipStream_t)stream>>>(p, g, m, v, n, (float)lr_t, b1, b2, eps);
    HIP_OK(hipGetLastError());
    return 0;
}

int oct_sgd_step(float* p, const float* g, float* mom, size_t n, float lr, float momentum, oct_stream_t stream) {
    if (!p || !g) return fail(-1, "sgd: bad arguments");
    const int grid = (int)std::min<size_t>((n + kBlock - 1) / kBlock, 4096);
    sgd_k<<<grid, kBlock, 0, (hipStream_t)stream>>>(p, g, mom, n, lr, momentum);
    HIP_OK(hipGetLastError());
    return 0;
}

int oct_unet_set_dropout_step(oct_unet* h, unsigned long long step) {
    if (!h) return fail(-1, "null handle");
    h->drop_step = step; h->drop_advance = 0;
    return 0;
}

int oct_unet_dropout_mask(oct_unet* h, int B, unsigned char* mask, oct_stream_t stream) {
    if (!h || !mask) return fail(-1, "null argument");
    const int P = h->cfg.pool_layers;
    const size_t n = (size_t)B * (h->cfg.H >> P) * (h->cfg.W >> P) * ((size_t)h->cfg.start_neurons << P);
    dropout_mask_k<<<(int)std::min<size_t>((n + 255) / 256, 4096), 256, 0, (hipStream_t)stream>>>(mask, n, make_drop(h));
    HIP_OK(hipGetLastError());
    return 0;
}

int oct_unet_graph_capture(oct_unet* h, const void* x, int x_is_u8, int B, const oct_unet_io* io, oct_stream_t stream) {
    if (!h || !x) return fail(-1, "null handle or input");
    if (B < 1 || B > h->cfg.max_batch) return fail(-1, "B out of range (1..max_batch)");
    hipStream_t s = (hipStream_t)stream;
    if (!s) return fail(-1, "graph capture needs a non-default stream");
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    t_prof = nullptr;  // no event records inside a capture
    HIP_OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = forward_impl(h, x, x_is_u8, B, 0, io, s);
    hipError_t e = hipStreamEndCapture(s, &h->graph);
    if (rc) return rc;
    if (e != hipSuccess) return fail(-5, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    HIP_OK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    return 0;
}

int oct_unet_graph_launch(oct_unet* h, oct_stream_t stream) {
    if (!h || !h->graph_exec) return fail(-1, "no captured graph");
    HIP_OK(hipGraphLaunch(h->graph_exec, (hipStream_t)stream));
    return 0;
}

int oct_unet_set_tail_event(oct_unet* h, void* hip_event) {
    if (!h) return fail(-1, "null handle");
    h->tail_event = (hipEvent_t)hip_event;
    return 0;
}

size_t oct_unet_grad_tail_offset(const oct_unet_cfg* c) {
    if (check_cfg(c)) return 0;
    const Plan pl = build_plan(*c);
    return pl.L[first_mid_layer(pl)].w_off;
}

int oct_unet_profile_begin(oct_unet* h) {
    if (!h) return fail(-1, "null handle");
    for (auto& r : h->prof.recs) { h->prof.pool.push_back(r.e0); h->prof.pool.push_back(r.e1); }
    h->prof.recs.clear();
    h->prof.on = true;
    return 0;
}

int oct_unet_profile_end(oct_unet* h, oct_profile_entry* out, int max_entries, int* n_out) {
    if (!h || !n_out) return fail(-1, "null argument");
    h->prof.on = false;
    HIP_OK(hipDeviceSynchronize());
    std::vector<oct_profile_entry> agg;
    for (auto& r : h->prof.recs) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, r.e0, r.e1));
        oct_profile_entry* e = nullptr;
        for (auto& a : agg) if (r.kernel == a.kernel && r.layer == a.layer) { e = &a; break; }
        if (!e) {
            oct_profile_entry n{}; snprintf(n.kernel, sizeof n.kernel, "%s", r.kernel.c_str());
            snprintf(n.layer, sizeof n.layer, "%s", r.layer.c_str());
            agg.push_back(n); e = &agg.back();
        }
        e->launches += 1; e->total_ms += ms; e->flops += r.flops; e->bytes += r.bytes;
    }
    *n_out = (int)agg.size();
    for (int i = 0; i < (int)agg.size() && i < max_entries && out; ++i) out[i] = agg[i];
    return 0;
}

int oct_boundary_maps(const unsigned char* labels, int B, int H, int W, int n_cls, int bg_ilm, int bg_csi,
                      unsigned char* maps, oct_stream_t stream) {
    if (!labels || !maps || B < 1 || H < 1 || W < 1 || n_cls < 2) return fail(-1, "boundary_maps: bad arguments");
    const size_t n = (size_t)B * H * W;
    boundary_maps_k<<<(int)std::min<size_t>((n + kBlock - 1) / kBlock, 8192), kBlock, 0, (hipStream_t)stream>>>(
        labels, maps, B, H, W, n_cls, bg_ilm, bg_csi);
    HIP_OK(hipGetLastError());
    return 0;
}

namespace {
struct Opt { const char* name; int Options::*var; int lo, hi; };     // values are clamped to [lo, hi]
const Opt k_opts[] = {
    {"igemm_persistent_min_tiles", &Options::persist_min_tiles, 1, 1 << 30}, {"dw32_blocks", &Options::dw32_blocks, 64, 1 << 20},
    {"dw16_blocks", &Options::dw16_blocks, 64, 1 << 20}, {"igemm_persistent_blocks", &Options::igemm_p_blocks, 8, 1 << 20},
    {"igemm_min_blocks", &Options::igemm_min_blocks, 1, 1 << 30}, {"dwpair8_enable", &Options::dwpair8, 0, 1},
    {"pair8_geometry", &Options::pair_geo, 111, 221}, {"pair8_min_tiles", &Options::pair_min_tiles, 1, 1 << 30},
    {"thin8_min_tiles", &Options::thin_min_tiles, 1, 1 << 30}, {"focal_clip_modulation", &Options::focal_clip_mod, 0, 1},
    {"mfma_mode", &Options::mfma_mode, 0, 1}, {"bx_min_blocks", &Options::bx_min_blocks, 1, 1 << 30},
    {"dwbx_blocks", &Options::dwbx_blocks, 8, 1 << 20}, {"bt_blocks_per_cu", &Options::bt_blocks_per_cu, 0, 8},
    {"dwbt_f32_all", &Options::dwbt_f32_all, 0, 1}, {"bt_m2", &Options::bt_m2, 0, 1},
    {"fuse_first_apply", &Options::fuse_first_apply, 0, 1}, {"fuse_bn_apply", &Options::fuse_bn_apply, 0, 1},
    {"fuse_bn_finalize", &Options::fuse_bn_finalize, 0, 1}, {"bx_waves", &Options::bx_waves, 4, 8},
    {"dw_side_stream", &Options::dw_side_stream, 0, 1}, {"timing_skip", &Options::timing_skip, 0, 255}, {"fuse_dw_thin", &Options::fuse_dw_thin, 0, 1},
    {"dwbx_enable", &Options::dwbx_enable, 0, 1},
    {"bx_two_blocks", &Options::bx_two_blocks, 0, 1}, {"fork_on_launch", &Options::fork_on_launch, 0, 1},
    {"event_sysfence", &Options::event_sysfence, 0, 1}, {"dw_fork_group", &Options::dw_fork_group, 1, 8}, {"fuse_bn_apply16", &Options::fuse_bn_apply16, 0, 1},
};
}  // namespace

int oct_get_option(const char* name, int* value) {
    if (!name || !value) return fail(-1, "null argument");
    for (const Opt& o : k_opts) if (!strcmp(name, o.name)) { *value = g_opt.*o.var; return 0; }
    return fail(-1, std::string("unknown option: ") + name);
}

int oct_set_option(const char* name, int value) {
    if (!name) return fail(-1, "null option name");
    if (!strcmp(name, "pair8_geometry") && value != 221 && value != 111) return fail(-1, "pair8_geometry must be 221 or 111");
    if (!strcmp(name, "bx_waves") && value != 4 && value != 8) return fail(-1, "bx_waves must be 4 or 8");
    for (const Opt& o : k_opts)
        if (!strcmp(name, o.name)) { g_opt.*o.var = value < o.lo ? o.lo : (value > o.hi ? o.hi : value); return 0; }
    return fail(-1, std::string("unknown option: ") + name);
}

int oct_unet_get_option(const oct_unet* h, const char* name, int* value) {
    if (!h || !name || !value) return fail(-1, "null argument");
    for (const Opt& o : k_opts) if (!strcmp(name, o.name)) { *value = h->opt.*o.var; return 0; }
    return fail(-1, std::string("unknown option: ") + name);
}

int oct_unet_set_option(oct_unet* h, const char* name, int value) {
    if (!h || !name) return fail(-1, "null argument");
    if (!strcmp(name, "bt_m2")) return fail(-1, "bt_m2 shapes the prepared weights: set the default before oct_unet_create");
    if (!strcmp(name, "pair8_geometry") && value != 221 && value != 111) return fail(-1, "pair8_geometry must be 221 or 111");
    if (!strcmp(name, "bx_waves") && value != 4 && value != 8) return fail(-1, "bx_waves must be 4 or 8");
    for (const Opt& o : k_opts)
        if (!strcmp(name, o.name)) { h->opt.*o.var = value < o.lo ? o.lo : (value > o.hi ? o.hi : value); return 0; }
    return fail(-1, std::string("unknown option: ") + name);
}

int oct_unet_debug_layer_fused(const oct_unet* h, int layer) {
    if (!h || layer < 0 || layer >= (int)h->plan.L.size()) return -1;
    return h->plan.L[layer].g_masked ? 1 : 0;
}

const void* oct_unet_debug_activation(oct_unet* h, int layer, int which) {
    if (!h || layer < 0 || layer >= (int)h->plan.L.size()) return nullptr;
    const Layer& l = h->plan.L[layer];
    return which == 0 ? (const void*)l.z : which == 1 ? (const void*)l.g : which == 2 ? (const void*)l.bn : nullptr;
}

}  // extern "C"
