// Forward kernels: conv (+bias, +BN-stat partials, input transform fused on load), BN finalize,
// max-pool (BN+ReLU fused on load), 1x1 head + softmax + Dice partial sums (+argmax).
//
// Fusion contract: a conv WRITES its raw output z (pre-BN); BatchNorm+ReLU (+dropout,
// +nearest-upsample, +concat) are applied by the CONSUMER when it loads z, using the per-channel
// affine (a, b) that bn_fwd_finalize derives from the conv's own per-block (sum, sum^2) partials.
#pragma once
#include "common.hpp"
#include "kernels_fin.hpp"

namespace oct {

struct ConvFwdArgs {
    const void* x0;    // source 0: u8 or f32 (B,Hs,Ws,C0)
    const float* ab0;  // BN record of source 0 (a at +0, b at +C0) when F_AFF
    int C0;
    const float* x1;   // source 1 (concat skip half) when F_TWO
    const float* ab1;
    int C1;
    const float* w;    // (KH,KW,Cin,Cout) HWIO
    const float* bias; // (Cout)
    void* z;           // (B,H,W,Cout), activation storage type
    float* part;       // [B*tiles][2*Cout] stat partials or nullptr
    int H, W, Cin, Cout, tiles_x, tiles;
    DropCfg drop;
    int act_bf16;      // activation storage type selector for the launcher
    FinDesc fin;       // conv_first_fwd_k: statistics finalized by the last block of the launch (kernels_fin.hpp)
};

// acc[j] += sum_c f(src[pix][c]) * w[c][j]   (w row stride = Cout; all weight addresses wave-uniform)
template <int CO_T, int FLAGS>
__device__ inline void accum_src(const void* __restrict__ src, const float* __restrict__ ab, int C, size_t pix,
                                 bool inb, const float* __restrict__ wk, int Cout, const DropCfg& drop,
                                 float (&acc)[CO_T]) {
    constexpr bool U8 = (FLAGS & F_U8) != 0, AFF = (FLAGS & F_AFF) != 0, DROP = (FLAGS & F_DROP) != 0;
    if constexpr (!U8) {
        if ((C & 3) == 0) {
            const float* p = reinterpret_cast<const float*>(src) + pix * C;
            for (int c = 0; c < C; c += 4) {
                float4 v = inb ? ld4(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (AFF) {
                    const float4 a = ld4(ab + c), b = ld4(ab + C + c);
                    v.x = inb ? fmaxf(fmaf(a.x, v.x, b.x), 0.f) : 0.f;
                    v.y = inb ? fmaxf(fmaf(a.y, v.y, b.y), 0.f) : 0.f;
                    v.z = inb ? fmaxf(fmaf(a.z, v.z, b.z), 0.f) : 0.f;
                    v.w = inb ? fmaxf(fmaf(a.w, v.w, b.w), 0.f) : 0.f;
                }
                if constexpr (DROP) {
                    const uint32_t e = (uint32_t)(pix * C + c);
                    v.x *= drop_mul(drop, e); v.y *= drop_mul(drop, e + 1);
                    v.z *= drop_mul(drop, e + 2); v.w *= drop_mul(drop, e + 3);
                }
                const float* wr = wk + (size_t)c * Cout;
#pragma unroll
                for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.x, wr[j], acc[j]);
#pragma unroll
                for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.y, wr[Cout + j], acc[j]);
#pragma unroll
                for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.z, wr[2 * Cout + j], acc[j]);
#pragma unroll
                for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v.w, wr[3 * Cout + j], acc[j]);
            }
            return;
        }
    }
    // scalar channel loop: u8 input or a channel count that is not a multiple of 4 (first layer)
    for (int c = 0; c < C; ++c) {
        float v = 0.f;
        if (inb) {
            if constexpr (U8) v = c_u8_lut[reinterpret_cast<const unsigned char*>(src)[pix * C + c]];
            else v = reinterpret_cast<const float*>(src)[pix * C + c];
            if constexpr (AFF) v = fmaxf(fmaf(ab[c], v, ab[C + c]), 0.f);
        }
        const float* wr = wk + (size_t)c * Cout;
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(v, wr[j], acc[j]);
    }
}

// One thread = one output pixel x CO_T output channels.  grid (tiles, Cout/CO_T, B), block 256.
template <int KH, int CO_T, int FLAGS, typename AT>
__global__ __launch_bounds__(kBlock) void conv_fwd_k(const ConvFwdArgs A) {
    constexpr int KW = KH, PT = (KH - 1) / 2;
    constexpr bool TWO = (FLAGS & F_TWO) != 0, UP = (FLAGS & F_UP) != 0;
    __shared__ float red[256];
    const int tx = threadIdx.x & (kTileX - 1), ty = threadIdx.x / kTileX;
    const int tile = blockIdx.x;
    const int x = (tile % A.tiles_x) * kTileX + tx, y = (tile / A.tiles_x) * kTileY + ty;
    const int b = blockIdx.z, co0 = blockIdx.y * CO_T;
    const bool valid = x < A.W && y < A.H;
    const int Hs = UP ? A.H >> 1 : A.H, Ws = UP ? A.W >> 1 : A.W;

    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = A.bias[co0 + j];

    for (int ky = 0; ky < KH; ++ky) {
        for (int kx = 0; kx < KW; ++kx) {
            const int iy = y + ky - PT, ix = x + kx - PT;
            const bool inb = valid && iy >= 0 && iy < A.H && ix >= 0 && ix < A.W;
            const int sy = UP ? iy >> 1 : iy, sx = UP ? ix >> 1 : ix;
            const size_t pix = inb ? ((size_t)b * Hs + sy) * Ws + sx : 0;
            const float* wk = A.w + (size_t)((ky * KW + kx) * A.Cin) * A.Cout + co0;
            accum_src<CO_T, FLAGS>(A.x0, A.ab0, A.C0, pix, inb, wk, A.Cout, A.drop, acc);
            if constexpr (TWO)
                accum_src<CO_T, FLAGS>(A.x1, A.ab1, A.C1, pix, inb, wk + (size_t)A.C0 * A.Cout, A.Cout, A.drop, acc);
        }
    }
    if (valid) {
        AT* zp = reinterpret_cast<AT*>(A.z) + (((size_t)b * A.H + y) * A.W + x) * A.Cout + co0;
#pragma unroll
        for (int j = 0; j < CO_T; j += 4) sta4<AT>(zp + j, make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]));
    }
    if (A.part) {  // per-block BatchNorm statistics of this conv's own output
        float s[CO_T], q[CO_T];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) { s[j] = valid ? acc[j] : 0.f; q[j] = s[j] * s[j]; }
        float* out = A.part + ((size_t)b * A.tiles + tile) * (2 * A.Cout);
        block_reduce_store<CO_T>(s, red, out + co0, CO_T);
        block_reduce_store<CO_T>(q, red, out + A.Cout + co0, CO_T);
    }
}

// ---- first layer, the real configuration (1 input channel -- uint8 through the /255 table, or f32 --, 8 output
// channels, 3x3).  Output-write bound (32 B/pixel): a PERSISTENT block walks 8 x 128 tiles, image tile in LDS, the 72
// weights wave-uniform, 4 pixels per thread with fully coalesced 32 B stores; BN statistics stay in registers across
// tiles and are reduced ONCE per block (the tile-per-block kernel pays two block reductions per 256 pixels and hands
// the finalize 16 384 partial rows at B = 32).  grid (nblk); statistics row per block.
template <typename AT>
__global__ __launch_bounds__(kBlock) void conv_first_fwd_k(const ConvFwdArgs A, int x_is_u8, int tiles_x, int tiles, int total_tiles,
                                                          const float* __restrict__ wgt, const float* __restrict__ bias) {
    constexpr int TH = 8, TW = 128, XH = TH + 2, XW = TW + 2;
    __shared__ float Xs[XH * XW];
    __shared__ float red[256];
    __shared__ float lut_s[256];          // the float32(i/255.0) table, copied once: lookups become LDS reads
    const int t = threadIdx.x, xl = t & 31, row = t >> 5;
    lut_s[t] = c_u8_lut[t];
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int tl = blockIdx.x; tl < total_tiles; tl += gridDim.x) {
        const int b = tl / tiles, tile = tl % tiles;
        const int x0 = (tile % tiles_x) * TW, y0 = (tile / tiles_x) * TH;
        __syncthreads();
        {   // all image loads of the tile are issued before any is used (the /255 table lookup is a dependent access)
            constexpr int NS = (XH * XW + kBlock - 1) / kBlock;
            unsigned int rb[NS]; float rf[NS]; bool in[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = t + k * kBlock, cy = i / XW, cx = i % XW, iy = y0 + cy - 1, ix = x0 + cx - 1;
                in[k] = i < XH * XW && iy >= 0 && iy < A.H && ix >= 0 && ix < A.W;
                const size_t pix = in[k] ? ((size_t)b * A.H + iy) * A.W + ix : 0;
                rb[k] = 0; rf[k] = 0.f;
                if (in[k]) { if (x_is_u8) rb[k] = reinterpret_cast<const unsigned char*>(A.x0)[pix]; else rf[k] = reinterpret_cast<const float*>(A.x0)[pix]; }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = t + k * kBlock;
                if (i < XH * XW) Xs[i] = in[k] ? (x_is_u8 ? lut_s[rb[k]] : rf[k]) : 0.f;
            }
        }
        __syncthreads();
        const int y = y0 + row;
#pragma unroll
        for (int k = 0; k < TW / 32; ++k) {
            const int xx = xl + 32 * k, x = x0 + xx;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = bias[j];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float xv = Xs[(row + tap / 3) * XW + xx + tap % 3];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wgt[tap * 8 + j], acc[j]);
            }
            if (y < A.H && x < A.W) {
                AT* zp = reinterpret_cast<AT*>(A.z) + (((size_t)b * A.H + y) * A.W + x) * 8;
                sta4<AT>(zp, make_float4(acc[0], acc[1], acc[2], acc[3]));
                sta4<AT>(zp + 4, make_float4(acc[4], acc[5], acc[6], acc[7]));
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], acc[j], s2[j]); }
            }
        }
    }
    if (A.part) {
        float* out = A.part + (size_t)blockIdx.x * 16;
        block_reduce_store<8, true>(s1, red, out, 8);
        block_reduce_store<8, true>(s2, red, out + 8, 8);
        if (A.fin.counter) finalize_in_launch(A.fin, A.part, gridDim.x, 8, gridDim.x, reinterpret_cast<char*>(Xs));      // (Xs: 5.2 KB, free now)
    }
}

// ---- BN finalize (training): partials -> mean / biased var -> (a, b); moving-stat update -----------------
struct BnFinArgs {
    const float* part;  // [nblk][2*C]
    int nblk, C;
    double count;       // B*H*W
    const float* gamma; const float* beta;
    float* bn;          // record (BN_ARRAYS * C)
    float* mm; float* mv;
    float eps, momentum; int unbiased;
};

static __global__ __launch_bounds__(kBlock) void bn_fwd_finalize_k(const BnFinArgs A) {
    __shared__ double sh[8];
    const int c = blockIdx.x;
    double s, q;
    column_sums_f64(A.part, A.nblk, A.C, c, sh, s, q);
    if (threadIdx.x == 0) bn_fwd_finalize_write(s, q, A.count, A.C, c, A.bn, A.gamma, A.beta, A.mm, A.mv, A.eps, A.momentum, A.unbiased);
}

// inference: (a, b) from the moving statistics, one thread per channel of one layer
static __global__ void bn_infer_coeffs_k(const float* gamma, const float* beta, const float* mm, const float* mv,
                                  float* bn, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float a = gamma[c] / sqrtf(mv[c] + eps);
    bn[BN_A * C + c] = a;
    bn[BN_B * C + c] = beta[c] - mm[c] * a;
    bn[BN_MEAN * C + c] = mm[c];
    bn[BN_RSTD * C + c] = 1.f / sqrtf(mv[c] + eps);
}

// all blocks at once (descriptor table by value): grid = number of BN blocks
struct BnInferAll {
    static constexpr int MAXL = 40;
    int n;
    struct Entry { const float* gamma; const float* beta; const float* mm; const float* mv; float* bn; int C; } L[MAXL];
};
static __global__ void bn_infer_all_k(const BnInferAll A, float eps) {
    const BnInferAll::Entry E = A.L[blockIdx.x];
    for (int c = threadIdx.x; c < E.C; c += blockDim.x) {
        const float rstd = 1.f / sqrtf(E.mv[c] + eps), a = E.gamma[c] * rstd;
        E.bn[BN_A * E.C + c] = a;
        E.bn[BN_B * E.C + c] = E.beta[c] - E.mm[c] * a;
        E.bn[BN_MEAN * E.C + c] = E.mm[c];
        E.bn[BN_RSTD * E.C + c] = rstd;
    }
}

// ---- max-pool 2x2 with BN+ReLU fused on load: p = max over window of relu(a*z+b) -------------------------
template <typename AT>
__global__ __launch_bounds__(kBlock) void pool_fwd_k(const AT* __restrict__ z, const float* __restrict__ ab,
                                                    AT* __restrict__ p, int B, int H, int W, int C) {
    // H, W: INPUT dims.  one thread = one pooled pixel x 4 channels
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const size_t n = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const int c = (int)(i % C4) * 4;
        size_t r = i / C4;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho); const int b = (int)(r / Ho);
        const float4 a = ld4(ab + c), bb = ld4(ab + C + c);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // relu output >= 0
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float4 v = lda4<AT>(z + (((size_t)b * H + 2 * yo + dy) * W + 2 * xo + dx) * C + c);
                m.x = fmaxf(m.x, fmaf(a.x, v.x, bb.x)); m.y = fmaxf(m.y, fmaf(a.y, v.y, bb.y));
                m.z = fmaxf(m.z, fmaf(a.z, v.z, bb.z)); m.w = fmaxf(m.w, fmaf(a.w, v.w, bb.w));
            }
        sta4<AT>(p + (((size_t)b * Ho + yo) * Wo + xo) * C + c, m);
    }
}

// ---- head: BN+ReLU on load, 1x1 conv, softmax, optional probs / argmax / Dice partial sums ---------------
constexpr int kDiceVals = 5;  // per class: I = sum y*p, T = sum y, P = sum p, Ih = sum y*[p>.5], Ph = sum [p>.5]
template <int C> struct DiceN { static constexpr int value = (5 * C <= 16) ? 16 : (5 * C <= 32 ? 32 : 64); };

struct HeadFwdArgs {
    const void* z; const float* ab;    // last conv's raw output (activation storage type) + BN record
    const float* w; const float* bias; // (CIN, C), (C)
    float* probs; unsigned char* argmax; const unsigned char* labels;
    float* dice_part;                  // [B][nblk][DiceN]; slot 5*C (always spare: 5*C < DiceN) = focal-loss sum
    int HW, nblk, act_bf16;
    // focal half of focal_dice_loss (custom_losses.py:98-178): cw[y] * (1 - p_y)^gamma * (-log p_y), p clipped to [1e-7, 1-1e-7]
    int focal_on; float focal_gamma; const float* focal_cw;   // class weights (C) or nullptr
    int focal_clip_mod;                // 1: the (1 - p)^gamma modulation also sees the clipped p; 0: only the logarithm does
};
constexpr float kFocalEps = 1e-7f;

// raw z of one pixel (CIN values) -> registers; split from the arithmetic so that the head kernels can request the next
// chunk's pixel before they work on the current one (one chunk of software prefetch)
template <int CIN, typename AT>
__device__ inline void head_load(const AT* __restrict__ zp, float (&zr)[CIN]) {
#pragma unroll
    for (int i = 0; i < CIN; i += 4) {
        const float4 v = lda4<AT>(zp + i);
        zr[i] = v.x; zr[i + 1] = v.y; zr[i + 2] = v.z; zr[i + 3] = v.w;
    }
}

template <int C, int CIN>
__device__ inline void head_logits(const float* __restrict__ ab, const float* __restrict__ w, const float* __restrict__ bias,
                                   float (&y)[CIN], const float (&zr)[CIN], float (&p)[C]) {
#pragma unroll
    for (int i = 0; i < CIN; ++i) y[i] = fmaxf(fmaf(ab[i], zr[i], ab[CIN + i]), 0.f);
    float mx = -3.4e38f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float l = bias[c];
#pragma unroll
        for (int i = 0; i < CIN; ++i) l = fmaf(y[i], w[i * C + c], l);
        p[c] = l; mx = fmaxf(mx, l);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - mx); sum += p[c]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
}

// grid (nblk, B): each block walks the 256-pixel chunks  blockIdx.x, blockIdx.x + nblk, ...  of ONE image and emits
// one row of Dice partial sums
template <int C, int CIN, typename AT>
__global__ __launch_bounds__(kBlock) void head_fwd_k(const HeadFwdArgs A) {
    constexpr int N = DiceN<C>::value;
    __shared__ float red[256];
    const int b = blockIdx.y;
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0.f;
    for (int chunk = blockIdx.x; chunk * kBlock < A.HW; chunk += gridDim.x) {
        const int px = chunk * kBlock + threadIdx.x;
        const bool valid = px < A.HW;
        const size_t pix = (size_t)b * A.HW + (valid ? px : 0);
        float y[CIN], zr[CIN], p[C];
        head_load<CIN, AT>(reinterpret_cast<const AT*>(A.z) + pix * CIN, zr);   // (a chunk of prefetch was measured here: slower)
        head_logits<C, CIN>(A.ab, A.w, A.bias, y, zr, p);
        if (valid) {
            if (A.probs) {
#pragma unroll
                for (int c = 0; c < C; ++c) A.probs[pix * C + c] = p[c];
            }
            if (A.argmax) {
                int am = 0; float best = p[0];
#pragma unroll
                for (int c = 1; c < C; ++c) if (p[c] > best) { best = p[c]; am = c; }  // first maximum, as np.argmax
                A.argmax[pix] = (unsigned char)am;
            }
            if (A.labels) {
                const int lab = A.labels[pix];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float yv = lab == c ? 1.f : 0.f, ph = p[c] > 0.5f ? 1.f : 0.f;
                    v[c * kDiceVals + 0] += yv * p[c]; v[c * kDiceVals + 1] += yv; v[c * kDiceVals + 2] += p[c];
                    v[c * kDiceVals + 3] += yv * ph;   v[c * kDiceVals + 4] += ph;
                }
                if (A.focal_on) {
                    float py = p[0];
#pragma unroll
                    for (int c = 1; c < C; ++c) py = lab == c ? p[c] : py;
                    const float pc = fminf(fmaxf(py, kFocalEps), 1.f - kFocalEps);
                    const float cw = A.focal_cw ? A.focal_cw[lab < C ? lab : 0] : 1.f;
                    v[C * kDiceVals] += cw * powf(1.f - (A.focal_clip_mod ? pc : py), A.focal_gamma) * -logf(pc);
                }
            }
        }
    }
    if (A.labels) block_reduce_store<N>(v, red, A.dice_part + ((size_t)b * gridDim.x + blockIdx.x) * N, N);
}

// Dice finalize: per-(b,c) sums in fp64 -> losses, metrics, and the per-(b,c) constants backward needs.
//   out4 = {dice_loss_macro, dice_loss_micro, dice_coef_macro, dice_coef_micro}
//   bc   = per (b,c): {Num = 2I+s, Den = T+P+s}; then the micro pair at [2*B*C], [2*B*C+1]
//   out8 = out4 + {focal mean, w*focal + (1-w)*dice_macro, w*focal + (1-w)*dice_micro, 0}   (focal_dice_loss)
struct DiceFinArgs {
    const float* part; int B, C, nblk, N;
    float smooth; float* out4; float* out4_user; double* bc;
    int n_user;            // floats copied to out4_user: 4 (loss_dice) or 8 (loss_focal_dice)
    double inv_count;      // 1 / (B*H*W)
    float focal_w;
};

static __global__ __launch_bounds__(kBlock) void dice_finalize_k(const DiceFinArgs A) {
    __shared__ double sh[8][kBlock];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // score_macro, coef_macro, I, T, P, Ih, Ph (micro sums), focal sum
    const int n = A.B * A.C;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int b = i / A.C, c = i % A.C;
        double v[kDiceVals] = {0, 0, 0, 0, 0};
        const float* p0 = A.part + (size_t)b * A.nblk * A.N + c * kDiceVals;
        int k = 0;
        for (; k + 8 <= A.nblk; k += 8) {          // 8 rows' loads issued together (one memory round trip), fixed order
            float t[8][kDiceVals];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < kDiceVals; ++j) t[r][j] = p0[(size_t)(k + r) * A.N + j];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < kDiceVals; ++j) v[j] += t[r][j];
        }
        for (; k < A.nblk; ++k)
            for (int j = 0; j < kDiceVals; ++j) v[j] += p0[(size_t)k * A.N + j];
        const double s = A.smooth;
        const double num = 2.0 * v[0] + s, den = v[1] + v[2] + s;
        A.bc[2 * i] = num; A.bc[2 * i + 1] = den;
        acc[0] += num / den;
        acc[1] += (2.0 * v[3] + 1e-5) / (v[1] + v[4] + 1e-5);   // dice_coef_macro eps (custom_metrics.py:50)
        acc[2] += v[0]; acc[3] += v[1]; acc[4] += v[2]; acc[5] += v[3]; acc[6] += v[4];
        if (c == 0)
            for (int k = 0; k < A.nblk; ++k) acc[7] += A.part[((size_t)b * A.nblk + k) * A.N + A.C * kDiceVals];
    }
    for (int j = 0; j < 8; ++j) sh[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) for (int j = 0; j < 8; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double s = A.smooth;
        const double num = 2.0 * sh[2][0] + s, den = sh[3][0] + sh[4][0] + s;
        A.bc[2 * n] = num; A.bc[2 * n + 1] = den;
        float o4[8];
        o4[0] = (float)(1.0 - sh[0][0] / n);
        o4[1] = (float)(1.0 - num / den);
        o4[2] = (float)(sh[1][0] / n);
        o4[3] = (float)(2.0 * sh[5][0] / (sh[3][0] + sh[6][0]));  // no epsilon: 0/0 -> nan as the reference
        const double focal = sh[7][0] * A.inv_count, w = A.focal_w;
        o4[4] = (float)focal;
        o4[5] = (float)(w * focal + (1.0 - w) * (1.0 - sh[0][0] / n));
        o4[6] = (float)(w * focal + (1.0 - w) * (1.0 - num / den));
        o4[7] = 0.f;
        for (int j = 0; j < 8; ++j) { A.out4[j] = o4[j]; if (A.out4_user && j < A.n_user) A.out4_user[j] = o4[j]; }
    }
}

// ---- post-step on device: class map -> boundary maps, the reference's
// convert_predictions_to_maps_semantic (common/utils.py:115-168) on the one-hot of the arg-max:
//   cur = [label == k],  g = 2*max(+-gradient_rows(cur), 0)  (np.gradient: central differences, one-sided at the
//   edges),  out = uint8(255 * max(g[r] - g[(r+1) mod H], 0))   -- exact in small integers.
// labels (B,H,W) u8 -> maps (B, C-1, H, W) u8.  One thread per (b, r, c); adjacent threads walk adjacent columns.
static __global__ __launch_bounds__(kBlock) void boundary_maps_k(const unsigned char* __restrict__ lab, unsigned char* __restrict__ out,
                                                         int B, int H, int W, int C, int bg_ilm, int bg_csi) {
    const size_t n = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const int c = (int)(i % W); size_t t = i / W;
        const int r = (int)(t % H), b = (int)(t / H);
        const unsigned char* col = lab + (size_t)b * H * W + c;
        auto L = [&](int rr) { return (int)col[(size_t)rr * W]; };
        // rows needed for g[r] and g[r1]: r-1, r, r+1 and r1-1, r1, r1+1 with r1 = (r+1) mod H
        const int r1 = r + 1 == H ? 0 : r + 1;
        int lv[6];
        const int rows[6] = {r > 0 ? r - 1 : 0, r, r + 1 < H ? r + 1 : H - 1, r1 > 0 ? r1 - 1 : 0, r1, r1 + 1 < H ? r1 + 1 : H - 1};
#pragma unroll
        for (int k = 0; k < 6; ++k) lv[k] = L(rows[k]);
        for (int m = 1; m < C; ++m) {
            const bool flip = (m == 1 && bg_ilm) || (m == C - 1 && bg_csi);
            const int k = flip ? m - 1 : m;
            auto grad2 = [&](int rr, int lo, int mid, int hi) {   // 2 * np.gradient at row rr, times 2 again = 4*grad
                // interior: (hi - lo)/2 ; first row: (hi - mid) ; last row: (mid - lo)   -> in units of 1/2
                int d2;
                if (H == 1) d2 = 0;
                else if (rr == 0) d2 = 2 * ((hi == k) - (mid == k));
                else if (rr == H - 1) d2 = 2 * ((mid == k) - (lo == k));
                else d2 = (hi == k) - (lo == k);
                if (flip) d2 = -d2;
                return d2 > 0 ? d2 : 0;                           // = max(grad,0)*2 expressed in units of 1/2... see below
            };
            // g = 2*max(grad,0) with grad in {0, 1/2, 1}: g in {0, 1, 2}; grad2() returns 2*max(grad,0) = g exactly
            const int g0 = grad2(r, lv[0], lv[1], lv[2]);
            const int g1 = grad2(r1, lv[3], lv[4], lv[5]);
            int v = g0 - g1; v = v > 0 ? v : 0;
            out[(((size_t)b * (C - 1) + (m - 1)) * H + r) * W + c] = (unsigned char)((v * 255) & 255);   // numpy's float->uint8 cast of 510 wraps to 254
        }
    }
}

// dropout keep-mask dump for parity tests
static __global__ void dropout_mask_k(unsigned char* out, size_t n, DropCfg d) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = drop_hash(d.seed, d.step, (uint32_t)i) >= d.thresh ? 1 : 0;
}

}  // namespace oct
