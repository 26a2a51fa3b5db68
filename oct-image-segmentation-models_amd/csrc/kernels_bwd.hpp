// Backward kernels.
//
// Gradient flow contract (DESIGN.md sections 5, 10): for a conv block  z = conv(x)+bias, y = relu(BN(z)):
//   * whoever produces dL/dy writes g' = dL/dy * [y>0] (* dropout) into the block's g buffer and emits
//     per-block partials of  sum(g')  and  sum(g' * xhat)  (the two BN-backward reductions = dbeta, dgamma);
//   * bn_bwd_finalize turns the partials into c1 = sum(g')/N, c2 = sum(g' xhat)/N, dgamma/dbeta and the two-fma form
//     dz = ga g' + (gb z + gd) of the BN-backward transform (record rows ga, gb, gd; common.hpp);
//   * the consumers of dz -- the backward-data launches (kernels_bx.hpp) and the backward-weights kernels (kernels_dw.hpp,
//     kernels_bx.hpp) -- apply that transform themselves while they stage g' and z (dz is never stored); bn_bwd_apply, which
//     rewrites g' in place into dz, runs only for the blocks whose consumers cannot (fp32-pipe conv kernels, two thin-kernel
//     instantiations): same two fmas, bit-identical results;
//   * the dW kernels reduce dW = sum x (x) dz into one partial slab per block; reduce_all_k sums the slabs.
#pragma once
#include "common.hpp"
#include "kernels_fwd.hpp"

namespace oct {

// ---- max-pool backward + skip-gradient merge + ReLU mask + BN-backward statistics ---------------------------
// The encoder block's output y feeds (i) the pool and (ii) the decoder concat.  gskip already holds the
// decoder's raw contribution; this kernel adds the pool's routed gradient, masks and emits the statistics.
struct PoolBwdArgs {
    const void* gp;    // (B,H/2,W/2,C) gradient wrt pooled tensor          (activation storage type)
    const void* z;     // (B,H,W,C) block's raw output
    const float* bn;
    void* g;           // (B,H,W,C): in = decoder contribution (raw), out = masked total
    float* part;       // [B*tiles][2*C]   tiles over the POOLED grid
    int H, W, C, tiles_x, tiles, act_bf16;
    FinDesc fin;       // pool_bwd_flat_k: BN-backward sums finalized by the last block (kernels_fin.hpp)
};

template <int C_T, typename AT>
__global__ __launch_bounds__(kBlock) void pool_bwd_k(const PoolBwdArgs A) {
    __shared__ float red[256];
    const int Ho = A.H >> 1, Wo = A.W >> 1;
    const int tx = threadIdx.x & (kTileX - 1), ty = threadIdx.x / kTileX;
    const int tile = blockIdx.x;
    const int xo = (tile % A.tiles_x) * kTileX + tx, yo = (tile / A.tiles_x) * kTileY + ty;
    const int b = blockIdx.z, c0 = blockIdx.y * C_T;
    const bool valid = xo < Wo && yo < Ho;
    float s1[C_T], s2[C_T];
#pragma unroll
    for (int i = 0; i < C_T; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    if (valid) {
        // every tensor element is touched exactly once: z (4 positions), gp, g in, g out
        float gpv[C_T], zv[4][C_T], gd[4][C_T];
        const AT* Az = reinterpret_cast<const AT*>(A.z); AT* Ag = reinterpret_cast<AT*>(A.g);
        const AT* gpp = reinterpret_cast<const AT*>(A.gp) + (((size_t)b * Ho + yo) * Wo + xo) * A.C + c0;
#pragma unroll
        for (int i = 0; i < C_T; i += 4) { const float4 t = lda4<AT>(gpp + i); gpv[i] = t.x; gpv[i + 1] = t.y; gpv[i + 2] = t.z; gpv[i + 3] = t.w; }
        size_t off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            off[q] = (((size_t)b * A.H + 2 * yo + (q >> 1)) * A.W + 2 * xo + (q & 1)) * A.C + c0;
#pragma unroll
            for (int i = 0; i < C_T; i += 4) {
                const float4 t = lda4<AT>(Az + off[q] + i); zv[q][i] = t.x; zv[q][i + 1] = t.y; zv[q][i + 2] = t.z; zv[q][i + 3] = t.w;
                const float4 u = lda4<AT>(Ag + off[q] + i); gd[q][i] = u.x; gd[q][i + 1] = u.y; gd[q][i + 2] = u.z; gd[q][i + 3] = u.w;
            }
        }
#pragma unroll
        for (int i = 0; i < C_T; ++i) {
            const int c = c0 + i;
            const float a = A.bn[BN_A * A.C + c], bb = A.bn[BN_B * A.C + c];
            const float mean = A.bn[BN_MEAN * A.C + c], rstd = A.bn[BN_RSTD * A.C + c];
            float yv[4]; int am = 0; float best = -1.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                yv[q] = fmaxf(fmaf(a, zv[q][i], bb), 0.f);
                if (yv[q] > best) { best = yv[q]; am = q; }   // first maximum in row-major window order
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gt = yv[q] > 0.f ? gd[q][i] + (q == am ? gpv[i] : 0.f) : 0.f;
                s1[i] += gt; s2[i] += gt * ((zv[q][i] - mean) * rstd);
                gd[q][i] = gt;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < C_T; i += 4) sta4<AT>(Ag + off[q] + i, make_float4(gd[q][i], gd[q][i + 1], gd[q][i + 2], gd[q][i + 3]));
    }
    float* out = A.part + ((size_t)b * A.tiles + tile) * (2 * A.C);
    block_reduce_store<C_T>(s1, red, out + c0, C_T);
    block_reduce_store<C_T>(s2, red, out + A.C + c0, C_T);
}

// Same operation with a FLAT thread mapping (pooled pixel x channel quad, quad fastest): a wave touches contiguous
// bytes whatever the channel count (the tiled kernel above gives a block one 4-channel slice, i.e. 16 useful bytes per
// 4*C-byte pixel -- 1.25 TB/s at C = 64).  Needs C/4 a power of two <= 64: a thread's quad is then the same in every
// grid-stride iteration, its statistics stay in registers, and lanes with equal quad are summed by xor-shuffles.
// grid (nblk); one statistics row per block.
// V = channels per thread (4, or 8 for bf16 storage: 16-byte accesses there too -- with 4 a lane moves 8 bytes per access and
// the kernel reaches 2.5 TB/s on the 537 MB tensors of configs[2] where the fp32 form reaches 4.4).
template <typename AT, int V = 4>
__global__ __launch_bounds__(kBlock) void pool_bwd_flat_k(const PoolBwdArgs A, int B) {
    static_assert(V == 4 || V == 8, "channels per thread");
    __shared__ float sh[4 * 64 * 8];                       // [wave][channel group][2 V]
    const int CV = A.C / V, Ho = A.H >> 1, Wo = A.W >> 1;
    const size_t n = (size_t)B * Ho * Wo * CV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = V * (tid % CV);
    float a[V], bb[V], mean[V], rstd[V], s1[V], s2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        a[k] = A.bn[BN_A * A.C + c + k]; bb[k] = A.bn[BN_B * A.C + c + k];
        mean[k] = A.bn[BN_MEAN * A.C + c + k]; rstd[k] = A.bn[BN_RSTD * A.C + c + k];
        s1[k] = 0.f; s2[k] = 0.f;
    }
    const AT* Az = reinterpret_cast<const AT*>(A.z); AT* Ag = reinterpret_cast<AT*>(A.g);
    const AT* Agp = reinterpret_cast<const AT*>(A.gp);
    for (size_t i = (size_t)blockIdx.x * kBlock + tid; i < n; i += (size_t)gridDim.x * kBlock) {
        size_t r = i / CV;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho); const int b = (int)(r / Ho);
        float gpv[V];
        size_t off[4]; float zv[4][V], gd[4][V];
        const size_t goff = (((size_t)b * Ho + yo) * Wo + xo) * A.C + c;
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const float4 gq = lda4<AT>(Agp + goff + j);
            gpv[j] = gq.x; gpv[j + 1] = gq.y; gpv[j + 2] = gq.z; gpv[j + 3] = gq.w;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            off[w] = (((size_t)b * A.H + 2 * yo + (w >> 1)) * A.W + 2 * xo + (w & 1)) * A.C + c;
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const float4 t = lda4<AT>(Az + off[w] + j), u = lda4<AT>(Ag + off[w] + j);
                zv[w][j] = t.x; zv[w][j + 1] = t.y; zv[w][j + 2] = t.z; zv[w][j + 3] = t.w;
                gd[w][j] = u.x; gd[w][j + 1] = u.y; gd[w][j + 2] = u.z; gd[w][j + 3] = u.w;
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float yv[4]; int am = 0; float best = -1.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                yv[w] = fmaxf(fmaf(a[k], zv[w][k], bb[k]), 0.f);
                if (yv[w] > best) { best = yv[w]; am = w; }     // first maximum in row-major window order
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float gt = yv[w] > 0.f ? gd[w][k] + (w == am ? gpv[k] : 0.f) : 0.f;
                s1[k] += gt; s2[k] += gt * ((zv[w][k] - mean[k]) * rstd[k]);
                gd[w][k] = gt;
            }
        }
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int j = 0; j < V; j += 4) sta4<AT>(Ag + off[w] + j, make_float4(gd[w][j], gd[w][j + 1], gd[w][j + 2], gd[w][j + 3]));
    }
    for (int o = CV; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < V; ++k) { s1[k] += __shfl_xor(s1[k], o, 64); s2[k] += __shfl_xor(s2[k], o, 64); }
    if (lane < CV) {
#pragma unroll
        for (int k = 0; k < V; ++k) { sh[(wave * CV + lane) * 2 * V + k] = s1[k]; sh[(wave * CV + lane) * 2 * V + V + k] = s2[k]; }
    }
    __syncthreads();
    if (tid < A.C) {
        const int qq = tid / V, k = tid % V;
        float* out = A.part + (size_t)blockIdx.x * (2 * A.C);
        auto at = [&](int w, int j) { return sh[(w * CV + qq) * 2 * V + j]; };
        part_store(out + tid, (at(0, k) + at(1, k)) + (at(2, k) + at(3, k)));           // fixed order
        part_store(out + A.C + tid, (at(0, V + k) + at(1, V + k)) + (at(2, V + k) + at(3, V + k)));
    }
    if (A.fin.counter) finalize_in_launch(A.fin, A.part, gridDim.x, A.C, gridDim.x, reinterpret_cast<char*>(sh));
}

// ---- BN backward finalize + apply -------------------------------------------------------------------------------
struct BnBwdFinArgs {
    const float* part; int nblk, C; double count;
    float* bn; const float* gamma; float* dgamma; float* dbeta;
};

static __global__ __launch_bounds__(kBlock) void bn_bwd_finalize_k(const BnBwdFinArgs A) {
    __shared__ double sh[8];
    const int c = blockIdx.x;
    double s, q;
    column_sums_f64(A.part, A.nblk, A.C, c, sh, s, q);
    if (threadIdx.x == 0) bn_bwd_finalize_write(s, q, A.count, A.C, c, A.bn, A.gamma, A.dgamma, A.dbeta);
}

// the BN-backward transform of one element: dz = ga g' + (gb z + gd) (record rows BN_GA / BN_GB / BN_GD), written as its
// two fmas so that the stand-alone pass and every stager that applies it on load produce the same bits
__device__ __forceinline__ float bn_bwd_apply1(float ga, float gb, float gd, float g, float z) {
    return fmaf(ga, g, fmaf(gb, z, gd));
}
// ... and rounded the way the stand-alone pass would have stored it (bf16 storage), for consumers that never store it
template <typename AT> __device__ __forceinline__ float dz_as_stored(float v) {
    if constexpr (sizeof(AT) == 2) return bf2f(f2bf(v)); else return v;
}

// dz = ga g' + gb z + gd, in place over g'
template <typename AT>
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_k(AT* __restrict__ g, const AT* __restrict__ z,
                                                        const float* __restrict__ bn, size_t n4, int C) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        const int c = (int)((i * 4) % C);
        const float4 gv = lda4<AT>(g + i * 4), zv = lda4<AT>(z + i * 4);
        const float4 ga = ld4(bn + BN_GA * C + c), gb = ld4(bn + BN_GB * C + c), gd = ld4(bn + BN_GD * C + c);
        float4 o;
        o.x = bn_bwd_apply1(ga.x, gb.x, gd.x, gv.x, zv.x);
        o.y = bn_bwd_apply1(ga.y, gb.y, gd.y, gv.y, zv.y);
        o.z = bn_bwd_apply1(ga.z, gb.z, gd.z, gv.z, zv.z);
        o.w = bn_bwd_apply1(ga.w, gb.w, gd.w, gv.w, zv.w);
        sta4<AT>(g + i * 4, o);
    }
}

// bf16 storage, channel count a multiple of 8: 8 elements per thread so that every access is 16 bytes per lane (the
// 4-element form moves 8 bytes per lane and reaches 4.5 TB/s instead of 6).  Same arithmetic, same operation order.
static __global__ __launch_bounds__(kBlock) void bn_bwd_apply8_bf16_k(bf16_t* __restrict__ g, const bf16_t* __restrict__ z,
                                                                     const float* __restrict__ bn, size_t n8, int C) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (size_t)gridDim.x * kBlock) {
        const int c = (int)((i * 8) % C);
        const uint4 gr = *reinterpret_cast<const uint4*>(g + i * 8), zr = *reinterpret_cast<const uint4*>(z + i * 8);
        const float4 gv[2] = {widen4(make_uint2(gr.x, gr.y)), widen4(make_uint2(gr.z, gr.w))};
        const float4 zv[2] = {widen4(make_uint2(zr.x, zr.y)), widen4(make_uint2(zr.z, zr.w))};
        bf16_t* out = g + i * 8;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int cc = c + 4 * hlf;
            const float4 ga = ld4(bn + BN_GA * C + cc), gb = ld4(bn + BN_GB * C + cc), gd = ld4(bn + BN_GD * C + cc);
            float4 o;
            o.x = bn_bwd_apply1(ga.x, gb.x, gd.x, gv[hlf].x, zv[hlf].x);
            o.y = bn_bwd_apply1(ga.y, gb.y, gd.y, gv[hlf].y, zv[hlf].y);
            o.z = bn_bwd_apply1(ga.z, gb.z, gd.z, gv[hlf].z, zv[hlf].z);
            o.w = bn_bwd_apply1(ga.w, gb.w, gd.w, gv[hlf].w, zv[hlf].w);
            sta4<bf16_t>(out + 4 * hlf, o);       // the two 8-byte halves of one 16-byte line: merged by the compiler / L2
        }
    }
}

// ---- head backward: Dice gradient -> softmax Jacobian -> 1x1 conv backward (data AND weights) -> mask + stats ------
// Everything the head needs is in registers here (y, p, dlogits), so its dW/db are accumulated in the same pass:
// no dlogits tensor, no second read of z.  grid (nblk, B); a block walks chunks of one image and emits one row of
// BN-backward statistics and one row of head-weight partials.
struct HeadBwdArgs {
    const void* z; const float* bn;      // last conv block (z: activation storage type)
    const float* w; const float* bias;   // head (CIN,C),(C)
    const unsigned char* labels;
    const double* bc;                    // Dice constants from dice_finalize_k
    void* g;                             // (B,H,W,CIN) masked gradient of the last conv block (activation storage type)
    float* part;                         // [B*nblk][2*CIN]       BN-backward statistics
    float* wpart;                        // [B*nblk][CIN*C + C]   head kernel / bias gradient partials
    int HW, nblk, B, macro; float loss_scale; int act_bf16;
    // focal_dice_loss: L = w * focal + (1 - w) * dice  (focal_w = 0: plain Dice)
    float focal_w, focal_gamma; const float* focal_cw; float inv_count;   // inv_count = 1 / (B*H*W)
    int focal_clip_mod;                // see HeadFwdArgs
    FinDesc fin;                       // BN-backward sums of the last conv block finalized by the last block (kernels_fin.hpp)
};

template <int C, int CIN, typename AT>
__global__ __launch_bounds__(kBlock) void head_bwd_k(const HeadBwdArgs A) {
    constexpr int NV = CIN * C + C, NG = (NV + 31) / 32;
    constexpr int CP = CIN <= 4 ? 4 : (CIN <= 8 ? 8 : (CIN <= 16 ? 16 : 32));     // the block reduction takes powers of two
    __shared__ float red[256];
    const int b = blockIdx.y;
    float s1[CP], s2[CP], wv[NG * 32];
#pragma unroll
    for (int i = 0; i < CP; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < NG * 32; ++i) wv[i] = 0.f;
    float num[C], den[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double* k = A.macro ? A.bc + 2 * (b * C + c) : A.bc + 2 * A.B * C;
        num[c] = (float)k[0]; den[c] = (float)k[1];
    }
    const float scale = (1.f - A.focal_w) * (A.macro ? A.loss_scale / (float)(A.B * C) : A.loss_scale);
    const float fscale = A.focal_w * A.loss_scale * A.inv_count;

    const AT* const zbase = reinterpret_cast<const AT*>(A.z) + (size_t)b * A.HW * CIN;
    auto pix_of = [&](int chunk) { const int px = chunk * kBlock + threadIdx.x; return px < A.HW ? px : 0; };
    float zn[CIN];
    int labn = 0;
    if ((int)blockIdx.x * kBlock < A.HW) {
        const int q = pix_of(blockIdx.x);
        head_load<CIN, AT>(zbase + (size_t)q * CIN, zn);
        labn = A.labels[(size_t)b * A.HW + q];
    }
    for (int chunk = blockIdx.x; chunk * kBlock < A.HW; chunk += gridDim.x) {
        const int px = chunk * kBlock + threadIdx.x;
        const bool valid = px < A.HW;
        const size_t pix = (size_t)b * A.HW + (valid ? px : 0);
        float y[CIN], zr[CIN], p[C];
#pragma unroll
        for (int i = 0; i < CIN; ++i) zr[i] = zn[i];
        const int lab = labn;
        if ((chunk + (int)gridDim.x) * kBlock < A.HW) {          // the next chunk's pixel is requested before this one is used
            const int q = pix_of(chunk + gridDim.x);
            head_load<CIN, AT>(zbase + (size_t)q * CIN, zn);
            labn = A.labels[(size_t)b * A.HW + q];
        }
        head_logits<C, CIN>(A.bn, A.w, A.bias, y, zr, p);
        float dp[C], dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float yv = lab == c ? 1.f : 0.f;
            dp[c] = -scale * (2.f * yv * den[c] - num[c]) / (den[c] * den[c]);
            if (fscale != 0.f && lab == c) {
                // d/dp [ cw (1-p)^g (-log pc) ],  pc = clip(p, eps, 1-eps):  cw ( g (1-p)^(g-1) log pc - (1-p)^g / pc * [p == pc] );
                // with the modulation clipped too (focal_clip_mod) both terms vanish where the clip is active
                const bool inr = p[c] >= kFocalEps && p[c] <= 1.f - kFocalEps;
                if (inr || !A.focal_clip_mod) {
                    const float pc = fminf(fmaxf(p[c], kFocalEps), 1.f - kFocalEps);
                    const float q = 1.f - p[c], cw = A.focal_cw ? A.focal_cw[c] : 1.f, qg1 = powf(q, A.focal_gamma - 1.f);
                    dp[c] += fscale * cw * (A.focal_gamma * qg1 * logf(pc) - (inr ? qg1 * q / pc : 0.f));
                }
            }
            dot = fmaf(p[c], dp[c], dot);
        }
        float dl[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { dl[c] = valid ? p[c] * (dp[c] - dot) : 0.f; wv[CIN * C + c] += dl[c]; }
        float g[CIN];
#pragma unroll
        for (int i = 0; i < CIN; ++i) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { a = fmaf(A.w[i * C + c], dl[c], a); wv[i * C + c] = fmaf(y[i], dl[c], wv[i * C + c]); }
            a = (valid && y[i] > 0.f) ? a : 0.f;
            const float xh = (zr[i] - A.bn[BN_MEAN * CIN + i]) * A.bn[BN_RSTD * CIN + i];
            g[i] = a; s1[i] += a; s2[i] += a * xh;
        }
        if (valid) {
#pragma unroll
            for (int i = 0; i < CIN; i += 4) sta4<AT>(reinterpret_cast<AT*>(A.g) + pix * CIN + i, make_float4(g[i], g[i + 1], g[i + 2], g[i + 3]));
        }
    }
    const size_t row = (size_t)b * gridDim.x + blockIdx.x;
    float* out = A.part + row * (2 * CIN);
    block_reduce_store<CP, true>(s1, red, out, CIN);
    block_reduce_store<CP, true>(s2, red, out + CIN, CIN);
    float* wout = A.wpart + row * NV;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        float t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) t[k] = wv[gi * 32 + k];
        block_reduce_store<32>(t, red, wout + gi * 32, NV - gi * 32 < 32 ? NV - gi * 32 : 32);
    }
    if (A.fin.counter) {
        __shared__ __attribute__((aligned(16))) char fin_lds[16 + kBlock * 16 + 2 * CP * 8];
        finalize_in_launch(A.fin, A.part, gridDim.x * gridDim.y, CIN, gridDim.x * gridDim.y, fin_lds);
    }
}

// ---- conv backward-weights (LDS-staged tiles, per-block partial dW, deterministic second-stage sum) ------------
struct ConvBwdWArgs {
    const void* x0; const float* ab0; int C0;   // conv input, same fetch semantics as the forward kernel
    const void* x1; const float* ab1; int C1;
    int flags;                                  // F_* (runtime here: staging is outside the FMA loop)
    const void* dz;                             // (B,H,W,Cout), activation storage type
    const void* zf; const float* bnf;           // BN-backward transform applied on load (zf != nullptr): `dz` is the masked
                                                // gradient g' of the layer, zf its raw output z, bnf its BN record; the
                                                // stager forms dz = ga g' + gb z + gd (common.hpp) instead of reading it
    float* part;                                // [npb][KH*KW*Cin*Cout + Cout]
    int B, H, W, Cin, Cout, tiles_x, tiles, total_tiles, npb;
    DropCfg drop;
    int act_bf16;
};

__device__ __forceinline__ float fetch_x(const ConvBwdWArgs& A, int b, int iy, int ix, int c) {
    // activated conv input at (b, iy, ix, channel c of the concatenated input); caller guarantees in-bounds
    const bool up = (A.flags & F_UP) != 0;
    const int Hs = up ? A.H >> 1 : A.H, Ws = up ? A.W >> 1 : A.W;
    const int sy = up ? iy >> 1 : iy, sx = up ? ix >> 1 : ix;
    const size_t pix = ((size_t)b * Hs + sy) * Ws + sx;
    if (A.flags & F_U8) return c_u8_lut[reinterpret_cast<const unsigned char*>(A.x0)[pix * A.C0 + c]];
    const float* src = reinterpret_cast<const float*>(A.x0); const float* ab = A.ab0; int C = A.C0, cc = c;
    if ((A.flags & F_TWO) && c >= A.C0) { src = reinterpret_cast<const float*>(A.x1); ab = A.ab1; C = A.C1; cc = c - A.C0; }
    float v = src[pix * C + cc];
    if (A.flags & F_AFF) v = fmaxf(fmaf(ab[cc], v, ab[C + cc]), 0.f);
    if (A.flags & F_DROP) v *= drop_mul(A.drop, (uint32_t)(pix * C + cc));
    return v;
}

// grid (npb, Cin/CI_T, ceil(Cout/CO_T)); thread t: entry e = t % (CI_T*CO_T) -> (ci, co); pixel split t / (CI_T*CO_T)
// (VALU kernel: first layer only -- its input is the caller's image, its dz the first block's gradient)
template <int KH, int CI_T, int CO_T, typename AT>
__global__ __launch_bounds__(kBlock) void conv_bwd_w_k(const ConvBwdWArgs A) {
    constexpr int KW = KH, PT = (KH - 1) / 2, NT = CI_T * CO_T, PS = kBlock / NT, TAPS = KH * KW;
    constexpr int XH = kTileY + KH - 1, XW = kTileX + KW - 1;
    static_assert(NT <= kBlock && kBlock % NT == 0, "entry tile must divide the block");
    __shared__ float Xs[XH * XW * CI_T];
    __shared__ float Ds[kTileY * kTileX * CO_T];
    __shared__ float Rs[kBlock * (TAPS + 1)];
    const int t = threadIdx.x, e = t % NT, ps = t / NT, ci = e / CO_T, co = e % CO_T;
    const int ci0 = blockIdx.y * CI_T, co0 = blockIdx.z * CO_T;
    float acc[TAPS], accb = 0.f;
#pragma unroll
    for (int k = 0; k < TAPS; ++k) acc[k] = 0.f;

    for (int tl = blockIdx.x; tl < A.total_tiles; tl += A.npb) {
        const int b = tl / A.tiles, tile = tl % A.tiles;
        const int x0 = (tile % A.tiles_x) * kTileX, y0 = (tile / A.tiles_x) * kTileY;
        __syncthreads();
        for (int i = t; i < XH * XW * CI_T; i += kBlock) {
            const int c = i % CI_T, r = i / CI_T, cx = r % XW, cy = r / XW;
            const int iy = y0 + cy - PT, ix = x0 + cx - PT;
            float v = 0.f;
            if (iy >= 0 && iy < A.H && ix >= 0 && ix < A.W && ci0 + c < A.Cin) v = fetch_x(A, b, iy, ix, ci0 + c);
            Xs[i] = v;
        }
        for (int i = t; i < kTileY * kTileX * CO_T; i += kBlock) {
            const int c = i % CO_T, r = i / CO_T, cx = r % kTileX, cy = r / kTileX;
            const int oy = y0 + cy, ox = x0 + cx;
            float v = 0.f;
            if (oy < A.H && ox < A.W && co0 + c < A.Cout) v = lda1<AT>(reinterpret_cast<const AT*>(A.dz) + (((size_t)b * A.H + oy) * A.W + ox) * A.Cout + co0 + c);
            Ds[i] = v;
        }
        __syncthreads();
        for (int p = ps; p < kTileY * kTileX; p += PS) {
            const int py = p / kTileX, pxx = p % kTileX;
            const float d = Ds[p * CO_T + co];
            accb += d;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
                    acc[ky * KW + kx] = fmaf(Xs[((py + ky) * XW + pxx + kx) * CI_T + ci], d, acc[ky * KW + kx]);
        }
    }
    // reduce the PS pixel-split groups
#pragma unroll
    for (int k = 0; k < TAPS; ++k) Rs[t * (TAPS + 1) + k] = acc[k];
    Rs[t * (TAPS + 1) + TAPS] = accb;
    __syncthreads();
    if (t < NT) {
        const size_t wsize = (size_t)TAPS * A.Cin * A.Cout;
        float* out = A.part + (size_t)blockIdx.x * (wsize + A.Cout);
        const bool ok = (ci0 + ci < A.Cin) && (co0 + co < A.Cout);
        for (int k = 0; k <= TAPS; ++k) {
            float s = 0.f;
            for (int q = 0; q < PS; ++q) s += Rs[(q * NT + t) * (TAPS + 1) + k];
            if (!ok) continue;
            if (k < TAPS) out[((size_t)k * A.Cin + ci0 + ci) * A.Cout + co0 + co] = s;
            else if (ci0 + ci == 0) out[wsize + co0 + co] = s;   // bias gradient = sum dz
        }
    }
}

// ---- first layer, the real configuration (1 input channel, 8 output channels, 3x3): dW[tap][co] = sum_px x(px+tap) dz(px)[co]
// Pure streaming reduction over dz (32 B/pixel, read straight from global memory, fully coalesced) against the image tile
// in LDS; every thread keeps all 72 weight + 8 bias sums in registers, one block reduction per persistent block.
// grid (npb); partial slab layout [tap][co] (72) + bias (8) = the generic [tap][ci][co] + bias layout for Cin = 1.
// FUSE: the BN-backward transform of this layer is applied here instead of by bn_bwd_apply_k (same expression, same
// operation order): nothing else consumes this layer's dz -- the input image has no gradient -- so the 3-tensor pass
// over the largest tensor of the net disappears from the backward-data chain (66 us of a 4.2 ms step).
template <typename AT, bool FUSE>
__global__ __launch_bounds__(kBlock) void conv_dw_first_k(const ConvBwdWArgs A, int tiles_x, int tiles, int total_tiles) {
    constexpr int TH = 8, TW = 128, XH = TH + 2, XW = TW + 2;
    __shared__ float Xs[XH * XW];
    __shared__ float red[256];
    __shared__ float lut_s[256];          // the float32(i/255.0) table, copied once: lookups become LDS reads
    const int t = threadIdx.x, xl = t & 31, row = t >> 5;
    lut_s[t] = c_u8_lut[t];
    float acc0[64], acc1[16];       // acc0 = taps 0..7 x 8 channels; acc1 = tap 8 x 8 channels, then the 8 bias sums
#pragma unroll
    for (int k = 0; k < 64; ++k) acc0[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc1[k] = 0.f;
    const bool u8 = (A.flags & F_U8) != 0;
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
                if (in[k]) { if (u8) rb[k] = reinterpret_cast<const unsigned char*>(A.x0)[pix]; else rf[k] = reinterpret_cast<const float*>(A.x0)[pix]; }
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = t + k * kBlock;
                if (i < XH * XW) Xs[i] = in[k] ? (u8 ? lut_s[rb[k]] : rf[k]) : 0.f;
            }
        }
        __syncthreads();
        const int y = y0 + row;
#pragma unroll
        for (int k = 0; k < TW / 32; ++k) {
            const int xx = xl + 32 * k, x = x0 + xx;
            if (y < A.H && x < A.W) {
                const size_t e8 = (((size_t)b * A.H + y) * A.W + x) * 8;
                const AT* dp = reinterpret_cast<const AT*>(A.dz) + e8;
                const float4 d0 = lda4<AT>(dp), d1 = lda4<AT>(dp + 4);
                float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                if constexpr (FUSE) {
                    const AT* zp = reinterpret_cast<const AT*>(A.zf) + e8;
                    const float4 z0 = lda4<AT>(zp), z1 = lda4<AT>(zp + 4);
                    const float zv[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
                    for (int co = 0; co < 8; ++co) {      // bn_bwd_apply_k's expression; bf16 storage: dz is rounded as it would be stored
                        const float o = bn_bwd_apply1(A.bnf[BN_GA * 8 + co], A.bnf[BN_GB * 8 + co], A.bnf[BN_GD * 8 + co], d[co], zv[co]);
                        d[co] = dz_as_stored<AT>(o);
                        asm volatile("" : "+v"(d[co]));      // dz is a ROUNDED value (as if stored): its last product must not be
                                                             // contracted into the sums below
                    }
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float xv = Xs[(row + tap / 3) * XW + xx + tap % 3];
#pragma unroll
                    for (int co = 0; co < 8; ++co) {
                        if (tap < 8) acc0[tap * 8 + co] = fmaf(xv, d[co], acc0[tap * 8 + co]);
                        else acc1[co] = fmaf(xv, d[co], acc1[co]);
                    }
                }
#pragma unroll
                for (int co = 0; co < 8; ++co) acc1[8 + co] += d[co];
            }
        }
    }
    float* out = A.part + (size_t)blockIdx.x * 80;
    block_reduce_store<64>(acc0, red, out, 64);
    block_reduce_store<16>(acc1, red, out + 64, 16);
}

// second stage: grads[j] = sum over pixel blocks of part[pb][j].  Block = JW consecutive j x (256/JW) slices of
// pb; fixed summation order (deterministic), double accumulation.
template <int JW>
__global__ __launch_bounds__(kBlock) void reduce_partials_k(const float* __restrict__ part, int npb, size_t stride,
                                                           size_t wsize, float* __restrict__ dw, float* __restrict__ db) {
    constexpr int NQ = kBlock / JW;
    __shared__ double sh[NQ][JW];
    const int col = threadIdx.x % JW, q = threadIdx.x / JW;
    const size_t j = (size_t)blockIdx.x * JW + col;
    double s = 0;
    if (j < stride)
        for (int p = q; p < npb; p += NQ) s += part[(size_t)p * stride + j];
    sh[q][col] = s;
    __syncthreads();
    if (q == 0 && j < stride) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < NQ; ++k) t += sh[k][col];
        if (j < wsize) dw[j] = (float)t; else db[j - wsize] = (float)t;
    }
}

// all layers' slab sums in ONE launch (the per-layer launches are latency-bound, ~20 us each): descriptor table by value
struct ReduceAllArgs {
    static constexpr int MAXL = 40;
    int n;
    struct Entry { const float* part; float* dw; float* db; int npb, jw; unsigned stride, wsize, blk_start; } L[MAXL];
};

static __global__ __launch_bounds__(kBlock) void reduce_all_k(const ReduceAllArgs A) {
    __shared__ double sh[kBlock];
    int d = 0;
    while (d + 1 < A.n && blockIdx.x >= A.L[d + 1].blk_start) ++d;
    const ReduceAllArgs::Entry E = A.L[d];
    const int jw = E.jw, nq = kBlock / jw, col = threadIdx.x % jw, q = threadIdx.x / jw;
    const size_t j = (size_t)(blockIdx.x - E.blk_start) * jw + col;
    double s = 0;
    if (j < E.stride) {
        const float* __restrict__ src = E.part + j;
        int p = q;
        for (; p + 7 * nq < E.npb; p += 8 * nq) {      // 8 independent loads in flight per thread, fixed summation order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(p + k * nq) * E.stride];
            s += (((double)v[0] + v[1]) + ((double)v[2] + v[3])) + (((double)v[4] + v[5]) + ((double)v[6] + v[7]));
        }
        for (; p < E.npb; p += nq) s += src[(size_t)p * E.stride];
    }
    sh[q * jw + col] = s;
    __syncthreads();
    if (q == 0 && j < E.stride) {
        double t = 0;
        for (int k = 0; k < nq; ++k) t += sh[k * jw + col];
        if (j < E.wsize) E.dw[j] = (float)t; else E.db[j - E.wsize] = (float)t;
    }
}

// ---- optimizers (Keras formulations) ----------------------------------------------------
static __global__ __launch_bounds__(kBlock) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float eps) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

static __global__ __launch_bounds__(kBlock) void sgd_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                                               size_t n, float lr, float momentum) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if (mom) { const float vi = momentum * mom[i] - lr * g[i]; mom[i] = vi; p[i] += vi; }
        else p[i] -= lr * g[i];
    }
}

}  // namespace oct
