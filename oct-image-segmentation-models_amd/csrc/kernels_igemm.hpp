// MFMA implicit-GEMM convolution for gfx950 (fp32 in / fp32 accumulate: v_mfma_f32_32x32x2_f32 and
// v_mfma_f32_16x16x4_f32 -- exact f32 FMA chains at the fp32 peak rate, 157 TF; MI355X_MICROARCH.md).
//
// One kernel template serves the forward convs AND backward-data:
//     out[b, y, x, m] = sum_{tap, c}  Wt[tap][c][m] * In(b, pixel(y, x, tap), c)
//   M (MFMA rows)  = output channels, tile 32 (C>=32) or 16 (thin layers)
//   N (MFMA cols)  = 32 / 16 consecutive output pixels of one image row
//   K              = taps x input channels, consumed in chunks of KC channels
// * The input tile (with halo) is staged into LDS in PLANAR layout [c][row][col] with the consumer-side
//   transform applied while loading (BN+ReLU affine, concat of two sources, dropout): 32 lanes of a B-operand
//   read hit 32 consecutive banks -> conflict-free ds_read_b32.
// * The weight chunk [tap][c][m] is staged into LDS with m fastest -> conflict-free A-operand reads.
// * Addressing modes: A_NORMAL (3x3 'same'), A_UPF (2x2 conv over a nearest-upsampled low-res tile: the
//   upsampled tensor is never materialised), A_DOWN2 (stride-2 3x3 gather over dz with effective weights =
//   backward-data of the up-conv with UpSampling2D's 2x2 window sum folded in).
// * Epilogues: EPI_FWD (bias, store z, per-channel sum / sum^2 partials), EPI_RAW (store gradient),
//   EPI_MASK (ReLU mask (+dropout) from the producer's z, BN-backward partial sums, store g').
#pragma once
#include "common.hpp"
#include "kernels_fin.hpp"

namespace oct {

enum { A_NORMAL = 0, A_UPF = 1, A_DOWN2 = 2 };
constexpr int kBxGbDown2MaxC = 192;   // conv_bx_k, stride-2 gather with the BN-backward transform on load: K channels whose three
                                      // coefficient rows still fit beside the LDS images (the host falls back to the separate pass)
enum { EPI_FWD = 0, EPI_RAW = 1, EPI_MASK = 2 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmArgs {
    const void* x0; const float* ab0; int C0;    // input source 0 (activation storage type; + BN record when F_AFF)
    const void* x1; const float* ab1; int C1;    // source 1 of a concat (F_TWO)
    int flags;                                   // F_AFF | F_TWO | F_DROP   (runtime: staging is not the hot loop)
    int Cin;                                     // K channels in total
    const float* w; int w_ld; int m_off;         // weights [taps][Cin][w_ld]; this launch covers columns m_off..m_off+Mout
    const float* bias;                           // EPI_FWD
    void* out; int Mout;                         // (B,Ho,Wo,Mout), activation storage type
    int Ho, Wo, Hi, Wi;                          // output / input-source spatial dims
    int tiles_x, tiles, total_tiles;            // total_tiles = B * tiles (persistent variant)
    float* part;                                 // [B*tiles][2*Mout] statistic partials (EPI_FWD / EPI_MASK) or nullptr
    const void* zin; const float* bnin;          // EPI_MASK: producer's raw output (B,Ho,Wo,Mout) + its BN record
    int act_bf16;                                // activation storage type selector for the launchers
    int drop_out;                                // EPI_MASK: the producer's output passes through dropout
    DropCfg drop;
    const void* wbx; int wbx_M;                  // bf16-pipe kernels (kernels_bx.hpp): split weights of this layer direction
                                                 // in prep_wbx_k layout, and the total M the layout was built for
    int bt_m2;                                   // ... prepared in the two-pixel form (8 output channels x 2 adjacent pixels = 16 rows)
    const void* wbt;                             // thin bf16-pipe kernel: prep_wbt_k slice for rows m_off .. m_off + 15
    const void* gb_z; const float* gb_bn;        // backward-data on the bf16-pipe kernels, BN-backward transform on load: x0 is
                                                 // the masked gradient g' of the layer, gb_z its raw output z, gb_bn its BN
                                                 // record (Cin channels): the stager forms dz = ga g' + gb z + gd (common.hpp)
    // conv_bt_k FDW (backward-data launch that also reduces the layer's backward-weights, kernels_bx.hpp): the conv INPUT
    // slice this launch's gradient belongs to -- dw_x = that producer's raw output z (B,Ho,Wo,Mout), dw_ab its BN record
    // (rows a, b) -- the layer's partial-slab base [grid][9 * dw_Cin * Cin + Cin], the layer's input-channel count, this
    // slice's first input channel, and whether this launch also writes the bias-gradient column sums
    const void* dw_x; const float* dw_ab; float* dw_part; int dw_Cin, dw_ci_off, dw_bias;
    FinDesc fin;                                 // conv_bt_k: the statistics this launch emits (EPI_FWD: of its own output;
                                                 // EPI_MASK: the BN-backward sums of the producer) are finalized by its last block
};

template <int SHAPE> struct MfmaShape;
template <> struct MfmaShape<32> {
    static constexpr int KS = 2, ACC = 16, QUADS = 4;
    using acc_t = f32x16;
    __device__ static inline acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    // channel (row of D) of quad q, element r for lane-half h:  (r&3) + 8*(reg>>2) + 4*h  with reg = 4q + r
    __device__ static inline int quad_base(int q, int h) { return 8 * q + 4 * h; }
};
template <> struct MfmaShape<16> {
    static constexpr int KS = 4, ACC = 4, QUADS = 1;
    using acc_t = f32x4;
    __device__ static inline acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static inline int quad_base(int q, int h) { return 4 * h; }
};

// sum N values over the lanes that share the upper lane bits (sub-group = lanes differing in bits < log2(2*OFF0)):
// halving butterfly from offset OFF0 down to 1.  On return v[0] holds the total of value index
// sub_chan<N,OFF0>(lane).
template <int NFULL, int N, int OFF>
__device__ inline void subgroup_reduce_rec(float (&v)[NFULL], int lane) {
    if constexpr (OFF >= 1) {
        if constexpr (N > 1) {
            const bool upper = (lane & OFF) != 0;
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const float send = upper ? v[i] : v[i + N / 2];
                const float keep = upper ? v[i + N / 2] : v[i];
                v[i] = keep + __shfl_xor(send, OFF, 64);
            }
            subgroup_reduce_rec<NFULL, N / 2, OFF / 2>(v, lane);
        } else {
            v[0] += __shfl_xor(v[0], OFF, 64);
            subgroup_reduce_rec<NFULL, 1, OFF / 2>(v, lane);
        }
    }
}
template <int N, int OFF0>
__device__ inline int sub_chan(int lane) {
    int c = 0, n = N, off = OFF0;
    while (n > 1) { if (lane & off) c += n / 2; n >>= 1; off >>= 1; }
    return c;
}

// ---- MFMA sweep over one staged K chunk: flat (tap x k-step) sequence, fully unrolled, operands double-buffered
// in registers so the LDS reads of step i+1 are in flight while the MFMAs of step i issue ----
template <int SHAPE, int KH, int AMODE, int KCX, int MB, int IW, int PLANE, int NPR, int MTW, int NTW, bool FULL>
__device__ __forceinline__ void mfma_sweep(const float* __restrict__ Ws, const float* __restrict__ Ib,
                                           typename MfmaShape<SHAPE>::acc_t (&acc)[MTW][NTW], const int (&boff)[NTW],
                                           int wn, int wm, int j, int kk, int nks) {
    using S = MfmaShape<SHAPE>;
    constexpr int MT = SHAPE, NT = SHAPE, KS = S::KS, NKS = KCX / KS, TAPS = KH * KH, STEPS = TAPS * NKS;
    auto load = [&](int i, float (&a)[MTW], float (&bv)[NTW]) {
        const int tap = i / NKS, ks = i % NKS, ky = tap / KH, kx = tap % KH, kc = ks * KS + kk;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) a[mt] = Ws[(tap * KCX + kc) * MB + (wm * MTW + mt) * MT + j];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            int off;
            if constexpr (AMODE == A_UPF) {
                const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
                off = ((r + ky) >> 1) * IW + ((xs + kx) >> 1);
            } else {
                off = boff[nt] + ky * IW + kx;
            }
            bv[nt] = Ib[kc * PLANE + off];
        }
    };
    auto fma = [&](const float (&a)[MTW], const float (&bv)[NTW]) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = S::mfma(a[mt], bv[nt], acc[mt][nt]);
    };
    if constexpr (FULL) {
        float a0[MTW], b0[NTW], a1[MTW], b1[NTW];
        __builtin_amdgcn_s_setprio(2);      // waves in their MFMA phase issue ahead of waves that are staging
        load(0, a0, b0);
#pragma unroll
        for (int i = 0; i < STEPS; i += 2) {
            if (i + 1 < STEPS) load(i + 1, a1, b1);
            fma(a0, b0);
            if (i + 2 < STEPS) load(i + 2, a0, b0);
            if (i + 1 < STEPS) fma(a1, b1);
        }
        __builtin_amdgcn_s_setprio(0);
    } else {   // fewer channels than the chunk holds (tiny test configurations): plain runtime loop
        for (int tap = 0; tap < TAPS; ++tap)
            for (int ks = 0; ks < nks; ++ks) {
                float a[MTW], bv[NTW];
                load(tap * NKS + ks, a, bv);
                fma(a, bv);
            }
    }
}

// grid (tiles, ceil(Mout/MB), B), block 256 = 4 waves arranged WN (pixel tiles) x WM (channel tiles)
template <int SHAPE, int KH, int AMODE, int EPI, int TH, int MB, int WN, typename AT>
__global__ __launch_bounds__(kBlock) void conv_igemm_k(const IgemmArgs A) {
    using S = MfmaShape<SHAPE>;
    constexpr int MT = SHAPE, NT = SHAPE, KS = S::KS, ACC = S::ACC, QUADS = S::QUADS;
    constexpr int TW = 32, NPR = TW / NT, NTB = TH * NPR, MTB = MB / MT, WM = 4 / WN;
    constexpr int NTW = NTB / WN, MTW = MTB / WM, TAPS = KH * KH;
    static_assert(NTB % WN == 0 && MTB % WM == 0 && NTW >= 1 && MTW >= 1, "bad wave arrangement");
    constexpr int KC = (AMODE == A_DOWN2 && TH >= 8) ? 8 : 16;
    constexpr int IH = AMODE == A_NORMAL ? TH + KH - 1 : (AMODE == A_UPF ? TH / 2 + 1 : 2 * TH + 1);
    constexpr int IW = AMODE == A_NORMAL ? TW + KH - 1 : (AMODE == A_UPF ? TW / 2 + 1 : 2 * TW + 1);
    constexpr int PLANE = ((IH * IW + 15) / 32) * 32 + 16;   // == 16 (mod 32): the two k-rows of a 16x16x4 read hit disjoint banks
    constexpr int RED = 4 * 2 * MTW * MT;

    __shared__ float Is[KC * PLANE];
    __shared__ float Ws[TAPS * KC * MB];
    __shared__ float red[RED];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int j = lane & (NT - 1), kk = lane / NT, h = kk;   // B/A operand: column j (or row i), k index kk; D: lane half/quarter h
    const int tile = blockIdx.x, b = blockIdx.z, m0 = blockIdx.y * MB;
    const int x0 = (tile % A.tiles_x) * TW, y0 = (tile / A.tiles_x) * TH;
    const int iy0 = AMODE == A_NORMAL ? y0 - (KH - 1) / 2 : (AMODE == A_UPF ? y0 / 2 : 2 * y0 - 1);
    const int ix0 = AMODE == A_NORMAL ? x0 - (KH - 1) / 2 : (AMODE == A_UPF ? x0 / 2 : 2 * x0 - 1);

    typename S::acc_t acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < ACC; ++r) acc[mt][nt][r] = 0.f;

    // per-lane LDS offsets of the B operand for each of this wave's pixel tiles (tap offset added in the loop)
    int boff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
        boff[nt] = AMODE == A_NORMAL ? r * IW + xs : (AMODE == A_UPF ? 0 : (2 * r) * IW + 2 * xs);
    }

    // K-chunk pipeline: the global loads of chunk c+1 (input tile + weight tile) are issued into registers before
    // the MFMA sweep of chunk c and written to LDS after it
    constexpr int NPI = ((KC / 4) * IH * IW + kBlock - 1) / kBlock;
    constexpr int NPW = (TAPS * KC * (MB / 4) + kBlock - 1) / kBlock;
    typename Raw4<AT>::type pin[NPI];            // raw prefetch registers (widened in store_chunk)
    float4 pw[NPW];
    // Tile geometry is the same for every chunk: per slot the image-relative pixel offset (-1 = outside the image or
    // past the tile) and the LDS offset are computed ONCE; a thread always serves channel quad q = tid % (KC/4), so per
    // chunk only its source tensor (concat aware), channel offset and BN affine pair change -- fetched once per chunk,
    // not once per element.
    constexpr int QC = KC / 4, PPI = kBlock / QC;
    static_assert(kBlock % QC == 0, "channel quad must be thread-invariant");
    const int q4 = 4 * (tid % QC);
    int goff[NPI], loff[NPI];
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
        const int p = tid / QC + k * PPI, lx = p % IW, ly = p / IW, gy = iy0 + ly, gx = ix0 + lx;
        const bool in = p < IH * IW && gy >= 0 && gy < A.Hi && gx >= 0 && gx < A.Wi;
        goff[k] = in ? gy * A.Wi + gx : -1;
        loff[k] = p < IH * IW ? ly * IW + lx : -1;
    }
    // weight tile: slot -> (tap, kc, m4) fixed; element offset relative to channel c0 computed once
    int woff[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int e = tid + k * kBlock, m4 = e % (MB / 4), r = e / (MB / 4), kc = r % KC, tap = r / KC, m = m0 + 4 * m4;
        woff[k] = (e < TAPS * KC * (MB / 4) && m < A.Mout) ? (tap * A.Cin + kc) * A.w_ld + A.m_off + m : -1;
    }
    const size_t img = (size_t)b * A.Hi * A.Wi;
    auto load_chunk = [&](int c0) {
        const int c = c0 + q4;
        const bool two = (A.flags & F_TWO) && c >= A.C0;
        const int C = two ? A.C1 : A.C0;
        const AT* __restrict__ src = (two ? reinterpret_cast<const AT*>(A.x1) + (c - A.C0) : reinterpret_cast<const AT*>(A.x0) + c) + img * C;
        const bool cok = c < A.Cin;
#pragma unroll
        for (int k = 0; k < NPI; ++k)
            pin[k] = (cok && goff[k] >= 0) ? ldraw4<AT>(src + (size_t)goff[k] * C) : raw_zero4<AT>();
        const float* __restrict__ wsrc = A.w + (size_t)c0 * A.w_ld;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            // (tap, kc) row exists for this chunk iff c0 + kc < Cin; kc = (slot / (MB/4)) % KC is slot-invariant
            const int kc = ((tid + k * kBlock) / (MB / 4)) % KC;
            pw[k] = (woff[k] >= 0 && c0 + kc < A.Cin) ? ld4(wsrc + woff[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int c0) {
        const int c = c0 + q4;
        const bool two = (A.flags & F_TWO) && c >= A.C0;
        const int C = two ? A.C1 : A.C0, cc = two ? c - A.C0 : c;
        const bool cok = c < A.Cin;
        float4 fa = make_float4(1.f, 1.f, 1.f, 1.f), fb = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((A.flags & F_AFF) && cok) { const float* ab = two ? A.ab1 : A.ab0; fa = ld4(ab + cc); fb = ld4(ab + C + cc); }
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            if (loff[k] < 0) continue;
            float4 v = widen4(pin[k]);
            const bool in = cok && goff[k] >= 0;
            if (A.flags & F_AFF) {       // zero padding is applied AFTER the activation: out-of-image stays 0
                v.x = in ? fmaxf(fmaf(fa.x, v.x, fb.x), 0.f) : 0.f; v.y = in ? fmaxf(fmaf(fa.y, v.y, fb.y), 0.f) : 0.f;
                v.z = in ? fmaxf(fmaf(fa.z, v.z, fb.z), 0.f) : 0.f; v.w = in ? fmaxf(fmaf(fa.w, v.w, fb.w), 0.f) : 0.f;
            }
            if ((A.flags & F_DROP) && in) {
                const uint32_t el = (uint32_t)((img + goff[k]) * C + cc);
                v.x *= drop_mul(A.drop, el); v.y *= drop_mul(A.drop, el + 1);
                v.z *= drop_mul(A.drop, el + 2); v.w *= drop_mul(A.drop, el + 3);
            }
            float* d = Is + q4 * PLANE + loff[k];
            d[0] = v.x; d[PLANE] = v.y; d[2 * PLANE] = v.z; d[3 * PLANE] = v.w;
        }
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int e = tid + k * kBlock;
            if (e < TAPS * KC * (MB / 4)) st4(Ws + (e / (MB / 4)) * MB + 4 * (e % (MB / 4)), pw[k]);
        }
    };

    load_chunk(0);
    for (int c0 = 0; c0 < A.Cin; c0 += KC) {
        __syncthreads();   // previous chunk fully consumed
        store_chunk(c0);
        __syncthreads();
        if (c0 + KC < A.Cin) load_chunk(c0 + KC);
        // ---- MFMA over taps x k-steps (only the channels that exist: thin layers have Cin < KC) ----
        const int nks = (A.Cin - c0 < KC ? A.Cin - c0 : KC) / KS;
        if (nks == KC / KS) mfma_sweep<SHAPE, KH, AMODE, KC, MB, IW, PLANE, NPR, MTW, NTW, true>(Ws, Is, acc, boff, wn, wm, j, kk, nks);
        else mfma_sweep<SHAPE, KH, AMODE, KC, MB, IW, PLANE, NPR, MTW, NTW, false>(Ws, Is, acc, boff, wn, wm, j, kk, nks);
    }

    // ---- epilogue ----
    // Per-channel constants of this M block (bias, or the producer's BN record a / b / mean / rstd) are copied ONCE into
    // the weight buffer, which is free now, and every z value the mask needs is requested up front into raw registers:
    // inside the output loop a global load cannot be moved above the preceding (possibly aliasing) store, so loading
    // there costs one memory round trip per quad.
    __syncthreads();                       // every wave has finished its last sweep over Ws / Is
    float* const epi = Ws;                 // [4][MB] (EPI_MASK: BN_A, BN_B, BN_MEAN, BN_RSTD rows) or [MB] (EPI_FWD: bias)
    if constexpr (EPI == EPI_FWD) {
        for (int e = tid; e < MB; e += kBlock) epi[e] = m0 + e < A.Mout ? A.bias[A.m_off + m0 + e] : 0.f;
    } else if constexpr (EPI == EPI_MASK) {
        for (int e = tid; e < 4 * MB; e += kBlock) {
            const int arr = e / MB, m = m0 + e % MB;
            epi[e] = m < A.Mout ? A.bnin[arr * A.Mout + m] : 0.f;
        }
    }
    typename Raw4<AT>::type zraw[EPI == EPI_MASK ? NTW : 1][EPI == EPI_MASK ? MTW : 1][EPI == EPI_MASK ? QUADS : 1];
    if constexpr (EPI == EPI_MASK) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
            const int y = y0 + r, x = x0 + xs;
            const bool pvalid = y < A.Ho && x < A.Wo;
            const size_t pix = pvalid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int q = 0; q < QUADS; ++q) {
                    const int m = m0 + (wm * MTW + mt) * MT + S::quad_base(q, h);
                    zraw[nt][mt][q] = (pvalid && m < A.Mout) ? ldraw4<AT>(reinterpret_cast<const AT*>(A.zin) + pix * A.Mout + m) : raw_zero4<AT>();
                }
        }
    }
    __syncthreads();                       // epi visible
    float s1[MTW][ACC], s2[MTW][ACC];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < ACC; ++r) { s1[mt][r] = 0.f; s2[mt][r] = 0.f; }

#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
        const int y = y0 + r, x = x0 + xs;
        const bool pvalid = y < A.Ho && x < A.Wo;
        const size_t pix = pvalid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
            for (int q = 0; q < QUADS; ++q) {
                const int ml = (wm * MTW + mt) * MT + S::quad_base(q, h), m = m0 + ml;     // channel within the block / overall
                const bool valid = pvalid && m < A.Mout;
                float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                if constexpr (EPI == EPI_FWD) {
                    if (valid) {
                        const float4 bs = ld4(epi + ml);
                        v[0] += bs.x; v[1] += bs.y; v[2] += bs.z; v[3] += bs.w;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float u = valid ? v[k] : 0.f;
                        s1[mt][4 * q + k] += u; s2[mt][4 * q + k] += u * u;
                    }
                } else if constexpr (EPI == EPI_MASK) {
                    const float4 zq = widen4(zraw[nt][mt][q]);
                    const float zz[4] = {zq.x, zq.y, zq.z, zq.w};
                    const float4 ea = ld4(epi + BN_A * MB + ml), eb = ld4(epi + BN_B * MB + ml);
                    const float4 em = ld4(epi + BN_MEAN * MB + ml), er = ld4(epi + BN_RSTD * MB + ml);
                    const float ka[4] = {ea.x, ea.y, ea.z, ea.w}, kb[4] = {eb.x, eb.y, eb.z, eb.w};
                    const float km[4] = {em.x, em.y, em.z, em.w}, kr[4] = {er.x, er.y, er.z, er.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = valid ? m + k : 0;
                        const float yv = fmaf(ka[k], zz[k], kb[k]);
                        float gv = v[k];
                        if (A.drop_out) gv *= drop_mul(A.drop, (uint32_t)(pix * A.Mout + c));
                        gv = (valid && yv > 0.f) ? gv : 0.f;
                        const float xh = (zz[k] - km[k]) * kr[k];
                        v[k] = gv; s1[mt][4 * q + k] += gv; s2[mt][4 * q + k] += gv * xh;
                    }
                }
                if (valid) sta4<AT>(reinterpret_cast<AT*>(A.out) + pix * A.Mout + m, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
    }

    if constexpr (EPI != EPI_RAW) {
        if (A.part) {
            // lanes sharing h hold the same channel set for different pixels: reduce over the pixel lanes
            constexpr int OFF0 = NT / 2;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                subgroup_reduce_rec<ACC, ACC, OFF0>(s1[mt], lane);
                subgroup_reduce_rec<ACC, ACC, OFF0>(s2[mt], lane);
                const int ci = sub_chan<ACC, OFF0>(lane);                      // value index this lane now holds
                const int mloc = S::quad_base(ci >> 2, h) + (ci & 3);          // channel within the M tile
                red[((wave * 2 + 0) * MTW + mt) * MT + mloc] = s1[mt][0];
                red[((wave * 2 + 1) * MTW + mt) * MT + mloc] = s2[mt][0];
            }
            __syncthreads();
            if (tid < 2 * MB) {
                const int stat = tid / MB, ml = tid % MB;            // channel ml of this block's MB
                const int mtile = ml / MT, wmm = mtile / MTW, mt = mtile % MTW, mloc = ml % MT;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) s += red[(((wmm * WN + w) * 2 + stat) * MTW + mt) * MT + mloc];
                if (m0 + ml < A.Mout)
                    A.part[((size_t)b * A.tiles + tile) * (2 * A.Mout) + (size_t)stat * A.Mout + m0 + ml] = s;
            }
        }
    }
}

// ---- XCD-aware persistent tile walk without divisions.  Workgroups are dealt round-robin to the 8 XCDs (each with its
// own L2), so block i serves XCD i % 8: the tile sequence is cut into 8 contiguous bands, one per XCD, and the blocks of
// an XCD stride through their band -- tiles that share halo rows / columns are then read through the SAME L2 and each
// XCD streams one contiguous region of the tensors.  A block visits tiles tl0, tl0 + step, ... < tlend; the origin
// (b, ty, tx) is advanced by the decomposed stride with two carries. ----
struct TileOrg { int b, ty, tx; };
template <int TH, int TW>
struct TileWalk {
    int tiles_x, tiles_y, sb, sy, sx, tl0, tlend, step;
    __device__ __forceinline__ void init(int tiles, int tx_, int total_tiles) {
        constexpr int NX = 8;
        if (gridDim.x % NX == 0) {
            const int xcd = blockIdx.x % NX, chunk = (total_tiles + NX - 1) / NX;
            step = gridDim.x / NX; tl0 = xcd * chunk + blockIdx.x / NX;
            tlend = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
        } else { step = gridDim.x; tl0 = blockIdx.x; tlend = total_tiles; }
        tiles_x = tx_; tiles_y = tiles / tx_;
        sb = step / tiles; sy = (step % tiles) / tiles_x; sx = step % tiles_x;
    }
    __device__ __forceinline__ TileOrg first(int tiles) const {
        TileOrg o; const int r = tl0 % tiles; o.b = tl0 / tiles; o.ty = r / tiles_x; o.tx = r % tiles_x; return o;
    }
    __device__ __forceinline__ TileOrg next(TileOrg o) const {
        o.tx += sx; int carry = 0;
        if (o.tx >= tiles_x) { o.tx -= tiles_x; carry = 1; }
        o.ty += sy + carry; carry = 0;
        if (o.ty >= tiles_y) { o.ty -= tiles_y; carry = 1; }
        o.b += sb + carry;
        return o;
    }
};

// ---- input-tile staging for the 8-output-channel kernels: global -> registers (prefetch) -> PLANAR LDS [c][row][col]
// with the consumer-side transform (BN affine + ReLU, dropout, concat of two sources, zero padding).
// Set up ONCE per thread: a thread always serves the same channel quad, so its source tensor, BN affine and LDS plane
// are tile-invariant; per slot k only the packed local pixel (ly, lx) is kept. ----
template <int CMAX, int IH, int IW, int IWP, int PLANE, int AMODE, int TH, int TW, typename AT, int NT = kBlock>
struct ThinStager {
    static constexpr int Q = CMAX / 4, PPI = NT / Q, NPIX = IH * IW, NPF = (NPIX + PPI - 1) / PPI;
    const AT* __restrict__ src; float* lds_q;
    int Csrc, cc; bool cok;
    float4 fa, fb;
    struct Buf { typename Raw4<AT>::type v[NPF]; };     // one tile's worth of RAW prefetched registers (widened in store())
    int lxy[NPF];

    __device__ __forceinline__ void init(const IgemmArgs& A, float* Is) {
        const int tid = threadIdx.x, q = tid % Q, c = 4 * q;
        const bool two = (A.flags & F_TWO) && c >= A.C0;
        Csrc = two ? A.C1 : A.C0; cc = two ? c - A.C0 : c;
        src = (two ? reinterpret_cast<const AT*>(A.x1) : reinterpret_cast<const AT*>(A.x0)) + cc;
        cok = c < A.Cin;
        fa = make_float4(1.f, 1.f, 1.f, 1.f); fb = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((A.flags & F_AFF) && cok) { const float* ab = two ? A.ab1 : A.ab0; fa = ld4(ab + cc); fb = ld4(ab + Csrc + cc); }
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int p = tid / Q + k * PPI;
            lxy[k] = p < NPIX ? ((p / IW) << 8) | (p % IW) : -1;
        }
        lds_q = Is + c * PLANE;
    }
    static __device__ __forceinline__ void origin(const TileOrg& o, int& iy0, int& ix0) {
        const int y0 = o.ty * TH, x0 = o.tx * TW;
        iy0 = AMODE == A_NORMAL ? y0 - 1 : y0 / 2; ix0 = AMODE == A_NORMAL ? x0 - 1 : x0 / 2;
    }
    // issue the global loads of tile o (out-of-image / absent channels load nothing and become zeros)
    __device__ __forceinline__ void load(const IgemmArgs& A, const TileOrg& o, Buf& pf) {
        int iy0, ix0; origin(o, iy0, ix0);
        const long long basepix = ((long long)o.b * A.Hi + iy0) * A.Wi + ix0;       // wave-uniform (may point into the halo)
        const AT* __restrict__ tb = src + basepix * Csrc;
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= A.Hi && ix0 + IW <= A.Wi;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int ly = lxy[k] >> 8, lx = lxy[k] & 255;
            bool ok = cok && lxy[k] >= 0;
            if (!interior) ok = ok && (unsigned)(iy0 + ly) < (unsigned)A.Hi && (unsigned)(ix0 + lx) < (unsigned)A.Wi;
            pf.v[k] = ok ? ldraw4<AT>(tb + (ly * A.Wi + lx) * Csrc) : raw_zero4<AT>();
        }
    }
    // write the prefetched tile to LDS with the transform applied
    __device__ __forceinline__ void store(const IgemmArgs& A, const TileOrg& o, const Buf& pf, int lds_off = 0) {
        int iy0, ix0; origin(o, iy0, ix0);
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= A.Hi && ix0 + IW <= A.Wi;
        const int basepix = (o.b * A.Hi + iy0) * A.Wi + ix0;                      // only used for the dropout element index
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            if (lxy[k] < 0) continue;
            const int ly = lxy[k] >> 8, lx = lxy[k] & 255;
            float4 v = widen4(pf.v[k]);
            bool in = cok;
            if (!interior) in = in && (unsigned)(iy0 + ly) < (unsigned)A.Hi && (unsigned)(ix0 + lx) < (unsigned)A.Wi;
            if (A.flags & F_AFF) {       // zero padding is applied AFTER the activation: out-of-image stays 0
                v.x = in ? fmaxf(fmaf(fa.x, v.x, fb.x), 0.f) : 0.f; v.y = in ? fmaxf(fmaf(fa.y, v.y, fb.y), 0.f) : 0.f;
                v.z = in ? fmaxf(fmaf(fa.z, v.z, fb.z), 0.f) : 0.f; v.w = in ? fmaxf(fmaf(fa.w, v.w, fb.w), 0.f) : 0.f;
            }
            if (A.flags & F_DROP) {
                const uint32_t el = (uint32_t)((basepix + ly * A.Wi + lx) * Csrc + cc);
                if (in) { v.x *= drop_mul(A.drop, el); v.y *= drop_mul(A.drop, el + 1);
                          v.z *= drop_mul(A.drop, el + 2); v.w *= drop_mul(A.drop, el + 3); }
            }
            float* d = lds_q + lds_off + ly * IWP + lx;
            d[0] = v.x; d[PLANE] = v.y; d[2 * PLANE] = v.z; d[3 * PLANE] = v.w;
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant for single-chunk layers (Cin <= KCP <= 16: the thin full-resolution layers,
// which are HBM/latency-bound rather than MFMA-bound).  One block walks a strided list of pixel tiles:
//   * the weight tile is staged ONCE per block;
//   * the input tile is double-buffered in LDS; the global loads of tile t+1 (and, for EPI_MASK, the producer's z of
//     tile t) are issued before the MFMA loop of tile t, so their latency hides under compute -- one barrier per tile;
//   * BN statistics accumulate in registers across all tiles of the block; ONE partial row per block.
// grid (nblk, ceil(Mout/MB), 1); A.tiles = pixel tiles per image, A.total_tiles = B * A.tiles.
// ------------------------------------------------------------------------------------------------------------------
template <int SHAPE, int KH, int AMODE, int EPI, int TH, int MB, int WN, int KCP, typename AT>
__global__ __launch_bounds__(kBlock) void conv_igemm_p_k(const IgemmArgs A) {
    using S = MfmaShape<SHAPE>;
    constexpr int MT = SHAPE, NT = SHAPE, KS = S::KS, ACC = S::ACC, QUADS = S::QUADS;
    constexpr int TW = 32, NPR = TW / NT, NTB = TH * NPR, MTB = MB / MT, WM = 4 / WN;
    constexpr int NTW = NTB / WN, MTW = MTB / WM, TAPS = KH * KH;
    static_assert(NTB % WN == 0 && MTB % WM == 0 && NTW >= 1 && MTW >= 1, "bad wave arrangement");
    constexpr int IH = AMODE == A_NORMAL ? TH + KH - 1 : (AMODE == A_UPF ? TH / 2 + 1 : 2 * TH + 1);
    constexpr int IW = AMODE == A_NORMAL ? TW + KH - 1 : (AMODE == A_UPF ? TW / 2 + 1 : 2 * TW + 1);
    constexpr int PLANE = ((IH * IW + 15) / 32) * 32 + 16;
    constexpr int NPF = ((KCP / 4) * IH * IW + kBlock - 1) / kBlock;     // prefetched float4 per thread
    constexpr int RED = 4 * 2 * MTW * MT;

    __shared__ float Is[2][KCP * PLANE];
    __shared__ float Ws[TAPS * KCP * MB];
    __shared__ float red[RED];
    __shared__ float epi[4 * MB];          // EPI_MASK: the producer's BN record rows a / b / mean / rstd of this M block

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int j = lane & (NT - 1), kk = lane / NT, h = kk;
    const int m0 = blockIdx.y * MB;
    const int nks = (A.Cin < KCP ? A.Cin : KCP) / KS;

    for (int e = tid; e < TAPS * KCP * (MB / 4); e += kBlock) {          // weights: once per block
        const int m4 = e % (MB / 4), r = e / (MB / 4), kc = r % KCP, tap = r / KCP, m = m0 + 4 * m4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kc < A.Cin && m < A.Mout) v = ld4(A.w + ((size_t)tap * A.Cin + kc) * A.w_ld + A.m_off + m);
        st4(Ws + (tap * KCP + kc) * MB + 4 * m4, v);
    }
    if constexpr (EPI == EPI_MASK) {        // once per persistent block; read from LDS in every tile's epilogue (a global
        for (int e = tid; e < 4 * MB; e += kBlock) {     // load there would sit behind the previous, possibly aliasing, store)
            const int arr = e / MB, m = m0 + e % MB;
            epi[e] = m < A.Mout ? A.bnin[arr * A.Mout + m] : 0.f;
        }
    }
    float4 bias[MTW][QUADS];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int q = 0; q < QUADS; ++q) {
            const int m = m0 + (wm * MTW + mt) * MT + S::quad_base(q, h);
            bias[mt][q] = (EPI == EPI_FWD && m < A.Mout) ? ld4(A.bias + A.m_off + m) : make_float4(0.f, 0.f, 0.f, 0.f);
        }

    // staging and tile walk shared with the 8-output-channel kernels: per-thread invariants set up once, interior
    // tiles skip bounds tests, XCD-aware division-free walk (ThinStager / TileWalk above)
    ThinStager<KCP, IH, IW, IW, PLANE, AMODE, TH, TW, AT> st;
    st.init(A, Is[0]);
    TileWalk<TH, TW> walk;
    walk.init(A.tiles, A.tiles_x, A.total_tiles);
    typename decltype(st)::Buf pf;

    int boff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
        boff[nt] = AMODE == A_NORMAL ? r * IW + xs : (AMODE == A_UPF ? 0 : (2 * r) * IW + 2 * xs);
    }
    float s1[MTW][ACC], s2[MTW][ACC];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < ACC; ++r) { s1[mt][r] = 0.f; s2[mt][r] = 0.f; }

    TileOrg cur = walk.first(A.tiles);
    if (walk.tl0 < walk.tlend) st.load(A, cur, pf);
    int buf = 0;
    for (int tl = walk.tl0; tl < walk.tlend; tl += walk.step, buf ^= 1) {
        const float* Ib = Is[buf];
        st.store(A, cur, pf, buf * (KCP * PLANE));
        __syncthreads();            // tile image (and, first time, the weights) visible; buffer buf^1 is free again
        const TileOrg nxt = walk.next(cur);
        if (tl + walk.step < walk.tlend) st.load(A, nxt, pf);
        const int b = cur.b, y0 = cur.ty * TH, x0 = cur.tx * TW;
        cur = nxt;

        // producer's z for the epilogue mask: issued now, consumed after the MFMA loop
        typename Raw4<AT>::type zq[NTW][MTW][QUADS];     // raw: widened after the sweep
        if constexpr (EPI == EPI_MASK) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
                const int y = y0 + r, x = x0 + xs;
                const bool pvalid = y < A.Ho && x < A.Wo;
                const size_t pix = pvalid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int q = 0; q < QUADS; ++q) {
                        const int m = m0 + (wm * MTW + mt) * MT + S::quad_base(q, h);
                        zq[nt][mt][q] = (pvalid && m < A.Mout) ? ldraw4<AT>(reinterpret_cast<const AT*>(A.zin) + pix * A.Mout + m) : raw_zero4<AT>();
                    }
            }
        }

        typename S::acc_t acc[MTW][NTW];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < ACC; ++r) acc[mt][nt][r] = 0.f;
        if (nks == KCP / KS) mfma_sweep<SHAPE, KH, AMODE, KCP, MB, IW, PLANE, NPR, MTW, NTW, true>(Ws, Ib, acc, boff, wn, wm, j, kk, nks);
        else mfma_sweep<SHAPE, KH, AMODE, KCP, MB, IW, PLANE, NPR, MTW, NTW, false>(Ws, Ib, acc, boff, wn, wm, j, kk, nks);

#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int t = wn * NTW + nt, r = t / NPR, xs = (t % NPR) * NT + j;
            const int y = y0 + r, x = x0 + xs;
            const bool pvalid = y < A.Ho && x < A.Wo;
            const size_t pix = pvalid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
                for (int q = 0; q < QUADS; ++q) {
                    const int m = m0 + (wm * MTW + mt) * MT + S::quad_base(q, h);
                    const bool valid = pvalid && m < A.Mout;
                    float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                    if constexpr (EPI == EPI_FWD) {
                        v[0] += bias[mt][q].x; v[1] += bias[mt][q].y; v[2] += bias[mt][q].z; v[3] += bias[mt][q].w;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float u = valid ? v[k] : 0.f;
                            s1[mt][4 * q + k] += u; s2[mt][4 * q + k] += u * u;
                        }
                    } else if constexpr (EPI == EPI_MASK) {
                        const float4 zw = widen4(zq[nt][mt][q]);
                        const float zz[4] = {zw.x, zw.y, zw.z, zw.w};
                        const int ml = m - m0;
                        const float4 ea = ld4(epi + BN_A * MB + ml), eb = ld4(epi + BN_B * MB + ml);
                        const float4 em = ld4(epi + BN_MEAN * MB + ml), er = ld4(epi + BN_RSTD * MB + ml);
                        const float ka[4] = {ea.x, ea.y, ea.z, ea.w}, kb[4] = {eb.x, eb.y, eb.z, eb.w};
                        const float km[4] = {em.x, em.y, em.z, em.w}, kr[4] = {er.x, er.y, er.z, er.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int c = valid ? m + k : 0;
                            const float yv = fmaf(ka[k], zz[k], kb[k]);
                            float gv = v[k];
                            if (A.drop_out) gv *= drop_mul(A.drop, (uint32_t)(pix * A.Mout + c));
                            gv = (valid && yv > 0.f) ? gv : 0.f;
                            const float xh = (zz[k] - km[k]) * kr[k];
                            v[k] = gv; s1[mt][4 * q + k] += gv; s2[mt][4 * q + k] += gv * xh;
                        }
                    }
                    if (valid) sta4<AT>(reinterpret_cast<AT*>(A.out) + pix * A.Mout + m, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
        }
    }

    if constexpr (EPI != EPI_RAW) {
        if (A.part) {
            constexpr int OFF0 = NT / 2;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                subgroup_reduce_rec<ACC, ACC, OFF0>(s1[mt], lane);
                subgroup_reduce_rec<ACC, ACC, OFF0>(s2[mt], lane);
                const int ci = sub_chan<ACC, OFF0>(lane);
                const int mloc = S::quad_base(ci >> 2, h) + (ci & 3);
                red[((wave * 2 + 0) * MTW + mt) * MT + mloc] = s1[mt][0];
                red[((wave * 2 + 1) * MTW + mt) * MT + mloc] = s2[mt][0];
            }
            __syncthreads();
            if (tid < 2 * MB) {
                const int stat = tid / MB, ml = tid % MB;
                const int mtile = ml / MT, wmm = mtile / MTW, mt = mtile % MTW, mloc = ml % MT;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) s += red[(((wmm * WN + w) * 2 + stat) * MTW + mt) * MT + mloc];
                if (m0 + ml < A.Mout)
                    A.part[(size_t)blockIdx.x * (2 * A.Mout) + (size_t)stat * A.Mout + m0 + ml] = s;
            }
        }
    }
}

// ---- per-step weight preparation for backward-data: transposed+flipped kernels, effective up-conv kernels ----
struct WtDesc {
    const float* w; float* wt;
    int kh, cin, cout, mode;     // mode 0: 3x3 -> WT[8-tap][co][ci];  mode 1: 2x2 -> Weff[a][b][co][ci] (3x3 stride-2 gather)
    unsigned start, count;       // element range of this layer in the flattened launch
};

static __global__ __launch_bounds__(kBlock) void prep_wt_k(const WtDesc* __restrict__ descs, int nd, unsigned total) {
    for (unsigned e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
        int d = 0;
        while (d + 1 < nd && e >= descs[d + 1].start) ++d;
        const WtDesc D = descs[d];
        const unsigned r = e - D.start;
        const int ci = r % D.cin, co = (r / D.cin) % D.cout, tap = r / (D.cin * D.cout);
        float v;
        if (D.mode == 0) {
            v = D.w[((size_t)(8 - tap) * D.cin + ci) * D.cout + co];
        } else {
            const int a = tap / 3, bq = tap % 3;
            v = 0.f;
            for (int ky = 0; ky < 2; ++ky) {
                const bool yin = (a == 0 && ky == 1) || a == 1 || (a == 2 && ky == 0);
                if (!yin) continue;
                for (int kx = 0; kx < 2; ++kx) {
                    const bool xin = (bq == 0 && kx == 1) || bq == 1 || (bq == 2 && kx == 0);
                    if (xin) v += D.w[((size_t)(ky * 2 + kx) * D.cin + ci) * D.cout + co];
                }
            }
        }
        D.wt[r] = v;
    }
}

}  // namespace oct
