// Implicit-GEMM convolution on the bf16 MFMA pipe of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for layers with
// >= 32 output channels -- forward convs and backward-data, the same addressing modes and epilogues as
// kernels_igemm.hpp (reference graph: models/unet.py:26-29,42-47).
//
// fp32 mode (NS = 3): every fp32 operand x is split EXACTLY into three bf16 terms x = x0 + x1 + x2 (round-to-nearest
// residuals: 3 x 8 significant bits = the 24 of an fp32) and a product a*b is formed from the six bf16 products
//     a0*b0 + a0*b1 + a1*b0 + a0*b2 + a1*b1 + a2*b0          (dropped: a1*b2, a2*b1, a2*b2 <= 2^-25 |a*b|)
// each of which the matrix core computes exactly and adds into the fp32 accumulator.  The result carries the same
// ~2^-24 relative error per product as an fp32 FMA chain, at 6 bf16 MFMAs (6 x 32 cycles for 32x32x16) instead of
// 8 fp32 MFMAs (8 x 64 cycles for the same 32x32x16 block): 2.67x the fp32-pipe rate.
// bf16 mode (NS = 1, BASELINE configs[2]): activations (after BN + ReLU) and weights are rounded once to bf16 and
// multiplied directly; accumulation, BN statistics and every epilogue stay fp32.
//
// Data movement:
//  * weights are split / rounded ONCE PER STEP by prep_wbx_k into MFMA A-operand order
//    [K chunk of 16][M block][tap row][term][tap col][m][16 k] (bf16), so that one (chunk, M block, tap row) slab is a
//    contiguous run that waves copy global -> LDS with global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip);
//    two slab slots: the DMA of slab g+1 is in flight while slab g is consumed;
//  * the input tile of a K chunk goes global -> registers (prefetched one chunk ahead) -> BN + ReLU (+ dropout, concat)
//    -> split -> LDS image [term][pixel][16 k] (bf16, 32 B per pixel: one ds_read_b128 per B fragment), double
//    buffered: chunk c+1 is converted and written while the last tap row of chunk c is multiplied;
//  * one barrier per tap row.  Block = 256 threads = 4 waves, one per SIMD (the tile sizes need most of the LDS);
//    a wave owns ALL M tiles of the block and TH/4 pixel rows, so A fragments are reused across its pixel tiles and
//    B fragments across its channel tiles.
#pragma once
#include <type_traits>

#include "kernels_bwd.hpp"
#include "kernels_igemm.hpp"

namespace oct {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_hw{a, b}, bf16x2_hw));
}

// "use" a prefetched raw register set without doing anything: pins the compiler's s_waitcnt for those loads HERE
__device__ __forceinline__ void touch_raw(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void touch_raw(uint2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }

// consumer-side transform of 8 staged values: y = max(fa * v + fb, lo) inside the image, exactly 0 outside (zero padding
// is applied AFTER the activation).  The affine goes as v_pk_fma_f32 (two channels per instruction), ReLU and padding as
// ONE v_med3_f32 per element: clamp to [lo, +big) inside, to [0, 0] outside.
__device__ __forceinline__ void act8(float (&v)[8], const float (&fa)[8], const float (&fb)[8], float lo, bool in) {
    const float cl = in ? lo : 0.f, ch = in ? 3.0e38f : 0.f;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_hw y = __builtin_elementwise_fma(f32x2_hw{fa[i], fa[i + 1]}, f32x2_hw{v[i], v[i + 1]}, f32x2_hw{fb[i], fb[i + 1]});
        v[i] = __builtin_amdgcn_fmed3f(y[0], cl, ch); v[i + 1] = __builtin_amdgcn_fmed3f(y[1], cl, ch);
    }
}

// BN-backward transform on load (common.hpp): v = g' (8 channels of one pixel), z = the layer's raw output there;
// v <- dz = ga g' + (gb z + gd) inside the image -- bn_bwd_apply1's two fmas as v_pk_fma_f32 pairs, rounded as the
// stand-alone pass would have stored it -- and exactly 0 outside (zero padding of dz)
template <typename AT>
__device__ __forceinline__ void gb8(float (&v)[8], const float (&z)[8], const float (&ga)[8], const float (&gb)[8], const float (&gd)[8], bool in) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_hw t = __builtin_elementwise_fma(f32x2_hw{gb[i], gb[i + 1]}, f32x2_hw{z[i], z[i + 1]}, f32x2_hw{gd[i], gd[i + 1]});
        const f32x2_hw y = __builtin_elementwise_fma(f32x2_hw{ga[i], ga[i + 1]}, f32x2_hw{v[i], v[i + 1]}, t);
        v[i] = in ? dz_as_stored<AT>(y[0]) : 0.f; v[i + 1] = in ? dz_as_stored<AT>(y[1]) : 0.f;
    }
}
__device__ __forceinline__ void load_gb8(const float* __restrict__ bn, int C, int c, float (&ga)[8], float (&gb)[8], float (&gd)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { ga[i] = bn[BN_GA * C + c + i]; gb[i] = bn[BN_GB * C + c + i]; gd[i] = bn[BN_GD * C + c + i]; }
}

// 8 floats -> NS planes of 8 bf16 (a uint4 each); planes 1 and 2 hold the exact residuals
template <int NS>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&pl)[NS]) {
    uint32_t w[NS][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const uint32_t p0 = pk_bf16(a, b);
        w[0][i] = p0;
        if constexpr (NS > 1) {
            const float a1 = a - __uint_as_float(p0 << 16), b1 = b - __uint_as_float(p0 & 0xFFFF0000u);
            const uint32_t p1 = pk_bf16(a1, b1);
            w[1][i] = p1;
            if constexpr (NS > 2) {
                const float a2 = a1 - __uint_as_float(p1 << 16), b2 = b1 - __uint_as_float(p1 & 0xFFFF0000u);
                w[2][i] = pk_bf16(a2, b2);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NS; ++p) pl[p] = make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]);
}

// ---- per-step weight preparation: fp32 [tap][k][m] -> split bf16 slabs in MFMA A-operand order --------------------
struct WbxDesc {
    const float* src; bf16_t* dst;
    int KH, Kc, M, ld;       // src[(tap * Kc + k) * ld + m], k < Kc, m < M
    int MB, NS;
    unsigned start, count;   // work items (one per (chunk, mblk, ky, kx, m, k-half)) of this entry in the flattened launch
};

static __global__ __launch_bounds__(kBlock) void prep_wbx_k(const WbxDesc* __restrict__ descs, int nd, unsigned total) {
    for (unsigned e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
        int d = 0;
        while (d + 1 < nd && e >= descs[d + 1].start) ++d;
        const WbxDesc D = descs[d];
        unsigned r = e - D.start;
        const int kh = r & 1; r >>= 1;
        const int m = r % D.MB; r /= D.MB;
        const int kx = r % D.KH; r /= D.KH;
        const int ky = r % D.KH; r /= D.KH;
        const int nmb = (D.M + D.MB - 1) / D.MB;
        const int mb = r % nmb, chunk = r / nmb;
        const int mg = mb * D.MB + m, tap = ky * D.KH + kx;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = chunk * 16 + kh * 8 + j;
            v[j] = (k < D.Kc && mg < D.M) ? D.src[((size_t)tap * D.Kc + k) * D.ld + mg] : 0.f;
        }
        const size_t slab = ((size_t)chunk * nmb + mb) * D.KH + ky;                 // (chunk, M block, tap row)
        bf16_t* base = D.dst + slab * ((size_t)D.NS * D.KH * D.MB * 16);
        const int khs = kh ^ ((m >> 3) & 1);          // bank swizzle of the 32-byte rows (see conv_bx_k)
        if (D.NS == 3) {
            uint4 pl[3]; split8<3>(v, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint4*>(base + (((size_t)p * D.KH + kx) * D.MB + m) * 16 + khs * 8) = pl[p];
        } else {
            uint4 pl[1]; split8<1>(v, pl);
            *reinterpret_cast<uint4*>(base + ((size_t)kx * D.MB + m) * 16 + khs * 8) = pl[0];
        }
    }
}

// bytes of the split weights of one layer direction
__host__ inline size_t wbx_bytes(int KH, int Kc, int M, int MB, int NS) {
    return (size_t)((Kc + 15) / 16) * ((M + MB - 1) / MB) * KH * NS * KH * MB * 16 * 2;
}

// ------------------------------------------------------------------------------------------------------------------
// grid (tiles, ceil(Mout/MB), B), block 256.  A.wbx = split weights of this layer direction (prep_wbx_k layout for MB),
// A.m_off must be a multiple of MB.  DROP: the input passes through dropout (the bottleneck output feeding dec0.up).
//
// LDS bank conflicts: a ds_read_b128 is served in groups of 16 lanes over 64 banks; with a 32-byte row pitch rows r and
// r + 8 (and r + 24) of a group share a 16-byte slot.  Both images therefore store the 8-channel half hh of row r at
// half-slot hh ^ ((r >> 3) & 1) -- prep_wbx_k swizzles the weights the same way -- which separates every such pair for
// any tap shift of the pixel rows: conflict-free fragment reads.
// ------------------------------------------------------------------------------------------------------------------
// NW = waves per block: 4 (one per SIMD) or 8 (two per SIMD: the conversion / LDS phases of one wave run under the MFMAs
// of its SIMD partner; each wave then owns TH / 8 pixel rows).
// GB (backward-data launches): the input x0 is the masked gradient g' of the layer whose dX this is; the BN-backward
// transform dz = ga g' + gb z + gd (A.gb_z, A.gb_bn; common.hpp) is applied while the tile is staged.
// NIMG = 1: ONE input image instead of the double buffer -- the block then fits twice per CU (4-wave blocks): the
// conversion of the next K chunk is no longer hidden behind this block's own MFMAs but behind the OTHER resident block's,
// and so are this block's prologue, barriers and epilogue (two independent instruction streams per SIMD).
template <int KH, int AMODE, int EPI, int TH, int MB, int NS, bool DROP, int NW, typename AT, bool GB = false, int NIMG = 2>
__global__ __launch_bounds__(64 * NW, NIMG == 1 ? (NW == 4 ? 2 : 1) : 1) void conv_bx_k(const IgemmArgs A) {
    constexpr int NTHR = 64 * NW;
    constexpr int TW = 32, MTW = MB / 32, NTW = TH / NW, ACC = 16, QUADS = 4, MT = 32;
    static_assert(TH % NW == 0 && NTW >= 1 && (MB == 32 || MB == 64), "tile geometry");
    constexpr int IH = AMODE == A_NORMAL ? TH + KH - 1 : (AMODE == A_UPF ? TH / 2 + 1 : 2 * TH + 1);
    constexpr int IW = AMODE == A_NORMAL ? TW + KH - 1 : (AMODE == A_UPF ? TW / 2 + 1 : 2 * TW + 1);
    constexpr int NPIX = IH * IW;
    constexpr int NSLOT = (NPIX * 2 + NTHR - 1) / NTHR;    // staging items (pixel, 8-channel half) per thread
    constexpr int NPIXP = NPIX + 64;                           // + a 64-pixel dump strip: items past the tile land there, so
                                                               // the conversion is branch-free (schedulable among MFMAs)
    constexpr int PLANE_B = NPIXP * 32;                        // bytes of one term's image
    constexpr int IN_B = NS * PLANE_B;                         // one buffer
    constexpr int SLAB_B = NS * KH * MB * 32;                  // one (chunk, M block, tap row) weight slab
    constexpr int PIECES = SLAB_B / 1024;                      // 1 KiB per LDS-DMA wave instruction
    static_assert(SLAB_B % 1024 == 0, "slab must be a whole number of DMA pieces");
    constexpr int MAXC = AMODE == A_DOWN2 ? (GB ? kBxGbDown2MaxC : 256) : 512;   // affine rows cached in LDS (K channels; launcher checks)
    constexpr int EPI_B = EPI == EPI_MASK ? 4 * MB * 4 : MB * 4;
    constexpr int RED_B = NW * 2 * MTW * MT * 4;
    constexpr int SCRATCH_B = (EPI_B + RED_B) > 2 * SLAB_B ? (EPI_B + RED_B) : 2 * SLAB_B;

    static_assert(!GB || (EPI != EPI_FWD && !DROP), "the BN-backward transform on load belongs to backward-data launches");
    static_assert(NIMG == 1 || NIMG == 2, "one input image or a double buffer");
    __shared__ __attribute__((aligned(1024))) char smem[NIMG * IN_B + SCRATCH_B + (GB ? 3 : 2) * MAXC * 4];
    char* const INs = smem;
    char* const WTs = smem + NIMG * IN_B;
    float* const ABs = reinterpret_cast<float*>(smem + NIMG * IN_B + SCRATCH_B);     // [2][MAXC]: a row, b row (GB: ga, gd, gb rows)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, b = blockIdx.z, m0 = blockIdx.y * MB;
    const int x0 = (tile % A.tiles_x) * TW, y0 = (tile / A.tiles_x) * TH;
    const int iy0 = AMODE == A_NORMAL ? y0 - (KH - 1) / 2 : (AMODE == A_UPF ? y0 / 2 : 2 * y0 - 1);
    const int ix0 = AMODE == A_NORMAL ? x0 - (KH - 1) / 2 : (AMODE == A_UPF ? x0 / 2 : 2 * x0 - 1);
    const int nch = (A.Cin + 15) / 16;
    const int nmb = (A.wbx_M + MB - 1) / MB, mblk = (A.m_off + m0) / MB;

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < ACC; ++r) acc[mt][nt][r] = 0.f;

    // ---- weight slabs by LDS-DMA: slab (chunk, ky) -> slot; wave w copies pieces w, w+4, ... ----
    const char* const wsrc = reinterpret_cast<const char*>(A.wbx);
    auto dma_slab = [&](int chunk, int ky, int slot) {
        const char* g = wsrc + (((size_t)chunk * nmb + mblk) * KH + ky) * SLAB_B + lane * 16;
        char* l = WTs + slot * SLAB_B;
#pragma unroll
        for (int p = 0; p < (PIECES + NW - 1) / NW; ++p) {
            const int piece = wave + NW * p;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024),
                                                 (__attribute__((address_space(3))) void*)(l + piece * 1024), 16, 0, 0);
        }
    };

    // ---- input staging: item = (tile pixel P, channel half hh); geometry computed once ----
    int goff[NSLOT], ldst[NSLOT];          // source pixel offset in the image (-1: zero); LDS byte offset of the item
    const int hh = tid & 1;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int P = (tid >> 1) + k * (NTHR / 2), lx = P % IW, ly = P / IW, gy = iy0 + ly, gx = ix0 + lx;
        const bool in = P < NPIX && gy >= 0 && gy < A.Hi && gx >= 0 && gx < A.Wi;
        goff[k] = in ? gy * A.Wi + gx : -1;
        const int Pd = P < NPIX ? P : NPIX + (P & 63);            // items past the tile: spread over the dump strip
        ldst[k] = Pd * 32 + ((hh ^ ((Pd >> 3) & 1)) * 16);
    }
    const size_t img = (size_t)b * A.Hi * A.Wi;
    // affine rows of the (possibly concatenated) input, once per block: (a, b) of y = max(a*z + b, lo); identity without BN
    const bool aff = (A.flags & F_AFF) != 0;
    const float lo = aff ? 0.f : -3.0e38f;
    // (the rows themselves are fetched in the pipeline prologue, BEHIND the first tile's loads: one latency, not two)
    typename Raw4<AT>::type pin[NSLOT][2], pz[GB ? NSLOT : 1][2];
    auto load_in = [&](int c0) {
        const int c = c0 + 8 * hh;
        const bool two = (A.flags & F_TWO) && c >= A.C0;
        const int C = two ? A.C1 : A.C0;
        const AT* __restrict__ src = (two ? reinterpret_cast<const AT*>(A.x1) + (c - A.C0) : reinterpret_cast<const AT*>(A.x0) + c) + img * C;
        const bool cok = c < A.Cin;           // channel counts are multiples of 8 on this path
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const bool ok = cok && goff[k] >= 0;
            const AT* p = src + (size_t)(ok ? goff[k] : 0) * C;
            pin[k][0] = ok ? ldraw4<AT>(p) : raw_zero4<AT>();
            pin[k][1] = ok ? ldraw4<AT>(p + 4) : raw_zero4<AT>();
            if constexpr (GB) {      // same element of the layer's z (same shape as g': one source, no concat)
                const AT* q = reinterpret_cast<const AT*>(A.gb_z) + c + img * C + (size_t)(ok ? goff[k] : 0) * C;
                pz[k][0] = ok ? ldraw4<AT>(q) : raw_zero4<AT>();
                pz[k][1] = ok ? ldraw4<AT>(q + 4) : raw_zero4<AT>();
            }
        }
    };
    // per-chunk constants of the conversion (set by begin_store, used by store_slot): branch-free so that the
    // conversion of slot k can be scheduled between the MFMAs of a tap
    float fa[8], fb[8], fz[GB ? 8 : 1]; bool cok_s = false; int el_c = 0, el_C = 1;
    auto begin_store = [&](int c0) {
        const int c = c0 + 8 * hh;
        cok_s = c < A.Cin;
        const float4 a0 = ld4(ABs + c), a1 = ld4(ABs + c + 4), b0 = ld4(ABs + MAXC + c), b1 = ld4(ABs + MAXC + c + 4);
        fa[0] = a0.x; fa[1] = a0.y; fa[2] = a0.z; fa[3] = a0.w; fa[4] = a1.x; fa[5] = a1.y; fa[6] = a1.z; fa[7] = a1.w;
        fb[0] = b0.x; fb[1] = b0.y; fb[2] = b0.z; fb[3] = b0.w; fb[4] = b1.x; fb[5] = b1.y; fb[6] = b1.z; fb[7] = b1.w;
        if constexpr (GB) {
            const float4 z0 = ld4(ABs + 2 * MAXC + c), z1 = ld4(ABs + 2 * MAXC + c + 4);
            fz[0] = z0.x; fz[1] = z0.y; fz[2] = z0.z; fz[3] = z0.w; fz[4] = z1.x; fz[5] = z1.y; fz[6] = z1.z; fz[7] = z1.w;
        }
        if constexpr (DROP) {
            const bool two = (A.flags & F_TWO) && c >= A.C0;
            el_C = two ? A.C1 : A.C0; el_c = two ? c - A.C0 : c;
        }
    };
    auto store_slot = [&](int k, int buf) {
        const float4 v0 = widen4(pin[k][0]), v1 = widen4(pin[k][1]);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const bool in = cok_s && goff[k] >= 0;
        if constexpr (GB) {
            const float4 z0 = widen4(pz[k][0]), z1 = widen4(pz[k][1]);
            const float zv[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)      // bn_bwd_apply1's two fmas (fa = ga, fz = gb, fb = gd); zero padding of dz
                v[i] = in ? dz_as_stored<AT>(fmaf(fa[i], v[i], fmaf(fz[i], zv[i], fb[i]))) : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {        // zero padding is applied AFTER the activation: out-of-image stays 0
                const float y = fmaxf(fmaf(fa[i], v[i], fb[i]), lo);     // (act8's med3 / pk_fma form measured slower HERE: the
                v[i] = in ? y : 0.f;                                      //  compiler's schedule among this kernel's MFMAs changes)
            }
        }
        if constexpr (DROP) {
            if (in) {
                const uint32_t el = (uint32_t)((img + goff[k]) * el_C + el_c);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= drop_mul(A.drop, el + i);
            }
        }
        uint4 pl[NS];
        split8<NS>(v, pl);
        char* d = INs + buf * IN_B + ldst[k];
#pragma unroll
        for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * PLANE_B) = pl[p];
    };

    // per-lane pixel offsets (in pixels of the tile image) of the B operand for each of this wave's pixel rows
    int boff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int r = wave * NTW + nt;
        boff[nt] = AMODE == A_NORMAL ? r * IW + j : (AMODE == A_UPF ? 0 : (2 * r) * IW + 2 * j);
    }
    const int a_lane = j * 32 + ((h ^ ((j >> 3) & 1)) * 16);    // A fragment: row j of an M tile, swizzled half

    struct Frag { bf16x8 a[MTW][NS], b[NTW][NS]; };
    // ---- one tap row: KH taps x MTW x NTW output tiles x (6 | 1) bf16 MFMAs; the fragments of tap kx+1 are read while
    // the MFMAs of tap kx issue; with STORE the next chunk's input slots are converted in the MFMAs' shadow ----
    auto sweep_row = [&](int ky, int slot, int buf, auto do_store, int sbuf) {
        const char* Wb = WTs + slot * SLAB_B + a_lane;
        const char* Ib = INs + buf * IN_B;
        auto load_frag = [&](int kx, Frag& f) {
#if defined(BX_DBG) && BX_DBG == 3      // timing experiment: no LDS fragment reads
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int p = 0; p < NS; ++p) asm volatile("" : "=v"(f.a[mt][p]));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) asm volatile("" : "=v"(f.b[nt][p]));
            return;
#endif
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    f.a[mt][p] = *reinterpret_cast<const bf16x8*>(Wb + ((p * KH + kx) * MB + mt * 32) * 32);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                int off;
                if constexpr (AMODE == A_UPF) {
                    const int r = wave * NTW + nt;
                    off = ((r + ky) >> 1) * IW + ((j + kx) >> 1);
                } else {
                    off = boff[nt] + ky * IW + kx;
                }
                const char* q = Ib + off * 32 + ((h ^ ((off >> 3) & 1)) * 16);
#pragma unroll
                for (int p = 0; p < NS; ++p) f.b[nt][p] = *reinterpret_cast<const bf16x8*>(q + p * PLANE_B);
            }
        };
        auto mfma_tap = [&](const Frag& f) {
#if defined(BX_DBG) && BX_DBG == 2      // timing experiment: no MFMAs (fragments kept alive)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int p = 0; p < NS; ++p) asm volatile("" :: "v"(f.a[mt][p]));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < NS; ++p) asm volatile("" :: "v"(f.b[nt][p]));
            return;
#endif
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    f32x16 c = acc[mt][nt];
                    if constexpr (NS == 3) {     // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][0], f.b[nt][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][2], f.b[nt][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][1], f.b[nt][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][0], f.b[nt][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][1], f.b[nt][0], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mt][0], f.b[nt][0], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
        };
        Frag f[2];
        load_frag(0, f[0]);
#pragma unroll
        for (int kx = 0; kx < KH; ++kx) {
            if (kx + 1 < KH) load_frag(kx + 1, f[(kx + 1) & 1]);
            mfma_tap(f[kx & 1]);
#if !(defined(BX_DBG) && BX_DBG == 4)      // (4: timing experiment without the in-loop conversions)
            if constexpr (decltype(do_store)::value) {
#pragma unroll
                for (int k = kx; k < NSLOT; k += KH) store_slot(k, sbuf);
            }
#endif
        }
#if !defined(BX_NOPIN)
        // Pin the order: left alone, hipcc sinks every fragment read to just in front of its first MFMA and waits
        // lgkmcnt(0) there -- an LDS round trip in front of every one to six MFMAs.  Tap 0's fragments first; then the reads
        // of tap kx+1 go between the MFMAs of tap kx, one group per MFMA.
        if constexpr (!decltype(do_store)::value) {
            constexpr int RD = (MTW + NTW) * NS, MF = MTW * NTW * (NS == 3 ? 6 : 1);
            __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
            for (int kx = 0; kx < KH; ++kx) {
                if (kx + 1 < KH) {
#pragma unroll
                    for (int i = 0; i < MF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < RD % MF) __builtin_amdgcn_sched_group_barrier(0x100, RD / MF + 1, 0);
                        else if (RD / MF > 0) __builtin_amdgcn_sched_group_barrier(0x100, RD / MF, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
                }
            }
        }
#endif
    };

    // ---- pipeline over (chunk, tap row) ----
    dma_slab(0, 0, 0);
    load_in(0);
    for (int c = tid; c < nch * 16; c += NTHR) {
        float av = 1.f, bv = 0.f;
        if (aff && c < A.Cin) {
            const bool two = (A.flags & F_TWO) && c >= A.C0;
            const float* ab = two ? A.ab1 : A.ab0; const int C = two ? A.C1 : A.C0, cc = two ? c - A.C0 : c;
            av = ab[cc]; bv = ab[C + cc];
        }
        if constexpr (GB) {
            const bool ok = c < A.Cin;
            av = ok ? A.gb_bn[BN_GA * A.Cin + c] : 0.f; bv = ok ? A.gb_bn[BN_GD * A.Cin + c] : 0.f;
            ABs[2 * MAXC + c] = ok ? A.gb_bn[BN_GB * A.Cin + c] : 0.f;
        }
        ABs[c] = av; ABs[MAXC + c] = bv;
    }
    __syncthreads();                 // ABs visible
    begin_store(0);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) store_slot(k, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // image 0 + slab (0,0) landed
    // EPI_MASK: the producer's z at this wave's output pixels (mask + BN-backward sums of the epilogue) is requested
    // before the LAST tap row of the last K chunk -- the input prefetch registers are dead by then -- instead of behind it
    typename Raw4<AT>::type zraw[EPI == EPI_MASK ? NTW : 1][EPI == EPI_MASK ? MTW : 1][EPI == EPI_MASK ? QUADS : 1];
    auto load_zraw = [&]() {
        if constexpr (EPI == EPI_MASK) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int y = y0 + wave * NTW + nt, x = x0 + j;
                const bool pvalid = y < A.Ho && x < A.Wo;
                const size_t pix = pvalid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int q = 0; q < QUADS; ++q) {
                        const int m = m0 + mt * MT + 8 * q + 4 * h;
                        zraw[nt][mt][q] = (pvalid && m < A.Mout) ? ldraw4<AT>(reinterpret_cast<const AT*>(A.zin) + pix * A.Mout + m) : raw_zero4<AT>();
                    }
            }
        }
    };
    // ... and so are the epilogue's per-channel constants (bias, or the producer's BN rows): one value per thread, kept in a
    // register until the weight slots they will be staged in are free
    constexpr int EPN = EPI == EPI_MASK ? 4 * MB : (EPI == EPI_FWD ? MB : 0), EPF = (EPN + NTHR - 1) / NTHR;
    float epi_pf[EPF > 0 ? EPF : 1];
    auto load_epi = [&]() {
#pragma unroll
        for (int i = 0; i < EPF; ++i) {
            const int e = tid + i * NTHR;
            float v = 0.f;
            if constexpr (EPI == EPI_FWD) { if (e < MB && m0 + e < A.Mout) v = A.bias[A.m_off + m0 + e]; }
            else if constexpr (EPI == EPI_MASK) { if (e < 4 * MB && m0 + e % MB < A.Mout) v = A.bnin[(e / MB) * A.Mout + m0 + e % MB]; }
            epi_pf[i] = v;
        }
    };
    int g = 0;
#if defined(BX_DBG) && (BX_DBG == 6 || BX_DBG == 7)      // timing experiments: prologue + epilogue only (6), prologue only (7)
    const int nch_run = 0;
    if constexpr (NW == 4) load_zraw();
    load_epi();
#if BX_DBG == 7
    if (A.Cin > 0) return;
#endif
#else
    const int nch_run = nch;
#endif
    for (int c = 0; c < nch_run; ++c) {
#pragma unroll
        for (int ky = 0; ky < KH; ++ky, ++g) {
            const bool last_row = ky == KH - 1, more = c + 1 < nch;
            if (last_row && !more) {
                if constexpr (NW == 4) load_zraw();      // (8-wave blocks live under the 256-register cap: there the 8 extra
                load_epi();                              //  float4 spill 100 registers -- their z is requested after the loop)
            }
            if (last_row && more) {
                // The conversion below consumes the input registers requested two tap rows ago.  hipcc waits vmcnt(0)
                // at their first use -- which would also wait for the weight DMA issued in THIS step (a full L2 round
                // trip in front of the step's MFMAs).  Touch the registers first: the wait lands here, where nothing
                // younger is in flight, and the DMA is issued behind it.
#pragma unroll
                for (int k = 0; k < NSLOT; ++k) {
                    touch_raw(pin[k][0]); touch_raw(pin[k][1]);
                    if constexpr (GB) { touch_raw(pz[k][0]); touch_raw(pz[k][1]); }
                }
                if constexpr (NIMG == 2) begin_store((c + 1) * 16);     // (one image: after the sweep -- 16-24 registers less in it)
            }
#if !(defined(BX_DBG) && BX_DBG == 5)      // (5: timing experiment without the in-loop weight DMA)
            if (!last_row) dma_slab(c, ky + 1, (g + 1) & 1);
            else if (more) dma_slab(c + 1, 0, (g + 1) & 1);
#endif
            if (ky == 0 && more) load_in((c + 1) * 16);
            if constexpr (NIMG == 2) {
                if (last_row && more) sweep_row(ky, g & 1, c & 1, std::true_type{}, (c + 1) & 1);
                else sweep_row(ky, g & 1, c & 1, std::false_type{}, 0);
            } else {
                sweep_row(ky, g & 1, 0, std::false_type{}, 0);
                if (last_row && more) {          // every wave is done with the image: overwrite it with the next K chunk
                    begin_store((c + 1) * 16);
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < NSLOT; ++k) store_slot(k, 0);
                }
            }
            // an LDS-DMA becomes visible to other waves' ds_reads only through the issuing wave's vmcnt wait followed by
            // a barrier: drain explicitly (hipcc also does before __syncthreads() while a DMA is in flight)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(defined(BX_DBG) && BX_DBG == 1)      // (1: timing experiment without the per-step barrier)
            __syncthreads();
#endif
        }
    }
#if defined(BX_DBG) && BX_DBG == 1
    __syncthreads();
#endif

    // ---- epilogue (same forms as conv_igemm_k; the weight slots are free now) ----
    float* const epi = reinterpret_cast<float*>(WTs);
    float* const red = reinterpret_cast<float*>(WTs + EPI_B);
    if constexpr (NW != 4) load_zraw();
    if constexpr (EPI != EPI_RAW) {
#pragma unroll
        for (int i = 0; i < EPF; ++i) { const int e = tid + i * NTHR; if (e < EPN) epi[e] = epi_pf[i]; }     // (requested before the last tap row)
    }
    __syncthreads();
    float s1[MTW][ACC], s2[MTW][ACC];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < ACC; ++r) { s1[mt][r] = 0.f; s2[mt][r] = 0.f; }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int y = y0 + wave * NTW + nt, x = x0 + j;
        const bool pvalid = y < A.Ho && x < A.Wo;
        const size_t pix = pvalid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
            for (int q = 0; q < QUADS; ++q) {
                const int ml = mt * MT + 8 * q + 4 * h, m = m0 + ml;
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
                        const int cc = valid ? m + k : 0;
                        const float yv = fmaf(ka[k], zz[k], kb[k]);
                        float gv = v[k];
                        if (A.drop_out) gv *= drop_mul(A.drop, (uint32_t)(pix * A.Mout + cc));
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
            // lanes sharing h hold the same 16 channels of an M tile for 32 different pixels: reduce over the pixel lanes
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                subgroup_reduce_rec<ACC, ACC, 16>(s1[mt], lane);
                subgroup_reduce_rec<ACC, ACC, 16>(s2[mt], lane);
                const int ci = sub_chan<ACC, 16>(lane);
                const int mloc = 8 * (ci >> 2) + 4 * h + (ci & 3);
                red[((wave * 2 + 0) * MTW + mt) * MT + mloc] = s1[mt][0];
                red[((wave * 2 + 1) * MTW + mt) * MT + mloc] = s2[mt][0];
            }
            __syncthreads();
            if (tid < 2 * MB) {
                const int stat = tid / MB, ml = tid % MB, mt = ml / MT, mloc = ml % MT;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) s += red[((w * 2 + stat) * MTW + mt) * MT + mloc];
                if (m0 + ml < A.Mout)
                    A.part[((size_t)b * A.tiles + tile) * (2 * A.Mout) + (size_t)stat * A.Mout + m0 + ml] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward-weights on the bf16 pipe:  dW[tap][ci][co] = sum_pixels X(pixel + tap)[ci] * dz(pixel)[co]  for layers with
// >= 32 input and >= 32 output channels (reference: the gradient of models/unet.py:26-29 that Keras' autodiff forms).
//   MFMA rows (M) = 32 input channels, columns (N) = 32 output channels, K = 16 consecutive pixels of an image row.
//   Both operands want K (pixels) contiguous per lane while the natural images are channel-fastest:
//   ds_read_b64_tr_b16 (transposing LDS read, 4 pixels x 16 channels per 16 lanes) delivers exactly that from the
//   NHWC images [term][pixel][32 channels] (64 B per pixel: the 32 lanes of a read cover 256 contiguous bytes --
//   conflict-free), for X at any tap offset because a tap shifts whole pixels.
//   X and dz are split into NS bf16 terms each while they are staged (6 products per fp32 product, see conv_bx_k).
// Block (256 threads, 1 per CU) = one (32 ci, 32 co) pair for a strided list of 4-row x 32-pixel tiles; wave w owns
// pixel row w of every tile and all taps (9 accumulator tiles); global loads run two tiles ahead in registers, the
// LDS images are double buffered (the split of tile t+1 is written while tile t is multiplied), one barrier per tile;
// a fixed-order 4-wave sum through LDS at the end writes ONE partial slab per block (reduce_all_k sums the slabs).
// UP: the conv input is the nearest-upsampled low-res tensor (2x2 up-conv): the VIRTUAL high-res tile is staged.
// grid (npb, Cin/32, Cout/32); A.tiles / A.tiles_x for 4 x 32 tiles.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// GB: A.dz is the masked gradient g' and the BN-backward transform is applied while it is staged (A.zf, A.bnf).
template <int KH, bool UP, int NS, typename AT, bool DROP = false, bool GB = false>
__global__ __launch_bounds__(kBlock, 1) void conv_dwbx_k(const ConvBwdWArgs A) {
    constexpr int TH = 4, TW = 32, TAPS = KH * KH;
    constexpr int IH = UP ? TH + 1 : TH + KH - 1, IW = UP ? TW + 1 : TW + KH - 1, PT = UP ? 0 : (KH - 1) / 2;
    constexpr int NPX = IH * IW, NPD = TH * TW;
    constexpr int NXS = (NPX * 4 + kBlock - 1) / kBlock, NDS = (NPD * 4) / kBlock;     // staging items per thread
    constexpr int NPXP = NXS * (kBlock / 4);                    // X image padded to whole slots (branch-free staging)
    constexpr int XPL = NPXP * 64, DPL = NPD * 64;              // bytes of one term's image (32 channels x bf16 per pixel)
    constexpr int BUF_B = NS * (XPL + DPL);
    static_assert((NPD * 4) % kBlock == 0, "dz tile items");
    constexpr int RED_B = TAPS * 4 * 16 * 64 * 4;              // the 4-wave sum of all taps (epilogue), then the bias scratch
    constexpr int SMEM_B = 2 * BUF_B > RED_B ? 2 * BUF_B : RED_B;
    static_assert(SMEM_B >= 4096 * 4 + kBlock * 8 * 4, "bias scratch fits");
    __shared__ __attribute__((aligned(256))) char smem[SMEM_B];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int Hs = UP ? A.H >> 1 : A.H, Ws = UP ? A.W >> 1 : A.W;

    // ---- staging set-up: a thread always serves channel octet o = tid & 3 of X and of dz (Cin, Cout multiples of 32) ----
    const int o8 = 8 * (tid & 3);
    const int cx = ci0 + o8;
    const bool two = (A.flags & F_TWO) && cx >= A.C0;
    const int Cs = two ? A.C1 : A.C0, ccx = two ? cx - A.C0 : cx;
    const AT* __restrict__ xsrc = (two ? reinterpret_cast<const AT*>(A.x1) : reinterpret_cast<const AT*>(A.x0)) + ccx;
    const AT* __restrict__ dsrc = reinterpret_cast<const AT*>(A.dz) + co0 + o8;
    const AT* __restrict__ zsrc = GB ? reinterpret_cast<const AT*>(A.zf) + co0 + o8 : nullptr;
    const bool aff = (A.flags & F_AFF) != 0;
    const float lo = aff ? 0.f : -3.0e38f;
    float fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = 1.f; fb[i] = 0.f; }
    if (aff) {
        const float* ab = two ? A.ab1 : A.ab0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = ab[ccx + i]; fb[i] = ab[Cs + ccx + i]; }
    }
    float ga[GB ? 8 : 1], gb[GB ? 8 : 1], gd[GB ? 8 : 1];
    if constexpr (GB) load_gb8(A.bnf, A.Cout, co0 + o8, ga, gb, gd);
    int xly[NXS], xlx[NXS];
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
        const int P = (tid >> 2) + k * (kBlock / 4);
        xly[k] = P < NPX ? P / IW : -1000000; xlx[k] = P % IW;      // pad pixels fall outside every image -> zeros
    }
    struct Regs { typename Raw4<AT>::type x[NXS][2], d[NDS][2], z[GB ? NDS : 1][2]; };
    float bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;

    auto tile_of = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / A.tiles; const int t = tl % A.tiles;
        x0 = (t % A.tiles_x) * TW; y0 = (t / A.tiles_x) * TH;
    };
    // Loads are UNCONDITIONAL (addresses clamped into the image; out-of-image values are discarded when stored): no
    // exec-mask branches, so staging and the MFMA loop share one basic block.  Staging is written per ITEM (one
    // (pixel, octet) of X or of dz per thread) so that compute() can place the items of the next tile between its steps.
    struct Geo { int b, y0, x0; };
    auto geo_of = [&](int tl) { Geo g; tile_of(tl, g.b, g.y0, g.x0); return g; };
    auto load_item = [&](int i, const Geo& g, Regs& R) {
        if (i < NXS) {
            const int k = i;
            int gy = g.y0 + xly[k] - PT, gx = g.x0 + xlx[k] - PT;         // position in the conv-input (high-res) image
            gy = gy < 0 ? 0 : (gy >= A.H ? A.H - 1 : gy); gx = gx < 0 ? 0 : (gx >= A.W ? A.W - 1 : gx);
            const int sy = UP ? gy >> 1 : gy, sx = UP ? gx >> 1 : gx;
            const AT* p = xsrc + (((size_t)g.b * Hs + sy) * Ws + sx) * Cs;
            R.x[k][0] = ldraw4<AT>(p); R.x[k][1] = ldraw4<AT>(p + 4);
        } else {
            const int k = i - NXS;
            const int P = (tid >> 2) + k * (kBlock / 4);
            int py = g.y0 + P / TW, px = g.x0 + P % TW;
            py = py >= A.H ? A.H - 1 : py; px = px >= A.W ? A.W - 1 : px;
            const size_t e = (((size_t)g.b * A.H + py) * A.W + px) * A.Cout;
            R.d[k][0] = ldraw4<AT>(dsrc + e); R.d[k][1] = ldraw4<AT>(dsrc + e + 4);
            if constexpr (GB) { R.z[k][0] = ldraw4<AT>(zsrc + e); R.z[k][1] = ldraw4<AT>(zsrc + e + 4); }
        }
    };
    // `live`: false for the dummy store issued behind the last tile (keeps the loop body branch-free)
    auto store_item = [&](int i, const Geo& g, const Regs& R, int buf, bool live) {
        if (i < NXS) {
            const int k = i;
            const float4 v0 = widen4(R.x[k][0]), v1 = widen4(R.x[k][1]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const int gy = g.y0 + xly[k] - PT, gx = g.x0 + xlx[k] - PT;
            const bool in = gy >= 0 && gy < A.H && gx >= 0 && gx < A.W;
            act8(v, fa, fb, lo, in);
            if constexpr (DROP) {              // (only the up-conv behind the bottleneck; compile-time: a runtime branch here
                                               //  would cut the conversion out of the basic block that holds the MFMAs)
                const int sy = UP ? gy >> 1 : gy, sx = UP ? gx >> 1 : gx;
                const uint32_t el = (uint32_t)((((size_t)g.b * Hs + sy) * Ws + sx) * Cs + ccx);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = in ? v[e] * drop_mul(A.drop, el + e) : 0.f;
            }
            uint4 pl[NS];
            split8<NS>(v, pl);
            const int P = (tid >> 2) + k * (kBlock / 4);
            char* d = smem + buf * BUF_B + o8 * 2 + P * 64;
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * XPL) = pl[p];
        } else {
            const int k = i - NXS;
            const int P = (tid >> 2) + k * (kBlock / 4);
            const bool in = live && g.y0 + P / TW < A.H && g.x0 + P % TW < A.W;
            const float4 v0 = widen4(R.d[k][0]), v1 = widen4(R.d[k][1]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (GB) {
                const float4 z0 = widen4(R.z[k][0]), z1 = widen4(R.z[k][1]);
                const float zv[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
                gb8<AT>(v, zv, ga, gb, gd, in);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] = in ? v[e] : 0.f; bsum[e] += v[e]; }
            }
            uint4 pl[NS];
            split8<NS>(v, pl);
            char* d = smem + buf * BUF_B + NS * XPL + o8 * 2 + P * 64;
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * DPL) = pl[p];
        }
    };
    constexpr int NIT = NXS + NDS;
    auto load = [&](int tl, Regs& R) {
        const Geo g = geo_of(tl);
#pragma unroll
        for (int i = 0; i < NIT; ++i) load_item(i, g, R);
    };
    auto store = [&](int tl, const Regs& R, int buf, bool live) {
        const Geo g = geo_of(tl);
#pragma unroll
        for (int i = 0; i < NIT; ++i) store_item(i, g, R, buf, live);
    };

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read lane offsets: 16-lane group g reads the 4 pixels x 16 channels block at pixel rows 8h + 4t + q,
    // channels 16 (g & 1) + 4 p  (h = lane >> 5, q = (lane & 15) >> 2, p = lane & 3); lane gets channel lane & 31
    const int laneoff = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto tr8 = [&](const char* base) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 4 * 64));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // One flat sequence of (16-pixel K step, tap) steps.  hipcc, left alone, emits the whole conversion of the next tile
    // first and then the MFMAs with every fragment read right in front of its use (one wave per SIMD here: nothing else
    // covers either).  So the A fragments of step s+2 are requested before the MFMAs of step s issue (ring of 3 sets, B
    // per K step two steps ahead of its first use) and the schedule of the block is pinned: per step the fragment reads,
    // the step's MFMAs, a share of the next tile's conversion VALU in their shadow, a share of its LDS writes.
    constexpr int KS = TW / 16, STEPS = KS * TAPS;
    auto compute = [&](int buf, auto&& between) {
        const char* Xl = smem + buf * BUF_B + laneoff;
        const char* Dl = smem + buf * BUF_B + NS * XPL + laneoff + (wave * TW) * 64;
        constexpr int DEPTH = 3;
        bf16x8 av[DEPTH][NS], bw[2][NS];
        auto fetch_a = [&](int st, bf16x8 (&f)[NS]) {
            const int ks = st / TAPS, t = st % TAPS, ky = t / KH, kx = t % KH;
#pragma unroll
            for (int p = 0; p < NS; ++p) f[p] = tr8(Xl + p * XPL + ((wave + ky) * IW + ks * 16 + kx) * 64);
        };
        auto fetch_b = [&](int ks, bf16x8 (&f)[NS]) {
#pragma unroll
            for (int p = 0; p < NS; ++p) f[p] = tr8(Dl + p * DPL + ks * 16 * 64);
        };
        __builtin_amdgcn_s_setprio(1);
        fetch_b(0, bw[0]); fetch_a(0, av[0]); fetch_a(1, av[1]);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 2 < STEPS) fetch_a(st + 2, av[(st + 2) % DEPTH]);
            if (st % TAPS == TAPS - 2 && st / TAPS + 1 < KS) fetch_b(st / TAPS + 1, bw[(st / TAPS + 1) & 1]);
            const int t = st % TAPS;
            const bf16x8 (&a)[NS] = av[st % DEPTH];
            const bf16x8 (&bv)[NS] = bw[(st / TAPS) & 1];
            f32x16 c = acc[t];
            if constexpr (NS == 3) {
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bv[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bv[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bv[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bv[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bv[0], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bv[0], c, 0, 0, 0);
            acc[t] = c;
            between(st);                      // staging items of the NEXT tile, in program order behind this step's reads
        }
        __builtin_amdgcn_s_setprio(0);
        constexpr int NPROD = NS == 3 ? 6 : 1, RPF = 2 * NS;                 // MFMAs per step; LDS reads per fragment fetch
        constexpr int VPS = NS == 3 ? (UP ? 40 : 26) : (UP ? 24 : 14);      // conversion VALU placed under a step's MFMAs
        constexpr int WPS = NS;                                             // LDS writes of one staging item
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * RPF, 0);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            if (st + 2 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, RPF, 0);
            if (st % TAPS == TAPS - 2 && st / TAPS + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, RPF, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPS, 0);
            if (((st + 1) * NIT) / STEPS != (st * NIT) / STEPS) {           // a staging item ends behind this step
                __builtin_amdgcn_sched_group_barrier(0x200, WPS, 0);
                if (GB && (st * NIT) / STEPS >= NXS) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // and its registers are
                else __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                  // re-loaded for tile t+2
            }
        }
    };

    // ---- tile pipeline: tile t in LDS buffer `buf`, tile t+1 in registers, tile t+2 requested; per iteration ONE
    // branch-free body: convert + write t+1 into the other buffer, request t+2, multiply t -- then one barrier ----
    Regs R;
    const int t0 = blockIdx.x, step = A.npb, tend = A.total_tiles, tlast = tend - 1;
    if (t0 < tend) {
        load(t0, R);
        store(t0, R, 0, true);
        load(t0 + step < tend ? t0 + step : tlast, R);
    }
    __syncthreads();
    int buf = 0;
    for (int tl = t0; tl < tend; tl += step, buf ^= 1) {
        const int t1 = tl + step, t2 = tl + 2 * step;
        const Geo g1 = geo_of(t1 < tend ? t1 : tlast), g2 = geo_of(t2 < tend ? t2 : tlast);
        const bool live = t1 < tend;
        compute(buf, [&](int st) {
#pragma unroll
            for (int i = 0; i < NIT; ++i)
                if (((st + 1) * NIT) / STEPS != (st * NIT) / STEPS && (st * NIT) / STEPS == i) {
                    store_item(i, g1, R, buf ^ 1, live);     // tile t+1: registers -> split -> LDS (other buffer)
                    load_item(i, g2, R);                     // tile t+2: the same registers, requested right away
                }
        });
        __syncthreads();
    }

    // ---- 4-wave sum of ALL taps in one pass through LDS (fixed order (w0 + w1) + (w2 + w3)), then the slab.  (Tap by tap
    // it was 2 barriers and a round trip per tap: the fixed cost of a launch -- prologue + this epilogue -- measured as
    // HALF of a 41 us launch.)  Image [wave][tap][register quad q][lane] of float4: 16-byte lanes, conflict-free. ----
    float* const red = reinterpret_cast<float*>(smem);
    float4* const red4 = reinterpret_cast<float4*>(smem);
    const size_t wsize = (size_t)TAPS * A.Cin * A.Cout;
    float* out = A.part + (size_t)blockIdx.x * (wsize + A.Cout);
    __syncthreads();                                              // every wave is done with the operand images
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            red4[((wave * TAPS + t) * 4 + q) * 64 + lane] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {                              // thread (wave q, lane): register quad q of every tap
        const int q = wave, e = (t * 4 + q) * 64 + lane;
        const float4 w0 = red4[e], w1 = red4[TAPS * 256 + e], w2 = red4[2 * TAPS * 256 + e], w3 = red4[3 * TAPS * 256 + e];
        const float sv[4] = {(w0.x + w1.x) + (w2.x + w3.x), (w0.y + w1.y) + (w2.y + w3.y),
                             (w0.z + w1.z) + (w2.z + w3.z), (w0.w + w1.w) + (w2.w + w3.w)};
        const int col = lane & 31, row0 = 8 * q + 4 * (lane >> 5);            // C/D layout: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
            if (ci0 + row0 + jr < A.Cin && co0 + col < A.Cout)
                out[((size_t)t * A.Cin + ci0 + row0 + jr) * A.Cout + co0 + col] = sv[jr];
    }
    __syncthreads();
    float* const bs = red + 4096;
#pragma unroll
    for (int i = 0; i < 8; ++i) bs[tid * 8 + i] = bsum[i];
    __syncthreads();
    if (tid < 32 && blockIdx.y == 0 && co0 + tid < A.Cout) {
        const int oct = tid >> 3, comp = tid & 7;
        float s = 0.f;
        for (int t = oct; t < kBlock; t += 4) s += bs[t * 8 + comp];
        out[wsize + co0 + tid] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// THIN layers (<= 16 output channels: the full- and half-resolution stages, where the tensors, not the FLOPs, are
// large) on the bf16 pipe: v_mfma_f32_16x16x32_bf16, M = 16 output channels (8 used by the 8-channel layers),
// N = 16 consecutive pixels of a row, K = 32 = (32 / CT) taps x CT input channels (CT = 8, 16 or 32 = Cin).
//   * persistent blocks walk pixel tiles of 8 x 32 (XCD-aware TileWalk); the split weights of the whole layer
//     (<= 9 MFMA K-groups x NS terms) live in REGISTERS for the lifetime of the block -- no weight traffic at all;
//   * the input tile goes global -> registers (one tile ahead) -> BN + ReLU -> split -> LDS image
//     [term][pixel][CT channels] (double buffered, branch-free so that it is scheduled among the MFMAs of the
//     current tile); a lane's B fragment is ONE ds_read_b128: 8 channels of pixel (x + tap offset) -- lanes 16 kg..16 kg+15
//     serve K-slice kg = (tap within the group, channel octet), so the tap offset is a per-lane constant;
//   * BN statistics accumulate in registers across all tiles of the block: ONE partial row per block.
// Same addressing modes / epilogues / argument block as conv_igemm_p_k.  A.wbx = prep_wbt_k output for this launch's
// 16-row slice (rows A.m_off .. A.m_off + 15 of the weight matrix).  grid (nblk, 1, 1).
// ------------------------------------------------------------------------------------------------------------------
struct WbtDesc {
    const float* src; bf16_t* dst;
    int KH, Kc, M, ld, m_off, CT, NS;    // src[(tap * Kc + c) * ld + m]; rows m_off .. m_off + 15
    int m2;                              // 8-row slice in the two-pixel form (see conv_bt_k M2): row = dx * 8 + channel
    unsigned start, count;               // work items: (K-group, row, K-slice)
};

static __global__ __launch_bounds__(kBlock) void prep_wbt_k(const WbtDesc* __restrict__ descs, int nd, unsigned total) {
    for (unsigned e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
        int d = 0;
        while (d + 1 < nd && e >= descs[d + 1].start) ++d;
        const WbtDesc D = descs[d];
        const unsigned r = e - D.start;
        const int kg = r & 3, m = (r >> 2) & 15, g = r >> 6;
        const int taps = D.KH * D.KH, tpm = 32 / D.CT;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k32 = kg * 8 + j, c = k32 % D.CT;
            int tap = g * tpm + k32 / D.CT, mg = D.m_off + m;
            bool ok = tap < taps;
            if (D.m2) {     // K runs over the (KH) x (KH + 1) window two adjacent output pixels share; row (dx, channel)
                const int e = tap, ky = e / (D.KH + 1), kx = e % (D.KH + 1) - (m >> 3);
                ok = e < D.KH * (D.KH + 1) && kx >= 0 && kx < D.KH;
                tap = ky * D.KH + kx; mg = D.m_off + (m & 7);
            }
            v[j] = (ok && c < D.Kc && mg < D.M) ? D.src[((size_t)tap * D.Kc + c) * D.ld + mg] : 0.f;
        }
        bf16_t* base = D.dst + ((size_t)g * D.NS * 16 + m) * 32 + kg * 8;
        if (D.NS == 3) {
            uint4 pl[3]; split8<3>(v, pl);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(base + (size_t)p * 16 * 32) = pl[p];
        } else {
            uint4 pl[1]; split8<1>(v, pl);
            *reinterpret_cast<uint4*>(base) = pl[0];
        }
    }
}
__host__ inline int wbt_groups(int KH, int CT, bool m2 = false) { const int tpm = 32 / CT; return (KH * (m2 ? KH + 1 : KH) + tpm - 1) / tpm; }
__host__ inline size_t wbt_bytes(int KH, int CT, int NS, bool m2 = false) { return (size_t)wbt_groups(KH, CT, m2) * NS * 16 * 32 * 2; }

// GB (backward-data launches): x0 is the masked gradient g' of the layer; dz = ga g' + gb z + gd is formed while the tile is
// staged (A.gb_z, A.gb_bn; common.hpp) -- the stand-alone bn_bwd_apply pass over g' is gone.
//
// FDW (3x3 layers with 8 output channels -- the full-resolution convs -- backward): the launch ALSO reduces the layer's
// backward-weights dW[tap][ci][co] = sum_q X(q)[ci] dz(q - tap + 1)[co] over its tiles, for the 8 input channels whose
// gradient it produces.  The dz halo tile is already in LDS, split, for the backward-data MFMAs; the conv input X of the
// tile's own pixels q (= relu(a z + b) of the producer, the tensor the mask epilogue reads anyway) is staged beside it, and
// both reach the matrix core through transposing LDS reads (K = the 32 pixels of a tile row): A = X^T (rows = ci; rows
// 8-15 repeat them), B = dz at two horizontally adjacent taps (columns (dx, co)): per kernel row the column pairs
// (kx 1, kx 0) and (kx 2, -).  g', z and the producer's z then leave HBM ONCE for dX and dW together -- the separate
// backward-weights kernel of these layers (3 more tensor passes over the largest tensors of the net) is gone.
// One partial slab per block, like conv_dwbt_k (A.dw_part; reduce_all_k sums them); two blocks per CU.
template <int KH, int AMODE, int EPI, int CT, int NS, typename AT, bool M2 = false, bool GB = false, bool FDW = false>
__global__ __launch_bounds__(kBlock, FDW ? 2 : (CT == 32 ? 1 : (CT == 16 ? 2 : 3))) void conv_bt_k(const IgemmArgs A) {
    static_assert(!GB || EPI != EPI_FWD, "the BN-backward transform on load belongs to backward-data launches");
    static_assert(!FDW || (GB && CT == 8 && KH == 3 && AMODE == A_NORMAL), "fused backward-weights: 3x3, 8 K channels, transform on load");
    static_assert(!M2 || ((CT == 8 || CT == 16) && AMODE != A_DOWN2), "two-pixel form: 8 or 16 K channels, unit-stride or upsampled input");
    constexpr int TH = 8, TW = 32, TAPS = KH * (M2 ? KH + 1 : KH), TPM = 32 / CT, NG = (TAPS + TPM - 1) / TPM, OCT = CT / 8;
    constexpr int KW = M2 ? KH + 1 : KH;                          // taps per kernel row of the (extended) window
    constexpr int NTW = M2 ? 2 : 4, ACC = 4, MB = 16;
    constexpr bool SW2 = M2 && AMODE == A_NORMAL;                // fragments read every second pixel: two-bit swizzle
    constexpr int IH = AMODE == A_NORMAL ? TH + KH - 1 : (AMODE == A_UPF ? TH / 2 + 1 : 2 * TH + 1);
    constexpr int IW = AMODE == A_NORMAL ? TW + KH - 1 : (AMODE == A_UPF ? TW / 2 + 1 : 2 * TW + 1);
    constexpr int NPIX = IH * IW, PPS = kBlock / OCT, NSLOT = (NPIX + PPS - 1) / PPS, NPIXP = NSLOT * PPS;
    constexpr int PIXB = CT * 2, PLANE_B = NPIXP * PIXB, IN_B = NS * PLANE_B;
    constexpr int SWS = OCT == 4 ? 2 : 3;                       // swizzle: octet ^= (pixel >> SWS) & (OCT - 1)
    // byte offset of octet o of pixel P inside a term's image.  Unit-stride readers (16 lanes = 16 consecutive pixels)
    // need the octet folded with a pixel bit; the two-pixel form reads pixels 2n + const: with Q = P >> 1 the 16-byte
    // chunk index must be a bijection of Q mod 16 -> pixel parity ^= bit 4 (8 channels), parity ^= bit 3 and octet ^=
    // bit 4 (16 channels).
    auto lds_off = [](int P, int o) {
        if constexpr (SW2) {
            if constexpr (OCT == 1) return (P ^ ((P >> 4) & 1)) * PIXB;
            else return (P ^ ((P >> 3) & 1)) * PIXB + ((o ^ ((P >> 4) & 1)) * 16);
        } else return P * PIXB + ((o ^ ((P >> SWS) & (OCT - 1))) * 16);
    };
    constexpr int XPL = TH * TW * 16, XIMG_B = FDW ? NS * XPL : 0;      // FDW: the tile's own X pixels, [term][pixel][8 ch]
    // GBL: the 16-K / 16-output mask launch with fp32 activations holds two raw register sets (g' and z), the mask z and 60
    // weight registers -- the 24 transform coefficients no longer fit under the 256-register budget of two blocks per CU
    // (21 spills); they live in LDS and are re-read by every staging item (6 ds_read_b128, hidden among the MFMAs)
    constexpr bool GBL = GB && CT == 16 && EPI == EPI_MASK && !FDW && sizeof(AT) == 4;
    __shared__ __attribute__((aligned(256))) char smem[2 * IN_B + 2 * XIMG_B + (4 * MB + 4 * 2 * MB) * 4 + (GBL ? 3 * CT * 4 : 0)];
    char* const Xs = smem + 2 * IN_B;
    float* const epi = reinterpret_cast<float*>(smem + 2 * IN_B + 2 * XIMG_B);     // EPI_MASK: producer's BN rows a / b / mean / rstd
    float* const red = epi + 4 * MB;
    float* const gbs = red + 4 * 2 * MB;                                           // GBL: rows ga, gb, gd of the K channels

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kg = lane >> 4;

    // ---- split weights of the whole layer slice -> registers (A operand: row px = output channel, K-slice kg) ----
    bf16x8 wa[NG][NS];
    {
        const char* w = reinterpret_cast<const char*>(A.wbt) + px * 64 + kg * 16;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int p = 0; p < NS; ++p) wa[g][p] = *reinterpret_cast<const bf16x8*>(w + (g * NS + p) * 16 * 64);
    }
    if constexpr (EPI == EPI_MASK) {
        for (int e = tid; e < 4 * MB; e += kBlock) {
            const int arr = e / MB, m = e % MB;
            epi[e] = m < A.Mout ? A.bnin[arr * A.Mout + m] : 0.f;
        }
    }
    // this lane's output channels m4 .. m4 + 3; two-pixel form: rows 0-7 = pixel 2 px, rows 8-15 = pixel 2 px + 1
    const int m4 = M2 ? 4 * (kg & 1) : 4 * kg, dx = M2 ? kg >> 1 : 0;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_FWD && m4 < A.Mout) bias = ld4(A.bias + A.m_off + m4);

    // ---- staging set-up: thread serves channel octet o of pixels P_k ----
    const int o = tid % OCT, c8 = 8 * o;
    const bool two = (A.flags & F_TWO) && c8 >= A.C0;
    const int Cs = two ? A.C1 : A.C0, ccx = two ? c8 - A.C0 : c8;
    const AT* __restrict__ xsrc = (two ? reinterpret_cast<const AT*>(A.x1) : reinterpret_cast<const AT*>(A.x0)) + ccx;
    const bool aff = (A.flags & F_AFF) != 0;
    const float lo = aff ? 0.f : -3.0e38f;
    float fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = 1.f; fb[i] = 0.f; }
    if (aff) {
        const float* ab = two ? A.ab1 : A.ab0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = ab[ccx + i]; fb[i] = ab[Cs + ccx + i]; }
    }
    float fz[GB ? 8 : 1];
    if constexpr (GBL) {
        if (tid < 3 * CT) {
            const int row = tid / CT, c = tid % CT;
            gbs[tid] = c < A.Cin ? A.gb_bn[(row == 0 ? BN_GA : (row == 1 ? BN_GB : BN_GD)) * A.Cin + c] : 0.f;
        }
        __syncthreads();
    } else if constexpr (GB) load_gb8(A.gb_bn, A.Cin, c8, fa, fz, fb);     // fa = ga, fz = gb, fb = gd (one source: no concat)
    // z of the layer sits at the same element offsets as g' (both are carved from the handle's workspace)
    const ptrdiff_t zdelta = GB ? reinterpret_cast<const AT*>(A.gb_z) - reinterpret_cast<const AT*>(A.x0) : 0;
    int sly[NSLOT], slx[NSLOT], sdst[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int P = tid / OCT + k * PPS;
        sly[k] = P < NPIX ? P / IW : -1000000; slx[k] = P % IW;                 // pad pixels fall outside every image
        sdst[k] = lds_off(P, o);
    }
    struct RegSet { typename Raw4<AT>::type v[NSLOT][2], z[GB ? NSLOT : 1][2], x[FDW ? 2 : 1]; };
    auto origin = [&](const TileOrg& t, int& iy0, int& ix0) {
        const int y0 = t.ty * TH, x0 = t.tx * TW;
        iy0 = AMODE == A_NORMAL ? y0 - (KH - 1) / 2 : (AMODE == A_UPF ? y0 / 2 : 2 * y0 - 1);
        ix0 = AMODE == A_NORMAL ? x0 - (KH - 1) / 2 : (AMODE == A_UPF ? x0 / 2 : 2 * x0 - 1);
    };
    int soff[NSLOT];                                             // element offset of slot k relative to the tile origin (interior tiles)
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int P = tid / OCT + k * PPS, Pc = P < NPIX ? P : 0;
        soff[k] = ((Pc / IW) * A.Wi + Pc % IW) * Cs;
    }
    // Staging in ITEM form (item k = pixel slot k of this thread): the address block is the only place with a branch
    // (interior tiles: a wave-uniform tile offset + a per-item constant; border tiles: positions clamped into the image);
    // loads are unconditional; store and load of an item are separate so that the tile loop can place them between
    // its MFMA steps.  Offsets are 32-bit elements relative to the image of the tile's B-scan.
    int aoff[NSLOT];
    auto addr = [&](const TileOrg& t) -> const AT* {
        int iy0, ix0; origin(t, iy0, ix0);
        if (iy0 >= 0 && ix0 >= 0 && iy0 + IH <= A.Hi && ix0 + IW <= A.Wi) {
            const int to = (iy0 * A.Wi + ix0) * Cs;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) aoff[k] = to + soff[k];
        } else {
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) {
                int gy = iy0 + sly[k], gx = ix0 + slx[k];
                gy = gy < 0 ? 0 : (gy >= A.Hi ? A.Hi - 1 : gy); gx = gx < 0 ? 0 : (gx >= A.Wi ? A.Wi - 1 : gx);
                aoff[k] = (gy * A.Wi + gx) * Cs;
            }
        }
        return xsrc + (size_t)(t.b * A.Hi * A.Wi) * Cs;
    };
    float xa[FDW ? 8 : 1], xb[FDW ? 8 : 1], bsum[FDW ? 8 : 1];
    bool stage_live = true;              // false while the tile loop re-stages its last tile as a dummy (branch-free body)
    if constexpr (FDW) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xa[i] = A.dw_ab[BN_A * A.Mout + i]; xb[i] = A.dw_ab[BN_B * A.Mout + i]; bsum[i] = 0.f; }
    }
    auto load_item = [&](int k, const AT* ib, RegSet& RS) {
        RS.v[k][0] = ldraw4<AT>(ib + aoff[k]); RS.v[k][1] = ldraw4<AT>(ib + aoff[k] + 4);
        if constexpr (GB) { RS.z[k][0] = ldraw4<AT>(ib + zdelta + aoff[k]); RS.z[k][1] = ldraw4<AT>(ib + zdelta + aoff[k] + 4); }
    };
    auto store_item = [&](int k, int iy0, int ix0, int buf, const RegSet& RS) {
        const float4 v0 = widen4(RS.v[k][0]), v1 = widen4(RS.v[k][1]);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const int gy = iy0 + sly[k], gx = ix0 + slx[k];
        const bool in = gy >= 0 && gy < A.Hi && gx >= 0 && gx < A.Wi;
        if constexpr (GB) {
            const float4 z0 = widen4(RS.z[k][0]), z1 = widen4(RS.z[k][1]);
            const float zv[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            if constexpr (GBL) {
                const float4 a0 = ld4(gbs + c8), a1 = ld4(gbs + c8 + 4), b0 = ld4(gbs + CT + c8), b1 = ld4(gbs + CT + c8 + 4);
                const float4 d0 = ld4(gbs + 2 * CT + c8), d1 = ld4(gbs + 2 * CT + c8 + 4);
                const float la[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, lb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                const float ld[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                gb8<AT>(v, zv, la, lb, ld, in);
            } else gb8<AT>(v, zv, fa, fz, fb, in);
            if constexpr (FDW) {      // bias gradient = column sums of dz over the tile's OWN pixels (the halo belongs to neighbours)
                const bool own = stage_live && sly[k] >= 1 && sly[k] <= TH && slx[k] >= 1 && slx[k] <= TW;
#pragma unroll
                for (int i = 0; i < 8; ++i) bsum[i] += own ? v[i] : 0.f;
            }
        } else act8(v, fa, fb, lo, in);
        uint4 pl[NS];
        split8<NS>(v, pl);
        char* d = smem + buf * IN_B + sdst[k];
#pragma unroll
        for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * PLANE_B) = pl[p];
    };
    // ---- FDW: the conv input of the tile's own pixels.  Thread t serves pixel (t >> 5, t & 31), all 8 channels ----
    auto x_load = [&](const TileOrg& t, RegSet& RS) {          // unconditional, position clamped into the image
        if constexpr (FDW) {
            int gy = t.ty * TH + (tid >> 5), gx = t.tx * TW + (tid & 31);
            gy = gy >= A.Ho ? A.Ho - 1 : gy; gx = gx >= A.Wo ? A.Wo - 1 : gx;
            const AT* p = reinterpret_cast<const AT*>(A.dw_x) + ((size_t)(t.b * A.Ho + gy) * A.Wo + gx) * A.Mout;
            RS.x[0] = ldraw4<AT>(p); RS.x[1] = ldraw4<AT>(p + 4);
        }
    };
    auto x_store = [&](const TileOrg& t, int buf, const RegSet& RS) {
        if constexpr (FDW) {
            const float4 v0 = widen4(RS.x[0]), v1 = widen4(RS.x[1]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const bool in = t.ty * TH + (tid >> 5) < A.Ho && t.tx * TW + (tid & 31) < A.Wo;
            act8(v, xa, xb, 0.f, in);                              // relu(a z + b) inside the image, exactly 0 outside
            uint4 pl[NS];
            split8<NS>(v, pl);
            char* d = Xs + buf * XIMG_B + tid * 16;
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * XPL) = pl[p];
        }
    };
    auto load = [&](const TileOrg& t, RegSet& RS) {
        const AT* ib = addr(t);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) load_item(k, ib, RS);
        x_load(t, RS);
    };
    auto store = [&](const TileOrg& t, int buf, const RegSet& RS) {
        int iy0, ix0; origin(t, iy0, ix0);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) store_item(k, iy0, ix0, buf, RS);
        x_store(t, buf, RS);
    };

    // ---- per-lane B-fragment geometry: K-slice kg of group g = tap g * TPM + kg / OCT, channel octet kg % OCT ----
    const int oq = kg % OCT;
    int tky[NG], tkx[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        int tap = g * TPM + kg / OCT;
        tap = tap < TAPS ? tap : TAPS - 1;                      // absent taps carry zero weights: any valid address will do
        tky[g] = tap / KW; tkx[g] = tap % KW;
    }
    int bofs[NTW][NG];                                          // byte offset of this lane's fragment in a term's image
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int r = M2 ? 2 * wave + nt : 2 * wave + (nt >> 1), x = M2 ? 2 * px : 16 * (nt & 1) + px;
            int P;
            if constexpr (AMODE == A_NORMAL) P = (r + tky[g]) * IW + x + tkx[g];
            else if constexpr (AMODE == A_UPF) P = ((r + tky[g]) >> 1) * IW + ((x + tkx[g]) >> 1);
            else P = (2 * r + tky[g]) * IW + 2 * x + tkx[g];
            bofs[nt][g] = lds_off(P, oq);
        }

    // output / mask-input element offsets of this lane's 4 pixel groups relative to the tile origin (32-bit; the tile
    // origin is a wave-uniform 64-bit base: no per-lane 64-bit multiplies in the tile loop)
    int ooff[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int r = M2 ? 2 * wave + nt : 2 * wave + (nt >> 1), x = M2 ? 2 * px + dx : 16 * (nt & 1) + px;
        ooff[nt] = (r * A.Wo + x) * A.Mout + (m4 < A.Mout ? m4 : 0);
    }

    // ---- FDW: accumulators (kernel row ky, column pair kb: base tap column 1 -> taps (1, 0); 2 -> (2, -)) and the lane
    // geometry of the transposing reads: lane 4q + pp of 16-lane group kg supplies row (= pixel) 8 kg + q of a 4 x 16
    // block, values 4 pp .. 4 pp + 3 of its 16; the two reads of a fragment are 4 pixels apart (K = 8 kg .. 8 kg + 7) ----
    f32x4 dacc[FDW ? 6 : 1];
    if constexpr (FDW) {
#pragma unroll
        for (int a = 0; a < 6; ++a) dacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int trq = (lane & 15) >> 2, trp = lane & 3;
    const int xlane = (8 * kg + trq) * 16 + 8 * (trp & 1);                       // X image: rows 8-15 of A repeat rows 0-7
    // dz halo-image pixel for X pixel (row 2 wave, column 8 kg + q) at tap (ky 2, kx 1): row qy + 2 - ky, column qx + 2 - kx,
    // + 1 for the right half of the 16 values (the next pixel = tap kx - 1)
    const int dpix = 2 * wave * IW + 8 * kg + trq + 1 + (trp >> 1);
    auto tr4 = [](const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p)); };

    float s1[ACC] = {0.f, 0.f, 0.f, 0.f}, s2[ACC] = {0.f, 0.f, 0.f, 0.f};
    TileWalk<TH, TW> walk;
    walk.init(A.tiles, A.tiles_x, A.total_tiles);
    // Global loads run one tile ahead in registers.  (Two tiles ahead was measured twice: the second register set costs
    // a resident block per CU at 8 channels and pushes the 16-channel mask epilogue past its register budget: 1.33 ->
    // 1.49 ms over the thin layers of a step; with the two-pixel form, where it fits: 1.12 -> 1.15.)
    RegSet R0;
    TileOrg cur = walk.first(A.tiles);
    if (walk.tl0 < walk.tlend) {
        const TileOrg n1 = walk.tl0 + walk.step < walk.tlend ? walk.next(cur) : cur;
        load(cur, R0);
        store(cur, 0, R0);
        load(n1, R0);
    }
    __syncthreads();
    int buf = 0;
    for (int tl = walk.tl0; tl < walk.tlend; tl += walk.step, buf ^= 1) {
        const TileOrg nxt = tl + walk.step < walk.tlend ? walk.next(cur) : cur;
        const TileOrg nx2 = tl + 2 * walk.step < walk.tlend ? walk.next(nxt) : nxt;
        const int b = cur.b, y0 = cur.ty * TH, x0 = cur.tx * TW;
        const size_t tbase = (((size_t)b * A.Ho + y0) * A.Wo + x0) * A.Mout;       // wave-uniform element offset of the tile
        bool pvalid[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            pvalid[nt] = M2 ? (y0 + 2 * wave + nt < A.Ho && x0 + 2 * px + dx < A.Wo && m4 < A.Mout)
                            : (y0 + 2 * wave + (nt >> 1) < A.Ho && x0 + 16 * (nt & 1) + px < A.Wo && m4 < A.Mout);
        // producer's z for the epilogue mask: requested now, consumed after the MFMAs
        typename Raw4<AT>::type zq[EPI == EPI_MASK ? NTW : 1];
        if constexpr (EPI == EPI_MASK) {
            const AT* zb = reinterpret_cast<const AT*>(A.zin) + tbase;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) zq[nt] = ldraw4<AT>(zb + (pvalid[nt] ? ooff[nt] : 0));   // (tile origin is in range)
        }
        const AT* const ib2 = addr(nx2);                       // item addresses of tile t+2 (the branch stays out of the MFMA block)
        int iy1, ix1; origin(nxt, iy1, ix1);                   // tile t+1 is in R0 (a dummy repeat behind the last tile: branch-free body)
        stage_live = tl + walk.step < walk.tlend;
        f32x4 acc[NTW];
        const char* Ib = smem + buf * IN_B;
        {   // flat sequence of (pixel group, K group) steps; the B fragments of step s+2 are requested before the MFMAs of
            // step s issue (ring of 3 fragment sets): an LDS round trip is longer than the 6 MFMAs of one step.  The staging
            // items of tile t+1 (registers -> LDS, other buffer) and the loads of tile t+2 into the freed registers sit in
            // program order BEHIND the fragment reads of their step: an LDS write may not move above a read it might alias.
            // FDW appends the backward-weights steps to the same sequence -- (X row r of this wave, kernel row ky, column
            // pair kb): A = the X row's fragments, B = dz at that tap pair, both through transposing reads -- so that their
            // LDS round trips hide behind the backward-data MFMAs and vice versa.
            // (look-ahead of the fragment ring: 2 steps; 1 where the 16-channel mask epilogue with the transform on load would
            //  otherwise spill 20 registers under its 256-register budget)
            constexpr int LA = (GB && CT == 16 && !M2 && EPI == EPI_MASK && NS == 3) ? 1 : 2;
            constexpr int SX = NTW * NG, SD = FDW ? 12 : 0, STEPS = SX + SD, DEPTH = LA + 1;
            constexpr int NITEM = NSLOT + (FDW ? 1 : 0);          // staging items: the dz slots, then the X pixel
            bf16x8 bv[DEPTH][NS], af[FDW ? 2 : 1][NS];
            const char* const Xb = Xs + buf * XIMG_B + xlane;
            auto fetch = [&](int st, bf16x8 (&f)[NS]) {
                if (st < SX) {
                    const char* q = Ib + bofs[st / NG][st % NG];
#pragma unroll
                    for (int p = 0; p < NS; ++p) f[p] = *reinterpret_cast<const bf16x8*>(q + p * PLANE_B);
                } else {
                    // dz at pixel q - tap + 1: halo row qy + 2 - ky, halo column qx + 2 - kx (+ 1 for the right half of the 16)
                    const int j = st - SX, r = j / 6, ky = (j % 6) / 2, kb = j % 2;
                    const int P0 = dpix + (r + 2 - ky) * IW - kb, P1 = P0 + 4;
                    const char* q0 = Ib + lds_off(P0, 0) + 8 * (trp & 1);
                    const char* q1 = Ib + lds_off(P1, 0) + 8 * (trp & 1);
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        const bf16x4 lo4 = tr4(q0 + p * PLANE_B), hi4 = tr4(q1 + p * PLANE_B);
                        f[p] = bf16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    }
                }
            };
            auto fetch_a = [&](int r, bf16x8 (&f)[NS]) {          // X row 2 wave + r: rows = ci (8-15 repeat 0-7), K = its 32 pixels
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    const char* q = Xb + p * XPL + (2 * wave + r) * (TW * 16);
                    const bf16x4 lo4 = tr4(q), hi4 = tr4(q + 4 * 16);
                    f[p] = bf16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                }
            };
            fetch(0, bv[0]);
            if (LA > 1 && STEPS > 1) fetch(1, bv[1]);
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (st + LA < STEPS) fetch(st + LA, bv[(st + LA) % DEPTH]);
                if constexpr (FDW) {
                    if (st == SX - 2) fetch_a(0, af[0]);           // two steps ahead of their first use
                    if (st == SX + 4) fetch_a(1, af[1]);
                }
                const bf16x8 (&b)[NS] = bv[st % DEPTH];
                if (st < SX) {
                    const int g = st % NG;
                    if constexpr (NS == 3) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g][0], b[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g][2], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g][1], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g][0], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g][1], b[0], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g][0], b[0], c, 0, 0, 0);
                    if (g == NG - 1) { acc[st / NG] = c; c = f32x4{0.f, 0.f, 0.f, 0.f}; }
                } else if constexpr (FDW) {
                    const int j = st - SX;
                    const bf16x8 (&xa_)[NS] = af[j / 6];
                    f32x4 d = dacc[j % 6];
                    if constexpr (NS == 3) {
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_[0], b[2], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_[2], b[0], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_[1], b[1], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_[0], b[1], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_[1], b[0], d, 0, 0, 0);
                    }
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa_[0], b[0], d, 0, 0, 0);
                    dacc[j % 6] = d;
                }
#pragma unroll
                for (int k = 0; k < NITEM; ++k)
                    if ((k * STEPS) / NITEM == st) {
                        if (k < NSLOT) { store_item(k, iy1, ix1, buf ^ 1, R0); load_item(k, ib2, R0); }
                        else { x_store(nxt, buf ^ 1, R0); x_load(nx2, R0); }
                    }
            }
            // pin the schedule of this basic block (hipcc otherwise sinks every LDS read next to its use and emits the
            // whole conversion in front of the MFMAs): per step the fragments two steps ahead, the step's MFMAs, a share
            // of the conversion VALU in their shadow, and behind an item its LDS writes and the loads that refill it
            constexpr int NPROD = NS == 3 ? 6 : 1;
            constexpr int VPS = (NSLOT * ((NS == 3 ? 60 : 36) + (GB ? (sizeof(AT) == 2 ? 16 : 4) : 0)) + (FDW ? (NS == 3 ? 56 : 32) : 0) + STEPS - 1) / STEPS;
            __builtin_amdgcn_sched_group_barrier(0x100, (SX > 0 ? NS : 2 * NS) + ((LA > 1 && STEPS > 1) ? (SX > 1 ? NS : 2 * NS) : 0), 0);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (st + LA < STEPS) { if (st + LA < SX) __builtin_amdgcn_sched_group_barrier(0x100, NS, 0); else __builtin_amdgcn_sched_group_barrier(0x100, 2 * NS, 0); }
                if (FDW && (st == SX - 2 || st == SX + 4)) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NS, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPS, 0);
                bool item = false, xitem = false;
#pragma unroll
                for (int k = 0; k < NITEM; ++k) { const bool hit = (k * STEPS) / NITEM == st; item = item || (hit && k < NSLOT); xitem = xitem || (hit && k >= NSLOT); }
                if (item) {
                    __builtin_amdgcn_sched_group_barrier(0x200, NS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, GB ? 4 : 2, 0);
                }
                if (xitem) {
                    __builtin_amdgcn_sched_group_barrier(0x200, NS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
            }
        }
        // ---- epilogue of this tile: lane holds channels m4..m4+3 of pixel (row 2 wave + nt/2, x = 16 (nt&1) + px) ----
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const bool valid = pvalid[nt];
            float v[4] = {acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
            if constexpr (EPI == EPI_FWD) {
                v[0] += bias.x; v[1] += bias.y; v[2] += bias.z; v[3] += bias.w;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float u = valid ? v[k] : 0.f; s1[k] += u; s2[k] += u * u; }
            } else if constexpr (EPI == EPI_MASK) {
                const float4 zw = widen4(zq[nt]);
                const float zz[4] = {zw.x, zw.y, zw.z, zw.w};
                const int ml = m4 < MB ? m4 : 0;
                const float4 ea = ld4(epi + BN_A * MB + ml), eb = ld4(epi + BN_B * MB + ml);
                const float4 em = ld4(epi + BN_MEAN * MB + ml), er = ld4(epi + BN_RSTD * MB + ml);
                const float ka[4] = {ea.x, ea.y, ea.z, ea.w}, kb[4] = {eb.x, eb.y, eb.z, eb.w};
                const float km[4] = {em.x, em.y, em.z, em.w}, kr[4] = {er.x, er.y, er.z, er.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float yv = fmaf(ka[k], zz[k], kb[k]);
                    float gv = v[k];
                    if (A.drop_out) gv *= drop_mul(A.drop, (uint32_t)tbase + (uint32_t)(valid ? ooff[nt] + k : 0));
                    gv = (valid && yv > 0.f) ? gv : 0.f;
                    const float xh = (zz[k] - km[k]) * kr[k];
                    v[k] = gv; s1[k] += gv; s2[k] += gv * xh;
                }
            }
            if (valid) sta4<AT>(reinterpret_cast<AT*>(A.out) + tbase + ooff[nt], make_float4(v[0], v[1], v[2], v[3]));
        }
        cur = nxt;
        __syncthreads();
    }

    if constexpr (EPI != EPI_RAW) {
        if (A.part) {
            // lanes sharing kg hold the same 4 channels for 16 different pixels: reduce over the pixel lanes, then waves
            subgroup_reduce_rec<ACC, ACC, 8>(s1, lane);
            subgroup_reduce_rec<ACC, ACC, 8>(s2, lane);
            const int ci = sub_chan<ACC, 8>(lane), mloc = 4 * kg + ci;
            red[(wave * 2 + 0) * MB + mloc] = s1[0];
            red[(wave * 2 + 1) * MB + mloc] = s2[0];
            __syncthreads();
            if (tid < 2 * MB) {
                const int stat = tid / MB, ml = tid % MB;
                float s = (red[(0 * 2 + stat) * MB + ml] + red[(1 * 2 + stat) * MB + ml]) +
                          (red[(2 * 2 + stat) * MB + ml] + red[(3 * 2 + stat) * MB + ml]);
                if constexpr (M2) {     // rows 8-15 are the same channels at the odd pixels
                    const int mh = (ml + 8) % MB;
                    s += (red[(0 * 2 + stat) * MB + mh] + red[(1 * 2 + stat) * MB + mh]) +
                         (red[(2 * 2 + stat) * MB + mh] + red[(3 * 2 + stat) * MB + mh]);
                }
                if (ml < A.Mout) part_store(A.part + (size_t)blockIdx.x * (2 * A.Mout) + (size_t)stat * A.Mout + ml, s);
            }
            // the last block of the launch turns the rows (one per block) into the layer's record (kernels_fin.hpp)
            if (A.fin.counter) finalize_in_launch(A.fin, A.part, gridDim.x, A.Mout, gridDim.x, smem);
        }
    }
    if constexpr (FDW) {
        // ---- the block's backward-weights slab: fixed-order 4-wave sum through LDS; D row = 4 (lane >> 4) + r is ci (rows
        // 8-15 repeat 0-7), column = lane & 15 is (dx, co): pair 0 -> taps kx = 1 - dx, pair 1 -> kx = 2 (dx = 1 unused) ----
        __syncthreads();                                       // every wave is done with the images
        float* const r4 = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) r4[((wave * 6 + a) * 4 + r) * 64 + lane] = dacc[a][r];
        __syncthreads();
        const int wsize = 9 * A.dw_Cin * 8;
        float* const out = A.dw_part + (size_t)blockIdx.x * (wsize + 8);
        for (int e = tid; e < 6 * 4 * 64; e += kBlock) {
            const float sv = (r4[e] + r4[1536 + e]) + (r4[3072 + e] + r4[4608 + e]);
            const int ln = e & 63, r = (e >> 6) & 3, a = e >> 8;
            const int m = 4 * (ln >> 4) + r, n = ln & 15, ky = a >> 1, kb = a & 1, dxn = n >> 3, co = n & 7;
            const int kx = kb == 0 ? 1 - dxn : 2;
            if (m < 8 && !(kb == 1 && dxn == 1)) out[((ky * 3 + kx) * A.dw_Cin + A.dw_ci_off + m) * 8 + co] = sv;
        }
        __syncthreads();
        block_reduce_store<8>(bsum, r4, out + wsize, A.dw_bias ? 8 : 0);      // bias gradient = sum of dz over the block's pixels
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward-weights of the THIN layers on the bf16 pipe (v_mfma_f32_16x16x32_bf16): cin = CI and cout = CO in
// {8, 16, 32}, not both 32 (those are conv_dwbx_k's).  K = 32 consecutive pixels of a row; both operands come out of
// the channel-fastest LDS images [term][pixel][C] through the transposing read (4 pixels x 16 consecutive bf16 per 16
// lanes).  "16 consecutive bf16" of a pixel row is what makes the thin shapes fit the 16-row / 16-column tile:
//   * CI = 16: one A tile per tap (rows = ci);   CI = 32: two (ci 0-15, 16-31);
//   * CI = 8:  the 16 values are (pixel, ci 0-7) and (pixel + 1, ci 0-7) = TWO horizontally adjacent taps in one tile:
//     per kernel row the tiles (kx 0, kx 1) and (kx 2, -) -- 6 A tiles instead of 9;
//   * CO = 8:  columns 8-15 of the B tile are dz of the next pixel: finite junk that is simply not written out.
// Block = all channels of the layer for a strided list of 4-row x 32-pixel tiles; wave w owns pixel row w; staging is
// branch-free (loads clamped, one tile ahead in registers, LDS images double buffered) so it is scheduled among the
// MFMAs; fixed-order 4-wave sum, ONE partial slab per block.  grid (npb, 1, 1).
// ------------------------------------------------------------------------------------------------------------------
// GB: A.dz is the masked gradient g' and the BN-backward transform is applied while it is staged (A.zf, A.bnf).
template <int KH, bool UP, int CI, int CO, int NS, typename AT, bool DROP = false, bool GB = false>
__global__ __launch_bounds__(kBlock, (CI + CO >= 48) ? 1 : 2) void conv_dwbt_k(const ConvBwdWArgs A) {
    constexpr int TH = 4, TW = 32, TAPS = KH * KH;
    constexpr int IH = UP ? TH + 1 : TH + KH - 1, IW = UP ? TW + 1 : TW + KH - 1, PT = UP ? 0 : (KH - 1) / 2;
    constexpr int NPX = IH * IW, NPD = TH * TW;
    constexpr int OX = CI / 8, OD = CO / 8;                                   // channel octets per pixel
    constexpr int PPX = kBlock / OX, NXS = (NPX + 4 + PPX - 1) / PPX, NPXP = NXS * PPX;      // (+4: junk taps read past the tile)
    constexpr int PPD = kBlock / OD, NDS = (NPD + (CO == 8 ? 1 : 0) + PPD - 1) / PPD, NPDP = NDS * PPD;   // (+1: junk columns at CO = 8)
    constexpr int XPB = CI * 2, DPB = CO * 2;                                 // bytes per pixel of a term's image
    constexpr int XPL = NPXP * XPB, DPL = NPDP * DPB, BUF_B = NS * (XPL + DPL);
    // A tiles: (tap group, column base); CI = 8 pairs horizontally adjacent taps
    constexpr int MTX = CI == 8 ? (KH + 1) / 2 : KH * (CI / 16);              // tiles per kernel row
    constexpr int MTILES = KH * MTX, NTILES = CO == 32 ? 2 : 1, NACC = MTILES * NTILES;
    constexpr int RED_B = 4 * 256 * 4;
    constexpr int SMEM_B = 2 * BUF_B > RED_B + kBlock * 8 * 4 ? 2 * BUF_B : RED_B + kBlock * 8 * 4;
    __shared__ __attribute__((aligned(256))) char smem[SMEM_B];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hs = UP ? A.H >> 1 : A.H, Ws = UP ? A.W >> 1 : A.W;

    // ---- staging set-up ----
    const int ox = tid % OX, cx = 8 * ox;
    const bool two = (A.flags & F_TWO) && cx >= A.C0;
    const int Cs = two ? A.C1 : A.C0, ccx = two ? cx - A.C0 : cx;
    const AT* __restrict__ xsrc = (two ? reinterpret_cast<const AT*>(A.x1) : reinterpret_cast<const AT*>(A.x0)) + ccx;
    const int od = tid % OD;
    const AT* __restrict__ dsrc = reinterpret_cast<const AT*>(A.dz) + 8 * od;
    const AT* __restrict__ zsrc = GB ? reinterpret_cast<const AT*>(A.zf) + 8 * od : nullptr;
    const bool aff = (A.flags & F_AFF) != 0;
    const float lo = aff ? 0.f : -3.0e38f;
    float fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = 1.f; fb[i] = 0.f; }
    if (aff) {
        const float* ab = two ? A.ab1 : A.ab0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = ab[ccx + i]; fb[i] = ab[Cs + ccx + i]; }
    }
    float ga[GB ? 8 : 1], gb[GB ? 8 : 1], gd[GB ? 8 : 1];
    if constexpr (GB) load_gb8(A.bnf, A.Cout, 8 * od, ga, gb, gd);
    int xly[NXS], xlx[NXS];
#pragma unroll
    for (int k = 0; k < NXS; ++k) {
        const int P = tid / OX + k * PPX;
        xly[k] = P < NPX ? P / IW : -1000000; xlx[k] = P % IW;               // pad pixels fall outside every image -> zeros
    }
    struct Regs { typename Raw4<AT>::type x[NXS][2], d[NDS][2], z[GB ? NDS : 1][2]; };
    float bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;

    auto tile_of = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / A.tiles; const int t = tl % A.tiles;
        x0 = (t % A.tiles_x) * TW; y0 = (t / A.tiles_x) * TH;
    };
    // staging in ITEM form (one (pixel, octet) of X or of dz per thread), unconditional loads with addresses clamped into
    // the image: the tile loop places the items of the next tiles between its MFMA steps (see conv_dwbx_k)
    struct Geo { int b, y0, x0; };
    auto geo_of = [&](int tl) { Geo g; tile_of(tl, g.b, g.y0, g.x0); return g; };
    constexpr int NIT = NXS + NDS;
    auto load_item = [&](int i, const Geo& g, Regs& R) {
        if (i < NXS) {
            const int k = i;
            int gy = g.y0 + xly[k] - PT, gx = g.x0 + xlx[k] - PT;
            gy = gy < 0 ? 0 : (gy >= A.H ? A.H - 1 : gy); gx = gx < 0 ? 0 : (gx >= A.W ? A.W - 1 : gx);
            const int sy = UP ? gy >> 1 : gy, sx = UP ? gx >> 1 : gx;
            const AT* p = xsrc + (((size_t)g.b * Hs + sy) * Ws + sx) * Cs;
            R.x[k][0] = ldraw4<AT>(p); R.x[k][1] = ldraw4<AT>(p + 4);
        } else {
            const int k = i - NXS;
            const int P = tid / OD + k * PPD;
            int py = g.y0 + P / TW, px = g.x0 + P % TW;
            py = py >= A.H ? A.H - 1 : py; px = px >= A.W ? A.W - 1 : px;
            const size_t e = (((size_t)g.b * A.H + py) * A.W + px) * A.Cout;
            R.d[k][0] = ldraw4<AT>(dsrc + e); R.d[k][1] = ldraw4<AT>(dsrc + e + 4);
            if constexpr (GB) { R.z[k][0] = ldraw4<AT>(zsrc + e); R.z[k][1] = ldraw4<AT>(zsrc + e + 4); }
        }
    };
    auto store_item = [&](int i, const Geo& g, const Regs& R, int buf, bool live) {
        if (i < NXS) {
            const int k = i;
            const float4 v0 = widen4(R.x[k][0]), v1 = widen4(R.x[k][1]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const int gy = g.y0 + xly[k] - PT, gx = g.x0 + xlx[k] - PT;
            const bool in = gy >= 0 && gy < A.H && gx >= 0 && gx < A.W;
            act8(v, fa, fb, lo, in);
            if constexpr (DROP) {              // (only the up-conv behind the bottleneck; compile-time: a runtime branch here
                                               //  would cut the conversion out of the basic block that holds the MFMAs)
                const int sy = UP ? gy >> 1 : gy, sx = UP ? gx >> 1 : gx;
                const uint32_t el = (uint32_t)((((size_t)g.b * Hs + sy) * Ws + sx) * Cs + ccx);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = in ? v[e] * drop_mul(A.drop, el + e) : 0.f;
            }
            uint4 pl[NS];
            split8<NS>(v, pl);
            const int P = tid / OX + k * PPX;
            char* d = smem + buf * BUF_B + cx * 2 + P * XPB;
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * XPL) = pl[p];
        } else {
            const int k = i - NXS;
            const int P = tid / OD + k * PPD;
            const bool in = live && P < NPD && g.y0 + P / TW < A.H && g.x0 + P % TW < A.W;
            const float4 v0 = widen4(R.d[k][0]), v1 = widen4(R.d[k][1]);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (GB) {
                const float4 z0 = widen4(R.z[k][0]), z1 = widen4(R.z[k][1]);
                const float zv[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
                gb8<AT>(v, zv, ga, gb, gd, in);
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[e] += v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] = in ? v[e] : 0.f; bsum[e] += v[e]; }
            }
            uint4 pl[NS];
            split8<NS>(v, pl);
            char* d = smem + buf * BUF_B + NS * XPL + od * 16 + P * DPB;
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(d + p * DPL) = pl[p];
        }
    };
    auto load = [&](int tl, Regs& R) {
        const Geo g = geo_of(tl);
#pragma unroll
        for (int i = 0; i < NIT; ++i) load_item(i, g, R);
    };
    auto store = [&](int tl, const Regs& R, int buf, bool live) {
        const Geo g = geo_of(tl);
#pragma unroll
        for (int i = 0; i < NIT; ++i) store_item(i, g, R, buf, live);
    };

    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing-read lane geometry: lanes 16 kg .. 16 kg + 15 read the 4 pixels x 16 values block at pixel rows
    // 8 kg + 4 t + q (q = (lane & 15) >> 2), values 4 p .. 4 p + 3 (p = lane & 3); lane receives value (lane & 15)
    const int kgq = 8 * (lane >> 4) + ((lane & 15) >> 2), p4 = 4 * (lane & 3);
    const int xlane = kgq * XPB + p4 * 2, dlane = kgq * DPB + p4 * 2;
    auto trx = [&](const char* base, int pitch) -> bf16x8 {
        const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base));
        const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(base + 4 * pitch));
        return bf16x8{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };
    // steps = A tiles; the A fragments of step s+2 are requested before the MFMAs of step s (ring of 3), the B fragments
    // of the tile up front; staging items of the next tiles sit behind the fragment reads of their step
    auto compute = [&](int buf, auto&& between) {
        const char* Xl = smem + buf * BUF_B + xlane;
        const char* Dl = smem + buf * BUF_B + NS * XPL + dlane + (wave * TW) * DPB;
        constexpr int DEPTH = 3;
        bf16x8 bv[NTILES][NS], av[DEPTH][NS];
        auto fetch_a = [&](int mt, bf16x8 (&f)[NS]) {
            const int ky = mt / MTX, mx = mt % MTX;
            const int kx = CI == 8 ? 2 * mx : mx / (CI / 16), cb = CI == 8 ? 0 : (mx % (CI / 16)) * 16;   // tap column, channel base
#pragma unroll
            for (int p = 0; p < NS; ++p) f[p] = trx(Xl + p * XPL + ((wave + ky) * IW + kx) * XPB + cb * 2, XPB);
        };
#pragma unroll
        for (int nt = 0; nt < NTILES; ++nt)
#pragma unroll
            for (int p = 0; p < NS; ++p) bv[nt][p] = trx(Dl + p * DPL + nt * 32, DPB);
        fetch_a(0, av[0]);
        if (MTILES > 1) fetch_a(1, av[1]);
#pragma unroll
        for (int mt = 0; mt < MTILES; ++mt) {
            if (mt + 2 < MTILES) fetch_a(mt + 2, av[(mt + 2) % DEPTH]);
            const bf16x8 (&a)[NS] = av[mt % DEPTH];
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                f32x4 c = acc[mt * NTILES + nt];
                if constexpr (NS == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bv[nt][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], bv[nt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bv[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bv[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], bv[nt][0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], bv[nt][0], c, 0, 0, 0);
                acc[mt * NTILES + nt] = c;
            }
            between(mt);
        }
        constexpr int NPROD = (NS == 3 ? 6 : 1) * NTILES, RPF = 2 * NS;
        constexpr int VPS = (NIT * (NS == 3 ? 60 : 36) + MTILES - 1) / MTILES, IPS = (NIT + MTILES - 1) / MTILES;   // items per step (at most)
        __builtin_amdgcn_sched_group_barrier(0x100, (NTILES + (MTILES > 1 ? 2 : 1)) * RPF, 0);
#pragma unroll
        for (int mt = 0; mt < MTILES; ++mt) {
            if (mt + 2 < MTILES) __builtin_amdgcn_sched_group_barrier(0x100, RPF, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NPROD, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPS, 0);
            if (((mt + 1) * NIT) / MTILES != (mt * NIT) / MTILES) {
                __builtin_amdgcn_sched_group_barrier(0x200, NS * IPS, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, (GB ? 4 : 2) * IPS, 0);
            }
        }
    };

    Regs R;
    const int t0 = blockIdx.x, step = A.npb, tend = A.total_tiles, tlast = tend - 1;
    if (t0 < tend) {
        load(t0, R);
        store(t0, R, 0, true);
        load(t0 + step < tend ? t0 + step : tlast, R);
    }
    __syncthreads();
    int buf = 0;
    for (int tl = t0; tl < tend; tl += step, buf ^= 1) {
        const int t1 = tl + step, t2 = tl + 2 * step;
        const Geo g1 = geo_of(t1 < tend ? t1 : tlast), g2 = geo_of(t2 < tend ? t2 : tlast);
        const bool live = t1 < tend;
        compute(buf, [&](int mt) {                           // items [mt NIT / MTILES, (mt + 1) NIT / MTILES) behind step mt
#pragma unroll
            for (int i = 0; i < NIT; ++i)
                if (i >= (mt * NIT) / MTILES && i < ((mt + 1) * NIT) / MTILES) {
                    store_item(i, g1, R, buf ^ 1, live);     // tile t+1: registers -> split -> LDS (other buffer)
                    load_item(i, g2, R);                     // tile t+2: the same registers, requested right away
                }
        });
        __syncthreads();
    }

    // ---- 4-wave sum, tile by tile, through LDS (fixed order), then the slab; bias gradient = column sums of dz ----
    float* const red = reinterpret_cast<float*>(smem);
    const size_t wsize = (size_t)TAPS * A.Cin * A.Cout;
    float* out = A.part + (size_t)blockIdx.x * (wsize + A.Cout);
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int mt = t / NTILES, nt = t % NTILES, ky = mt / MTX, mx = mt % MTX;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        {
            const int r = tid >> 6, ln = tid & 63;                           // 256 threads = 4 regs x 64 lanes
            const float sv = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
            const int col = ln & 15, row = 4 * (ln >> 4) + r;
            int kx, ci;
            if constexpr (CI == 8) { kx = 2 * mx + (row >> 3); ci = row & 7; }
            else { kx = mx / (CI / 16); ci = (mx % (CI / 16)) * 16 + row; }
            const int co = nt * 16 + col;
            if (kx < KH && co < CO)
                out[((size_t)(ky * KH + kx) * A.Cin + ci) * A.Cout + co] = sv;
        }
    }
    __syncthreads();
    float* const bs = red + 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) bs[tid * 8 + i] = bsum[i];
    __syncthreads();
    if (tid < CO) {
        const int oct = tid >> 3, comp = tid & 7;
        float sv = 0.f;
        for (int t = oct; t < kBlock; t += OD) sv += bs[t * 8 + comp];
        out[wsize + tid] = sv;
    }
}

}  // namespace oct
