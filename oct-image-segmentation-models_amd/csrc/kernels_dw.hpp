// MFMA backward-weights kernels:  dW[tap][ci][co] = sum_pixels  X(pixel + tap)[ci] * dz(pixel)[co]
//   MFMA rows (M) = input channels (or flattened (tap, ci) for thin layers), cols (N) = output channels,
//   K = pixels.  Both operands are staged NHWC in LDS (channel fastest), which is exactly the operand order the
//   f32 MFMA wants: 32 (16) consecutive lanes read 32 (16) consecutive floats -> conflict-free ds_read_b32.
// Each block walks a strided set of pixel tiles accumulating in registers and writes ONE partial
// [tap][ci][co](+bias) slab; reduce_partials_k sums the slabs in a fixed order (deterministic, no atomics).
#pragma once
#include "common.hpp"
#include "kernels_bwd.hpp"
#include "kernels_igemm.hpp"

namespace oct {

// ---- software-pipelined staging: the global loads of tile t+1 are issued (into registers) before the MFMA loop of
// tile t and written to LDS (with the consumer-side transform) after it, so HBM/L2 latency hides under compute ----

// Per-thread invariants are set up ONCE (init): a thread always serves the same channel quad of X and of dz (kBlock
// is a multiple of CIC/4 and COC/4), so its source tensor (concat aware), BN affine and LDS column never change; per
// slot only the packed local pixel (ly, lx) is kept.  Per tile: one wave-uniform base pointer, and interior tiles skip
// every bounds test.
// GB: `dz` is the masked gradient g'; the BN-backward transform dz = ga g' + gb z + gd is applied while the tile is stored.
template <int CIC, int COC, int IH, int IW, bool UP, int KH, int TH, typename AT, int XSTRIDE = CIC, bool GB = false>
struct TileStager {
    static constexpr int QX = CIC / 4, PPX = kBlock / QX, NPX = IH * IW, NX = (NPX + PPX - 1) / PPX;   // X tile
    static constexpr int QD = COC / 4, PPD = kBlock / QD, NPD = TH * 32, ND = (NPD + PPD - 1) / PPD;   // dz tile
    typename Raw4<AT>::type xr[NX], dr[ND], zr[GB ? ND : 1];     // raw prefetch registers (widened in store())
    float4 fa, fb, ga, gb, gd;
    const AT* __restrict__ xsrc; const AT* __restrict__ dsrc; const AT* __restrict__ zsrc;
    int lxy[NX], dxy[ND];
    int Cs, cc, xq4, dq4; bool cok, dok;

    __device__ __forceinline__ void init(const ConvBwdWArgs& A, int ci0, int co0) {
        const int tid = threadIdx.x, c = ci0 + 4 * (tid % QX);
        const bool two = (A.flags & F_TWO) && c >= A.C0;
        Cs = two ? A.C1 : A.C0; cc = two ? c - A.C0 : c; cok = c < A.Cin; xq4 = 4 * (tid % QX);
        xsrc = (two ? reinterpret_cast<const AT*>(A.x1) : reinterpret_cast<const AT*>(A.x0)) + cc;
        fa = make_float4(1.f, 1.f, 1.f, 1.f); fb = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((A.flags & F_AFF) && cok) { const float* ab = two ? A.ab1 : A.ab0; fa = ld4(ab + cc); fb = ld4(ab + Cs + cc); }
#pragma unroll
        for (int k = 0; k < NX; ++k) { const int p = tid / QX + k * PPX; lxy[k] = p < NPX ? ((p / IW) << 8) | (p % IW) : -1; }
        const int dc = co0 + 4 * (tid % QD);
        dok = dc < A.Cout; dq4 = 4 * (tid % QD);
        dsrc = reinterpret_cast<const AT*>(A.dz) + dc;
        zsrc = GB ? reinterpret_cast<const AT*>(A.zf) + dc : nullptr;
        ga = gb = gd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (GB && dok) { ga = ld4(A.bnf + BN_GA * A.Cout + dc); gb = ld4(A.bnf + BN_GB * A.Cout + dc); gd = ld4(A.bnf + BN_GD * A.Cout + dc); }
#pragma unroll
        for (int k = 0; k < ND; ++k) { const int p = tid / QD + k * PPD; dxy[k] = p < NPD ? ((p / 32) << 8) | (p % 32) : -1; }
    }
    static __device__ __forceinline__ void origin(int y0, int x0, int& iy0, int& ix0) {
        iy0 = UP ? y0 / 2 : y0 - (KH - 1) / 2; ix0 = UP ? x0 / 2 : x0 - (KH - 1) / 2;
    }
    // issue the global loads of tile (b, y0, x0); out-of-range elements load nothing and become zeros
    __device__ __forceinline__ void load(const ConvBwdWArgs& A, int b, int y0, int x0, int /*ci0*/, int /*co0*/) {
        const int Hs = UP ? A.H >> 1 : A.H, Ws = UP ? A.W >> 1 : A.W;
        int iy0, ix0; origin(y0, x0, iy0, ix0);
        const AT* __restrict__ tb = xsrc + (((long long)b * Hs + iy0) * Ws + ix0) * Cs;     // wave-uniform (may point into the halo)
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= Hs && ix0 + IW <= Ws;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int ly = lxy[k] >> 8, lx = lxy[k] & 255;
            bool ok = cok && lxy[k] >= 0;
            if (!interior) ok = ok && (unsigned)(iy0 + ly) < (unsigned)Hs && (unsigned)(ix0 + lx) < (unsigned)Ws;
            xr[k] = ok ? ldraw4<AT>(tb + (ly * Ws + lx) * Cs) : raw_zero4<AT>();
        }
        const long long doff = (((long long)b * A.H + y0) * A.W + x0) * A.Cout;
        const AT* __restrict__ db = dsrc + doff;
        const bool dint = y0 + TH <= A.H && x0 + 32 <= A.W;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int py = dxy[k] >> 8, px = dxy[k] & 255;
            bool ok = dok && dxy[k] >= 0;
            if (!dint) ok = ok && y0 + py < A.H && x0 + px < A.W;
            dr[k] = ok ? ldraw4<AT>(db + (py * A.W + px) * A.Cout) : raw_zero4<AT>();
            if constexpr (GB) zr[k] = ok ? ldraw4<AT>(zsrc + doff + (py * A.W + px) * A.Cout) : raw_zero4<AT>();
        }
    }
    // write the loaded tile to LDS, applying the input transform; accumulates dz column sums (bias gradient)
    __device__ __forceinline__ void store(const ConvBwdWArgs& A, float* Xs, float* Ds, int b, int y0, int x0, int /*ci0*/,
                                          float4& bsum) {
        const int Hs = UP ? A.H >> 1 : A.H, Ws = UP ? A.W >> 1 : A.W;
        int iy0, ix0; origin(y0, x0, iy0, ix0);
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= Hs && ix0 + IW <= Ws;
        const int basepix = (b * Hs + iy0) * Ws + ix0;                              // dropout element index only
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            if (lxy[k] < 0) continue;
            const int ly = lxy[k] >> 8, lx = lxy[k] & 255;
            float4 v = widen4(xr[k]);
            bool in = cok;
            if (!interior) in = in && (unsigned)(iy0 + ly) < (unsigned)Hs && (unsigned)(ix0 + lx) < (unsigned)Ws;
            if (A.flags & F_AFF) {       // out-of-range stays exactly zero (padding is applied after the activation)
                v.x = in ? fmaxf(fmaf(fa.x, v.x, fb.x), 0.f) : 0.f; v.y = in ? fmaxf(fmaf(fa.y, v.y, fb.y), 0.f) : 0.f;
                v.z = in ? fmaxf(fmaf(fa.z, v.z, fb.z), 0.f) : 0.f; v.w = in ? fmaxf(fmaf(fa.w, v.w, fb.w), 0.f) : 0.f;
            }
            if ((A.flags & F_DROP) && in) {
                const uint32_t el = (uint32_t)((basepix + ly * Ws + lx) * Cs + cc);
                v.x *= drop_mul(A.drop, el); v.y *= drop_mul(A.drop, el + 1);
                v.z *= drop_mul(A.drop, el + 2); v.w *= drop_mul(A.drop, el + 3);
            }
            st4(Xs + (ly * IW + lx) * XSTRIDE + xq4, v);
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (dxy[k] < 0) continue;
            const int py = dxy[k] >> 8, px = dxy[k] & 255;
            float4 d = widen4(dr[k]);
            if constexpr (GB) {       // (elements past the image / channel range: g' = z = 0 and ga = gb = gd = 0 -> 0 only if
                                      //  the coefficients are zero too, so out-of-image pixels are forced to zero explicitly)
                const float4 z = widen4(zr[k]);
                const bool in = dok && y0 + py < A.H && x0 + px < A.W;
                d.x = in ? dz_as_stored<AT>(bn_bwd_apply1(ga.x, gb.x, gd.x, d.x, z.x)) : 0.f;
                d.y = in ? dz_as_stored<AT>(bn_bwd_apply1(ga.y, gb.y, gd.y, d.y, z.y)) : 0.f;
                d.z = in ? dz_as_stored<AT>(bn_bwd_apply1(ga.z, gb.z, gd.z, d.z, z.z)) : 0.f;
                d.w = in ? dz_as_stored<AT>(bn_bwd_apply1(ga.w, gb.w, gd.w, d.w, z.w)) : 0.f;
            }
            st4(Ds + (py * 32 + px) * COC + dq4, d);
            bsum.x += d.x; bsum.y += d.y; bsum.z += d.z; bsum.w += d.w;
        }
    }
};

// bias gradient = column sums of dz: reduce the per-thread quads through LDS (threads with equal tid % (COC/4) share a quad)
template <int COC>
__device__ __forceinline__ void bias_reduce(const ConvBwdWArgs& A, float* scratch /*>= 1024 floats*/, const float4& bsum,
                                   float* out_bias, int co0, bool write) {
    __syncthreads();
    st4(scratch + threadIdx.x * 4, bsum);
    __syncthreads();
    if ((int)threadIdx.x < COC) {
        const int quad = threadIdx.x / 4, comp = threadIdx.x % 4;
        float s = 0.f;
        for (int t = quad; t < kBlock; t += COC / 4) s += scratch[t * 4 + comp];
        if (write && co0 + (int)threadIdx.x < A.Cout) out_bias[co0 + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// thin layers: v_mfma_f32_16x16x4_f32, M = flattened (tap, ci) rows in tiles of 16, N = 16 output channels.
// Every wave owns ALL M tiles for a quarter of the tile's pixel rows; 4-wave sum through LDS at the end.
// grid (npb, ceil(Cin/CIC), ceil(Cout/16))
// ---------------------------------------------------------------------------------------------------------------
// DC = channels of dz that are STAGED (16, or 8 for layers with 8 output channels): with DC = 8 a pixel of the dz tile
// is 8 floats, lanes 8-15 of a B operand read the next pixel's channels (finite values: they fill MFMA columns 8-15,
// which are never written out), no staging slot and no prefetch register is spent on absent channels.
template <int KH, int CIC, bool UP, typename AT, bool GB = false, int DC = 16>
__global__ __launch_bounds__(kBlock, DC == 8 ? 4 : 3) void conv_dw16_k(const ConvBwdWArgs A) {   // 3 (4) waves/SIMD: <= 168 (128) VGPRs
    constexpr int TH = 8, TW = 32, TAPS = KH * KH, MROWS = TAPS * CIC, MTILES = (MROWS + 15) / 16;
    constexpr int IH = UP ? TH / 2 + 1 : TH + KH - 1, IW = UP ? TW / 2 + 1 : TW + KH - 1;
    constexpr int XS = IH * IW * CIC, DS = TH * TW * DC + (16 - DC), RED = 4 * MTILES * 256;
    static_assert(DC == 16 || DC == 8, "dz staging width");
    constexpr int LDSN = (XS + DS > RED ? XS + DS : RED) > 1024 ? (XS + DS > RED ? XS + DS : RED) : 1024;
    __shared__ float lds[LDSN];
    float* Xs = lds; float* Ds = lds + XS;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * CIC, co0 = blockIdx.z * 16;

    int aoff[MTILES], aky[MTILES], akx[MTILES], aci[MTILES]; bool aval[MTILES];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) {
        const int mrow = mt * 16 + i, tap = mrow / CIC;
        aval[mt] = mrow < MROWS; aci[mt] = mrow % CIC; aky[mt] = tap / KH; akx[mt] = tap % KH;
        aoff[mt] = (aky[mt] * IW + akx[mt]) * CIC + aci[mt];
    }
    f32x4 acc[MTILES];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    TileStager<CIC, DC, IH, IW, UP, KH, TH, AT, CIC, GB> st;
    st.init(A, ci0, co0);
    if constexpr (DC < 16) { if (tid < 16 - DC) Ds[TH * TW * DC + tid] = 0.f; }     // the last pixel's "columns 8-15"
    auto tile_of = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / A.tiles; const int tile = tl % A.tiles;
        x0 = (tile % A.tiles_x) * TW; y0 = (tile / A.tiles_x) * TH;
    };
    {
        int b, y0, x0;
        if ((int)blockIdx.x < A.total_tiles) { tile_of(blockIdx.x, b, y0, x0); st.load(A, b, y0, x0, ci0, co0); }
    }
    for (int tl = blockIdx.x; tl < A.total_tiles; tl += A.npb) {
        int b, y0, x0;
        tile_of(tl, b, y0, x0);
        __syncthreads();                                   // every wave is done with the previous tile's LDS image
        st.store(A, Xs, Ds, b, y0, x0, ci0, bsum);
        __syncthreads();
        if (tl + A.npb < A.total_tiles) {                  // next tile's loads fly while this tile computes
            int nb, ny0, nx0;
            tile_of(tl + A.npb, nb, ny0, nx0);
            st.load(A, nb, ny0, nx0, ci0, co0);
        }
        // operand address = lane base (once per tile) + row offset + an immediate of the unrolled k-step loop;
        // the operands of step ks+1 are read while the MFMAs of step ks issue
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) {
            const int rr = wave + 4 * rs;
            const float* dbr = Ds + (rr * TW + kk) * DC + i;
            const float* xr[MTILES];
#pragma unroll
            for (int mt = 0; mt < MTILES; ++mt) {
                if constexpr (UP) xr[mt] = Xs + (((rr + aky[mt]) >> 1) * IW + ((kk + akx[mt]) >> 1)) * CIC + aci[mt];
                else xr[mt] = Xs + aoff[mt] + (rr * IW + kk) * CIC;
            }
            auto load = [&](int ks, float (&a)[MTILES], float& bv) {
                bv = dbr[4 * ks * DC];
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt) a[mt] = xr[mt][(UP ? 2 * ks : 4 * ks) * CIC];
            };
            float an[MTILES], bn;
            load(0, an, bn);
#pragma unroll
            for (int ks = 0; ks < TW / 4; ++ks) {
                float a[MTILES]; const float bv = bn;
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt) a[mt] = aval[mt] ? an[mt] : 0.f;
                if (ks + 1 < TW / 4) load(ks + 1, an, bn);
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    // ---- 4-wave sum through LDS, then the slab ----
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(wave * MTILES + mt) * 256 + r * 64 + lane] = acc[mt][r];
    __syncthreads();
    const size_t wsize = (size_t)TAPS * A.Cin * A.Cout;
    float* out = A.part + (size_t)blockIdx.x * (wsize + A.Cout);
    for (int idx = tid; idx < MTILES * 256; idx += kBlock) {
        const int mt = idx / 256, rem = idx % 256, r = rem / 64, ln = rem % 64;
        const float s = (lds[(0 * MTILES + mt) * 256 + rem] + lds[(1 * MTILES + mt) * 256 + rem]) +
                        (lds[(2 * MTILES + mt) * 256 + rem] + lds[(3 * MTILES + mt) * 256 + rem]);
        const int col = ln & 15, row = 4 * (ln >> 4) + r, mrow = mt * 16 + row;
        const int tap = mrow / CIC, ci = mrow % CIC;
        if (mrow < MROWS && ci0 + ci < A.Cin && co0 + col < A.Cout)
            out[((size_t)tap * A.Cin + ci0 + ci) * A.Cout + co0 + col] = s;
    }
    bias_reduce<DC>(A, lds, bsum, out + wsize, co0, blockIdx.y == 0);
}

// ---------------------------------------------------------------------------------------------------------------
// 3x3 layers with 8 OUTPUT channels (full resolution): pixel-pair packing, the backward-weights counterpart of
// kernels_pair.hpp.  With N = 8 output channels a 16-column MFMA is half padding; here the 16 columns are
// (pixel parity j, co) and K runs over pixel PAIRS (2n, 2n+1) of a row:
//     D[(ky,u,ci)][(j,co)] = sum_{y,n} X(y+ky-1, 2n+u-1)[ci] * dz(y, 2n+j)[co],   u = 0..3 (window column of the pair)
//     dW[ky][kx][ci][co]   = D[(ky,kx,ci)][(0,co)] + D[(ky,kx+1,ci)][(1,co)]
// 12*CIC rows (6 or 12 M tiles of 16), every column useful: 9/12 = 75 % useful MFMA work (45-50 % for the padded
// form), i.e. 0.75 (CIC = 8) / 1.5 (CIC = 16) MFMAs per pixel instead of 1.25 / 2.25.
// Operands straight from the NHWC LDS tiles, conflict-free: B = dz[(row, 2n+j)][co] -- 64 lanes read 64 consecutive
// floats; A = X[(row+ky, 2n+u)][ci] -- consecutive for CIC = 8, pixel stride padded to 20 floats for CIC = 16.
// Every wave owns all M tiles for a quarter of the tile's pixel rows; 4-wave sum through LDS at the end.
// grid (npb, ceil(Cin/CIC), 1); requires Cout == 8, KH == 3, not an up-conv.
// ---------------------------------------------------------------------------------------------------------------
template <int CIC, typename AT, bool GB = false>
__global__ __launch_bounds__(kBlock) void conv_dwpair8_k(const ConvBwdWArgs A) {
    constexpr int KH = 3, TH = 8, TW = 32, MROWS = 12 * CIC, MTILES = MROWS / 16, XST = CIC == 16 ? 20 : CIC;
    constexpr int IH = TH + 2, IW = TW + 2;
    constexpr int XS = IH * IW * XST, DS = TH * TW * 8, RED = MROWS * 16;
    constexpr int LDSN = (XS + DS > 4 * RED ? XS + DS : 4 * RED) > 1024 ? (XS + DS > 4 * RED ? XS + DS : 4 * RED) : 1024;
    __shared__ float lds[LDSN];
    float* Xs = lds; float* Ds = lds + XS;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * CIC;

    // A row of M tile mt held by this lane: m = mt*16 + i -> (ky, u, ci);  LDS offset of X(ky, u)[ci] relative to the pair
    int aoff[MTILES];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) {
        const int m = mt * 16 + i, ci = m % CIC, ku = m / CIC, u = ku & 3, ky = ku >> 2;
        aoff[mt] = (ky * IW + u) * XST + ci;
    }
    f32x4 acc[MTILES];
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    TileStager<CIC, 8, IH, IW, false, KH, TH, AT, XST, GB> st;
    st.init(A, ci0, 0);
    auto tile_of = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / A.tiles; const int tile = tl % A.tiles;
        x0 = (tile % A.tiles_x) * TW; y0 = (tile / A.tiles_x) * TH;
    };
    {
        int b, y0, x0;
        if ((int)blockIdx.x < A.total_tiles) { tile_of(blockIdx.x, b, y0, x0); st.load(A, b, y0, x0, ci0, 0); }
    }
    for (int tl = blockIdx.x; tl < A.total_tiles; tl += A.npb) {
        int b, y0, x0;
        tile_of(tl, b, y0, x0);
        __syncthreads();                                   // every wave is done with the previous tile's LDS image
        st.store(A, Xs, Ds, b, y0, x0, ci0, bsum);
        __syncthreads();
        if (tl + A.npb < A.total_tiles) {                  // next tile's loads fly while this tile computes
            int nb, ny0, nx0;
            tile_of(tl + A.npb, nb, ny0, nx0);
            st.load(A, nb, ny0, nx0, ci0, 0);
        }
        // lane's K slot = pixel pair n = 4 ks + kk; address = lane base + row offset + an immediate of the unrolled loop
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) {
            const int rr = wave + 4 * rs;
            const float* dbr = Ds + (rr * TW + 2 * kk) * 8 + i;               // i = (j, co): pixel 2n + j, channel co
            const float* xr[MTILES];
#pragma unroll
            for (int mt = 0; mt < MTILES; ++mt) xr[mt] = Xs + (rr * IW + 2 * kk) * XST + aoff[mt];
            auto load = [&](int ks, float (&a)[MTILES], float& bv) {
                bv = dbr[8 * ks * 8];
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt) a[mt] = xr[mt][8 * ks * XST];
            };
            float an[MTILES], bn;
            load(0, an, bn);
#pragma unroll
            for (int ks = 0; ks < TW / 8; ++ks) {
                float a[MTILES]; const float bv = bn;
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt) a[mt] = an[mt];
                if (ks + 1 < TW / 8) load(ks + 1, an, bn);
#pragma unroll
                for (int mt = 0; mt < MTILES; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bv, acc[mt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    // ---- 4-wave sum through LDS into D[m][16], then fold the two pixel parities into the 3x3 taps ----
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MTILES; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[wave * RED + (mt * 16 + 4 * kk + r) * 16 + i] = acc[mt][r];    // D row = 4*(lane>>4)+r, col = lane&15
    __syncthreads();
    for (int idx = tid; idx < RED; idx += kBlock) lds[idx] = (lds[idx] + lds[RED + idx]) + (lds[2 * RED + idx] + lds[3 * RED + idx]);
    __syncthreads();
    const size_t wsize = (size_t)9 * A.Cin * A.Cout;
    float* out = A.part + (size_t)blockIdx.x * (wsize + A.Cout);
    for (int idx = tid; idx < 9 * CIC * 8; idx += kBlock) {
        const int co = idx & 7, ci = (idx >> 3) % CIC, tap = idx / (8 * CIC), ky = tap / 3, kx = tap % 3;
        if (ci0 + ci < A.Cin)
            out[((size_t)tap * A.Cin + ci0 + ci) * A.Cout + co] =
                lds[((ky * 4 + kx) * CIC + ci) * 16 + co] + lds[((ky * 4 + kx + 1) * CIC + ci) * 16 + 8 + co];
    }
    bias_reduce<8>(A, lds, bsum, out + wsize, 0, blockIdx.y == 0);
}

// ---------------------------------------------------------------------------------------------------------------
// wide layers: v_mfma_f32_32x32x2_f32, block = CIC (32|64) input channels x 32 output channels x all taps.
// The (ci-tile, tap) units are dealt round-robin to the 4 waves; every wave sweeps all pixels of the tile for
// its own units, so no cross-wave reduction is needed.   grid (npb, Cin/CIC, Cout/32)
// ---------------------------------------------------------------------------------------------------------------
template <int KH, int CIC, bool UP, int TH, typename AT, bool GB = false>
__global__ __launch_bounds__(kBlock) void conv_dw32_k(const ConvBwdWArgs A) {
    constexpr int TW = 32, COC = 32, TAPS = KH * KH, MTB = CIC / 32, UNITS = MTB * TAPS, UPW = (UNITS + 3) / 4;
    constexpr int IH = UP ? TH / 2 + 1 : TH + KH - 1, IW = UP ? TW / 2 + 1 : TW + KH - 1;
    constexpr int XS = IH * IW * CIC, DS = TH * TW * COC;
    __shared__ float lds[(XS + DS) > 1024 ? (XS + DS) : 1024];
    float* Xs = lds; float* Ds = lds + XS;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, kk = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: unit guards become scalar branches
    const int ci0 = blockIdx.y * CIC, co0 = blockIdx.z * COC;

    int uky[UPW], ukx[UPW], umt[UPW];
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = (wave + 4 * k < UNITS) ? wave + 4 * k : UNITS - 1, tap = u / MTB;
        umt[k] = u % MTB; uky[k] = tap / KH; ukx[k] = tap % KH;
    }
    f32x16 acc[UPW];
#pragma unroll
    for (int k = 0; k < UPW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // lane bases of the MFMA operands (lane = (i, kk): row/column i of the 32-tile, pixel kk of the pair)
    const float* xb[UPW];
#pragma unroll
    for (int k = 0; k < UPW; ++k)
        xb[k] = Xs + (UP ? ((kk + ukx[k]) >> 1) : uky[k] * IW + ukx[k] + kk) * CIC + umt[k] * 32 + i;
    const float* const db = Ds + kk * COC + i;

    TileStager<CIC, COC, IH, IW, UP, KH, TH, AT, CIC, GB> st;
    st.init(A, ci0, co0);
    auto tile_of = [&](int tl, int& b, int& y0, int& x0) {
        b = tl / A.tiles; const int tile = tl % A.tiles;
        x0 = (tile % A.tiles_x) * TW; y0 = (tile / A.tiles_x) * TH;
    };
    {
        int b, y0, x0;
        if ((int)blockIdx.x < A.total_tiles) { tile_of(blockIdx.x, b, y0, x0); st.load(A, b, y0, x0, ci0, co0); }
    }
    for (int tl = blockIdx.x; tl < A.total_tiles; tl += A.npb) {
        int b, y0, x0;
        tile_of(tl, b, y0, x0);
        __syncthreads();                                   // every wave is done with the previous tile's LDS image
        st.store(A, Xs, Ds, b, y0, x0, ci0, bsum);
        __syncthreads();
        if (tl + A.npb < A.total_tiles) {                  // next tile's loads fly while this tile computes
            int nb, ny0, nx0;
            tile_of(tl + A.npb, nb, ny0, nx0);
            st.load(A, nb, ny0, nx0, ci0, co0);
        }
        {   // TH rows x 16 pixel pairs.  Every operand address is (per-unit lane base, set up once) + (row offset,
            // scalar) + (pair offset, an immediate of the unrolled pair loop): no per-step index arithmetic; the
            // operands of pair pp+1 are read from LDS while the MFMAs of pair pp issue.
            __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
            for (int rr = 0; rr < TH; ++rr) {
                const float* dbr = db + rr * TW * COC;
                const float* xr[UPW];
#pragma unroll
                for (int k = 0; k < UPW; ++k) xr[k] = xb[k] + (UP ? ((rr + uky[k]) >> 1) : rr) * (IW * CIC);
                auto load = [&](int pp, float (&a)[UPW], float& bv) {
                    bv = dbr[2 * pp * COC];
#pragma unroll
                    for (int k = 0; k < UPW; ++k) a[k] = xr[k][(UP ? pp : 2 * pp) * CIC];
                };
                float an[UPW], bn;
                load(0, an, bn);
#pragma unroll
                for (int pp = 0; pp < TW / 2; ++pp) {
                    float a[UPW]; const float bv = bn;
#pragma unroll
                    for (int k = 0; k < UPW; ++k) a[k] = an[k];
                    if (pp + 1 < TW / 2) load(pp + 1, an, bn);
#pragma unroll
                    for (int k = 0; k < UPW; ++k) {
                        if (4 * k + 3 < UNITS || wave + 4 * k < UNITS)   // compile-time true except for the last unit
                            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], bv, acc[k], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
    const size_t wsize = (size_t)TAPS * A.Cin * A.Cout;
    float* out = A.part + (size_t)blockIdx.x * (wsize + A.Cout);
#pragma unroll
    for (int k = 0; k < UPW; ++k) {
        const int u = wave + 4 * k;
        if (u < UNITS) {
            const int tap = u / MTB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;    // D: row = ci within the 32-tile, col = co
                const int ci = ci0 + umt[k] * 32 + row, co = co0 + i;
                if (ci < A.Cin && co < A.Cout) out[((size_t)tap * A.Cin + ci) * A.Cout + co] = acc[k][r];
            }
        }
    }
    bias_reduce<COC>(A, lds, bsum, out + wsize, co0, blockIdx.y == 0);
}

}  // namespace oct
