// VALU convolution for the THIN full-resolution layers (8 output channels).
//
// With M = 8 every 16x16 MFMA is half padding, while the f32 VALU has the same peak rate as the f32 MFMA on
// gfx950 (64 flop/clk/SIMD) and no tile padding -- so these HBM-bound layers run on the VALU:
//   * block tile 8 rows x 64 cols, one thread = 2 adjacent pixels x 8 output channels (16 accumulators);
//   * input tile (all <=16 input channels, halo included) in PLANAR LDS [c][row][col], transform applied on load;
//     a thread reads its 3x4 window per channel with 8-byte ds_reads, conflict-free;
//   * weights are wave-uniform -> scalar loads, used as SGPR operands of v_fma (no LDS, no VGPRs);
//   * persistent over pixel tiles with the next tile's global loads in flight during the FMA loop;
//   * same addressing modes / epilogues / argument block as the MFMA kernel (kernels_igemm.hpp).
#pragma once
#include "common.hpp"
#include "kernels_igemm.hpp"

namespace oct {

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
    struct Buf { float4 v[NPF]; };     // one tile's worth of prefetched registers
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
            pf.v[k] = ok ? lda4<AT>(tb + (ly * A.Wi + lx) * Csrc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // write the prefetched tile to LDS with the transform applied
    __device__ __forceinline__ void store(const IgemmArgs& A, const TileOrg& o, const Buf& pf) {
        int iy0, ix0; origin(o, iy0, ix0);
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= A.Hi && ix0 + IW <= A.Wi;
        const int basepix = (o.b * A.Hi + iy0) * A.Wi + ix0;                      // only used for the dropout element index
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            if (lxy[k] < 0) continue;
            const int ly = lxy[k] >> 8, lx = lxy[k] & 255;
            float4 v = pf.v[k];
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
            float* d = lds_q + ly * IWP + lx;
            d[0] = v.x; d[PLANE] = v.y; d[2 * PLANE] = v.z; d[3 * PLANE] = v.w;
        }
    }
};

// grid (nblk, 1, 1); requires Mout == 8, Cin <= 16 (Cin % 4 == 0), AMODE in {A_NORMAL (KH=3), A_UPF (KH=2)}
template <int KH, int AMODE, int EPI, int CMAX, typename AT>
__global__ __launch_bounds__(kBlock) void conv_thin8_k(const IgemmArgs A, const float* __restrict__ wgt, AT* __restrict__ outp) {
    constexpr int TH = 8, TW = 64, M = 8, PX = 2;
    constexpr int IH = AMODE == A_NORMAL ? TH + KH - 1 : TH / 2 + 1;
    constexpr int IW = AMODE == A_NORMAL ? TW + KH - 1 : TW / 2 + 1;
    constexpr int IWP = (IW + 1) & ~1;                     // even row stride: 8-byte aligned window reads
    constexpr int PLANE = IH * IWP;
    __shared__ float Is[CMAX * PLANE];
    __shared__ float red[256];

    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;     // thread -> pixels (ty, 2*tx .. 2*tx+1)

    ThinStager<CMAX, IH, IW, IWP, PLANE, AMODE, TH, TW, AT> st;
    st.init(A, Is);
    TileWalk<TH, TW> walk;
    walk.init(A.tiles, A.tiles_x, A.total_tiles);

    float bias[M];
#pragma unroll
    for (int m = 0; m < M; ++m) bias[m] = EPI == EPI_FWD ? A.bias[A.m_off + m] : 0.f;
    float s1[M], s2[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { s1[m] = 0.f; s2[m] = 0.f; }

    TileOrg cur = walk.first(A.tiles);
    typename decltype(st)::Buf pf;
    if (walk.tl0 < walk.tlend) st.load(A, cur, pf);
    for (int tl = walk.tl0; tl < walk.tlend; tl += walk.step) {
        __syncthreads();                    // every wave has finished reading the previous tile image
        st.store(A, cur, pf);
        __syncthreads();
        const TileOrg nxt = walk.next(cur);
        if (tl + walk.step < walk.tlend) st.load(A, nxt, pf);
        const int b = cur.b, y0 = cur.ty * TH, x0 = cur.tx * TW;
        cur = nxt;
        const int y = y0 + ty, x = x0 + 2 * tx;
        const bool rowok = y < A.Ho;
        float4 zq[PX][2];
        if constexpr (EPI == EPI_MASK) {    // producer's z for the ReLU mask: in flight during the FMA loop
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const bool ok = rowok && x + p < A.Wo;
                const AT* zp = reinterpret_cast<const AT*>(A.zin) + (((size_t)b * A.Ho + (ok ? y : 0)) * A.Wo + (ok ? x + p : 0)) * M;
                zq[p][0] = ok ? lda4<AT>(zp) : make_float4(0.f, 0.f, 0.f, 0.f);
                zq[p][1] = ok ? lda4<AT>(zp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }

        float acc[PX][M];
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int m = 0; m < M; ++m) acc[p][m] = bias[m];

        for (int c = 0; c < A.Cin; ++c) {
            const float* plane = Is + c * PLANE;
            if constexpr (AMODE == A_NORMAL) {
                float xw[KH][PX + KH - 1];   // 3 x 4 window of this thread's two pixels
#pragma unroll
                for (int r = 0; r < KH; ++r) {
                    const float2 lo = *reinterpret_cast<const float2*>(plane + (ty + r) * IWP + 2 * tx);
                    const float2 hi = *reinterpret_cast<const float2*>(plane + (ty + r) * IWP + 2 * tx + 2);
                    xw[r][0] = lo.x; xw[r][1] = lo.y; xw[r][2] = hi.x; xw[r][3] = hi.y;
                }
#pragma unroll
                for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KH; ++kx) {
                        const float* wr = wgt + ((size_t)(ky * KH + kx) * A.Cin + c) * A.w_ld + A.m_off;   // wave-uniform, noalias -> s_load
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            const float wv = wr[m];
                            acc[0][m] = fmaf(xw[ky][kx], wv, acc[0][m]);
                            acc[1][m] = fmaf(xw[ky][kx + 1], wv, acc[1][m]);
                        }
                    }
            } else {   // A_UPF: 2x2 conv over the nearest-upsampled tensor; low-res window 2 x 2
                float xl[2][2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int lr = (ty + r) >> 1;                  // rows (ty, ty+1) >> 1
                    xl[r][0] = plane[lr * IWP + tx]; xl[r][1] = plane[lr * IWP + tx + 1];
                }
#pragma unroll
                for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) {
                        const float* wr = wgt + ((size_t)(ky * 2 + kx) * A.Cin + c) * A.w_ld + A.m_off;
                        // pixel p = 2*tx + p reads up-sampled column (2*tx + p + kx) >> 1 = tx + ((p + kx) >> 1)
#pragma unroll
                        for (int m = 0; m < M; ++m) {
                            const float wv = wr[m];
                            acc[0][m] = fmaf(xl[ky][kx >> 1], wv, acc[0][m]);
                            acc[1][m] = fmaf(xl[ky][(1 + kx) >> 1], wv, acc[1][m]);
                        }
                    }
            }
        }

#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const bool valid = rowok && x + p < A.Wo;
            const size_t pix = valid ? ((size_t)b * A.Ho + y) * A.Wo + x + p : 0;
            float v[M];
#pragma unroll
            for (int m = 0; m < M; ++m) v[m] = acc[p][m];
            if constexpr (EPI == EPI_FWD) {
#pragma unroll
                for (int m = 0; m < M; ++m) { const float u = valid ? v[m] : 0.f; s1[m] += u; s2[m] += u * u; }
            } else if constexpr (EPI == EPI_MASK) {
                const float zz[M] = {zq[p][0].x, zq[p][0].y, zq[p][0].z, zq[p][0].w, zq[p][1].x, zq[p][1].y, zq[p][1].z, zq[p][1].w};
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float yv = fmaf(A.bnin[BN_A * M + m], zz[m], A.bnin[BN_B * M + m]);
                    float gv = v[m];
                    if (A.drop_out) gv *= drop_mul(A.drop, (uint32_t)(pix * M + m));
                    gv = (valid && yv > 0.f) ? gv : 0.f;
                    const float xh = (zz[m] - A.bnin[BN_MEAN * M + m]) * A.bnin[BN_RSTD * M + m];
                    v[m] = gv; s1[m] += gv; s2[m] += gv * xh;
                }
            }
            if (valid) {
                sta4<AT>(outp + pix * M, make_float4(v[0], v[1], v[2], v[3]));
                sta4<AT>(outp + pix * M + 4, make_float4(v[4], v[5], v[6], v[7]));
            }
        }
    }
    if constexpr (EPI != EPI_RAW) {
        if (A.part) {
            float* out = A.part + (size_t)blockIdx.x * (2 * M);
            block_reduce_store<M>(s1, red, out, M);
            block_reduce_store<M>(s2, red, out + M, M);
        }
    }
}

}  // namespace oct
