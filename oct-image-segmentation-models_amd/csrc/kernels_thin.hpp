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

// (TileWalk / ThinStager: kernels_igemm.hpp -- shared with the persistent MFMA kernel)

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
        typename Raw4<AT>::type zq[PX][2];   // raw: widened after the FMA loop
        if constexpr (EPI == EPI_MASK) {    // producer's z for the ReLU mask: in flight during the FMA loop
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const bool ok = rowok && x + p < A.Wo;
                const AT* zp = reinterpret_cast<const AT*>(A.zin) + (((size_t)b * A.Ho + (ok ? y : 0)) * A.Wo + (ok ? x + p : 0)) * M;
                zq[p][0] = ok ? ldraw4<AT>(zp) : raw_zero4<AT>();
                zq[p][1] = ok ? ldraw4<AT>(zp + 4) : raw_zero4<AT>();
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
                const float4 z0 = widen4(zq[p][0]), z1 = widen4(zq[p][1]);
                const float zz[M] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
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
