// Pixel-pair-packed MFMA convolution for the 3x3 layers with 8 OUTPUT channels (the full-resolution layers).
//
// A 16x16 MFMA tile with 8 output channels as its rows is half padding.  Here the 16 rows are
// (pixel parity j, output channel co): one MFMA produces the 8 channels of TWO horizontally adjacent pixels
// (x, x+1), x even, for 16 such pairs (its 16 columns).  The two pixels share a 3-row x 4-column input window, so
//     K = (ky, u, ci)  with u = 0..3 the window column,      A[(j,co)][(ky,u,ci)] = w[ky][u-j][ci][co]  (0 if u-j not in 0..2)
//     B[(ky,u,ci)][pair n] = X[y+ky-1][2n+u-1][ci]
// 9 of the 12 window positions of every row carry a weight: 75 % of the MFMA is useful work (50 % for the padded
// tile), which puts the MFMA time of an 8->8 layer at the HBM time of its tensors, and the staging / epilogue VALU
// work runs beside the matrix pipe instead of competing with the FMAs for the vector ALU.
//   * a wave owns RPW groups of 4 output rows x 32 columns (4 accumulators per group: the 4 rows share the A operand
//     and, through the 3 kernel rows, their B operands); a block is NWY x NWX waves on a (4 RPW NWY) x (32 NWX) tile.
//     Measured (8->8 forward, B=32, 256x512): 2x2 waves x 1 group (8x64 tile) 83 us; 2x1 88; 1x1 (no cross-wave
//     barrier at all) 100; 2x2 x 2 groups (16x64) 91; forcing 4 waves/SIMD (spills) 101.  Knock-outs: without the MFMA
//     sweep 50 us (5.4 TB/s), without global loads/stores 89 us -- the LDS/MFMA/epilogue chain of a wave, not memory,
//     is what bounds it (MFMA pipe 55 % busy at 3 waves/SIMD);
//   * A operand (the weight matrix with its structural zeros) is built ONCE per persistent block in LDS, in lane
//     order: one conflict-free ds_read_b32 serves the 4 MFMAs (4 output rows) of a K-step;
//   * B operand: one ds_read_b32 per (window column, channel quad, input row) from the PLANAR LDS tile [c][row][col];
//     PLANE is odd so the four channel planes of a K-step spread over all banks (2 lanes per bank = the minimum);
//   * staging, tile walk, addressing modes, epilogues and the argument block are those of kernels_thin.hpp /
//     kernels_igemm.hpp; D layout: lane (n = lane&15, g = lane>>4) holds channels 4*(g&1)..+3 of pixel 2n + (g>>1).
#pragma once
#include "common.hpp"
#include "kernels_igemm.hpp"
#include "kernels_thin.hpp"

namespace oct {

// grid (nblk, 1, 1), block 64*NWY*NWX; requires KH == 3, A_NORMAL, Mout == 8, Cin <= CMAX (Cin % 4 == 0)
template <int EPI, int CMAX, int DEPTH, int NWY, int NWX, int RPW, typename AT>
__global__ __launch_bounds__(64 * NWY * NWX) void conv_pair8_k(const IgemmArgs A, const float* __restrict__ wgt, AT* __restrict__ outp) {
    constexpr int TH = 4 * RPW * NWY, TW = 32 * NWX, M = 8, NW = NWY * NWX, NT = 64 * NW;
    constexpr int IH = TH + 2, IW = TW + 2, IWP = IW;
    constexpr int PLANE = (IH * IWP) | 1;
    constexpr int Q = CMAX / 4, NSTEP = 3 * 4 * Q;
    __shared__ float Is[CMAX * PLANE];
    __shared__ float Aw[NSTEP * 64];
    __shared__ float red[NW * 16];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ry0 = 4 * RPW * (wave / NWX), cx0 = 32 * (wave % NWX);
    const int ch0 = 4 * (g & 1), pj = g >> 1;                  // this lane's output: channels ch0..ch0+3 of pixel 2n + pj

    ThinStager<CMAX, IH, IW, IWP, PLANE, A_NORMAL, TH, TW, AT, NT> st;
    st.init(A, Is);
    TileWalk<TH, TW> walk;
    walk.init(A.tiles, A.tiles_x, A.total_tiles);

    // A operand: row (j, co) = n, k = g within the step (ky, u, q); identical for the 4 waves -> wave 0..3 build a quarter each
    for (int e = tid; e < NSTEP * 64; e += NT) {
        const int sidx = e >> 6, ln = e & 63, nn = ln & 15, gg = ln >> 4;
        const int q = sidx % Q, u = (sidx / Q) & 3, ky = sidx / (4 * Q);
        const int kx = u - (nn >> 3), ci = 4 * q + gg;
        Aw[e] = (kx >= 0 && kx < 3 && ci < A.Cin) ? wgt[((size_t)(ky * 3 + kx) * A.Cin + ci) * A.w_ld + A.m_off + (nn & 7)] : 0.f;
    }
    float bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = EPI == EPI_FWD ? A.bias[A.m_off + ch0 + i] : 0.f;
    float bna[4], bnb[4], bnm[4], bnr[4];                      // producer's BN record of this lane's channels (EPI_MASK)
    if constexpr (EPI == EPI_MASK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bna[i] = A.bnin[BN_A * M + ch0 + i]; bnb[i] = A.bnin[BN_B * M + ch0 + i];
            bnm[i] = A.bnin[BN_MEAN * M + ch0 + i]; bnr[i] = A.bnin[BN_RSTD * M + ch0 + i];
        }
    }
    float s1[4], s2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[i] = 0.f; s2[i] = 0.f; }

    const float* const bbase = Is + g * PLANE + ry0 * IWP + cx0 + 2 * n;

    // one tile: MFMA sweep over the LDS image + epilogue
    auto process = [&](const TileOrg& o) {
#pragma unroll 1
      for (int rg = 0; rg < RPW; ++rg) {
        const int b = o.b, y0 = o.ty * TH + ry0 + 4 * rg, x = o.tx * TW + cx0 + 2 * n + pj;
        const float* const bb = bbase + 4 * rg * IWP;
        typename Raw4<AT>::type zq[4];       // raw: widened after the sweep
        if constexpr (EPI == EPI_MASK) {    // producer's z for the ReLU mask: in flight during the MFMA loop
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = y0 + r < A.Ho && x < A.Wo;
                const AT* zp = reinterpret_cast<const AT*>(A.zin) + (((size_t)b * A.Ho + (ok ? y0 + r : 0)) * A.Wo + (ok ? x : 0)) * M + ch0;
                zq[r] = ok ? ldraw4<AT>(zp) : raw_zero4<AT>();
            }
        }

        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = f32x4{bias[0], bias[1], bias[2], bias[3]};
        __builtin_amdgcn_s_setprio(2);      // waves in their MFMA phase issue ahead of waves that are staging
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float xin[6], aw[3];
#pragma unroll
                for (int t = 0; t < 6; ++t) xin[t] = bb[4 * q * PLANE + t * IWP + u];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) aw[ky] = Aw[((ky * 4 + u) * Q + q) * 64 + lane];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ky], xin[r + ky], acc[r], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);

#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + r;
            const bool valid = y < A.Ho && x < A.Wo;
            const size_t pix = valid ? ((size_t)b * A.Ho + y) * A.Wo + x : 0;
            float v[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            if constexpr (EPI == EPI_FWD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float t = valid ? v[i] : 0.f; s1[i] += t; s2[i] += t * t; }
            } else if constexpr (EPI == EPI_MASK) {
                const float4 zw = widen4(zq[r]);
                const float zz[4] = {zw.x, zw.y, zw.z, zw.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float yv = fmaf(bna[i], zz[i], bnb[i]);
                    float gv = v[i];
                    if (A.drop_out) gv *= drop_mul(A.drop, (uint32_t)(pix * M + ch0 + i));
                    gv = (valid && yv > 0.f) ? gv : 0.f;
                    const float xh = (zz[i] - bnm[i]) * bnr[i];
                    v[i] = gv; s1[i] += gv; s2[i] += gv * xh;
                }
            }
            if (valid) sta4<AT>(outp + pix * M + ch0, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    };

    // software pipeline over the block's tiles: while tile t is computed from LDS, tile t+1 waits in one register
    // buffer and (DEPTH 2) the loads of tile t+2 are in flight into the other -- the loop is unrolled by two so both
    // buffers are statically indexed
    typename decltype(st)::Buf pf0, pf1;
    const int stp = walk.step, end = walk.tlend;
    TileOrg o0 = walk.first(A.tiles), o1 = walk.next(o0);
    if (walk.tl0 < end) st.load(A, o0, pf0);
    if constexpr (DEPTH == 2) { if (walk.tl0 + stp < end) st.load(A, o1, pf1); }
    for (int tl = walk.tl0; tl < end; tl += 2 * stp) {
        __syncthreads();                    // every wave has finished reading the previous tile image
        st.store(A, o0, pf0);
        __syncthreads();
        const TileOrg o2 = walk.next(o1);
        if constexpr (DEPTH == 2) { if (tl + 2 * stp < end) st.load(A, o2, pf0); }
        else { if (tl + stp < end) st.load(A, o1, pf1); }
        process(o0);
        if (tl + stp >= end) break;
        __syncthreads();
        st.store(A, o1, pf1);
        __syncthreads();
        const TileOrg o3 = walk.next(o2);
        if constexpr (DEPTH == 2) { if (tl + 3 * stp < end) st.load(A, o3, pf1); }
        else { if (tl + 2 * stp < end) st.load(A, o2, pf0); }
        process(o1);
        o0 = o2; o1 = o3;
    }
    if constexpr (EPI != EPI_RAW) {
        if (A.part) {
            // lanes with equal (g & 1) hold the same channels: sum over n (lane bits 0-3) and pj (lane bit 5)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int o = 1; o <= 8; o <<= 1) { s1[i] += __shfl_xor(s1[i], o, 64); s2[i] += __shfl_xor(s2[i], o, 64); }
                s1[i] += __shfl_xor(s1[i], 32, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
            }
            __syncthreads();
            if ((lane & 0x2F) == 0) {       // lanes 0 (channels 0-3) and 16 (channels 4-7) of every wave
#pragma unroll
                for (int i = 0; i < 4; ++i) { red[wave * 16 + ch0 + i] = s1[i]; red[wave * 16 + 8 + ch0 + i] = s2[i]; }
            }
            __syncthreads();
            if (tid < 2 * M) {             // [0,8) = sum, [8,16) = second statistic; fixed summation order
                float t = red[tid];
#pragma unroll
                for (int w = 1; w < NW; ++w) t += red[w * 16 + tid];
                A.part[(size_t)blockIdx.x * (2 * M) + tid] = t;
            }
        }
    }
}

}  // namespace oct
