// Launchers of the MFMA conv kernels (forward and backward-data): pick the kernel family from the channel counts and the
// pixel tile from the grid size.  Included by the tu_conv_*.hip files, each of which instantiates launch_igemm for one
// (kernel size, addressing mode, epilogue) triple -- the device code of those instantiations is most of the library's
// compile time, so they are built in parallel.
#pragma once
#include <algorithm>
#include <cstdio>

#include "host.hpp"
#include "kernels_bx.hpp"
#include "kernels_igemm.hpp"
#include "kernels_pair.hpp"
#include "kernels_thin.hpp"

namespace octh {
using namespace oct;

// ---- fp32-pipe implicit GEMM ----
template <int SHAPE, int KH, int AMODE, int EPI, int TH, int MB, int WN>
int launch_igemm_geo(IgemmArgs a, const LaunchCtx& c, int* rows) {
    a.tiles_x = cdiv(a.Wo, 32); a.tiles = a.tiles_x * cdiv(a.Ho, TH);
    dim3 grid(a.tiles, cdiv(a.Mout, MB), c.B), block(kBlock);
    char nm[64]; snprintf(nm, sizeof nm, "conv_igemm_k<%d,%d,%d,%d,%d,%d,%d,%s>", SHAPE, KH, AMODE, EPI, TH, MB, WN, AT_NAME(a.act_bf16));
    ProfScope ps(c.s, nm, c.layer, c.flops, c.bytes);
    AT_DISPATCH(a.act_bf16, conv_igemm_k<SHAPE, KH, AMODE, EPI, TH, MB, WN, AT><<<grid, block, 0, c.s>>>(a));
    HIP_OK(hipGetLastError());
    *rows = c.B * a.tiles;   // one statistic partial row per (image, pixel tile)
    return 0;
}

// persistent pipelined variant (single K chunk): returns the number of statistic partial rows (= blocks along x)
template <int SHAPE, int KH, int AMODE, int EPI, int TH, int MB, int WN, int KCP>
int launch_igemm_p(IgemmArgs a, const LaunchCtx& c, int* rows) {
    a.tiles_x = cdiv(a.Wo, 32); a.tiles = a.tiles_x * cdiv(a.Ho, TH); a.total_tiles = c.B * a.tiles;
    const int nblk = std::min(a.total_tiles, c.o->igemm_p_blocks);    // ~5 resident blocks per CU
    dim3 grid(nblk, cdiv(a.Mout, MB), 1), block(kBlock);
    char nm[64]; snprintf(nm, sizeof nm, "conv_igemm_p_k<%d,%d,%d,%d,%d,%d,%d,%d,%s>", SHAPE, KH, AMODE, EPI, TH, MB, WN, KCP, AT_NAME(a.act_bf16));
    ProfScope ps(c.s, nm, c.layer, c.flops, c.bytes);
    AT_DISPATCH(a.act_bf16, conv_igemm_p_k<SHAPE, KH, AMODE, EPI, TH, MB, WN, KCP, AT><<<grid, block, 0, c.s>>>(a));
    HIP_OK(hipGetLastError());
    *rows = nblk;       // one statistic partial row per block
    return 0;
}

#ifndef PAIR_DEPTH
#define PAIR_DEPTH 1     // register prefetch depth of the 8-input-channel pair kernel (tiles in flight beyond the one in LDS)
#endif
// pixel-pair MFMA kernel for 3x3 layers with 8 output channels (see kernels_pair.hpp)
template <int EPI, int CMAX, int NWY, int NWX, int RPW>
int launch_pair8_geo(IgemmArgs a, const LaunchCtx& c, int* rows) {
    constexpr int NT = 64 * NWY * NWX, DEPTH = PAIR_DEPTH;
    a.tiles_x = cdiv(a.Wo, 32 * NWX); a.tiles = a.tiles_x * cdiv(a.Ho, 4 * RPW * NWY); a.total_tiles = c.B * a.tiles;
    static int occ[2] = {0, 0};
    const int bf = a.act_bf16 ? 1 : 0;
    if (!occ[bf]) {
        int nb = 0;
        AT_DISPATCH(bf, if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_pair8_k<EPI, CMAX, DEPTH, NWY, NWX, RPW, AT>, NT, 0) != hipSuccess) nb = 0);
        occ[bf] = nb < 1 ? 2 : nb;
    }
    const int nblk = std::min(a.total_tiles, occ[bf] * 256);
    char nm[80]; snprintf(nm, sizeof nm, "conv_pair8_k<%d,%d,%d,%d,%d,%d,%s>", EPI, CMAX, DEPTH, NWY, NWX, RPW, AT_NAME(a.act_bf16));
    ProfScope ps(c.s, nm, c.layer, c.flops, c.bytes);
    AT_DISPATCH(bf, conv_pair8_k<EPI, CMAX, DEPTH, NWY, NWX, RPW, AT><<<nblk, NT, 0, c.s>>>(a, a.w, reinterpret_cast<AT*>(a.out)));
    HIP_OK(hipGetLastError());
    *rows = nblk;
    return 0;
}
template <int EPI>
int launch_pair8(const IgemmArgs& a, const LaunchCtx& c, int* rows) {
    const int geo = c.o->pair_geo;    // NWY*100 + NWX*10 + RPW
    if (a.Cin <= 8) {
        if (geo == 111) return launch_pair8_geo<EPI, 8, 1, 1, 1>(a, c, rows);
        return launch_pair8_geo<EPI, 8, 2, 2, 1>(a, c, rows);
    }
    if (geo == 111) return launch_pair8_geo<EPI, 16, 1, 1, 1>(a, c, rows);
    return launch_pair8_geo<EPI, 16, 2, 2, 1>(a, c, rows);
}

// VALU kernel for 8-output-channel layers (see kernels_thin.hpp)
template <int KH, int AMODE, int EPI>
int launch_thin8(IgemmArgs a, const LaunchCtx& c, int* rows) {
    a.tiles_x = cdiv(a.Wo, 64); a.tiles = a.tiles_x * cdiv(a.Ho, 8); a.total_tiles = c.B * a.tiles;
    const int nblk = std::min(a.total_tiles, 1536);
    char nm[64]; snprintf(nm, sizeof nm, "conv_thin8_k<%d,%d,%d,%d,%s>", KH, AMODE, EPI, a.Cin <= 8 ? 8 : 16, AT_NAME(a.act_bf16));
    ProfScope ps(c.s, nm, c.layer, c.flops, c.bytes);
    if (a.Cin <= 8) AT_DISPATCH(a.act_bf16, conv_thin8_k<KH, AMODE, EPI, 8, AT><<<nblk, kBlock, 0, c.s>>>(a, a.w, reinterpret_cast<AT*>(a.out)));
    else AT_DISPATCH(a.act_bf16, conv_thin8_k<KH, AMODE, EPI, 16, AT><<<nblk, kBlock, 0, c.s>>>(a, a.w, reinterpret_cast<AT*>(a.out)));
    HIP_OK(hipGetLastError());
    *rows = nblk;
    return 0;
}

// ---- bf16-pipe implicit GEMM (kernels_bx.hpp): NS = 3 split products in fp32 mode, 1 in bf16 mode ----
// GB: the input is a masked gradient g' whose BN-backward transform is applied while it is staged (backward-data only)
template <int KH, int AMODE, int EPI, int TH, int MB, int NW, int NIMG = 2>
int launch_bx_nw(IgemmArgs a, const LaunchCtx& c, int* rows) {
    a.tiles_x = cdiv(a.Wo, 32); a.tiles = a.tiles_x * cdiv(a.Ho, TH);
    dim3 grid(a.tiles, cdiv(a.Mout, MB), c.B), block(64 * NW);
    const bool gb = a.gb_z != nullptr;
    char nm[80]; snprintf(nm, sizeof nm, "conv_bx_k<%d,%d,%d,%d,%d,%d,%d,%s%s%s>", KH, AMODE, EPI, TH, MB, a.act_bf16 ? 1 : 3, NW, AT_NAME(a.act_bf16), gb ? ",gb" : "", NIMG == 1 ? ",1img" : "");
    ProfScope ps(c.s, nm, c.layer, c.flops, c.bytes);
    *rows = c.B * a.tiles;
    if constexpr (AMODE == A_UPF && EPI == EPI_FWD && NIMG == 2) {
        if (a.flags & F_DROP) {
            if (a.act_bf16) conv_bx_k<KH, AMODE, EPI, TH, MB, 1, true, NW, bf16_t><<<grid, block, 0, c.s>>>(a);
            else conv_bx_k<KH, AMODE, EPI, TH, MB, 3, true, NW, float><<<grid, block, 0, c.s>>>(a);
            HIP_OK(hipGetLastError());
            return 0;
        }
    }
    if (a.flags & F_DROP) return fail(-3, "conv_bx_k: dropout on the input is only built for the up-conv forward (double-buffered form)");
    if constexpr (EPI != EPI_FWD) {
        if (gb) {
            if (a.act_bf16) conv_bx_k<KH, AMODE, EPI, TH, MB, 1, false, NW, bf16_t, true, NIMG><<<grid, block, 0, c.s>>>(a);
            else conv_bx_k<KH, AMODE, EPI, TH, MB, 3, false, NW, float, true, NIMG><<<grid, block, 0, c.s>>>(a);
            HIP_OK(hipGetLastError());
            return 0;
        }
    }
    if (gb) return fail(-3, "conv_bx_k: the BN-backward transform on load is only built for backward-data launches");
    if (a.act_bf16) conv_bx_k<KH, AMODE, EPI, TH, MB, 1, false, NW, bf16_t, false, NIMG><<<grid, block, 0, c.s>>>(a);
    else conv_bx_k<KH, AMODE, EPI, TH, MB, 3, false, NW, float, false, NIMG><<<grid, block, 0, c.s>>>(a);
    HIP_OK(hipGetLastError());
    return 0;
}
template <int KH, int AMODE, int EPI, int TH, int MB>
int launch_bx_geo(const IgemmArgs& a, const LaunchCtx& c, int* rows) {
    if constexpr (TH % 8 == 0) {
        if (c.o->bx_waves == 8) return launch_bx_nw<KH, AMODE, EPI, TH, MB, 8>(a, c, rows);
    }
    return launch_bx_nw<KH, AMODE, EPI, TH, MB, 4>(a, c, rows);
}
template <int KH, int AMODE, int EPI>
int launch_bx(const IgemmArgs& a, const LaunchCtx& c, int* rows) {
    auto blocks = [&](int th, int mb) { return (long)c.B * cdiv(a.Ho, th) * cdiv(a.Wo, 32) * cdiv(a.Mout, mb); };
    if constexpr (AMODE == A_DOWN2) {      // 2x-strided input tile: only the 4-row tile fits the LDS double buffer
        if (a.Mout % 64 == 0) return launch_bx_geo<KH, AMODE, EPI, 4, 64>(a, c, rows);
        return launch_bx_geo<KH, AMODE, EPI, 4, 32>(a, c, rows);
    } else {
        // bx_two_blocks: 4-row tiles, 4-wave blocks with ONE input image, two blocks per CU (see conv_bx_k NIMG)
        if (c.o->bx_two_blocks && !(a.flags & F_DROP)) {
            if (a.Mout % 64 == 0) return launch_bx_nw<KH, AMODE, EPI, 4, 64, 4, 1>(a, c, rows);
            return launch_bx_nw<KH, AMODE, EPI, 8, 32, 4, 1>(a, c, rows);
        }
        if (a.Mout % 64 == 0) {
            if (blocks(8, 64) >= c.o->bx_min_blocks) return launch_bx_geo<KH, AMODE, EPI, 8, 64>(a, c, rows);
            return launch_bx_geo<KH, AMODE, EPI, 4, 64>(a, c, rows);
        }
        if (blocks(16, 32) >= c.o->bx_min_blocks) return launch_bx_geo<KH, AMODE, EPI, 16, 32>(a, c, rows);
        return launch_bx_geo<KH, AMODE, EPI, 8, 32>(a, c, rows);
    }
}

// ---- thin bf16-pipe kernel (conv_bt_k): persistent, weights in registers ----
template <int KH, int AMODE, int EPI>
int launch_bt(IgemmArgs a, const LaunchCtx& c, int* rows) {
    a.tiles_x = cdiv(a.Wo, 32); a.tiles = a.tiles_x * cdiv(a.Ho, 8); a.total_tiles = c.B * a.tiles;
    const bool fdw = a.dw_part != nullptr;                          // the launch also reduces the layer's backward-weights (FDW)
    const int per_cu = fdw ? 2 : (a.Cin == 32 ? 1 : (a.Cin == 16 ? 2 : 3));     // what the LDS images and registers of the instantiation allow
    const int want = c.o->bt_blocks_per_cu;
    const int nblk = std::min(a.total_tiles, want > 0 ? 256 * std::min(want, per_cu) : 256 * per_cu);
    const int bf = a.act_bf16 ? 1 : 0;
    const bool m2 = a.bt_m2 && a.Mout == 8 && AMODE != A_DOWN2 && (a.Cin == 8 || a.Cin == 16);
    const bool gb = a.gb_z != nullptr;
    char nm[72]; snprintf(nm, sizeof nm, "conv_bt_k<%d,%d,%d,%d,%d,%s%s%s%s>", KH, AMODE, EPI, a.Cin, bf ? 1 : 3, AT_NAME(bf), m2 ? ",2px" : "", gb ? ",gb" : "", fdw ? ",dw" : "");
    ProfScope ps(c.s, nm, c.layer, c.flops, c.bytes);
    if (a.bt_m2 && !m2) return fail(-3, "conv_bt_k: weights were prepared in the two-pixel form for a launch that cannot use it");
    if (gb && EPI == EPI_FWD) return fail(-3, "conv_bt_k: the BN-backward transform on load is only built for backward-data launches");
    *rows = nblk;
    if (fdw) {
        if constexpr (KH == 3 && AMODE == A_NORMAL && EPI != EPI_FWD) {
            if (!(gb && a.Cin == 8 && a.Mout == 8 && a.dw_x && a.dw_ab))
                return fail(-3, "conv_bt_k: fused backward-weights needs 8 K channels, 8 output channels and the transform on load");
            if (m2) { if (bf) conv_bt_k<KH, AMODE, EPI, 8, 1, bf16_t, true, true, true><<<nblk, kBlock, 0, c.s>>>(a);
                      else conv_bt_k<KH, AMODE, EPI, 8, 3, float, true, true, true><<<nblk, kBlock, 0, c.s>>>(a); }
            else { if (bf) conv_bt_k<KH, AMODE, EPI, 8, 1, bf16_t, false, true, true><<<nblk, kBlock, 0, c.s>>>(a);
                   else conv_bt_k<KH, AMODE, EPI, 8, 3, float, false, true, true><<<nblk, kBlock, 0, c.s>>>(a); }
            HIP_OK(hipGetLastError());
            return 0;
        }
        return fail(-3, "conv_bt_k: fused backward-weights is only built for 3x3 backward-data launches");
    }
#define BT_LAUNCH(CT, M2V, GBV) do { if (bf) conv_bt_k<KH, AMODE, EPI, CT, 1, bf16_t, M2V, GBV><<<nblk, kBlock, 0, c.s>>>(a); \
                                     else conv_bt_k<KH, AMODE, EPI, CT, 3, float, M2V, GBV><<<nblk, kBlock, 0, c.s>>>(a); } while (0)
#define BT_CASE(CT, M2V) case CT: if constexpr (EPI != EPI_FWD) { if (gb) { BT_LAUNCH(CT, M2V, true); break; } } BT_LAUNCH(CT, M2V, false); break;
    if constexpr (AMODE != A_DOWN2) {
        if (m2) {
            switch (a.Cin) { BT_CASE(8, true) BT_CASE(16, true) default: break; }
            HIP_OK(hipGetLastError());
            return 0;
        }
    }
    if constexpr (AMODE == A_DOWN2) {     // the 2x-strided input tile only fits the LDS double buffer at 8 channels
        switch (a.Cin) { BT_CASE(8, false) default: return fail(-3, "conv_bt_k: stride-2 gather needs 8 K channels"); }
    } else {
        switch (a.Cin) {
            BT_CASE(8, false) BT_CASE(16, false)
            case 32:      // (no transform on load here: the staging registers for g' AND z do not fit -- the host plan knows)
                if (gb) return fail(-3, "conv_bt_k: the BN-backward transform on load is not built for 32 K channels");
                BT_LAUNCH(32, false, false); break;
            default: return fail(-3, "conv_bt_k: K channels must be 8, 16 or 32");
        }
    }
#undef BT_CASE
#undef BT_LAUNCH
    HIP_OK(hipGetLastError());
    return 0;
}

template <int KH, int AMODE, int EPI>
int launch_igemm(const IgemmArgs& a, const LaunchCtx& c, int* rows) {
    const ConvRoute r = conv_route(a, AMODE, *c.o);
    if (r == ROUTE_BT) return launch_bt<KH, AMODE, EPI>(a, c, rows);
    if (r == ROUTE_BX) return launch_bx<KH, AMODE, EPI>(a, c, rows);
    if (a.gb_z) return fail(-3, "fp32-pipe conv kernels do not apply the BN-backward transform on load");
    const Options& o = *c.o;
    auto blocks = [&](int th, int mb) { return (long)c.B * cdiv(a.Ho, th) * cdiv(a.Wo, 32) * cdiv(a.Mout, mb); };
    if constexpr (AMODE == A_NORMAL && KH == 3) {
        // 8 output channels, 3x3: two adjacent pixels share one 16-row MFMA tile (75 % useful instead of 50 %)
        if (a.Mout == 8 && a.Cin <= 16 && blocks(8, 16) >= o.pair_min_tiles)
            return launch_pair8<EPI>(a, c, rows);
    }
    if constexpr (AMODE != A_DOWN2) {
        // 8 output channels: a 16-row MFMA tile would be half padding -> VALU kernel (same f32 peak, no padding)
        if (a.Mout == 8 && a.Cin <= 16 && blocks(8, 16) >= o.thin_min_tiles)
            return launch_thin8<KH, AMODE, EPI>(a, c, rows);
        // thin single-chunk layers with plenty of pixel tiles: persistent software-pipelined kernel
        if (a.Cin <= 16 && a.Mout <= 16 && blocks(8, 16) >= o.persist_min_tiles) {
            if (a.Cin <= 8) return launch_igemm_p<16, KH, AMODE, EPI, 8, 16, 4, 8>(a, c, rows);
            return launch_igemm_p<16, KH, AMODE, EPI, 8, 16, 4, 16>(a, c, rows);
        }
    }
    if (a.Mout <= 16) {
        if (blocks(8, 16) >= o.igemm_min_blocks) return launch_igemm_geo<16, KH, AMODE, EPI, 8, 16, 4>(a, c, rows);
        return launch_igemm_geo<16, KH, AMODE, EPI, 4, 16, 4>(a, c, rows);
    }
    if (a.Mout <= 32) {
        if (blocks(8, 32) >= o.igemm_min_blocks) return launch_igemm_geo<32, KH, AMODE, EPI, 8, 32, 4>(a, c, rows);
        return launch_igemm_geo<32, KH, AMODE, EPI, 4, 32, 4>(a, c, rows);
    }
    if (blocks(4, 64) >= o.igemm_min_blocks) return launch_igemm_geo<32, KH, AMODE, EPI, 4, 64, 4>(a, c, rows);
    return launch_igemm_geo<32, KH, AMODE, EPI, 2, 64, 2>(a, c, rows);
}

}  // namespace octh
