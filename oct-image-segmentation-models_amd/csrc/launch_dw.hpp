// Launchers of the MFMA backward-weights kernels.  tu_dw_bf16.hip instantiates the bf16-pipe kernels (kernels_bx.hpp:
// conv_dwbx_k, conv_dwbt_k), tu_dw_f32.hip the fp32-pipe ones (kernels_dw.hpp) -- see host.hpp.
#pragma once
#include <cstdio>

#include "host.hpp"

namespace octh {
using namespace oct;

#ifdef OCT_TU_DW_BF16
int launch_dw_bf16pipe(const ConvBwdWArgs& a, const DwPlan& p, int kh, bool up, const LaunchCtx& c) {
    hipStream_t s = c.s;
    const int bf = a.act_bf16 ? 1 : 0;
    const bool dr = (a.flags & F_DROP) != 0;          // dropout on the input: only the up-conv behind the bottleneck
    const bool gb = a.zf != nullptr;                  // dz = BN-backward transform of the masked gradient, applied on load
    if (p.kind == 34) {
        char nm[72]; snprintf(nm, sizeof nm, "conv_dwbt_k<%d,%s,%d,%d,%d,%s%s>", kh, up ? "true" : "false", a.Cin, a.Cout, bf ? 1 : 3, AT_NAME(bf), gb ? ",gb" : "");
        ProfScope ps(s, nm, c.layer, c.flops, c.bytes);
        if (dr && !up) return fail(-3, "conv_dwbt_k: dropout on the input is only built for the up-conv");
#define DWBT_G(KHV, UPV, CI, CO, DR, GBV) { \
            if (bf) conv_dwbt_k<KHV, UPV, CI, CO, 1, bf16_t, DR, GBV><<<p.npb, kBlock, 0, s>>>(a); \
            else conv_dwbt_k<KHV, UPV, CI, CO, 3, float, DR, GBV><<<p.npb, kBlock, 0, s>>>(a); }
#define DWBT_D(KHV, UPV, CI, CO, DR) { if (gb) DWBT_G(KHV, UPV, CI, CO, DR, true) else DWBT_G(KHV, UPV, CI, CO, DR, false) }
#define DWBT(KHV, UPV, CI, CO) if (a.Cin == CI && a.Cout == CO) { if (UPV && dr) DWBT_D(KHV, UPV, CI, CO, UPV) else DWBT_D(KHV, UPV, CI, CO, false) }
        if (up) { DWBT(2, true, 16, 8) else DWBT(2, true, 32, 16) else return fail(-3, "conv_dwbt_k: up-conv shape not instantiated"); }
        else { DWBT(3, false, 8, 8) else DWBT(3, false, 8, 16) else DWBT(3, false, 16, 8) else DWBT(3, false, 16, 16)
               else DWBT(3, false, 16, 32) else DWBT(3, false, 32, 16) else return fail(-3, "conv_dwbt_k: shape not instantiated"); }
#undef DWBT
#undef DWBT_D
#undef DWBT_G
        HIP_OK(hipGetLastError());
        return 0;
    }
    if (p.kind != 33) return fail(-3, "launch_dw_bf16pipe: bad kind");
    dim3 grid(p.npb, a.Cin / 32, a.Cout / 32), block(kBlock);
    char nm[72]; snprintf(nm, sizeof nm, "conv_dwbx_k<%d,%s,%d,%s%s>", kh, up ? "true" : "false", bf ? 1 : 3, AT_NAME(bf), gb ? ",gb" : "");
    ProfScope ps(s, nm, c.layer, c.flops, c.bytes);
    if (dr && !up) return fail(-3, "conv_dwbx_k: dropout on the input is only built for the up-conv");
#define DWBX(KHV, UPV, DR) { if (gb) { if (bf) conv_dwbx_k<KHV, UPV, 1, bf16_t, DR, true><<<grid, block, 0, s>>>(a); else conv_dwbx_k<KHV, UPV, 3, float, DR, true><<<grid, block, 0, s>>>(a); } \
                             else { if (bf) conv_dwbx_k<KHV, UPV, 1, bf16_t, DR, false><<<grid, block, 0, s>>>(a); else conv_dwbx_k<KHV, UPV, 3, float, DR, false><<<grid, block, 0, s>>>(a); } }
    if (up && dr) DWBX(2, true, true)
    else if (up) DWBX(2, true, false)
    else DWBX(3, false, false)
#undef DWBX
    HIP_OK(hipGetLastError());
    return 0;
}
#endif

#ifdef OCT_TU_DW_F32
int launch_dw_f32pipe(const ConvBwdWArgs& a, const DwPlan& p, int kh, bool up, const LaunchCtx& c) {
    hipStream_t s = c.s;
    dim3 grid(p.npb, cdiv(a.Cin, p.cic), cdiv(a.Cout, p.coc)), block(kBlock);
    const bool pair8 = p.kind == 16 && !up && a.Cout == 8 && kh == 3 && c.o->dwpair8;
    const bool gb = a.zf != nullptr;
    char nm[72];
    if (pair8) snprintf(nm, sizeof nm, "conv_dwpair8_k<%d,%s%s>", p.cic, AT_NAME(a.act_bf16), gb ? ",gb" : "");
    else if (p.kind == 16) snprintf(nm, sizeof nm, "conv_dw16_k<%d,%d,%s,%s%s%s>", kh, p.cic, up ? "true" : "false", AT_NAME(a.act_bf16), gb ? ",gb" : "",
                                    (up && p.cic == 16 && a.Cout == 8) ? ",dz8" : "");
    else snprintf(nm, sizeof nm, "conv_dw32_k<%d,%d,%s,%d,%s%s>", kh, p.cic, up ? "true" : "false", p.th, AT_NAME(a.act_bf16), gb ? ",gb" : "");
    ProfScope ps(s, nm, c.layer, c.flops, c.bytes);
#define GBD(...) do { if (gb) { constexpr bool GBV = true; AT_DISPATCH(a.act_bf16, __VA_ARGS__); } else { constexpr bool GBV = false; AT_DISPATCH(a.act_bf16, __VA_ARGS__); } } while (0)
    if (pair8) {
        grid = dim3(p.npb, cdiv(a.Cin, p.cic), 1);
        if (p.cic == 16) GBD(conv_dwpair8_k<16, AT, GBV><<<grid, block, 0, s>>>(a));
        else GBD(conv_dwpair8_k<8, AT, GBV><<<grid, block, 0, s>>>(a));
    } else if (p.kind == 16) {
        if (up && p.cic == 16 && a.Cout == 8) GBD(conv_dw16_k<2, 16, true, AT, GBV, 8><<<grid, block, 0, s>>>(a));     // dz staged 8 wide
        else if (up) { if (p.cic == 16) GBD(conv_dw16_k<2, 16, true, AT, GBV><<<grid, block, 0, s>>>(a)); else GBD(conv_dw16_k<2, 8, true, AT, GBV><<<grid, block, 0, s>>>(a)); }
        else { if (p.cic == 16) GBD(conv_dw16_k<3, 16, false, AT, GBV><<<grid, block, 0, s>>>(a)); else GBD(conv_dw16_k<3, 8, false, AT, GBV><<<grid, block, 0, s>>>(a)); }
    } else if (p.kind == 32) {
        if (up) { if (p.cic == 64) GBD(conv_dw32_k<2, 64, true, 2, AT, GBV><<<grid, block, 0, s>>>(a)); else GBD(conv_dw32_k<2, 32, true, 4, AT, GBV><<<grid, block, 0, s>>>(a)); }
        else { if (p.cic == 64) GBD(conv_dw32_k<3, 64, false, 2, AT, GBV><<<grid, block, 0, s>>>(a)); else GBD(conv_dw32_k<3, 32, false, 4, AT, GBV><<<grid, block, 0, s>>>(a)); }
    } else return fail(-3, "launch_dw_f32pipe: bad kind");
#undef GBD
    HIP_OK(hipGetLastError());
    return 0;
}
#endif

}  // namespace octh
