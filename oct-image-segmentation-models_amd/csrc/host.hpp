// Host-side plumbing shared by the translation units of liboct_unet_hip.so: error channel, tuning options, the per-launch
// HIP-event profiler and the signatures of the conv launchers.  The library is built from several .hip files compiled in
// parallel (build.sh): oct_unet.hip holds the plan, the C ABI and the streaming kernels; tu_*.hip each instantiate one
// family of the MFMA conv kernels behind the launcher declared here.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "common.hpp"
#include "kernels_bwd.hpp"
#include "kernels_igemm.hpp"

namespace octh {

// ---- tuning options.  oct_set_option edits the process-wide DEFAULTS (g_opt); a handle snapshots them when it is
// created and every launch of that handle reads the snapshot, so two handles with different settings can live in one
// process (train + eval engines) and a knob that shapes prepared weights or workspace rows (bt_m2, dw*_blocks) cannot
// change under a live handle. ----
struct Options {
    int thin_min_tiles = 2048;      // pixel tiles from which 8-channel layers use the VALU thin kernel
    int dw32_blocks = 512, dw16_blocks = 768;   // target resident blocks of a backward-weights launch (wide / thin fp32-pipe kernels)
    int igemm_p_blocks = 1280;      // persistent igemm grid
    int igemm_min_blocks = 512;     // a layer takes the taller pixel tile only if that still yields this many blocks
    int dwpair8 = 1;                // 3x3 layers with 8 output channels: pixel-pair backward-weights kernel (0 = padded 16-column kernel)
    int pair_geo = 221;             // pixel-pair kernel geometry NWY*100 + NWX*10 + RPW
    int pair_min_tiles = 2048;      // pixel tiles from which 3x3 layers with 8 output channels use the pixel-pair MFMA kernel
    int dw_side_stream = 1;         // backward-weights kernels on the handle's side stream beside the backward-data chain
    int bx_two_blocks = 1;          // wide bf16-pipe launches (3x3, 2x2-over-upsample) as 4-wave blocks with ONE input image, two per CU
                                    // (cfg-A +1.4 %, cfg-C bf16 +5.3 % over the 8-wave double-buffered blocks, which 0 selects)
    int bx_waves = 8;               // waves per block of conv_bx_k where the tile has >= 8 rows
    int fuse_first_apply = 1;       // the first conv's BN-backward transform is applied inside its backward-weights kernel
    int fuse_bn_apply = 1;          // every other block: the transform is applied by the dX / dW kernels while they stage g'
    int fuse_bn_apply16 = 1;        // ... also for the 16-K / 16-output thin backward-data launches (coefficients kept in LDS there)
    int fuse_dw_thin = 1;           // 3x3 layers with 8 output channels: backward-weights reduced inside the backward-data launches
    int timing_skip = 0;            // TIMING EXPERIMENTS ONLY (results become wrong): bit 0 / 1 = skip the forward / backward BN finalize launches after step 2
    int fuse_bn_finalize = 0;       // 1: the BN records of the thin layers are written by the last block of the launch that emits the
                                    // partial rows (kernels_fin.hpp) instead of by a bn_*_finalize launch.  Built, tested -- and measured
                                    // 0.5-1 % SLOWER per step than the 5 us finalize launches it removes: off
    int bt_m2 = 1;                  // conv_bt_k: 8-output-channel launches in the two-pixel form
    int dwbt_f32_all = 0;           // 1: fp32 mode also takes conv_dwbt_k for every thin shape
    int bt_blocks_per_cu = 0;       // thin bf16-pipe kernel: persistent blocks per CU (0 = what its LDS allows)
    int dwbx_enable = 1;            // wide backward-weights on the bf16 pipe (0: the fp32-pipe conv_dw32_k; experiments)
    int dwbx_blocks = 256;          // target grid of a bf16-pipe backward-weights launch
    int bx_min_blocks = 256;        // a bf16-pipe launch takes the taller pixel tile only if that still yields this many blocks
    int mfma_mode = 1;              // 1: bf16 MFMA pipe (split products in fp32 mode); 0: the fp32-pipe kernels everywhere
    int focal_clip_mod = 0;         // focal loss: 1 = the (1-p)^gamma modulation sees the clipped p too
    int fork_on_launch = 1;         // the event a forked backward-weights kernel waits for is the completion signal of the preceding
                                    // launch itself (0: a recorded marker behind it -- ~6 us in front of every backward-data launch)
    int dw_fork_group = 1;          // blocks whose backward-weights launches share one fork to the side stream.  Measured: 2-4 halve the
                                    // caller stream's idle time (119 -> 71 us) and still lose 1.3-1.7 % per step -- the late kernels
                                    // pair with backward-data launches of the next level, which they slow down more
    int event_sysfence = 0;         // 1: the handle's fork/join events carry a system-scope fence (set before oct_unet_create)
    int persist_min_tiles = 2048;   // pixel tiles from which thin single-chunk convs use the persistent pipelined kernel
};
extern Options g_opt;

int fail(int code, const std::string& msg);

#define HIP_OK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return ::octh::fail(-5, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---- per-launch HIP-event profiler (off by default; bench.py turns it on for a few untimed steps) ---------
struct ProfRec { std::string kernel, layer; double flops, bytes; hipEvent_t e0, e1; };
struct Profiler {
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
};
extern thread_local Profiler* t_prof;

// RAII: records an event pair around the launch(es) issued in its scope, on the launch stream itself
struct ProfScope {
    Profiler* p; hipStream_t s; size_t idx;
    ProfScope(hipStream_t st, const char* kernel, const char* layer, double flops, double bytes) : p(t_prof), s(st), idx(0) {
        if (!p || !p->on) { p = nullptr; return; }
        ProfRec r{kernel, layer, flops, bytes, p->get(), p->get()};
        (void)hipEventRecord(r.e0, s);
        idx = p->recs.size(); p->recs.push_back(r);
    }
    ~ProfScope() { if (p) (void)hipEventRecord(p->recs[idx].e1, s); }
};

// run a launch statement with `AT` bound to the activation storage type
#define AT_DISPATCH(bf, ...) do { if (bf) { using AT = ::oct::bf16_t; __VA_ARGS__; } else { using AT = float; __VA_ARGS__; } } while (0)
#define AT_NAME(bf) ((bf) ? "unsigned short" : "float")

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int bx_mb(int M) { return M % 64 == 0 ? 64 : 32; }
inline bool bt_k_ok(int k) { return k == 8 || k == 16 || k == 32; }

// what a conv launch needs besides its argument block
struct LaunchCtx { const Options* o; int B; hipStream_t s; const char* layer; double flops, bytes; };

// kernel family a conv launch is routed to (launch_conv.hpp::route): the host plan asks before it launches, because only
// the bf16-pipe kernels can apply the BN-backward transform on load and finalize statistics in the launch
enum ConvRoute { ROUTE_BT = 0, ROUTE_BX = 1, ROUTE_F32 = 2 };
ConvRoute conv_route(const oct::IgemmArgs& a, int amode, const Options& o);

// MFMA conv launchers (forward and backward-data); *rows = statistic partial rows the launch writes.
// Defined in launch_conv.hpp, instantiated one per tu_conv_*.hip.
template <int KH, int AMODE, int EPI>
int launch_igemm(const oct::IgemmArgs& a, const LaunchCtx& c, int* rows);

// MFMA backward-weights launchers (launch_dw.hpp; tu_dw_*.hip).  `kind` as in DwPlan.
struct DwPlan { int kind;  /* 0 = VALU (1-channel input / head), 16, 32 = fp32 pipe, 33 = conv_dwbx_k, 34 = conv_dwbt_k */ int cic, coc, th, chunks, npb, tiles; };
int launch_dw_bf16pipe(const oct::ConvBwdWArgs& a, const DwPlan& p, int kh, bool up, const LaunchCtx& c);
int launch_dw_f32pipe(const oct::ConvBwdWArgs& a, const DwPlan& p, int kh, bool up, const LaunchCtx& c);

}  // namespace octh
