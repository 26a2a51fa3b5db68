// Shared device helpers for the OCT U-Net HIP engine (gfx950 / CDNA4, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oct {

constexpr int kTileX = 32;   // pixel tile of the thread-per-pixel kernels: 32 (x, fastest across lanes) x 8 (y)
constexpr int kTileY = 8;
constexpr int kBlock = 256;  // 4 waves

// per-layer BN device record: 9 arrays of C floats each, in this order.  a, b, mean, rstd: forward (y = relu(a z + b));
// c1, c2: the BN-backward means of g' and g' xhat; ga, gb, gd: the BN-backward transform as two fmas,
//     dz = gamma rstd (g' - c1 - xhat c2) = ga g' + (gb z + gd),   ga = gamma rstd, gb = -ga rstd c2, gd = -ga c1 - gb mean
// (bn_bwd_finalize writes c1 .. gd; every consumer of dz -- the stand-alone pass and the stagers of the backward-data /
// backward-weights kernels that apply it on load -- evaluates exactly fmaf(ga, g', fmaf(gb, z, gd)): same bits everywhere)
enum { BN_A = 0, BN_B = 1, BN_MEAN = 2, BN_RSTD = 3, BN_C1 = 4, BN_C2 = 5, BN_GA = 6, BN_GB = 7, BN_GD = 8, BN_ARRAYS = 9 };

// input-fetch flags of the conv kernels
enum { F_U8 = 1, F_AFF = 2, F_TWO = 4, F_UP = 8, F_DROP = 16 };

static __constant__ float c_u8_lut[256];  // float32(i / 255.0): bit-identical to the reference's x/255.0 path (static: one
                                          // copy per translation unit; only oct_unet.hip's is filled and read)

// Counter-based dropout stream: keep(element) = hash(seed, step, idx) >= thresh.  Regenerated in
// forward, dX and dW, so no mask tensor is ever stored.
__host__ __device__ inline uint32_t drop_hash(uint64_t seed, uint64_t step, uint32_t idx) {
    uint64_t x = seed ^ (step * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)idx * 0xD1B54A32D192ED03ull);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return (uint32_t)(x >> 32);
}

struct DropCfg {
    unsigned long long seed, step;
    unsigned thresh;   // keep iff hash >= thresh  (rate * 2^32)
    float scale;       // 1 / (1 - rate)
};

__device__ inline float drop_mul(const DropCfg& d, uint32_t idx) {
    return drop_hash(d.seed, d.step, idx) >= d.thresh ? d.scale : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Wavefront-level multi-value reduction (64 lanes, N values per lane, N a power of two <= 64).
// Halving butterfly: at each step a lane keeps half of its values and hands the other half to its
// partner, so N values cost N-1 (+ log2(64/N)) shuffles instead of 6N.  On return v[0] of EVERY
// lane holds the wave total of channel chan_of_lane<N>(lane); lane c*(64/N) is a holder of channel c.
// ------------------------------------------------------------------------------------------------
template <int NFULL, int N, int OFF>
__device__ inline void wave_reduce_rec(float (&v)[NFULL], int lane) {
    if constexpr (N > 1) {
        const bool upper = (lane & OFF) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const float send = upper ? v[i] : v[i + N / 2];
            const float keep = upper ? v[i + N / 2] : v[i];
            v[i] = keep + __shfl_xor(send, OFF, 64);
        }
        wave_reduce_rec<NFULL, N / 2, OFF / 2>(v, lane);
    } else if constexpr (OFF >= 1) {
        v[0] += __shfl_xor(v[0], OFF, 64);
        wave_reduce_rec<NFULL, 1, OFF / 2>(v, lane);
    }
}

template <int N>
__device__ inline void wave_reduce_multi(float (&v)[N]) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two <= 64");
    wave_reduce_rec<N, N, 32>(v, threadIdx.x & 63);
}

// Block (4 waves) reduction of N per-thread values -> out[0..N) written by threads 0..N-1.
// `red` is 4*64 floats of LDS.  Contains two __syncthreads(); every thread of the block must call it.
// WT: the outputs are stored write-through (agent scope): rows another block of the same launch reads (kernels_fin.hpp).
template <int N, bool WT = false>
__device__ inline void block_reduce_store(float (&v)[N], float* red, float* out, int nvalid) {
    wave_reduce_multi<N>(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    red[wave * 64 + lane] = v[0];
    __syncthreads();
    if ((int)threadIdx.x < N && (int)threadIdx.x < nvalid) {
        const int l = threadIdx.x * (64 / N);
        const float r = (red[l] + red[64 + l]) + (red[128 + l] + red[192 + l]);
        if constexpr (WT) __hip_atomic_store(out + threadIdx.x, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[threadIdx.x] = r;
    }
    __syncthreads();
}

// which of the N channels lane `lane` holds after wave_reduce_multi<N> (used by tests of the helper)
template <int N>
__host__ __device__ inline int chan_of_lane(int lane) {
    int c = 0, n = N, off = 32;
    while (n > 1) { if (lane & off) c += n / 2; n >>= 1; off >>= 1; }
    return c;
}

// ---- column sums of a [rows][2*C] partial table for channel c, in fp64, by one block of kBlock threads ----
// The loads of 8 rows are issued together (independent, so one memory round trip covers 8 rows per thread instead of a
// dependent chain), the block sum is a wave shuffle tree + one LDS step.  Fixed summation order: deterministic.
// Returns the two sums in every thread of wave 0 (use thread 0).  `sh` = 8 doubles of LDS.
__device__ inline void column_sums_f64(const float* __restrict__ part, int rows, int C, int c, double* sh, double& s, double& q) {
    s = 0; q = 0;
    const size_t ld = 2 * (size_t)C;
    int i = threadIdx.x;
    for (; i + 7 * kBlock < rows; i += 8 * kBlock) {
        float a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float* r = part + (size_t)(i + k * kBlock) * ld + c; a[k] = r[0]; b[k] = r[C]; }
        s += (((double)a[0] + a[1]) + ((double)a[2] + a[3])) + (((double)a[4] + a[5]) + ((double)a[6] + a[7]));
        q += (((double)b[0] + b[1]) + ((double)b[2] + b[3])) + (((double)b[4] + b[5]) + ((double)b[6] + b[7]));
    }
    {   // tail: up to 7 rows per thread, still issued together
        float a[7], b[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int r_ = i + k * kBlock; const bool ok = r_ < rows;
            const float* r = part + (size_t)(ok ? r_ : 0) * ld + c;
            a[k] = ok ? r[0] : 0.f; b[k] = ok ? r[C] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) { s += a[k]; q += b[k]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wave] = s; sh[4 + wave] = q; }
    __syncthreads();
    s = (sh[0] + sh[1]) + (sh[2] + sh[3]); q = (sh[4] + sh[5]) + (sh[6] + sh[7]);
}

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// ---- activation storage type: float (dtype 0) or bfloat16 (dtype 1: bf16 STORAGE of activations and activation
// gradients; all arithmetic, BN statistics, parameters and parameter gradients stay fp32) ----
typedef unsigned short bf16_t;
__device__ inline float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ inline bf16_t f2bf(float f) {   // round to nearest even (finite inputs; the network never stores NaN/Inf)
    const uint32_t u = __float_as_uint(f);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
template <typename AT> __device__ inline float4 lda4(const AT* p);
template <> __device__ inline float4 lda4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ inline float4 lda4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
// Raw prefetch registers: what a 4-element activation load leaves in VGPRs BEFORE widening (float4 for f32 storage,
// uint2 for bf16 storage).  Prefetched tiles are kept raw and widened when they are consumed: widening at the load
// would put a use of the loaded data next to the load, which makes the compiler wait for every prefetch load in place
// (measured: 3x SQ_WAIT_ANY in the bf16 dW kernels) -- and the raw form needs half the registers in bf16 mode.
template <typename AT> struct Raw4;
template <> struct Raw4<float> { using type = float4; };
template <> struct Raw4<bf16_t> { using type = uint2; };
template <typename AT> __device__ inline typename Raw4<AT>::type ldraw4(const AT* p);
template <> __device__ inline float4 ldraw4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ inline uint2 ldraw4<bf16_t>(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
template <typename AT> __device__ inline typename Raw4<AT>::type raw_zero4();
template <> __device__ inline float4 raw_zero4<float>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ inline uint2 raw_zero4<bf16_t>() { return make_uint2(0u, 0u); }
__device__ inline float4 widen4(const float4& r) { return r; }
__device__ inline float4 widen4(const uint2& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
template <typename AT> __device__ inline void sta4(AT* p, const float4& v);
template <> __device__ inline void sta4<float>(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// gfx950 converts a pair per instruction (v_cvt_pk_bf16_f32, round to nearest even like f2bf above)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
template <> __device__ inline void sta4<bf16_t>(bf16_t* p, const float4& v) {
    uint2 u;
    u.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_hw{v.x, v.y}, bf16x2_hw));
    u.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_hw{v.z, v.w}, bf16x2_hw));
    *reinterpret_cast<uint2*>(p) = u;
}
template <typename AT> __device__ inline float lda1(const AT* p);
template <> __device__ inline float lda1<float>(const float* p) { return *p; }
template <> __device__ inline float lda1<bf16_t>(const bf16_t* p) { return bf2f(*p); }

}  // namespace oct
