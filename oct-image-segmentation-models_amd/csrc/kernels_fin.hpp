// BatchNorm statistics finalized INSIDE the launch that produces the partial rows (reference: the batch statistics of
// models/unet.py:20-23 `BatchNormalization()` and their gradient): the block that arrives last reduces the rows and writes
// the layer's record, instead of a bn_*_finalize launch behind every conv (44 launches of ~5 us per training step).
//
// Hand-off between workgroups of one launch (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility", the
// measured form: write-through stores + arrival counter, told by the value the add returned; no spinning anywhere):
//   * every partial-row element is stored write-through (`part_store`: global_store sc1 -- nothing stays dirty in the
//     writer's L2), every storing wave drains its stores (s_waitcnt vmcnt(0)), the block meets at a barrier, then ONE lane
//     adds 1 to the layer's counter (agent scope, returning);
//   * the block whose add returned nblocks - 1 is the last: after a barrier (its other waves learn it through LDS) it reads
//     ALL rows with sc1 loads (never served by its own L1 / a stale L2 line), sums them in double in a FIXED order
//     (deterministic whichever block happens to be last), writes the record and resets the counter for the next launch;
//   * the record is consumed by LATER launches (kernel boundary: ordinary visibility).
// Only producers with few, short rows use it (the persistent thin conv kernel, the streaming first-layer / head / pool
// kernels: <= ~1k rows of <= 32 floats, one ~2 us reduction at the tail); the wide conv kernels write one row per
// (image, tile) of up to 256 floats -- there a single block would take longer than the C-block finalize launch it replaces.
#pragma once
#include "common.hpp"

namespace oct {

struct FinDesc {
    unsigned* counter;                 // nullptr: the statistics are finalized by a separate launch
    int bwd;                           // 0: forward statistics (sum z, sum z^2); 1: backward (sum g', sum g' xhat)
    double count;                      // B * H * W
    float* bn;                         // the layer's record (BN_ARRAYS rows of C floats)
    const float* gamma; const float* beta;
    float* mm; float* mv; float eps, momentum; int unbiased;      // forward: moving statistics
    float* dgamma; float* dbeta;                                  // backward
};

// forward: column sums (s = sum z, q = sum z^2) of channel c -> (a, b, mean, rstd) and the moving-statistics update
__device__ __forceinline__ void bn_fwd_finalize_write(double s, double q, double count, int C, int c, float* __restrict__ bn,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ mm, float* __restrict__ mv, float eps, float momentum,
                                                      int unbiased) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double a = (double)gamma[c] * rstd;
    bn[BN_A * C + c] = (float)a;
    bn[BN_B * C + c] = (float)((double)beta[c] - mean * a);
    bn[BN_MEAN * C + c] = (float)mean;
    bn[BN_RSTD * C + c] = (float)rstd;
    const double m = momentum;
    const double uv = (unbiased && count > 1) ? var * (count / (count - 1.0)) : var;
    mm[c] = (float)((double)mm[c] * m + mean * (1.0 - m));
    mv[c] = (float)((double)mv[c] * m + uv * (1.0 - m));
}

// backward: partial sums (s = sum g', q = sum g' xhat) of channel c -> dbeta, dgamma, the record's c1, c2 and the two-fma
// form of the BN-backward transform (common.hpp)
__device__ __forceinline__ void bn_bwd_finalize_write(double s, double q, double count, int C, int c, float* __restrict__ bn,
                                                      const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta) {
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    const double c1 = s / count, c2 = q / count;
    const double rstd = (double)bn[BN_RSTD * C + c], mean = (double)bn[BN_MEAN * C + c];
    const double ga = (double)gamma[c] * rstd, gb = -ga * rstd * c2, gd = -ga * c1 - gb * mean;
    bn[BN_C1 * C + c] = (float)c1; bn[BN_C2 * C + c] = (float)c2;
    bn[BN_GA * C + c] = (float)ga; bn[BN_GB * C + c] = (float)gb; bn[BN_GD * C + c] = (float)gd;
}

// a partial-row element that another block of the SAME launch may read: write-through
__device__ __forceinline__ void part_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by EVERY thread of EVERY block of the launch, after the block's part_store()s.  `part`: [rows][2 * C] floats, C a
// multiple of 2; `lds`: >= 16 + blockDim.x * 16 + 2 * C * 8 bytes, 16-byte aligned, free for this call.  Contains barriers.
__device__ inline void finalize_in_launch(const FinDesc& F, const float* __restrict__ part, int rows, int C, unsigned nblocks, char* lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's row stores have left (write-through)
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(lds);
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(F.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = prev == nblocks - 1 ? 1 : 0;
    }
    __syncthreads();
    if (*flag) {                                               // block-uniform
        double* const sh = reinterpret_cast<double*>(lds + 16);                   // [G][C pairs][2]
        double* const tot = sh + 2 * (size_t)blockDim.x;                          // [2 C]
        const int Wp = C, G = (int)blockDim.x / Wp, t = threadIdx.x;             // pair column q = t % Wp, row group g = t / Wp
        if (t < G * Wp) {
            const int q = t % Wp, g = t / Wp;
            const unsigned long long* col = reinterpret_cast<const unsigned long long*>(part) + q;
            double s0 = 0, s1 = 0;
            int r = g;
            // every load of a batch is issued before the first is used: the whole table (<= ~1k rows) costs one or two
            // memory round trips, not one per few rows -- this block is the tail of the launch.  Fixed summation order.
            for (; r + 31 * G < rows; r += 32 * G) {
                unsigned long long v[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) v[k] = __hip_atomic_load(col + (size_t)(r + k * G) * Wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 32; ++k) { s0 += (double)__uint_as_float((unsigned)v[k]); s1 += (double)__uint_as_float((unsigned)(v[k] >> 32)); }
            }
            for (; r + 7 * G < rows; r += 8 * G) {
                unsigned long long v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __hip_atomic_load(col + (size_t)(r + k * G) * Wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s0 += (double)__uint_as_float((unsigned)v[k]); s1 += (double)__uint_as_float((unsigned)(v[k] >> 32)); }
            }
            for (; r < rows; r += G) {
                const unsigned long long v = __hip_atomic_load(col + (size_t)r * Wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s0 += (double)__uint_as_float((unsigned)v); s1 += (double)__uint_as_float((unsigned)(v >> 32));
            }
            sh[2 * t] = s0; sh[2 * t + 1] = s1;
        }
        __syncthreads();
        if (t < 2 * C) {                                       // column j = t of the rows: pair t / 2, component t & 1
            double a = 0;
            for (int g = 0; g < G; ++g) a += sh[2 * (g * Wp + (t >> 1)) + (t & 1)];
            tot[t] = a;
        }
        __syncthreads();
        if (t < C) {
            if (F.bwd) bn_bwd_finalize_write(tot[t], tot[C + t], F.count, C, t, F.bn, F.gamma, F.dgamma, F.dbeta);
            else bn_fwd_finalize_write(tot[t], tot[C + t], F.count, C, t, F.bn, F.gamma, F.beta, F.mm, F.mv, F.eps, F.momentum, F.unbiased);
        }
        if (t == 0) __hip_atomic_store(F.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    __syncthreads();                                           // `lds` may be reused by the caller
}

}  // namespace oct
