// Microbenchmark: issue rate of f32 FMA flavours on gfx950 (decides VALU vs MFMA for the 8-channel layers).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));

constexpr int ITERS = 4096;

// 16 independent scalar FMAs per iteration, VGPR operands
__global__ void k_fma(float* out, float a, float b) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// same with an SGPR multiplier
__global__ void k_fma_s(float* out, float a, float b) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    float x = a + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "s"(b));
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 8 independent packed FMAs (16 FMAs) per iteration, VGPR operands
__global__ void k_pk(float* out, float a, float b) {
    float2_ acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float2_{threadIdx.x * 1e-3f + i, 1.f};
    float2_ x = {a + threadIdx.x, a}, y = {b, b};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// packed FMA, SGPR-pair multiplier, op_sel broadcast of one half of x (the thin-layer kernel's inner loop form)
__global__ void k_pk_s(float* out, float a, float b) {
    float2_ acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float2_{threadIdx.x * 1e-3f + i, 1.f};
    float2_ x = {a + threadIdx.x, a};
    float2_ y = {b, b * 0.5f};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(y));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// packed FMA, all-VGPR, with op_sel broadcast
__global__ void k_pk_sel(float* out, float a, float b) {
    float2_ acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float2_{threadIdx.x * 1e-3f + i, 1.f};
    float2_ x = {a + threadIdx.x, a}, y = {b, b * 0.5f};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "v"(y));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMA 4x4x1 (16 blocks): 512 flop / instruction; 4 independent accumulators
__global__ void k_mfma4(float* out, float a, float b) {
    float4_ acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = float4_{0, 0, 0, 0};
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMA 16x16x4: 2048 flop / instruction
__global__ void k_mfma16(float* out, float a, float b) {
    float4_ acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = float4_{0, 0, 0, 0};
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMA 32x32x2: 4096 flop / instruction
__global__ void k_mfma32(float* out, float a, float b) {
    float16_ acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
int run(const char* name, K kern, double flop_per_thread_iter, int waves_per_simd, float* out) {
    const int block = 256, blocks = 256 * waves_per_simd * 8;    // 256 CUs x (waves_per_simd blocks resident) x 8 rounds
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), 0, 0, out, 1.0f, 0.999f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), 0, 0, out, 1.0f, 0.999f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double flop = (double)blocks * block * ITERS * flop_per_thread_iter;
    printf("%-28s waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", name, waves_per_simd, ms, flop / ms / 1e9);
    return 0;
}

int main() {
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 256 * 256 * 8 * 8));
    for (int w : {1, 2, 4}) {
        run("v_fma_f32 vgpr", k_fma, 32.0, w, out);
        run("v_fma_f32 sgpr", k_fma_s, 32.0, w, out);
        run("v_pk_fma_f32 vgpr", k_pk, 32.0, w, out);
        run("v_pk_fma_f32 vgpr op_sel", k_pk_sel, 32.0, w, out);
        run("v_pk_fma_f32 sgpr op_sel", k_pk_s, 32.0, w, out);
        run("v_mfma_f32_4x4x1 (x4)", k_mfma4, 4 * 512.0 / 64, w, out);
        run("v_mfma_f32_16x16x4 (x4)", k_mfma16, 4 * 2048.0 / 64, w, out);
        run("v_mfma_f32_32x32x2 (x2)", k_mfma32, 2 * 4096.0 / 64, w, out);
    }
    CHECK(hipFree(out));
    return 0;
}
