// Micro-benchmark (runs on the GPU box): does hipStreamWaitValue32 on signal memory, satisfied by a store from a KERNEL of
// another stream, order two streams without a marker / completion signal in the writing stream?
//   stream A: spin(20 us) -> flagging kernel (stores seq to the signal word, then works 20 us) -> spin ...
//   stream B: hipStreamWaitValue32(sig >= seq) -> check kernel (reads a word the pre-flag kernel of A wrote)
// Reports: correctness of the ordering, and the time of N rounds on stream A with / without the cross-stream hand-off
// (event record + wait for comparison).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work_k(unsigned* data, unsigned v, int iters) {
    unsigned x = v;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0 && blockIdx.x == 0) data[0] = v + (x == 0xdeadbeefu);
}
__global__ void flag_k(unsigned* sig, unsigned v, unsigned* sink, int iters) {
    if (sig && threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(sig, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned x = v;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[1] = x;
}
__global__ void check_k(const unsigned* data, unsigned expect, unsigned* bad) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && data[0] < expect) atomicAdd(bad, 1u);
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    unsigned* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    *sig = 0;
    unsigned *data, *bad;
    CK(hipMalloc(&data, 64)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(data, 0, 64));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int N = 200, iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {      // 0: no hand-off, 1: event record + wait, 2: kernel store + wait value
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 1; r <= N; ++r) {
            const unsigned seq = 1000u * (unsigned)mode + (unsigned)r;
            work_k<<<256, 256, 0, A>>>(data, seq, iters);
            if (mode == 1) { CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0)); }
            if (mode == 2) CK(hipStreamWaitValue32(B, sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            flag_k<<<256, 256, 0, A>>>(mode == 2 ? sig : nullptr, seq, data, iters);
            if (mode) check_k<<<1, 64, 0, B>>>(data, seq, bad);
        }
        CK(hipStreamSynchronize(A));
        auto t1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(B));
        unsigned nb = 0; CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
        printf("mode %d: stream A %.1f us per round, ordering violations so far %u\n", mode,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N, nb);
    }
    // (the check counts a violation when it runs too EARLY: data[0] still holds a smaller sequence number than the one its
    // wait was for; running late is fine)
    return 0;
}
