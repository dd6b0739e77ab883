// ROCm 7.2: an out-of-memory hipMalloc after a CU-masked stream was used.  `bash tools/round.sh oomrepro`
//   hipcc --offload-arch=gfx950 -O2 tools/oom_cumask_repro.hip -o tools/bin/oom_cumask_repro
//   oom_cumask_repro <0|1|2>    0: plain stream; 1: CU-masked stream, alive at the failing hipMalloc; 2: CU-masked stream, destroyed before it
// A hipMalloc that cannot be satisfied must return hipErrorOutOfMemory (2).  ffv1_check.hip keeps the hash off the decoder's SIMDs with such
// streams, and callers rely on that error (route C halves its batch on it; rcgpu_ffv1_set_run_on falls back to one batch at a time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_touch(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
int main(int argc, char** argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    unsigned* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 4) != hipSuccess) return 1;
    hipStream_t s = nullptr;
    uint32_t mask[8] = { 0xFFu, 0, 0, 0, 0, 0, 0, 0 };
    if ((mode ? hipExtStreamCreateWithCUMask(&s, 8, mask) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_touch, dim3(8), dim3(64), 0, s, d);
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (mode == 2) (void)hipStreamDestroy(s);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    void* filler = nullptr; void* big = nullptr;
    if (hipMalloc(&filler, free_b - (size_t(96) << 20)) != hipSuccess) { printf("filler failed\n"); return 1; }      // 96 MiB left
    const hipError_t e = hipMalloc(&big, size_t(1) << 30);                   // fits the device, not what is free
    printf("mode %d: hipMalloc of 1 GiB with 96 MiB free returned %d (%s)\n", mode, int(e), hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? 0 : 1;
}
