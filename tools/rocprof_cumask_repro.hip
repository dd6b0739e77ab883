// rocprofv3 and CU-masked streams (ROCm 7.2): the smallest program that shows what tools/profile_check.sh works around.
//   hipcc --offload-arch=gfx950 -O2 tools/rocprof_cumask_repro.hip -o tools/bin/rocprof_cumask_repro
//   tools/bin/rocprof_cumask_repro [0|1|2]      1 (default): the second stream is made with hipExtStreamCreateWithCUMask; 0: a plain stream;
//                                               2: CU-masked and still alive when the process ends (a library's static stream)
// Alone it prints "ok"; `bash tools/round.sh cumask` runs it under the rocprofv3 modes and records which of them survive
// (profiles/r05_rocprof_cumask.txt).  ffv1_check.hip makes such streams to keep the hash off the decoder's SIMDs (partition_streams).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_touch(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }
#define T(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int mode = argc < 2 ? 1 : atoi(argv[1]);
    const bool masked = mode != 0;
    unsigned* d = nullptr; unsigned h = 0;
    T(hipMalloc(reinterpret_cast<void**>(&d), 4)); T(hipMemset(d, 0, 4));
    hipStream_t a = nullptr, b = nullptr;
    T(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    uint32_t mask[8] = { 0xFFu, 0, 0, 0, 0, 0, 0, 0 };          // one CU of every XCD (bit i = a CU of XCD i % 8)
    if (masked) T(hipExtStreamCreateWithCUMask(&b, 8, mask)); else T(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) { hipLaunchKernelGGL(k_touch, dim3(64), dim3(64), 0, a, d); hipLaunchKernelGGL(k_touch, dim3(8), dim3(64), 0, b, d); }
    T(hipStreamSynchronize(a)); T(hipStreamSynchronize(b));
    T(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("ok: %u launches counted, second stream %s\n", h, masked ? "CU-masked" : "plain");
    T(hipStreamDestroy(a)); if (mode != 2) T(hipStreamDestroy(b)); T(hipFree(d));
    return h == 4 * (64 + 8) ? 0 : 1;
}
