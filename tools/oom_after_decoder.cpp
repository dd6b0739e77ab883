// Scratch (round 5): decoder create + destroy, then an out-of-memory hipMalloc -- in C++, so that tools/bin/segv_trace.so can say where it dies.
//   g++ -O1 -g -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/oom_after_decoder.cpp -Lrawcooked_amd -lrcgpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/rawcooked_amd -Wl,-rpath,/opt/rocm/lib -o tools/bin/oom_after_decoder
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstring>
#include "rcgpu.h"
int main(int argc, char** argv)
{
    const int with_decoder = argc > 1 ? atoi(argv[1]) : 1;
    if (with_decoder) {
        rcgpu_ffv1_config c; memset(&c, 0, sizeof c);
        c.width = 192; c.height = 96; c.pixfmt = RCGPU_PIX_RGB16_BE; c.line_bytes = 192 * 6; c.num_h_slices = 2; c.num_v_slices = 2; c.slicecrc = 1; c.context = 1; c.max_batch = 3; c.coder = 1; c.level = 3;
        rcgpu_ffv1_decoder* d = nullptr;
        const int r = rcgpu_ffv1_decoder_create(&c, &d);
        printf("decoder_create: %d %s\n", r, r ? rcgpu_last_error() : "");
        if (!r) rcgpu_ffv1_decoder_destroy(d);
    }
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    void* filler = nullptr; void* big = nullptr;
    if (hipMalloc(&filler, free_b - (size_t(96) << 20)) != hipSuccess) { printf("filler failed\n"); return 1; }
    const hipError_t e = hipMalloc(&big, size_t(1) << 30);
    printf("with_decoder %d: hipMalloc of 1 GiB with 96 MiB free returned %d (%s)\n", with_decoder, int(e), hipGetErrorString(e));
    return 0;
}
