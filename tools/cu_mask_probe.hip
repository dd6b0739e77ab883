// cu_mask_probe.hip -- which CU does bit b of hipExtStreamCreateWithCUMask's mask stand for?  One launch per bit with only that bit set;
// the kernel reports XCC_ID and HW_ID of where it ran.  hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(uint32_t* out)
{
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);        // HW_REG_XCC_ID, bits 3:0
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);    // HW_REG_HW_ID
    }
}
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t ncu = uint32_t(prop.multiProcessorCount), words = (ncu + 31) / 32;
    uint32_t* d = nullptr; CHECK(hipMalloc(reinterpret_cast<void**>(&d), 8 * 64));
    for (uint32_t b = 0; b < ncu; b++) {
        std::vector<uint32_t> mask(words, 0); mask[b / 32] = 1u << (b % 32);
        hipStream_t st; CHECK(hipExtStreamCreateWithCUMask(&st, words, mask.data()));
        CHECK(hipMemsetAsync(d, 0xFF, 8 * 64, st));
        hipLaunchKernelGGL(k_where, dim3(16), dim3(64), 0, st, d);
        uint32_t h[32]; CHECK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
        uint32_t xcc_set = 0, cu_lo = 99, cu_hi = 0, se_set = 0;
        for (int i = 0; i < 16; i++) { xcc_set |= 1u << (h[i * 2] & 15); const uint32_t cu = (h[i * 2 + 1] >> 8) & 15, se = (h[i * 2 + 1] >> 13) & 7; cu_lo = cu < cu_lo ? cu : cu_lo; cu_hi = cu > cu_hi ? cu : cu_hi; se_set |= 1u << se; }
        printf("bit %3u: xcc mask %02x  se mask %02x  cu_id %u..%u  hw_id[0] %08x\n", b, xcc_set, se_set, cu_lo, cu_hi, h[1]);
        CHECK(hipStreamDestroy(st));
    }
    return 0;
}
