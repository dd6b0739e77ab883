// cu_mask_probe.hip -- what does the mask of hipExtStreamCreateWithCUMask mean on an MI355X (8 XCDs x 32 CUs)?
//   1. one launch per bit with only that bit set; the kernel reports XCC_ID and HW_ID of where it ran (finding: everywhere -- see 2);
//   2. patterns, 4096 workgroups each, counting the distinct CUs and the CUs per XCD they landed on (finding: bit i is a CU of XCD i % 8, so
//      32 / 64 / 128 evenly set bits give 4 / 8 / 16 CUs of every XCD; a mask that leaves an XCD without a CU -- one bit, every second bit --
//      is ignored as a whole);
//   3. a spin kernel on two streams with disjoint masks, alone and together (finding: they run side by side).
// What ffv1_check.hip's partition_streams relies on.  hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <chrono>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(uint32_t* out)
{
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);        // HW_REG_XCC_ID, bits 3:0
        out[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);    // HW_REG_HW_ID
    }
}
__global__ void k_spin(uint32_t n, uint32_t* out)
{
    uint32_t a = threadIdx.x;
    for (uint32_t i = 0; i < n; i++) a = a * 1664525u + 1013904223u;
    if (a == 12345u) out[0] = a;
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
    // patterns: how many distinct CUs (XCC, shader engine, CU id) do 4096 workgroups of a masked stream land on, and on which XCCs?
    uint32_t* d2 = nullptr; CHECK(hipMalloc(reinterpret_cast<void**>(&d2), 8 * 4096));
    struct pat { const char* name; uint32_t word_even, word_odd; uint32_t words_set; };
    const pat pats[] = { { "all bits", 0xFFFFFFFFu, 0xFFFFFFFFu, words }, { "every second bit", 0x55555555u, 0x55555555u, words }, { "low half of every word", 0x0000FFFFu, 0x0000FFFFu, words },
                         { "first half of the words", 0xFFFFFFFFu, 0xFFFFFFFFu, words / 2 }, { "even words only", 0xFFFFFFFFu, 0u, words }, { "first word only", 0xFFFFFFFFu, 0xFFFFFFFFu, 1 },
                         { "low byte of every word", 0x000000FFu, 0x000000FFu, words } };
    for (const pat& q : pats) {
        std::vector<uint32_t> mask(words, 0);
        for (uint32_t w = 0; w < q.words_set; w++) mask[w] = (w & 1) ? q.word_odd : q.word_even;
        hipStream_t st; const hipError_t e = hipExtStreamCreateWithCUMask(&st, words, mask.data());
        if (e != hipSuccess) { printf("pattern %-26s: %s\n", q.name, hipGetErrorString(e)); continue; }
        CHECK(hipMemsetAsync(d2, 0xFF, 8 * 4096, st));
        hipLaunchKernelGGL(k_where, dim3(4096), dim3(64), 0, st, d2);
        std::vector<uint32_t> h(8192); CHECK(hipMemcpyAsync(h.data(), d2, 8 * 4096, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
        std::vector<uint32_t> ids; uint32_t per_xcc[16] = { 0 };
        for (int i = 0; i < 4096; i++) ids.push_back((h[i * 2] & 15) << 16 | ((h[i * 2 + 1] >> 13) & 7) << 8 | ((h[i * 2 + 1] >> 8) & 15));
        std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        for (uint32_t id : ids) per_xcc[id >> 16]++;
        printf("pattern %-26s: %3zu distinct CUs; per XCC", q.name, ids.size());
        for (int x = 0; x < 8; x++) printf(" %u", per_xcc[x]);
        printf("\n");
        CHECK(hipStreamDestroy(st));
    }
    // do two kernels on two CU-masked streams with disjoint masks run side by side?  A spin of ~200 ms on each, alone and together.
    {
        uint32_t ma[8] = { 0xFFu, 0, 0, 0, 0, 0, 0, 0 }, mb[8] = { ~0xFFu, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u };
        hipStream_t sa, sb, sp; CHECK(hipExtStreamCreateWithCUMask(&sa, 8, ma)); CHECK(hipExtStreamCreateWithCUMask(&sb, 8, mb)); CHECK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
        auto timed = [&](hipStream_t s1, uint32_t g1, hipStream_t s2, uint32_t g2) -> double {
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            if (s1) hipLaunchKernelGGL(k_spin, dim3(g1), dim3(64), 0, s1, 40000000u, d2);
            if (s2) hipLaunchKernelGGL(k_spin, dim3(g2), dim3(64), 0, s2, 40000000u, d2);
            (void)hipDeviceSynchronize();
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
        };
        printf("spin: hash-mask stream alone %.0f ms, decode-mask stream alone %.0f ms, both %.0f ms; plain stream + hash-mask stream %.0f ms; plain + decode-mask %.0f ms\n",
               timed(sa, 16, nullptr, 0), timed(sb, 1024, nullptr, 0), timed(sa, 16, sb, 1024), timed(sp, 1024, sa, 16), timed(sp, 1024, sb, 1024));
    }
    return 0;
}
