// scalar_gather.hip -- is there request capacity beside the vector memory path?  k_resolve is bound by the random 32-byte state records its
// CU can have in flight through the vector L1 (TCP): 0.1 G records/s per CU gathered and written back, whatever the region
// (tools/gather_region.hip, profiles/r04_gather_region.jsonl).  A record is exactly what ONE s_load_dwordx8 fetches, and scalar loads go
// through the scalar data cache, not the TCP.  Three kernels over the encoder's pattern (a 316 KB array per wavefront, arrays side by side):
//   vector   64 records per turn gathered by their lanes and written back by pairs of lanes            (= gather_region's wave pattern)
//   scalar   64 records per turn fetched by s_load_dwordx8, eight in flight, no write-back             (what the path sustains alone)
//   mixed    64 - K by the vector path (gathered + written back) and K by scalar loads (+ written back by the vector path)
// Output: one JSON object per line.   hipcc --offload-arch=gfx950 -O2 tools/scalar_gather.hip -o tools/bin/scalar_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) u32x8* kptr;

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// K of the 64 records of a turn come through the scalar cache (lanes 64 - K .. 63 name them), the others through the vector path.
// WB: every record is written back by a pair of lanes (the encoder's write-back).  INV: s_dcache_inv before a turn's scalar loads (what a
// real kernel needs: the records it reads were written by vector stores, the scalar cache does not see those).
template <int K, bool WB, bool INV>
__global__ __launch_bounds__(64) void k_turns(uint4* __restrict__ region, uint32_t span, unsigned long long stride, uint32_t iters, uint32_t seed, unsigned long long* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint4* const base = region + size_t(wave) * stride * 2;
    uint32_t s = mix(seed + wave * 64u + lane);
    auto draw = [&]() -> uint32_t { s = mix(s + 0x9e3779b9u); return __umulhi(s, span); };
    uint32_t at = draw();
    const bool vec = lane < uint32_t(64 - K);
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (vec) { a = base[size_t(at) * 2]; b = base[size_t(at) * 2 + 1]; }
    uint32_t acc = 0, sacc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 va = a, vb = b;
        const uint32_t was = at;
        acc += va.x ^ vb.w; va.x += 1; vb.w += 1;
        at = draw();
        if (vec) { a = base[size_t(at) * 2]; b = base[size_t(at) * 2 + 1]; }
        if (K > 0) {
            if (INV) __builtin_amdgcn_s_dcache_inv();
            // eight scalar loads in flight at a time
#pragma unroll
            for (int g = 0; g < K; g += 8) {
                u32x8 r[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t rec = uint32_t(__builtin_amdgcn_readlane(int(was), 64 - K + g + j));
                    r[j] = *(kptr)(uintptr_t(base) + size_t(rec) * 32);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) sacc += r[j].s0 ^ r[j].s7;
            }
        }
        if (WB) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t rec = uint32_t(__shfl(int(was), h * 32 + int(lane >> 1)));
                base[size_t(rec) * 2 + (lane & 1)] = (lane & 1) ? vb : va;
            }
        }
    }
    acc += a.y ^ b.z;
    if (acc + sacc == 0x12345678u) sink[0] = acc;
}

template <int K, bool WB, bool INV>
static double run(uint4* region, uint32_t span, unsigned long long stride, uint32_t waves, uint32_t iters, unsigned long long* sink)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_turns<K, WB, INV>), dim3(waves), dim3(64), 0, nullptr, region, span, stride, iters / 8, 1u, sink);
    CHECK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((k_turns<K, WB, INV>), dim3(waves), dim3(64), 0, nullptr, region, span, stride, iters, 7u, sink);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return double(ms) * 1e-3;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t ncu = uint32_t(prop.multiProcessorCount);
    const uint32_t blk = 2u * 5063u;                     // records per wavefront's array
    const uint32_t wpc_max = 12;
    uint4* region = nullptr; unsigned long long* sink = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&region), size_t(ncu) * wpc_max * blk * 32));
    CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 4096));
    CHECK(hipMemset(region, 0x80, size_t(ncu) * wpc_max * blk * 32));
    CHECK(hipDeviceSynchronize());
    const unsigned long long total = 1ull << 30;
    for (uint32_t wpc : { 8u, 10u, 12u }) {
        const uint32_t waves = ncu * wpc, iters = uint32_t(total / (size_t(waves) * 64));
        const double recs = double(waves) * 64 * iters;
#define ROW(name, K, WB, INV) { const double t = run<K, WB, INV>(region, blk, blk, waves, iters, sink); \
        printf("{\"kernel\": \"%s\", \"scalar_records_per_turn\": %d, \"write_back\": %s, \"s_dcache_inv_per_turn\": %s, \"waves_per_cu\": %u, \"G_records_per_s\": %.2f}\n", \
               name, K, WB ? "true" : "false", INV ? "true" : "false", wpc, recs / t * 1e-9); fflush(stdout); }
        ROW("vector", 0, true, false)
        ROW("vector", 0, false, false)
        ROW("scalar", 64, false, false)
        ROW("scalar", 64, false, true)
        ROW("mixed", 8, true, true)
        ROW("mixed", 16, true, true)
        ROW("mixed", 24, true, true)
        ROW("mixed", 32, true, true)
        ROW("mixed", 16, true, false)
    }
    return 0;
}
