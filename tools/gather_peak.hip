// gather_peak.hip -- what does an MI355X sustain in RANDOM 32-byte records, gathered and written back?
//
// k_resolve (rawcooked_amd/csrc/ffv1_gpu.hip) fetches one 32-byte context-state record per sample from a 7 GB array at an address that is
// random for 16-bit content, and writes it back: its byte roofline fraction is 1 % while TCP_PENDING_STALL_CYCLES is 83 % of its L1's
// cycles (profiles/r03_tcp_counters.txt).  The roofline that binds it is therefore the chip's ceiling in such REQUESTS, which
// MI355X_MICROARCH.md does not state.  This program measures it with k_resolve's own access pattern and nothing else:
//   * every lane of a wavefront reads one record (two 16-byte loads of one 32-byte record, as k_resolve's prefetch does);
//   * write-back in k_resolve's two forms: the lane's own record in two 16-byte stores (round 2), or a PAIR of lanes writing one record,
//     its halves side by side in one store instruction (round 3);
//   * `depth` independent gathers in flight per wavefront (k_resolve: the prefetch of chunk k+1 is issued while chunk k is binarised: 1);
//   * W wavefronts per CU (k_resolve runs 10-12), each a workgroup of its own, over a region of R bytes;
//   * (the run "from half of the CUs" at the end sets every second bit of a CU mask; tools/cu_mask_probe.hip shows that bit i stands for a CU
//     of XCD i % 8 and that a mask which leaves an XCD empty is ignored: that run used all CUs and says nothing.  tools/gather_region.hip
//     repeats it with one workgroup of 16 wavefronts per CU in use.)
//   * plus the latency of one dependent random read, unloaded (one wavefront on the chip) and beside the loaded chip.
// Output: one JSON object per line.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/gather_peak.hip -o tools/bin/gather_peak && tools/bin/gather_peak [region GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)           // a cheap full-period scrambler: every lane walks its own sequence
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: gather only.  1: gather + the lane's own record back in two 16-byte stores.  2: gather + a pair of lanes per record.
// DEPTH gathers are in flight per lane before the first one is used.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(64) void k_gather(uint4* __restrict__ region, uint32_t nrec_mask, uint32_t iters, uint32_t seed, unsigned long long* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint32_t s = mix(seed + wave * 64u + lane);
    uint32_t idx[DEPTH]; uint4 a[DEPTH], b[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { s = mix(s + 0x9e3779b9u); idx[d] = s & nrec_mask; a[d] = region[size_t(idx[d]) * 2]; b[d] = region[size_t(idx[d]) * 2 + 1]; }
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            // use the oldest gather, then replace it: DEPTH - 1 others stay in flight
            uint4 va = a[d], vb = b[d];
            const uint32_t at = idx[d];
            acc += va.x ^ vb.w;
            va.x += 1; vb.w += 1;                                       // "the states moved"
            s = mix(s + 0x9e3779b9u); idx[d] = s & nrec_mask;
            a[d] = region[size_t(idx[d]) * 2]; b[d] = region[size_t(idx[d]) * 2 + 1];
            if (MODE == 1) { region[size_t(at) * 2] = va; region[size_t(at) * 2 + 1] = vb; }
            if (MODE == 2) {
#pragma unroll
                for (int h = 0; h < 2; h++) {                           // records of lanes h * 32 .. h * 32 + 31, two lanes each
                    const uint32_t rec = uint32_t(__shfl(int(at), h * 32 + int(lane >> 1)));
                    region[size_t(rec) * 2 + (lane & 1)] = (lane & 1) ? vb : va;
                }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc += a[d].y ^ b[d].z;
    if (acc == 0x12345678u) sink[0] = acc;                               // keeps the loads alive
}

// The same with records that own a whole 64-byte line (the 32 state bytes + 32 of padding): does a write-back of the FULL line cost the
// memory less than the 32-byte half (no read-modify-write of a 64-byte ECC word)?  FULL = 0: 32 bytes read, 32 written (a pair of lanes),
// at a 64-byte stride; FULL = 1: 32 bytes read, the whole 64-byte line written (a quad of lanes); FULL = 2: 64 read, 64 written.
template <int FULL>
__global__ __launch_bounds__(64) void k_gather64(uint4* __restrict__ region, uint32_t nrec_mask, uint32_t iters, uint32_t seed, unsigned long long* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint32_t s = mix(seed + wave * 64u + lane);
    s = mix(s + 0x9e3779b9u);
    uint32_t idx = s & nrec_mask;
    uint4 a = region[size_t(idx) * 4], b = region[size_t(idx) * 4 + 1], c = make_uint4(0, 0, 0, 0), d = c;
    if (FULL == 2) { c = region[size_t(idx) * 4 + 2]; d = region[size_t(idx) * 4 + 3]; }
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 va = a, vb = b, vc = c, vd = d;
        const uint32_t at = idx;
        acc += va.x ^ vb.w ^ vc.y ^ vd.z;
        va.x += 1; vb.w += 1;
        s = mix(s + 0x9e3779b9u); idx = s & nrec_mask;
        a = region[size_t(idx) * 4]; b = region[size_t(idx) * 4 + 1];
        if (FULL == 2) { c = region[size_t(idx) * 4 + 2]; d = region[size_t(idx) * 4 + 3]; }
        if (FULL == 0) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t rec = uint32_t(__shfl(int(at), h * 32 + int(lane >> 1)));
                region[size_t(rec) * 4 + (lane & 1)] = (lane & 1) ? vb : va;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 4; h++) {                               // records of lanes h * 16 .. h * 16 + 15, four lanes each
                const uint32_t rec = uint32_t(__shfl(int(at), h * 16 + int(lane >> 2)));
                region[size_t(rec) * 4 + (lane & 3)] = (lane & 3) == 0 ? va : (lane & 3) == 1 ? vb : (lane & 3) == 2 ? vc : vd;
            }
        }
    }
    acc += a.y ^ b.z ^ c.x ^ d.w;
    if (acc == 0x12345678u) sink[0] = acc;
}

// one dependent chain per lane: the next address comes out of the record just read
__global__ __launch_bounds__(64) void k_chase(const uint4* __restrict__ region, uint32_t nrec_mask, uint32_t iters, uint32_t seed, unsigned long long* __restrict__ cycles)
{
    const uint32_t lane = threadIdx.x;
    uint32_t idx = mix(seed + blockIdx.x * 64u + lane) & nrec_mask;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; it++) {
        const uint4 v = region[size_t(idx) * 2];
        idx = mix(idx + v.x + it) & nrec_mask;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0 + (idx == 0xFFFFFFFFu);
}

template <int MODE, int DEPTH>
static double run(uint4* region, uint32_t mask, uint32_t waves, uint32_t iters, unsigned long long* sink, hipStream_t st)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<MODE, DEPTH>), dim3(waves), dim3(64), 0, st, region, mask, iters / 8, 1u, sink);      // warm-up
    CHECK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_gather<MODE, DEPTH>), dim3(waves), dim3(64), 0, st, region, mask, iters, 7u, sink);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return double(ms) * 1e-3;
}

int main(int argc, char** argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 7.0;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t ncu = uint32_t(prop.multiProcessorCount);
    // records: the largest power of two that fits the region (a mask instead of a modulo in the kernel)
    uint64_t nrec = 1; while (nrec * 2 * 32 <= uint64_t(gb * 1e9)) nrec *= 2;
    uint4* region = nullptr; unsigned long long* sink = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&region), nrec * 32));
    CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 8 * 4096));
    CHECK(hipMemset(region, 0x80, nrec * 32));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    const uint32_t mask = uint32_t(nrec - 1);
    printf("{\"device\": \"%s\", \"cus\": %u, \"clock_mhz\": %d, \"region_gb\": %.2f, \"records\": %llu}\n", prop.name, ncu, prop.clockRate / 1000, double(nrec * 32) / 1e9, (unsigned long long)nrec);
    const char* mode_name[3] = { "gather only", "gather + own record back (2 x 16 B stores per lane)", "gather + a pair of lanes per record (1 request per record)" };
    for (int mode = 0; mode < 3; mode++)
        for (uint32_t wpc : { 4u, 8u, 12u, 16u, 24u, 32u })
            for (int depth : { 1, 2, 4 }) {
                const uint32_t waves = ncu * wpc;
                // ~2 G records per run at full occupancy
                const uint32_t iters = uint32_t(std::max<uint64_t>(64, (uint64_t(1) << 31) / (uint64_t(waves) * 64 * depth)));
                double t = 0;
#define CASE(M, D) if (mode == M && depth == D) t = run<M, D>(region, mask, waves, iters, sink, st);
                CASE(0, 1) CASE(0, 2) CASE(0, 4) CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4)
#undef CASE
                const double recs = double(waves) * 64 * depth * iters;
                const double rd_bytes = recs * 64, wr_bytes = mode ? recs * 32 : 0;      // a 32-byte read costs a 64-byte HBM burst (FETCH_SIZE counts it so)
                printf("{\"mode\": %d, \"what\": \"%s\", \"waves_per_cu\": %u, \"depth\": %d, \"records\": %.0f, \"seconds\": %.4f, \"G_records_per_s\": %.2f, "
                       "\"G_requests_per_s\": %.2f, \"hbm_TBps_at_64B_reads\": %.3f, \"useful_TBps\": %.3f}\n",
                       mode, mode_name[mode], wpc, depth, recs, t, recs / t * 1e-9, recs * (mode == 0 ? 1 : mode == 1 ? 3 : 2) / t * 1e-9,
                       (rd_bytes + wr_bytes) / t * 1e-12, recs * (mode ? 64 : 32) / t * 1e-12);
                fflush(stdout);
            }
    // ---- how the ceiling moves with the REGION (does the 256 MB Infinity Cache serve such records faster than HBM?) and with the number
    // of CUs that ask (is it a per-CU queue or the memory's own limit?): pair write-back and gather only, 12 wavefronts per CU, depth 1
    for (double rgb : { 0.03125, 0.0625, 0.125, 0.1875, 0.25, 0.5, 1.0, 2.0 }) {
        uint64_t n = 1; while (n * 2 * 32 <= uint64_t(rgb * (1ull << 30))) n *= 2;
        if (rgb == 0.1875) n = (uint64_t(192) << 20) / 32;                      // not a power of two: the kernel masks, so use 128 MB + a second run below
        if (n > nrec) continue;
        const uint32_t m = uint32_t((rgb == 0.1875 ? (uint64_t(128) << 20) / 32 : n) - 1);
        const uint32_t waves = ncu * 12;
        const uint32_t iters = uint32_t((uint64_t(1) << 31) / (uint64_t(waves) * 64));
        for (int mode : { 0, 2 }) {
            const double t = mode == 0 ? run<0, 1>(region, m, waves, iters, sink, st) : run<2, 1>(region, m, waves, iters, sink, st);
            const double recs = double(waves) * 64 * iters;
            printf("{\"region_sweep_mb\": %.0f, \"mode\": %d, \"waves_per_cu\": 12, \"G_records_per_s\": %.2f}\n", double(uint64_t(m) + 1) * 32 / double(1 << 20), mode, recs / t * 1e-9);
            fflush(stdout);
        }
    }
    {   // 64-byte records over the same region (half as many records)
        const uint32_t m64 = uint32_t(nrec / 2 - 1);
        const uint32_t waves = ncu * 12;
        const uint32_t iters = uint32_t((uint64_t(1) << 31) / (uint64_t(waves) * 64));
        const char* what[3] = { "64-byte stride: 32 B read, 32 B written (pair of lanes)", "64-byte records: 32 B read, the whole 64 B line written (quad of lanes)", "64-byte records: 64 B read, 64 B written" };
        for (int full = 0; full < 3; full++) {
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            auto launch = [&](uint32_t it, uint32_t seed) {
                if (full == 0) hipLaunchKernelGGL(k_gather64<0>, dim3(waves), dim3(64), 0, st, region, m64, it, seed, sink);
                else if (full == 1) hipLaunchKernelGGL(k_gather64<1>, dim3(waves), dim3(64), 0, st, region, m64, it, seed, sink);
                else hipLaunchKernelGGL(k_gather64<2>, dim3(waves), dim3(64), 0, st, region, m64, it, seed, sink);
            };
            launch(iters / 8, 1u);
            CHECK(hipEventRecord(e0, st)); launch(iters, 7u); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"records64\": \"%s\", \"waves_per_cu\": 12, \"G_records_per_s\": %.2f}\n", what[full], double(waves) * 64 * iters / (double(ms) * 1e-3) * 1e-9);
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        }
    }
    {   // half of the CUs (every second one) through a CU-masked stream
        hipStream_t sm = nullptr;
        std::vector<uint32_t> cumask((ncu + 31) / 32, 0x55555555u);
        if (hipExtStreamCreateWithCUMask(&sm, uint32_t(cumask.size()), cumask.data()) == hipSuccess) {
            const uint32_t waves = ncu / 2 * 12;
            const uint32_t iters = uint32_t((uint64_t(1) << 30) / (uint64_t(waves) * 64));
            for (int mode : { 0, 2 }) {
                const double t = mode == 0 ? run<0, 1>(region, mask, waves, iters, sink, sm) : run<2, 1>(region, mask, waves, iters, sink, sm);
                const double recs = double(waves) * 64 * iters;
                printf("{\"half_of_the_cus\": true, \"mode\": %d, \"waves_per_cu\": 12, \"G_records_per_s\": %.2f}\n", mode, recs / t * 1e-9);
            }
            CHECK(hipStreamDestroy(sm));
        }
    }
    // latency of one dependent random read: alone on the chip, then with every CU busy gathering
    unsigned long long* cyc = nullptr; CHECK(hipMalloc(reinterpret_cast<void**>(&cyc), 8 * 4096));
    for (int loaded = 0; loaded < 2; loaded++) {
        hipStream_t s2; CHECK(hipStreamCreate(&s2));
        if (loaded) hipLaunchKernelGGL((k_gather<2, 1>), dim3(ncu * 12), dim3(64), 0, st, region, mask, 40000u, 3u, sink);
        const uint32_t iters = 20000;
        hipLaunchKernelGGL(k_chase, dim3(loaded ? ncu : 1), dim3(64), 0, s2, region, mask, iters, 11u, cyc);
        CHECK(hipStreamSynchronize(s2));
        std::vector<unsigned long long> h(loaded ? ncu : 1);
        CHECK(hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost));
        double sum = 0; for (auto v : h) sum += double(v);
        CHECK(hipDeviceSynchronize());
        printf("{\"latency\": \"%s\", \"cycles_per_dependent_read\": %.0f}\n", loaded ? "beside 12 gathering wavefronts per CU (pair write-back)" : "one wavefront alone on the chip",
               sum / double(h.size()) / iters);
        CHECK(hipStreamDestroy(s2));
    }
    return 0;
}
