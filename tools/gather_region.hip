// gather_region.hip -- does the ceiling of tools/gather_peak.hip (19.7 G records/s over 4.3 GB) hold for the regions the kernels really
// walk?  profiles/r04_gather_peak.jsonl falls from 31 G/s (128 MB) over 23 (1 GB) to 19.7 (4.3 GB) and stops there because the program
// does.  The decoder (k_dec_slices, a lane per slice) has every slice of a batch in flight: 1600 frames x 64 slices x 316 KB of states
// (2 plane sets x 5063 contexts x 32 bytes) are 33 GB, 336 frames of 576 slices are 63 GB, and the 64 lanes of one load instruction are in 64 different slices' arrays.  The encoder
// (k_resolve, a wavefront per slice) has ~2048 slices in flight whose arrays lie side by side: 0.66 GB, all 64 lanes in one 316 KB array.
// Three patterns, each gather + a pair of lanes per record written back (gather_peak's mode 2) and gather only:
//   flat     every lane draws from the whole region                     (gather_peak, larger regions)
//   lane     every lane draws from its own block of B bytes             (the decoder)
//   wave     the lanes of a wavefront draw from the wavefront's block   (the encoder)
// Then: the encoder's pattern with its arrays 316 KB to 48 MiB apart; the same beside a copy kernel that streams 0.2 to 4 TB/s; and the
// uniform pattern from 256, 128 and 64 CUs (one workgroup of 16 wavefronts each, their CUs read back from HW_ID).
// Output: one JSON object per line.  hipcc --offload-arch=gfx950 -O2 tools/gather_region.hip -o tools/bin/gather_region
// Run: tools/bin/gather_region [GiB to allocate, default 100]   (with three more arguments: a fine sweep of the array distance, lo hi step in KB;
//      with the one word "rec": the record-size rows of round 6 -- 32 / 64 / 128-byte records gathered and written back whole)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// records are 32 bytes.  A lane's record index = base + (draw % span): base and span in records, chosen by the pattern.
// PAT 0 flat: base 0, span = all.  1 lane: base = (wave * 64 + lane) * block.  2 wave: base = wave * block.
// `stride` (records) lets consecutive owners lie `stride` apart instead of side by side (stride >= block), not used by flat.
template <int PAT, bool WB>
__global__ __launch_bounds__(64) void k_walk(uint4* __restrict__ region, unsigned long long span, unsigned long long stride, uint32_t iters, uint32_t seed,
                                             unsigned long long* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const unsigned long long base = PAT == 0 ? 0ull : PAT == 1 ? (size_t(wave) * 64 + lane) * stride : size_t(wave) * stride;
    uint32_t s = mix(seed + wave * 64u + lane);
    auto draw = [&]() -> unsigned long long {
        s = mix(s + 0x9e3779b9u);
        if (PAT == 0) { const uint32_t t = mix(s ^ 0x51ed270bu); return __umul64hi((static_cast<unsigned long long>(t) << 32) | s, span); }
        return __umulhi(s, uint32_t(span));
    };
    unsigned long long at = base + draw();
    uint4 a = region[at * 2], b = region[at * 2 + 1];
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 va = a, vb = b;
        const unsigned long long was = at;
        acc += va.x ^ vb.w;
        va.x += 1; vb.w += 1;
        at = base + draw();
        a = region[at * 2]; b = region[at * 2 + 1];
        if (WB) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t lo = uint32_t(__shfl(int(uint32_t(was)), h * 32 + int(lane >> 1)));
                const uint32_t hi = uint32_t(__shfl(int(uint32_t(was >> 32)), h * 32 + int(lane >> 1)));
                const unsigned long long rec = (static_cast<unsigned long long>(hi) << 32) | lo;
                region[rec * 2 + (lane & 1)] = (lane & 1) ? vb : va;
            }
        }
    }
    acc += a.y ^ b.z;
    if (acc == 0x12345678u) sink[0] = acc;
}

// Round 6: the record size as a template parameter.  REC = 32 is k_walk's wave pattern again; REC = 64 / 128 put every record in a
// sector pair / a whole 128-byte line of its own (aligned), gathered whole by its lane (REC / 16 loads of 16 bytes, all in one line) and
// written back whole by REC / 16 neighbouring lanes in ONE store instruction -- no partial sector, no partial line ever leaves the CU.
// The question ("what's weak" 1 i): the 32-byte write-back costs half the gather-only rate; does a whole-sector or
// whole-line write-back cost less?  If the ceiling rose by a quarter, the encoder's state records would be laid out that way.
// PAT 2 (wave: the encoder, all lanes in the wavefront's block) or 0 (flat).
template <int REC, int PAT, bool WB>
__global__ __launch_bounds__(64) void k_walk_rec(uint4* __restrict__ region, unsigned long long span, unsigned long long stride, uint32_t iters, uint32_t seed,
                                                 unsigned long long* __restrict__ sink)
{
    constexpr int V = REC / 16;                      // 16-byte vectors per record = lanes that share a write-back
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const unsigned long long base = PAT == 0 ? 0ull : size_t(wave) * stride;
    uint32_t s = mix(seed + wave * 64u + lane);
    auto draw = [&]() -> unsigned long long {
        s = mix(s + 0x9e3779b9u);
        if (PAT == 0) { const uint32_t t = mix(s ^ 0x51ed270bu); return __umul64hi((static_cast<unsigned long long>(t) << 32) | s, span); }
        return __umulhi(s, uint32_t(span));
    };
    unsigned long long at = base + draw();
    uint4 r[V];
#pragma unroll
    for (int k = 0; k < V; k++) r[k] = region[at * V + k];
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[V];
#pragma unroll
        for (int k = 0; k < V; k++) { v[k] = r[k]; acc += v[k].x ^ v[k].w; v[k].x += 1; }
        const unsigned long long was = at;
        at = base + draw();
#pragma unroll
        for (int k = 0; k < V; k++) r[k] = region[at * V + k];
        if (WB) {
            // store instruction h writes the records of lanes h * (64 / V) .. : V lanes side by side cover one record
            uint4 mine = v[0];
#pragma unroll
            for (int k = 1; k < V; k++) mine = (lane % V) == uint32_t(k) ? v[k] : mine;
#pragma unroll
            for (int h = 0; h < V; h++) {
                const int srcl = h * (64 / V) + int(lane / V);
                const uint32_t lo = uint32_t(__shfl(int(uint32_t(was)), srcl));
                const uint32_t hi = uint32_t(__shfl(int(uint32_t(was >> 32)), srcl));
                const unsigned long long rec = (static_cast<unsigned long long>(hi) << 32) | lo;
                region[rec * V + (lane % V)] = mine;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < V; k++) acc += r[k].y ^ r[k].z;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int REC, int PAT, bool WB>
static double run_rec(uint4* region, unsigned long long span, unsigned long long stride, uint32_t waves, uint32_t iters, unsigned long long* sink)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_walk_rec<REC, PAT, WB>), dim3(waves), dim3(64), 0, nullptr, region, span, stride, iters / 8, 1u, sink);
    CHECK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((k_walk_rec<REC, PAT, WB>), dim3(waves), dim3(64), 0, nullptr, region, span, stride, iters, 7u, sink);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return double(ms) * 1e-3;
}

template <int REC>
static void record_rows(uint4* region, unsigned long long nbytes, uint32_t ncu, unsigned long long* sink)
{
    const unsigned long long total = 1ull << 30;                               // records per run
    const unsigned long long blk = 2ull * 5063;                                // records per wavefront's array: two plane sets x 5063 contexts
    for (uint32_t wpc : { 8u, 10u, 12u }) {
        const uint32_t waves = ncu * wpc, iters = uint32_t(total / (size_t(waves) * 64));
        if (size_t(waves) * blk * REC > nbytes) continue;
        const double t0 = run_rec<REC, 2, false>(region, blk, blk, waves, iters, sink), t2 = run_rec<REC, 2, true>(region, blk, blk, waves, iters, sink);
        const double recs = double(waves) * 64 * iters;
        printf("{\"pattern\": \"wave\", \"record_bytes\": %d, \"lanes_per_write_back\": %d, \"block_kb\": %.0f, \"waves_per_cu\": %u, \"region_gib\": %.2f, "
               "\"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f}\n", REC, REC / 16, double(blk) * REC / 1024, wpc, double(waves) * blk * REC / double(1ull << 30),
               recs / t0 * 1e-9, recs / t2 * 1e-9);
        fflush(stdout);
    }
    for (double r : { 0.125, 4.0, 64.0 }) {                                    // flat: is it the pattern or the memory?
        const unsigned long long span = static_cast<unsigned long long>(r * double(1ull << 30)) / REC;
        if (span * REC > nbytes) continue;
        const uint32_t waves = ncu * 8, iters = uint32_t(total / (size_t(waves) * 64));
        const double t0 = run_rec<REC, 0, false>(region, span, 0, waves, iters, sink), t2 = run_rec<REC, 0, true>(region, span, 0, waves, iters, sink);
        const double recs = double(waves) * 64 * iters;
        printf("{\"pattern\": \"flat\", \"record_bytes\": %d, \"lanes_per_write_back\": %d, \"region_gib\": %.3f, \"waves_per_cu\": 8, "
               "\"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f}\n", REC, REC / 16, r, recs / t0 * 1e-9, recs / t2 * 1e-9);
        fflush(stdout);
    }
}

// A stream beside the gathers: `groups` workgroups of 256 copy `bytes_each` bytes each, 16 bytes per lane and turn, `turns` times over
// (what k_resolve's decision stream and k_rangecode's reading of it are to the memory: whole lines, one after the other)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned long long vec_each, uint32_t turns)
{
    const u32x4_t* s = reinterpret_cast<const u32x4_t*>(src) + size_t(blockIdx.x) * vec_each;
    u32x4_t* d = reinterpret_cast<u32x4_t*>(dst) + size_t(blockIdx.x) * vec_each;
    for (uint32_t t = 0; t < turns; t++)
        for (unsigned long long i = threadIdx.x; i < vec_each; i += 256) {
            u32x4_t v = __builtin_nontemporal_load(s + i);
            v.x += t;
            __builtin_nontemporal_store(v, d + i);
        }
}

// The flat pattern from workgroups of 16 wavefronts (so that a launch of N workgroups keeps N CUs busy and no more), with where each ran
template <bool WB>
__global__ __launch_bounds__(1024) void k_walk_wg(uint4* __restrict__ region, unsigned long long span, uint32_t iters, uint32_t seed, unsigned long long* __restrict__ sink,
                                                  uint32_t* __restrict__ where)
{
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20), hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        where[blockIdx.x] = (xcc & 15) << 16 | ((hw >> 13) & 7) << 8 | ((hw >> 8) & 15);         // XCC, shader engine, CU
    }
    uint32_t s = mix(seed + wave * 64u + lane);
    auto draw = [&]() -> unsigned long long { s = mix(s + 0x9e3779b9u); const uint32_t t = mix(s ^ 0x51ed270bu); return __umul64hi((static_cast<unsigned long long>(t) << 32) | s, span); };
    unsigned long long at = draw();
    uint4 a = region[at * 2], b = region[at * 2 + 1];
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 va = a, vb = b;
        const unsigned long long was = at;
        acc += va.x ^ vb.w; va.x += 1; vb.w += 1;
        at = draw();
        a = region[at * 2]; b = region[at * 2 + 1];
        if (WB) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t lo = uint32_t(__shfl(int(uint32_t(was)), h * 32 + int(lane >> 1), 64));
                const uint32_t hi = uint32_t(__shfl(int(uint32_t(was >> 32)), h * 32 + int(lane >> 1), 64));
                const unsigned long long rec = (static_cast<unsigned long long>(hi) << 32) | lo;
                region[rec * 2 + (lane & 1)] = (lane & 1) ? vb : va;
            }
        }
    }
    acc += a.y ^ b.z;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT, bool WB>
static double run(uint4* region, unsigned long long span, unsigned long long stride, uint32_t waves, uint32_t iters, unsigned long long* sink)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_walk<PAT, WB>), dim3(waves), dim3(64), 0, nullptr, region, span, stride, iters / 8, 1u, sink);
    CHECK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((k_walk<PAT, WB>), dim3(waves), dim3(64), 0, nullptr, region, span, stride, iters, 7u, sink);
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return double(ms) * 1e-3;
}

int main(int argc, char** argv)
{
    const double gb = argc > 1 ? atof(argv[1]) : 100.0;                       // bytes to allocate, in GiB
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t ncu = uint32_t(prop.multiProcessorCount);
    const unsigned long long nrec = static_cast<unsigned long long>(gb * double(1ull << 30)) / 32;
    uint4* region = nullptr; unsigned long long* sink = nullptr;
    CHECK(hipMalloc(reinterpret_cast<void**>(&region), nrec * 32));
    CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 4096));
    CHECK(hipMemset(region, 0x80, nrec * 32));
    CHECK(hipDeviceSynchronize());
    printf("{\"cus\": %u, \"allocated_gib\": %.1f}\n", ncu, double(nrec) * 32 / double(1ull << 30));
    if (argc > 2 && argv[2][0] == 'r') {                                        // tools/bin/gather_region <GiB> rec: the record-size rows of round 6 only
        record_rows<32>(region, nrec * 32, ncu, sink);
        record_rows<64>(region, nrec * 32, ncu, sink);
        record_rows<128>(region, nrec * 32, ncu, sink);
        return 0;
    }
    const unsigned long long total = 1ull << 31;                              // records per run
    // ---- flat: the whole chip draws from one region of R GiB
    for (double r : { 0.125, 1.0, 4.0, 8.0, 16.0, 32.0, 64.0, 96.0 }) {
        const unsigned long long span = static_cast<unsigned long long>(r * double(1ull << 30)) / 32;
        if (span > nrec) continue;
        for (uint32_t wpc : { 6u, 12u }) {
            const uint32_t waves = ncu * wpc, iters = uint32_t(total / (size_t(waves) * 64));
            const double t0 = run<0, false>(region, span, 0, waves, iters, sink), t2 = run<0, true>(region, span, 0, waves, iters, sink);
            const double recs = double(waves) * 64 * iters;
            printf("{\"pattern\": \"flat\", \"region_gib\": %.3f, \"waves_per_cu\": %u, \"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f}\n",
                   r, wpc, recs / t0 * 1e-9, recs / t2 * 1e-9);
            fflush(stdout);
        }
    }
    // ---- lane: a block per lane (the decoder: 316 KB of states per slice), blocks side by side; waves per CU as the decoder's batch has them
    const unsigned long long blk = 2ull * 5063;                                // records: two plane sets x 5063 contexts (324 032 bytes)
    for (uint32_t wpc : { 2u, 4u, 6u, 12u }) {
        const uint32_t waves = ncu * wpc;
        if (size_t(waves) * 64 * blk > nrec) continue;
        const uint32_t iters = uint32_t(total / (size_t(waves) * 64));
        const double t0 = run<1, false>(region, blk, blk, waves, iters, sink), t2 = run<1, true>(region, blk, blk, waves, iters, sink);
        const double recs = double(waves) * 64 * iters;
        printf("{\"pattern\": \"lane\", \"block_kb\": 316, \"waves_per_cu\": %u, \"region_gib\": %.2f, \"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f}\n",
               wpc, double(waves) * 64 * blk * 32 / double(1ull << 30), recs / t0 * 1e-9, recs / t2 * 1e-9);
        fflush(stdout);
    }
    // ---- lane with smaller blocks (what a slice really touches, were its records packed): does the region or the page count matter?
    for (unsigned long long kb : { 32ull, 79ull, 158ull }) {
        const uint32_t wpc = 6, waves = ncu * wpc, iters = uint32_t(total / (size_t(waves) * 64));
        const unsigned long long b = kb * 1024 / 32;
        const double t2 = run<1, true>(region, b, b, waves, iters, sink);
        const double t2s = size_t(waves) * 64 * blk <= nrec ? run<1, true>(region, b, blk, waves, iters, sink) : 0;
        const double recs = double(waves) * 64 * iters;
        printf("{\"pattern\": \"lane\", \"block_kb\": %llu, \"waves_per_cu\": %u, \"region_gib\": %.2f, \"gather_writeback_G_per_s\": %.2f, \"same_blocks_an_array_apart_G_per_s\": %.2f}\n",
               kb, wpc, double(waves) * 64 * b * 32 / double(1ull << 30), recs / t2 * 1e-9, t2s > 0 ? recs / t2s * 1e-9 : 0.0);
        fflush(stdout);
    }
    // ---- wave: a block per wavefront (the encoder), 8 and 12 wavefronts per CU
    for (uint32_t wpc : { 8u, 12u }) {
        const uint32_t waves = ncu * wpc, iters = uint32_t(total / (size_t(waves) * 64));
        const double t0 = run<2, false>(region, blk, blk, waves, iters, sink), t2 = run<2, true>(region, blk, blk, waves, iters, sink);
        // the same wavefronts, their blocks spread over the allocation (a batch's arrays are 10 GB: what if the ones in flight were not neighbours?)
        const unsigned long long far = nrec / waves;
        const double t2f = run<2, true>(region, blk, far, waves, iters, sink);
        const double recs = double(waves) * 64 * iters;
        printf("{\"pattern\": \"wave\", \"block_kb\": 316, \"waves_per_cu\": %u, \"region_gib\": %.2f, \"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f, "
               "\"blocks_spread_over_the_allocation_G_per_s\": %.2f}\n", wpc, double(waves) * blk * 32 / double(1ull << 30), recs / t0 * 1e-9, recs / t2 * 1e-9, recs / t2f * 1e-9);
        fflush(stdout);
    }
    // ---- wave: how far apart must the blocks in flight lie?  strides in KB, 8 wavefronts per CU
    for (unsigned long long skb : { 317ull, 512ull, 726ull, 968ull, 1024ull, 1452ull, 1936ull, 2048ull, 2112ull, 3872ull, 4096ull, 5082ull, 7744ull, 15488ull, 30976ull, 49152ull }) {
        const uint32_t wpc = 8, waves = ncu * wpc, iters = uint32_t(total / (size_t(waves) * 64));
        const unsigned long long st = skb * 1024 / 32;
        if (size_t(waves) * st > nrec) continue;
        const double t0 = run<2, false>(region, blk, st, waves, iters, sink), t2 = run<2, true>(region, blk, st, waves, iters, sink);
        const double recs = double(waves) * 64 * iters;
        printf("{\"pattern\": \"wave\", \"block_kb\": 316, \"stride_kb\": %llu, \"waves_per_cu\": %u, \"extent_gib\": %.2f, \"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f}\n",
               skb, wpc, double(waves) * st * 32 / double(1ull << 30), recs / t0 * 1e-9, recs / t2 * 1e-9);
        fflush(stdout);
    }
    // ---- the encoder's mix: wave-pattern gathers + write-backs (8 wavefronts per CU, blocks side by side) while `groups` workgroups
    // stream a copy through the upper half of the allocation.  Two passes per setting: the first finds the rates, the second sizes the
    // copy so that both end together.
    if (argc <= 2) {
        hipStream_t sg, sc; CHECK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
        const uint32_t waves = ncu * 8, iters = uint32_t(total / (size_t(waves) * 64));
        const double recs = double(waves) * 64 * iters;
        uint4* half = region + (nrec / 2) * 2;
        for (uint32_t groups : { 0u, 32u, 64u, 128u, 256u, 512u, 1024u }) {
            const unsigned long long vec_each = groups ? std::min<unsigned long long>((nrec / 4) * 2 / groups, (64ull << 20) / 16) : 0;   // <= 64 MB per group
            uint32_t turns = 4; double g_rate = 0, c_rate = 0;
            for (int pass = 0; pass < 3; pass++) {
                hipEvent_t g0, g1, c0, c1; CHECK(hipEventCreate(&g0)); CHECK(hipEventCreate(&g1)); CHECK(hipEventCreate(&c0)); CHECK(hipEventCreate(&c1));
                CHECK(hipDeviceSynchronize());
                if (groups) { CHECK(hipEventRecord(c0, sc)); hipLaunchKernelGGL(k_copy, dim3(groups), dim3(256), 0, sc, half, half + size_t(groups) * vec_each, vec_each, turns); CHECK(hipEventRecord(c1, sc)); }
                CHECK(hipEventRecord(g0, sg));
                hipLaunchKernelGGL((k_walk<2, true>), dim3(waves), dim3(64), 0, sg, region, blk, blk, iters, 7u + pass, sink);
                CHECK(hipEventRecord(g1, sg));
                CHECK(hipDeviceSynchronize());
                float gm = 0, cm = 0; CHECK(hipEventElapsedTime(&gm, g0, g1)); if (groups) CHECK(hipEventElapsedTime(&cm, c0, c1));
                g_rate = recs / (double(gm) * 1e-3) * 1e-9;
                c_rate = groups ? double(groups) * vec_each * 16 * turns * 2 / (double(cm) * 1e-3) * 1e-12 : 0;     // read + written, TB/s
                if (groups && cm > 0) turns = std::max<uint32_t>(1, uint32_t(double(turns) * gm / cm + 0.5));
                CHECK(hipEventDestroy(g0)); CHECK(hipEventDestroy(g1)); CHECK(hipEventDestroy(c0)); CHECK(hipEventDestroy(c1));
            }
            printf("{\"mix\": \"wave gathers + write-backs beside a streaming copy\", \"copy_groups\": %u, \"gather_writeback_G_per_s\": %.2f, \"copy_TB_per_s_read_plus_written\": %.3f}\n",
                   groups, g_rate, c_rate);
            fflush(stdout);
        }
        CHECK(hipStreamDestroy(sg)); CHECK(hipStreamDestroy(sc));
    }
    if (argc > 2) {                                                        // fine sweep: strides from argv[2] to argv[3] KB in steps of argv[4] KB
        const unsigned long long lo = strtoull(argv[2], nullptr, 10), hi = strtoull(argv[3], nullptr, 10), step = strtoull(argv[4], nullptr, 10);
        const bool lane_pat = argc > 5 && argv[5][0] == 'l';                 // fifth argument "lane": the decoder's pattern, strides in BYTES
        for (unsigned long long skb = lo; skb <= hi; skb += step) {
            const uint32_t wpc = lane_pat ? 6 : 8, waves = ncu * wpc, iters = uint32_t((total / 4) / (size_t(waves) * 64));
            const unsigned long long st = lane_pat ? skb / 32 : skb * 1024 / 32;
            if (size_t(waves) * (lane_pat ? 64 : 1) * st > nrec) break;
            const double t2 = lane_pat ? run<1, true>(region, blk, st, waves, iters, sink) : run<2, true>(region, blk, st, waves, iters, sink);
            if (lane_pat) printf("{\"sweep\": \"lane\", \"stride_bytes\": %llu, \"gather_writeback_G_per_s\": %.2f}\n", skb, double(waves) * 64 * iters / t2 * 1e-9);
            else printf("{\"sweep\": \"wave\", \"stride_kb\": %llu, \"gather_writeback_G_per_s\": %.2f}\n", skb, double(waves) * 64 * iters / t2 * 1e-9);
            fflush(stdout);
        }
    }
    // ---- from half of the CUs, from a quarter: workgroups of 16 wavefronts, one per CU as the dispatcher deals them, as many as there
    // are to be CUs in use; where they ran is read back (XCC_ID, HW_ID).  (gather_peak's CU-masked stream was not it: its mask, every second bit,
    // names four of the eight XCDs and such a mask is ignored -- tools/cu_mask_probe.)
    if (argc <= 2) {
        uint32_t* where = nullptr; CHECK(hipMalloc(reinterpret_cast<void**>(&where), 4 * 4096));
        for (uint32_t wgs : { 256u, 128u, 64u }) {
            const uint32_t iters = uint32_t((total / 2) / (size_t(wgs) * 1024));
            const unsigned long long span = (4ull << 30) / 32;
            double t[2];
            for (int wb = 0; wb < 2; wb++) {
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipMemset(where, 0xFF, 4 * 4096));
                if (wb) hipLaunchKernelGGL((k_walk_wg<true>), dim3(wgs), dim3(1024), 0, nullptr, region, span, iters / 8, 1u, sink, where);
                else hipLaunchKernelGGL((k_walk_wg<false>), dim3(wgs), dim3(1024), 0, nullptr, region, span, iters / 8, 1u, sink, where);
                CHECK(hipEventRecord(e0, nullptr));
                if (wb) hipLaunchKernelGGL((k_walk_wg<true>), dim3(wgs), dim3(1024), 0, nullptr, region, span, iters, 7u, sink, where);
                else hipLaunchKernelGGL((k_walk_wg<false>), dim3(wgs), dim3(1024), 0, nullptr, region, span, iters, 7u, sink, where);
                CHECK(hipEventRecord(e1, nullptr)); CHECK(hipEventSynchronize(e1));
                float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); t[wb] = double(ms) * 1e-3;
                CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
            }
            std::vector<uint32_t> h(wgs); CHECK(hipMemcpy(h.data(), where, 4 * wgs, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end()); const size_t distinct = size_t(std::unique(h.begin(), h.end()) - h.begin());
            const double recs = double(wgs) * 1024 * iters;
            printf("{\"pattern\": \"flat\", \"region_gib\": 4, \"workgroups_of_16_wavefronts\": %u, \"distinct_cus_they_ran_on\": %zu, \"gather_only_G_per_s\": %.2f, \"gather_writeback_G_per_s\": %.2f}\n",
                   wgs, distinct, recs / t[0] * 1e-9, recs / t[1] * 1e-9);
            fflush(stdout);
        }
    }
    return 0;
}
