// flac_gpu.hip -- FLAC encoder for MI355X (gfx950, wave64): replaces FFmpeg's flacenc on RAWcooked's WAV path
// (`-c:a flac`, CLI/Global.cpp:951-952; executed at CLI/Output.cpp:356).  The bitstream is what the reference's
// vendored libFLAC decoder parses (Lib/ThirdParty/flac/src/libFLAC/stream_decoder.c:2012-2788) behind flac_wrapper
// (Lib/CoDec/Wrapper.cpp:131-373).  No CPU fallback: without a HIP device rcgpu_flac_create() fails.
//
// Two kernels, one wavefront per work item:
//   k_flac_plan   (block, signal): a signal is a channel -- or, for two channels, one of left, right, mid = (l + r) >> 1, side = l - r (one
//                 bit wider), so that k_flac_write can choose between the plain pair and left/side, side/right, mid/side (channel
//                 assignments 8, 9, 10; stream_decoder.c:2299-2321 reads them, :2036-2062 undoes them).  Per signal: constant / verbatim / fixed 0-4 / LPC 1..max search.  Integer-exact Welch-windowed
//                 autocorrelation (wave reduction of int64), Levinson-Durbin + coefficient quantiser in IEEE double with a
//                 fixed operation order (-ffp-contract=off), exhaustive order search by exact Rice bit counts,
//                 partition-order search by the integer cost m*(k+1) + (U >> k).
//   k_flac_write  (block): frame header + CRC-8, every subframe bit-packed at its prefix-summed bit offset (wave scan of
//                 code lengths, atomicOr into big-endian words), zero padding, CRC-16.
// FLAC leaves all predictor choices to the encoder; the fixed rule above is also stated in scalar C by the oracle
// (oracle/flac_oracle.c) so that device bytes can be compared bit for bit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "rc_common.h"

using namespace rc;

namespace {

constexpr int kMaxOrder = 32;
constexpr int kQlpPrecision = 15;
constexpr int kMaxPartOrder = 8;
constexpr int kMaxParts = 1 << kMaxPartOrder;

// (a failed call also leaves the runtime's sticky "last error" behind, which the NEXT user of the runtime in this thread -- torch, say -- would take for
// its own: it is read out here, the error travels in the return code)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); return fail(100, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)

struct flac_const {
    uint32_t channels, sample_rate, bps, block_size, max_order;
    uint32_t nblocks; unsigned long long total_samples;
    uint32_t frame_slot;        // bytes reserved per frame in the output area
};

struct sub_plan {                // one per (block, channel)
    int32_t type;                // 0 constant, 1 verbatim, 2 fixed, 3 lpc
    int32_t order, shift, part_order;
    int32_t qlp[kMaxOrder];
    uint8_t rice_k[kMaxParts];
    unsigned long long bits;     // exact subframe size
};

__device__ __forceinline__ int32_t load_pcm(const uint8_t* p, uint32_t bps)
{
    if (bps == 8) return int32_t(p[0]) - 128;
    if (bps == 16) return int16_t(uint16_t(p[0]) | (uint16_t(p[1]) << 8));
    return int32_t((uint32_t(p[0]) << 8) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 24)) >> 8;
}
// Signal `sg` of a frame: channel sg, or for two channels 0 left, 1 right, 2 mid, 3 side.
__device__ __forceinline__ int32_t load_signal(const uint8_t* pcm, unsigned long long sample, uint32_t nch, uint32_t sg, uint32_t bps)
{
    const uint32_t bytes_ps = bps / 8;
    if (nch != 2) return load_pcm(pcm + (sample * nch + sg) * bytes_ps, bps);
    const int32_t l = load_pcm(pcm + sample * 2 * bytes_ps, bps), r = load_pcm(pcm + (sample * 2 + 1) * bytes_ps, bps);
    return sg == 0 ? l : sg == 1 ? r : sg == 2 ? (l + r) >> 1 : l - r;
}
__device__ __forceinline__ uint32_t zigzag(int32_t r) { return r >= 0 ? uint32_t(r) << 1 : ((uint32_t(-(r + 1))) << 1) | 1u; }

template <typename T> __device__ __forceinline__ T wave_sum(T v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ bool wave_any(bool b) { return __ballot(b) != 0; }

// Residual of sample i for a candidate predictor; x in LDS.  Returns false if it does not fit the format.
__device__ __forceinline__ bool residual_at(const int32_t* x, uint32_t i, int type, int order, const int32_t* q, int shift, int32_t& r)
{
    long long v;
    if (type == 2) {
        switch (order) {
        case 0: v = x[i]; break;
        case 1: v = (long long)x[i] - x[i - 1]; break;
        case 2: v = (long long)x[i] - 2ll * x[i - 1] + x[i - 2]; break;
        case 3: v = (long long)x[i] - 3ll * x[i - 1] + 3ll * x[i - 2] - x[i - 3]; break;
        default: v = (long long)x[i] - 4ll * x[i - 1] + 6ll * x[i - 2] - 4ll * x[i - 3] + x[i - 4]; break;
        }
        r = int32_t(v);
        return r != INT32_MIN;
    }
    long long acc = 0;
    for (int j = 0; j < order; j++) acc += (long long)q[j] * x[i - 1 - j];
    v = (long long)x[i] - (acc >> shift);
    r = int32_t(v);
    return v <= 0x7FFFFFFFll && v >= -0x7FFFFFFFll;
}

__device__ __forceinline__ unsigned long long rice_cost(unsigned long long U, unsigned long long m, int kmax, int& kbest)
{
    unsigned long long best = ~0ull; int kb = 0;
    for (int k = 0; k <= kmax; k++) {
        const unsigned long long c = m * (unsigned long long)(k + 1) + (U >> k);
        if (c < best) { best = c; kb = k; }
    }
    kbest = kb;
    return best;
}

// Evaluates one candidate (residual already in res[order..n)): picks partition order + Rice parameters and returns the
// exact residual bit count.  sums / ks are LDS scratch (kMaxParts entries).  All lanes return the same values.
__device__ unsigned long long plan_residual(const int32_t* res, uint32_t n, int order, int kmax, int param_bits,
                                            unsigned long long* sums, uint8_t* ks, uint8_t* ks_best, int& po_best, int lane)
{
    int pmax = 0;
    while (pmax < kMaxPartOrder && !((n >> pmax) & 1) && (n >> (pmax + 1)) > uint32_t(order)) pmax++;
    unsigned long long best = ~0ull;
    for (int po = pmax; po >= 0; po--) {
        const uint32_t parts = 1u << po, plen = n >> po;
        if (po == pmax) {
            for (uint32_t p = lane; p < parts; p += 64) {
                unsigned long long U = 0;
                for (uint32_t i = p ? p * plen : uint32_t(order); i < (p + 1) * plen; i++) U += zigzag(res[i]);
                sums[p] = U;
            }
        } else {
            unsigned long long t[4];
            for (uint32_t p = lane, c = 0; p < parts; p += 64, c++) t[c] = sums[2 * p] + sums[2 * p + 1];
            __syncthreads();
            for (uint32_t p = lane, c = 0; p < parts; p += 64, c++) sums[p] = t[c];
        }
        __syncthreads();
        unsigned long long bits = 0;
        for (uint32_t p = lane; p < parts; p += 64) {
            const unsigned long long m = p ? plen : plen - uint32_t(order);
            int k; bits += (unsigned long long)param_bits + rice_cost(sums[p], m, kmax, k);
            ks[p] = uint8_t(k);
        }
        bits = wave_sum(bits) + 6;
        __syncthreads();
        if (bits < best) {
            best = bits; po_best = po;
            for (uint32_t p = lane; p < parts; p += 64) ks_best[p] = ks[p];
        }
        __syncthreads();
    }
    // exact size with the chosen parameters
    const uint32_t parts = 1u << po_best, plen = n >> po_best;
    unsigned long long bits = 0;
    for (uint32_t i = uint32_t(order) + lane; i < n; i += 64) {
        const int k = ks_best[i / plen];
        bits += (zigzag(res[i]) >> k) + 1 + (unsigned long long)k;
    }
    return wave_sum(bits) + 6 + (unsigned long long)parts * param_bits;
}

__global__ __launch_bounds__(64) void k_flac_plan(const flac_const* __restrict__ C, const uint8_t* __restrict__ pcm,
                                                  sub_plan* __restrict__ plans, int32_t* __restrict__ residuals)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t B = C->block_size;
    int32_t* x = reinterpret_cast<int32_t*>(smem);                       // B samples
    int32_t* res = x + B;                                                // B residuals of the candidate under test
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(res + B);   // kMaxParts
    long long* xs = reinterpret_cast<long long*>(sums + kMaxParts);      // B windowed samples
    uint8_t* ks = reinterpret_cast<uint8_t*>(xs + B);                    // kMaxParts
    uint8_t* ks_cand = ks + kMaxParts;
    __shared__ sub_plan best;
    __shared__ double lp[kMaxOrder][kMaxOrder];
    __shared__ int32_t qcand[kMaxOrder];
    __shared__ int s_shift, s_ok, s_usable;

    const int lane = threadIdx.x;
    const uint32_t blk = blockIdx.x, sg = blockIdx.y, nch = C->channels, nsig = gridDim.y;
    const uint32_t bps = C->bps + (nch == 2 && sg == 3 ? 1u : 0u);          // the side signal has one bit more
    const unsigned long long first = (unsigned long long)blk * B;
    const uint32_t n = uint32_t(first + B <= C->total_samples ? B : C->total_samples - first);
    const int param_bits = bps > 16 ? 5 : 4, kmax = bps > 16 ? 30 : 14;

    for (uint32_t i = lane; i < n; i += 64) x[i] = load_signal(pcm, first + i, nch, sg, C->bps);
    __syncthreads();

    bool differs = false;
    for (uint32_t i = 1 + lane; i < n; i += 64) differs |= x[i] != x[0];
    const bool constant = !wave_any(differs);
    if (lane == 0) {
        best.type = constant ? 0 : 1; best.order = 0; best.shift = 0; best.part_order = 0;
        best.bits = constant ? 8 + bps : 8 + (unsigned long long)n * bps;
    }
    __syncthreads();
    sub_plan* out = plans + size_t(blk) * nsig + sg;
    int32_t* rout = residuals + (size_t(blk) * nsig + sg) * B;

    if (!constant) {
        int maxorder = int(C->max_order);
        if (maxorder > kMaxOrder) maxorder = kMaxOrder;
        if (uint32_t(maxorder) >= n) maxorder = int(n) - 1;
        int usable = 0;
        if (maxorder > 0) {
            // Welch window in Q15, exact integer autocorrelation (same arithmetic as oracle/flac_oracle.c autocorrelate())
            const long long d = (long long)n + 1, d2 = d * d;
            for (uint32_t i = lane; i < n; i += 64) {
                const long long c = 2ll * i - ((long long)n - 1);
                const long long w = ((d2 - c * c) << 15) / d2;
                xs[i] = ((long long)x[i] * w) >> 15;
            }
            __syncthreads();
            __shared__ long long ac[kMaxOrder + 1];
            for (int l = 0; l <= maxorder; l++) {
                long long s = 0;
                for (uint32_t i = uint32_t(l) + lane; i < n; i += 64) s += xs[i] * xs[i - uint32_t(l)];
                s = wave_sum(s);
                if (lane == 0) ac[l] = s;
            }
            __syncthreads();
            if (lane == 0) {
                // Levinson-Durbin, fixed operation order (oracle levinson())
                double a[kMaxOrder], t[kMaxOrder];
                double err = double(ac[0]);
                int m = 1, got = 0;
                if (err > 0.0) {
                    for (m = 1; m <= maxorder; m++) {
                        double acc = double(ac[m]);
                        for (int j = 1; j < m; j++) acc = acc - a[j - 1] * double(ac[m - j]);
                        const double k = acc / err;
                        for (int j = 1; j < m; j++) t[j - 1] = a[j - 1] - k * a[m - j - 1];
                        for (int j = 1; j < m; j++) a[j - 1] = t[j - 1];
                        a[m - 1] = k;
                        err = err * (1.0 - k * k);
                        for (int j = 0; j < m; j++) lp[m - 1][j] = a[j];
                        if (!(err > 0.0)) { m++; break; }
                    }
                    got = m - 1;
                }
                s_usable = got;
            }
            __syncthreads();
            usable = s_usable;
        }
        // candidates in the oracle's order: fixed 0..4, then LPC 1..usable
        for (int cand = 0; cand < 5 + usable; cand++) {
            const int type = cand < 5 ? 2 : 3;
            const int order = cand < 5 ? cand : cand - 4;
            if (uint32_t(order) >= n && type == 2) continue;
            if (type == 3) {
                if (lane == 0) {
                    // coefficient quantiser (oracle quantize())
                    const double* c = lp[order - 1];
                    double cmax = 0.0;
                    for (int j = 0; j < order; j++) { const double av = c[j] < 0 ? -c[j] : c[j]; if (av > cmax) cmax = av; }
                    int ok = (cmax > 0.0) && !(cmax > 65536.0);
                    int shift = 0;
                    if (ok) {
                        const int e = int((__double_as_longlong(cmax) >> 52) & 0x7FF) - 1022;
                        shift = kQlpPrecision - 1 - e;
                        if (shift > 15) shift = 15;
                        if (shift < 0) ok = 0;
                    }
                    if (ok) {
                        const int qmax = (1 << (kQlpPrecision - 1)) - 1, qmin = -(1 << (kQlpPrecision - 1));
                        const double scale = double(1 << shift);
                        double error = 0.0;
                        for (int j = 0; j < order; j++) {
                            error = error + c[j] * scale;
                            const double rr = error < 0 ? -error : error;
                            const double fl = double((long long)(rr + 0.5));
                            long long qi = (long long)(error < 0 ? -fl : fl);
                            if (qi > qmax) qi = qmax;
                            if (qi < qmin) qi = qmin;
                            error = error - double(qi);
                            qcand[j] = int32_t(qi);
                        }
                    }
                    s_shift = shift; s_ok = ok;
                }
                __syncthreads();
                if (!s_ok) { __syncthreads(); continue; }
            }
            bool bad = false;
            for (uint32_t i = uint32_t(order) + lane; i < n; i += 64) {
                int32_t r;
                if (!residual_at(x, i, type, order, qcand, s_shift, r)) bad = true;
                res[i] = r;
            }
            __syncthreads();
            if (wave_any(bad)) { __syncthreads(); continue; }
            int po = 0;
            const unsigned long long rbits = plan_residual(res, n, order, kmax, param_bits, sums, ks, ks_cand, po, lane);
            const unsigned long long bits = 8 + (unsigned long long)order * bps + (type == 3 ? 4 + 5 + (unsigned long long)order * kQlpPrecision : 0) + rbits;
            __syncthreads();
            if (bits < best.bits) {
                if (lane == 0) {
                    best.type = type; best.order = order; best.shift = type == 3 ? s_shift : 0; best.part_order = po; best.bits = bits;
                    for (int j = 0; j < order && type == 3; j++) best.qlp[j] = qcand[j];
                }
                for (uint32_t p = lane; p < (1u << po); p += 64) best.rice_k[p] = ks_cand[p];
                for (uint32_t i = uint32_t(order) + lane; i < n; i += 64) rout[i] = res[i];
            }
            __syncthreads();
        }
    }
    // publish
    const uint32_t words = sizeof(sub_plan) / 4;
    for (uint32_t i = lane; i < words; i += 64) reinterpret_cast<uint32_t*>(out)[i] = reinterpret_cast<const uint32_t*>(&best)[i];
}

// ------------------------------------------------------------------------------------------------------------
// Bit packing
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_bits(uint32_t* words, unsigned long long pos, uint32_t v, int n)   // n in 1..32, MSB first
{
    const uint32_t w = uint32_t(pos >> 5), o = uint32_t(pos & 31);
    const unsigned long long val = (unsigned long long)(n == 32 ? v : (v & ((1u << n) - 1))) << (64 - o - n);
    const uint32_t hi = uint32_t(val >> 32), lo = uint32_t(val);
    if (hi) atomicOr(&words[w], hi);
    if (lo) atomicOr(&words[w + 1], lo);
}
__device__ __forceinline__ uint8_t crc8_step(uint8_t c, uint8_t b) { c ^= b; for (int k = 0; k < 8; k++) c = uint8_t((c & 0x80) ? (c << 1) ^ 0x07 : c << 1); return c; }

__device__ int blocksize_code(uint32_t bs)
{
    switch (bs) {
    case 192: return 1; case 576: return 2; case 1152: return 3; case 2304: return 4; case 4608: return 5;
    case 256: return 8; case 512: return 9; case 1024: return 10; case 2048: return 11; case 4096: return 12;
    case 8192: return 13; case 16384: return 14; case 32768: return 15;
    default: return bs <= 256 ? 6 : 7;
    }
}

__global__ __launch_bounds__(64) void k_flac_write(const flac_const* __restrict__ C, const uint8_t* __restrict__ pcm,
                                                   const sub_plan* __restrict__ plans, const int32_t* __restrict__ residuals,
                                                   uint32_t* __restrict__ work /* zeroed big-endian words, frame_slot bytes per frame */,
                                                   uint8_t* __restrict__ frames, uint32_t* __restrict__ frame_sizes)
{
    __shared__ uint16_t T16[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) { uint16_t c = uint16_t(i << 8); for (int k = 0; k < 8; k++) c = uint16_t((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1); T16[i] = c; }
    const uint32_t blk = blockIdx.x, B = C->block_size, nch = C->channels, nsig = nch == 2 ? 4u : nch;
    const unsigned long long first = (unsigned long long)blk * B;
    const uint32_t n = uint32_t(first + B <= C->total_samples ? B : C->total_samples - first);
    // two channels: the pair of signals with the fewest bits, the plain pair on a tie (the rule oracle/flac_oracle.c states)
    uint32_t assignment = nch - 1, sig0 = 0, sig1 = 1;
    if (nch == 2) {
        const sub_plan* q = plans + size_t(blk) * 4;
        const unsigned long long lr = q[0].bits + q[1].bits, ls = q[0].bits + q[3].bits, rs = q[3].bits + q[1].bits, ms = q[2].bits + q[3].bits;
        unsigned long long best = lr; assignment = 1;
        if (ls < best) { best = ls; assignment = 8; }
        if (rs < best) { best = rs; assignment = 9; }
        if (ms < best) { best = ms; assignment = 10; }
        if (assignment == 8) sig1 = 3; else if (assignment == 9) sig0 = 3; else if (assignment == 10) { sig0 = 2; sig1 = 3; }
    }
    uint32_t* words = work + size_t(blk) * (C->frame_slot / 4);
    uint8_t* fout = frames + size_t(blk) * C->frame_slot;

    // frame header (stream_decoder.c:2159-2466), fixed blocking strategy, then CRC-8
    __shared__ uint8_t hdr[16]; __shared__ int hlen;
    if (lane == 0) {
        int h = 0;
        hdr[h++] = 0xFF; hdr[h++] = 0xF8;
        const int bsc = blocksize_code(n);
        const int src = C->sample_rate == 44100 ? 9 : C->sample_rate == 48000 ? 10 : C->sample_rate == 96000 ? 11 : 0;
        hdr[h++] = uint8_t((bsc << 4) | src);
        hdr[h++] = uint8_t((assignment << 4) | ((C->bps == 8 ? 1 : C->bps == 16 ? 4 : 6) << 1));
        unsigned long long v = blk;                                 // UTF-8 coded frame number
        if (v < 0x80) hdr[h++] = uint8_t(v);
        else {
            const int cnt = v < 0x800 ? 2 : v < 0x10000 ? 3 : v < 0x200000 ? 4 : v < 0x4000000 ? 5 : v < 0x80000000ull ? 6 : 7;
            const uint8_t lead[8] = { 0, 0, 0xC0, 0xE0, 0xF0, 0xF8, 0xFC, 0xFE };
            for (int i = cnt - 1; i > 0; i--) { hdr[h + i] = uint8_t(0x80 | (v & 0x3F)); v >>= 6; }
            hdr[h] = uint8_t(lead[cnt] | v); h += cnt;
        }
        if (bsc == 6) hdr[h++] = uint8_t(n - 1);
        else if (bsc == 7) { hdr[h++] = uint8_t((n - 1) >> 8); hdr[h++] = uint8_t(n - 1); }
        uint8_t c = 0; for (int i = 0; i < h; i++) c = crc8_step(c, hdr[i]);
        hdr[h++] = c;
        hlen = h;
    }
    __syncthreads();
    unsigned long long pos = (unsigned long long)hlen * 8;
    for (int i = lane; i < hlen; i += 64) put_bits(words, (unsigned long long)i * 8, hdr[i], 8);

    for (uint32_t ch = 0; ch < nch; ch++) {
        const uint32_t sg = nch == 2 ? (ch ? sig1 : sig0) : ch;
        const uint32_t bps = C->bps + (nch == 2 && sg == 3 ? 1u : 0u);
        const int param_bits = bps > 16 ? 5 : 4;
        const sub_plan* sp = plans + size_t(blk) * nsig + sg;
        const int32_t* res = residuals + (size_t(blk) * nsig + sg) * B;
        const int type = sp->type, order = sp->order;
        auto sample = [&](uint32_t i) { return load_signal(pcm, first + i, nch, sg, C->bps); };
        if (type == 0) {
            if (lane == 0) { put_bits(words, pos, 0x00, 8); put_bits(words, pos + 8, uint32_t(sample(0)), int(bps)); }
        } else if (type == 1) {
            if (lane == 0) put_bits(words, pos, 0x02, 8);
            for (uint32_t i = lane; i < n; i += 64) put_bits(words, pos + 8 + (unsigned long long)i * bps, uint32_t(sample(i)), int(bps));
        } else {
            unsigned long long p = pos;
            if (lane == 0) put_bits(words, p, type == 2 ? uint32_t(0x10 | (order << 1)) : uint32_t(0x40 | ((order - 1) << 1)), 8);
            p += 8;
            for (int i = lane; i < order; i += 64) put_bits(words, p + (unsigned long long)i * bps, uint32_t(sample(uint32_t(i))), int(bps));
            p += (unsigned long long)order * bps;
            if (type == 3) {
                if (lane == 0) { put_bits(words, p, kQlpPrecision - 1, 4); put_bits(words, p + 4, uint32_t(sp->shift), 5); }
                for (int j = lane; j < order; j += 64) put_bits(words, p + 9 + (unsigned long long)j * kQlpPrecision, uint32_t(sp->qlp[j]), kQlpPrecision);
                p += 9 + (unsigned long long)order * kQlpPrecision;
            }
            if (lane == 0) { put_bits(words, p, param_bits == 5 ? 1 : 0, 2); put_bits(words, p + 2, uint32_t(sp->part_order), 4); }
            p += 6;
            // residual codes: every lane takes a contiguous run of samples; code lengths are prefix-summed across lanes
            const uint32_t plen = n >> sp->part_order;
            const uint32_t nres = n - uint32_t(order);
            const uint32_t per = (nres + 63) / 64;
            const uint32_t i0 = min(n, uint32_t(order) + lane * per), i1 = min(n, i0 + per);
            unsigned long long mine = 0;
            for (uint32_t i = i0; i < i1; i++) {
                const uint32_t pi = i / plen; const int k = sp->rice_k[pi];
                if (i == (pi ? pi * plen : uint32_t(order))) mine += param_bits;
                mine += (zigzag(res[i]) >> k) + 1 + (unsigned long long)k;
            }
            unsigned long long incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            unsigned long long q = p + incl - mine;
            for (uint32_t i = i0; i < i1; i++) {
                const uint32_t pi = i / plen; const int k = sp->rice_k[pi];
                if (i == (pi ? pi * plen : uint32_t(order))) { put_bits(words, q, uint32_t(k), param_bits); q += param_bits; }
                const uint32_t u = zigzag(res[i]);
                q += u >> k;                                                  // unary zeros are already there
                put_bits(words, q, (1u << k) | (k ? (u & ((1u << k) - 1)) : 0u), k + 1);
                q += k + 1;
            }
        }
        pos += sp->bits;
    }
    __threadfence_block();
    __syncthreads();
    // words -> bytes, zero padding to a byte boundary is implicit; CRC-16 over everything before it
    const uint32_t nbytes = uint32_t((pos + 7) / 8);
    for (uint32_t i = lane; i < (nbytes + 3) / 4; i += 64) {
        const uint32_t w = __builtin_bswap32(__hip_atomic_load(&words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        reinterpret_cast<uint32_t*>(fout)[i] = w;
    }
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {
        uint16_t c = 0;
        for (uint32_t i = 0; i < nbytes; i++) c = uint16_t((c << 8) ^ T16[(c >> 8) ^ fout[i]]);
        fout[nbytes] = uint8_t(c >> 8); fout[nbytes + 1] = uint8_t(c);
        frame_sizes[blk] = nbytes + 2;
    }
}

uint32_t default_block_size(uint32_t rate)
{
    static const uint32_t tab[] = { 192, 256, 512, 576, 1024, 1152, 2048, 2304, 4096, 4608, 8192, 16384 };
    const uint64_t target = uint64_t(rate) * 105 / 1000;
    uint32_t best = 192;
    for (uint32_t t : tab) if (t <= target) best = t;
    return best;
}

}  // namespace

struct rcgpu_flac {
    rcgpu_flac_config cfg{};
    uint32_t block_size = 0;
    uint64_t total_samples = 0;
    uint32_t min_frame = 0, max_frame = 0;
    uint8_t md5[16] = { 0 };
    bool encoded = false;
};

extern "C" int rcgpu_flac_create(const rcgpu_flac_config* cfg, rcgpu_flac** out)
{
    clear_error();
    if (!cfg || !out) return fail(1, "flac: null argument");
    *out = nullptr;
    if (cfg->channels < 1 || cfg->channels > 8) return fail(2, "flac: %u channels", cfg->channels);
    if (cfg->bits_per_sample != 8 && cfg->bits_per_sample != 16 && cfg->bits_per_sample != 24) return fail(2, "flac: %u bits per sample", cfg->bits_per_sample);
    if (!cfg->sample_rate || cfg->sample_rate > 655350) return fail(2, "flac: sample rate %u", cfg->sample_rate);
    if (cfg->max_lpc_order > kMaxOrder) return fail(2, "flac: LPC order %u > 32", cfg->max_lpc_order);
    const uint32_t B = cfg->block_size ? cfg->block_size : default_block_size(cfg->sample_rate);
    if (B < 16 || B > 8192) return fail(2, "flac: block size %u outside 16..8192 (LDS budget of k_flac_plan; the reference reader allows up to 16384, Wrapper.cpp:249-251)", B);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(3, "flac: no HIP device available -- this encoder has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(3, "flac: device %d out of range", cfg->device);
    rcgpu_flac* e = new rcgpu_flac;
    e->cfg = *cfg; e->block_size = B;
    *out = e;
    return 0;
}

extern "C" void rcgpu_flac_destroy(rcgpu_flac* e) { delete e; }

extern "C" int rcgpu_flac_encode_host(rcgpu_flac* e, const uint8_t* pcm, uint64_t pcm_bytes, uint8_t* frames_out, size_t cap,
                                      uint32_t* frame_sizes, uint32_t frame_cap, uint32_t* n_frames)
{
    clear_error();
    if (!e || !pcm || !frames_out || !frame_sizes || !n_frames) return fail(1, "flac: null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint32_t ch = e->cfg.channels, bps = e->cfg.bits_per_sample, B = e->block_size;
    const uint64_t total = pcm_bytes / (uint64_t(bps / 8) * ch);
    const uint64_t nblocks = (total + B - 1) / B;
    *n_frames = 0;
    if (!total) { e->total_samples = 0; e->encoded = true; return 0; }
    if (nblocks > frame_cap) return fail(2, "flac: %llu frames do not fit frame_sizes[%u]", (unsigned long long)nblocks, frame_cap);

    flac_const hc{};
    hc.channels = ch; hc.sample_rate = e->cfg.sample_rate; hc.bps = bps; hc.block_size = B; hc.max_order = e->cfg.max_lpc_order;
    hc.nblocks = uint32_t(nblocks); hc.total_samples = total;
    hc.frame_slot = uint32_t((16 + size_t(B) * ch * (bps / 8) + ch + 2 + 8 + 15) & ~size_t(15));

    flac_const* d_c = nullptr; uint8_t* d_pcm = nullptr; sub_plan* d_plans = nullptr; int32_t* d_res = nullptr;
    uint32_t* d_work = nullptr; uint8_t* d_frames = nullptr; uint32_t* d_sizes = nullptr;
    auto cleanup = [&]() { for (void* p : { (void*)d_c, (void*)d_pcm, (void*)d_plans, (void*)d_res, (void*)d_work, (void*)d_frames, (void*)d_sizes }) if (p) (void)hipFree(p); };
    hipError_t he = hipSuccess;
    const size_t slots = size_t(nblocks) * hc.frame_slot;
#define DM(p, b) if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&(p)), (b))
    const uint32_t nsig = ch == 2 ? 4u : ch;              // two channels: left, right, mid, side
    DM(d_c, sizeof hc); DM(d_pcm, pcm_bytes + 16); DM(d_plans, sizeof(sub_plan) * nblocks * nsig); DM(d_res, size_t(nblocks) * nsig * B * 4);
    DM(d_work, slots + 16); DM(d_frames, slots + 16); DM(d_sizes, nblocks * 4);
#undef DM
    if (he == hipSuccess) he = hipMemcpy(d_c, &hc, sizeof hc, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(d_pcm, pcm, pcm_bytes, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemset(d_work, 0, slots + 16);
    if (he != hipSuccess) { (void)hipGetLastError(); cleanup(); return fail(100, "flac: device setup failed: %s", hipGetErrorString(he)); }
    const size_t lds = size_t(B) * 4 * 2 + kMaxParts * 8 + size_t(B) * 8 + 2 * kMaxParts;
    he = hipFuncSetAttribute(reinterpret_cast<const void*>(k_flac_plan), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (he == hipSuccess) {
        hipLaunchKernelGGL(k_flac_plan, dim3(uint32_t(nblocks), nsig), dim3(64), lds, 0, d_c, d_pcm, d_plans, d_res);
        hipLaunchKernelGGL(k_flac_write, dim3(uint32_t(nblocks)), dim3(64), 0, 0, d_c, d_pcm, d_plans, d_res, d_work, d_frames, d_sizes);
        he = hipGetLastError();
    }
    std::vector<uint32_t> sizes(nblocks);
    if (he == hipSuccess) he = hipMemcpy(sizes.data(), d_sizes, nblocks * 4, hipMemcpyDeviceToHost);
    std::vector<uint8_t> slotbuf;
    if (he == hipSuccess) { slotbuf.resize(slots); he = hipMemcpy(slotbuf.data(), d_frames, slots, hipMemcpyDeviceToHost); }
    cleanup();
    if (he != hipSuccess) return fail(100, "flac: device encode failed: %s", hipGetErrorString(he));
    size_t pos = 0; uint32_t mn = ~0u, mx = 0;
    for (uint64_t i = 0; i < nblocks; i++) {
        if (pos + sizes[i] > cap) return fail(2, "flac: output buffer too small");
        memcpy(frames_out + pos, slotbuf.data() + i * hc.frame_slot, sizes[i]);
        frame_sizes[i] = sizes[i]; pos += sizes[i];
        mn = std::min(mn, sizes[i]); mx = std::max(mx, sizes[i]);
    }
    *n_frames = uint32_t(nblocks);
    e->total_samples = total; e->min_frame = mn; e->max_frame = mx; e->encoded = true;
    // STREAMINFO MD5 of the signed little-endian samples (8-bit WAV is offset binary)
    if (bps == 8) {
        std::vector<uint8_t> s(pcm, pcm + total * ch);
        for (uint8_t& b : s) b = uint8_t(b - 128);
        rcgpu_md5(s.data(), s.size(), e->md5);
    } else
        rcgpu_md5(pcm, size_t(total) * ch * (bps / 8), e->md5);
    return 0;
}

extern "C" size_t rcgpu_flac_codec_private(const rcgpu_flac* e, uint8_t* out, size_t cap)
{
    if (!e || cap < 42 || !out) return 0;
    uint8_t* p = out;
    memcpy(p, "fLaC", 4); p += 4;
    *p++ = 0x80; *p++ = 0; *p++ = 0; *p++ = 34;
    const uint32_t B = e->block_size;
    *p++ = uint8_t(B >> 8); *p++ = uint8_t(B); *p++ = uint8_t(B >> 8); *p++ = uint8_t(B);
    *p++ = uint8_t(e->min_frame >> 16); *p++ = uint8_t(e->min_frame >> 8); *p++ = uint8_t(e->min_frame);
    *p++ = uint8_t(e->max_frame >> 16); *p++ = uint8_t(e->max_frame >> 8); *p++ = uint8_t(e->max_frame);
    const uint64_t v = (uint64_t(e->cfg.sample_rate) << 44) | (uint64_t(e->cfg.channels - 1) << 41) | (uint64_t(e->cfg.bits_per_sample - 1) << 36) |
                       (e->total_samples & 0xFFFFFFFFFull);
    for (int s = 56; s >= 0; s -= 8) *p++ = uint8_t(v >> s);
    memcpy(p, e->md5, 16); p += 16;
    return size_t(p - out);
}
