// ffv1_check.hip -- the `--check` half on the device (BASELINE config 5): FFV1 packets -> rebuilt payload bytes,
// byte comparison with the source and MD5, for MI355X (gfx950).
//
// What it restates: ffv1_frame::Process (Lib/CoDec/FFV1/FFV1_Frame.cpp:134-228: slice split from the tail sizes),
// slice::Parse / SliceHeader / Line (FFV1_Slice.cpp:113-177,210-318,447-472), rangecoder::b/u/s
// (FFV1_RangeCoder.cpp:71-305), the slice CRC (FFV1_Slice.cpp:247-249), Transform::From (Lib/Transform/Transform.cpp:
// inverse RCT + packers) and the compare / MD5 of frame_writer (Lib/Utils/FileIO/FileWriter.cpp:448-463,596-727).
//
// Decoding is serial per slice in both recurrences AND in the context model (the context of a sample depends on the
// sample decoded just before it), so there is nothing for a wavefront to share: the mapping is one LANE per slice,
// 64 slices per wavefront, thousands of slices in flight.  Context states live in HBM (32 bytes per context, one
// gather + one write-back per sample into eight registers); the previous lines of a slice are kept in three-line rings
// interleaved per wavefront; the slice's bytes come through a register window that is taken from without a look.
// What bounds it: the memory system's rate of random 32-byte gathers + write-backs in this pattern, every lane in a state array of
// its own (tools/gather_region: 19.6 G records/s; 1600 4K frames in flight run at 0.99 of it alone, 0.85-0.9 with the hash of the
// previous batch beside them -- DESIGN.md section 5).  The kernel ends with its slowest wavefront, so nothing may share a SIMD with
// some of its wavefronts only: it raises its wave priority, and the hash runs on CUs of its own (partition_streams below).
// Host side: a batch's payloads stay on the device in one of three sets of slots (current / under verification / decoded ahead:
// rcgpu_ffv1_decoder_decode_keep_hint decodes the next batch on a thread of the library's own while the caller verifies this one).
//   k_dec_split   thread / frame    walk the 24-bit slice sizes from the packet tail
//   k_dec_crc     block  / slice    CRC-32 over the whole slice must be 0 (ec = 1)
//   k_dec_slices  LANE   / slice    range decoder + median predictor + contexts; whole-byte layouts: inverse RCT + pack of every
//                                   finished line by the same lane (three-line rings, no planes); word-stream layouts: planar int32
//   k_pack_words  thread / word     word-stream layouts only: inverse RCT + pack, one thread per 32-bit word of the payload
//   k_compare     grid-stride       byte compare of two buffers -> first mismatch
//   k_md5         LANE   / buffer   RFC 1321, one buffer per lane
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <mutex>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "ffv1_host.h"
#include "rc_common.h"
#include "crc_dev.h"

using namespace rc;

namespace {

// (a failed call also leaves the runtime's sticky "last error" behind, which the NEXT user of the runtime in this thread -- torch, say -- would take for
// its own: it is read out here, the error travels in the return code)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); return fail(100, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)

struct dec_const {
    uint32_t W, H, line_bytes, pixfmt;
    uint32_t planes, bps, bits, rgb, gb_swap, big_endian, bytes_pp, overflow16;
    uint32_t fields, fill, vflip, altern;  // payload layout of the bit-packed DPX flavors (rc_common.h kFields*), RCGPU_FLAG_*
    uint32_t num_h, num_v, S, ngroups, ec, index_count;
    // per plane group (0 = Y, 1 = Cb and Cr, 2 = alpha; Line(), FFV1_Slice.cpp:447-455): the quant_table_set_index every slice header must
    // name, whether its table set looks at five neighbours (QuantTables[3][127] != 0, :453), its contexts and where they start in a lane's states
    uint32_t idx[3], is5[3], nctx[3], kbase[3];
    uint32_t v1, hdr_n;                    // FFV1 version 0 / 1: one slice = the packet, hdr_n header decisions in front of it, no footer
    uint32_t win_cap;                      // bytes a lane's window is filled up to (7; rcgpu_ffv1_decoder_debug_window makes it less, for the tests of the careful path)
    uint32_t qslot[3], nqslots;            // the DISTINCT table sets of the plane groups, q[0 .. nqslots), and which of them each group uses: the kernel
                                           // stages only those in LDS (2.5 KB each; one for every stream this encoder writes)
    int16_t  q[3][5][256];
    uint8_t  one_state[256], zero_state[256];
};

// ------------------------------------------------------------------------------------------------------------
__global__ void k_dec_split(const dec_const* __restrict__ C, const uint8_t* const* __restrict__ packets,
                            const unsigned long long* __restrict__ sizes, uint32_t n,
                            unsigned long long* __restrict__ slice_start, uint32_t* __restrict__ slice_len, uint32_t* __restrict__ err)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const uint8_t* p = packets[f];
    const uint32_t tail = C->ec ? 8 : 3, S = C->S;
    if (C->v1) { slice_start[f] = 0; slice_len[f] = uint32_t(sizes[f]); return; }       // version 1: the packet is the one slice
    unsigned long long pos = sizes[f];
    uint32_t count = 0;
    while (pos && count < S) {                               // FFV1_Frame.cpp:177-198
        if (pos < tail) { atomicOr(err, 1u); break; }
        const unsigned long long sz = ((unsigned long long)p[pos - tail] << 16 | (unsigned long long)p[pos - tail + 1] << 8 | p[pos - tail + 2]) + tail;
        if (sz > pos) { atomicOr(err, 1u); break; }
        pos -= sz;
        slice_start[f * S + count] = pos; slice_len[f * S + count] = uint32_t(sz);
        count++;
    }
    if (pos || count != S) atomicOr(err, 2u);
    for (; count < S; count++) { slice_start[f * S + count] = 0; slice_len[f * S + count] = 0; }
}

__global__ __launch_bounds__(256) void k_dec_crc(const dec_const* __restrict__ C, const uint8_t* const* __restrict__ packets,
                                                 const unsigned long long* __restrict__ slice_start, const uint32_t* __restrict__ slice_len,
                                                 uint32_t* __restrict__ err)
{
    // CRC over the whole slice incl. its stored CRC must be 0 (FFV1_Slice.cpp:248).  Slices start anywhere in the packet: the
    // bytes in front of the first 16-byte boundary go to thread 0, the aligned body to the tiled block CRC of crc_dev.h.
    __shared__ uint32_t T[4][256]; __shared__ uint32_t TM[4][256]; __shared__ uint32_t part[4];
    const int tid = threadIdx.x;
    crc_tables(T, tid);
    const uint32_t chain = blockIdx.x, f = chain / C->S;
    const uint8_t* p = packets[f] + slice_start[chain];
    const uint32_t total = slice_len[chain];
    const uint32_t head = min(total, uint32_t((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15));
    uint32_t c = block_crc_share(p + head, total - head, T, TM, tid);
    if (tid == 0 && head) {
        uint32_t h = 0;
        for (uint32_t i = 0; i < head; i++) h = (h << 8) ^ T[0][(h >> 24) ^ p[i]];
        c ^= gf_mulmod(h, gf_xpow8(total - head));
    }
    for (int o = 32; o; o >>= 1) c ^= __shfl_xor(c, o);
    if ((tid & 63) == 0) part[tid >> 6] = c;
    __syncthreads();
    if (tid == 0 && (part[0] ^ part[1] ^ part[2] ^ part[3])) atomicOr(err, 4u);       // FFV1-SLICE-slice_crc_parity
}

// ------------------------------------------------------------------------------------------------------------
// Range decoder state of one lane (rangecoder, FFV1_RangeCoder.cpp:21-102)
// The compressed bytes reach the decoder through a register window: `win` holds the next <= 8 bytes of the slice (left aligned),
// `pend` a further load that was issued one sample earlier.  A renormalisation is then pure ALU work -- with 64 slices decoding in
// lock-step some lane renormalises at almost every decision, and a global byte load there stalls the whole wavefront every time.
struct rd_lane {
    uint32_t current, mask;
    uint32_t pos, n;                  // bytes consumed as of the last refill (Buffer_Cur - Buffer), size of the slice's coded data
    // The window: the next `have` (<= 7) bytes of the slice, left aligned in win_hi:win_lo, then a SENTINEL byte 0x80, then zeros.  A
    // renormalisation shifts the whole window left by a byte and keeps no count: how many bytes are left is read off the sentinel -- the
    // lowest set bit of the window -- when somebody asks (once per sample).  A window of all zeros means the sentinel itself was consumed as
    // data: the sample read more bytes than the window held (rd_underflow), and is decoded again by the careful decoder below.
    uint32_t win_hi, win_lo, have, cap;           // cap: the window is filled up to this many bytes (7)
    unsigned long long pend; uint32_t npend;      // the next bytes behind the window, big end first
    // what the last load brought, AS LOADED (nraw halves of 8 bytes, little-endian dwords): nobody touches it in the sample that issued the
    // load -- a use right behind the load (the byte swap, say) would put the whole memory latency into that sample
    uint32_t raw[4]; uint32_t nraw;
    const uint8_t* next; const uint8_t* end;      // first byte not yet loaded, end of the coded data
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) uint8_t* gbytes_t;      // global memory, said so: a generic pointer makes these loads flat_load, which count as LDS traffic too

// The slice's bytes come 16 at a time (one request per lane and 16 bytes: every lane reads a line of its own, and a line that is asked
// for eight bytes at a time is fetched eight times once thousands of wavefronts share the caches); the last bytes one by one.
__device__ __forceinline__ void rd_load(rd_lane& r)
{
    const uint32_t avail = r.next < r.end ? uint32_t(r.end - r.next) : 0u;
    gbytes_t g = (gbytes_t)r.next;
    if (avail >= 16) {
        const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(1), aligned(1))) u32x4_t*>(g);     // unaligned loads are fine in global memory
        r.raw[0] = v.x; r.raw[1] = v.y; r.raw[2] = v.z; r.raw[3] = v.w;
        r.nraw = 2; r.next += 16;
    } else if (avail >= 8) {
        r.raw[0] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(g);
        r.raw[1] = *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(g + 4);
        r.nraw = 1; r.next += 8;
    } else {
        uint32_t lo = 0, hi = 0;
        for (uint32_t i = 0; i < avail; i++) { const uint32_t b = g[i]; if (i < 4) lo |= b << (8 * i); else hi |= b << (8 * (i - 4)); }   // last bytes of the slice: never read past them
        r.raw[0] = lo; r.raw[1] = hi;                                                                 // ... and zeros after them (FFV1_RangeCoder.cpp:79-85)
        r.nraw = 1; r.next += avail;
    }
}
// the next 8 loaded bytes become `pend` (this is where loaded data is first looked at: a sample or more after its load was issued)
__device__ __forceinline__ void rd_promote(rd_lane& r)
{
    r.pend = (unsigned long long)__builtin_bswap32(r.raw[0]) << 32 | __builtin_bswap32(r.raw[1]);
    r.raw[0] = r.raw[2]; r.raw[1] = r.raw[3];
    r.nraw--; r.npend = 8;
}
__device__ __forceinline__ bool rd_underflow(const rd_lane& r) { return (r.win_hi | r.win_lo) == 0; }
// valid bytes left in the window: the sentinel's 0x80 is the lowest set bit, at bit 63 - 8 * valid
__device__ __forceinline__ uint32_t rd_valid(const rd_lane& r)
{
    const uint32_t low = r.win_lo ? uint32_t(__builtin_ctz(r.win_lo)) : 32u + uint32_t(__builtin_ctz(r.win_hi | 0x80000000u));
    return (63u - low) >> 3;
}
// At a sample boundary (all lanes converged, no underflow): account for what was consumed, move arrived bytes into the window behind the
// valid ones, set the sentinel anew, then put the next load in flight.
__device__ __forceinline__ void rd_refill(rd_lane& r)
{
    if (!r.npend && r.nraw) rd_promote(r);
    const uint32_t v = rd_valid(r);
    r.pos += r.have - v;
    const uint32_t take = min(r.npend, r.cap > v ? r.cap - v : 0u);
    unsigned long long w = (unsigned long long)r.win_hi << 32 | r.win_lo;
    w &= w - 1;                                                                   // the sentinel goes
    const uint32_t nv = v + take;                                                 // <= 7
    w |= r.pend >> (8 * v);
    w &= nv ? ~0ull << (8 * (8 - nv)) : 0ull;                                     // only the bytes taken
    w |= 0x80ull << (8 * (7 - nv));                                               // ... and the sentinel behind them
    r.win_hi = uint32_t(w >> 32); r.win_lo = uint32_t(w);
    r.have = nv;
    r.pend = take < 8 ? r.pend << (8 * take) : 0ull;
    r.npend -= take;
    if (!r.npend && !r.nraw) rd_load(r);                                          // (issued here, looked at a sample later)
}
// exact count of consumed bytes right now (end of slice)
__device__ __forceinline__ uint32_t rd_pos(const rd_lane& r) { return r.pos + r.have - rd_valid(r); }

// A lane's 32 context states live in LDS as eight dwords of a [8][64] array (dword k of lane l at (k*64 + l)*4): for a state index
// that is uniform over the wavefront every lane touches its own dword -- no bank conflicts, where a [lane][32] layout gives 16-way.
#define ST_AT(base, k) ((base)[(uint32_t(k) >> 2) * 256 + (uint32_t(k) & 3)])

// rangecoder::b (FFV1_RangeCoder.cpp:71-102), the CAREFUL form: before a renormalisation it looks whether the window still holds a byte and
// refills on the spot if not.  Slice headers, the end-of-slice bit, and the rare sample that outruns its window (rd_s_careful).
__device__ __forceinline__ uint32_t rd_bit(rd_lane& r, uint8_t* base, int k, const uint8_t* trans)
{
    const bool need = r.mask < 0x100;
    const bool empty = r.win_hi == 0x80000000u && r.win_lo == 0;       // nothing but the sentinel
    if (__builtin_expect(__ballot(need && empty) != 0, 0)) {
        if (need && empty) { rd_refill(r); if (r.win_hi == 0x80000000u && r.win_lo == 0) rd_refill(r); }      // (the first call may only have fetched; cap >= 1 byte then arrives)
    }
    const uint32_t b = r.win_hi >> 24;
    r.current = need ? (r.current << 8) | b : r.current;
    r.mask = need ? r.mask << 8 : r.mask;
    const uint32_t nhi = __builtin_amdgcn_alignbit(r.win_hi, r.win_lo, 24);      // (hi:lo) << 8
    r.win_hi = need ? nhi : r.win_hi;
    r.win_lo = need ? r.win_lo << 8 : r.win_lo;
    const uint32_t s = ST_AT(base, k);
    const uint32_t m2 = (r.mask * s) >> 8;
    const uint32_t nm = r.mask - m2;
    const bool bit = r.current >= nm;
    r.current -= bit ? nm : 0u;
    r.mask = bit ? m2 : nm;
    ST_AT(base, k) = trans[(bit ? 256u : 0u) + s];
    return bit ? 1u : 0u;
}
// (slice header only; inlined so that the lane's coder state never has its address taken and stays in registers)
__device__ __forceinline__ uint32_t rd_u(rd_lane& r, uint8_t* st, const uint8_t* trans)
{
    if (rd_bit(r, st, 0, trans)) return 0;
    int e = 0;
    while (rd_bit(r, st, 1 + (e < 9 ? e : 9), trans)) { if (++e > 31) return 0; }
    uint32_t a = 1;
    for (int i = e - 1; i >= 0; i--) a = (a << 1) | rd_bit(r, st, 22 + (i < 9 ? i : 9), trans);
    return a;
}
// rangecoder::s (FFV1_RangeCoder.cpp:206-236) in the careful form, states in the lane's LDS slot: for the sample that outran its window
__device__ __forceinline__ int32_t rd_s_careful(rd_lane& r, uint8_t* st, const uint8_t* trans)
{
    if (rd_bit(r, st, 0, trans)) return 0;
    int e = 0;
    while (rd_bit(r, st, 1 + (e < 9 ? e : 9), trans)) { if (++e > 31) return 0; }
    int32_t a = 1;
    for (int i = e - 1; i >= 0; i--) a = (a << 1) | int32_t(rd_bit(r, st, 22 + (i < 9 ? i : 9), trans));
    return rd_bit(r, st, 11 + (e < 10 ? e : 10), trans) ? -a : a;
}
// ---- the same symbol decoder for the samples, with the context's 32 states in eight REGISTERS (state k = byte k & 3 of w[k >> 2], the
// record as it lies in HBM).  A chain's time is its latency: ~32 decision slots per sample (the wavefront walks the longest lane's
// exponent), and with the states in LDS a slot was two dependent LDS round trips (state, then the transition table) plus the arithmetic.
// Here every state index is a compile-time constant -- the exponent loop and the mantissa loop are unrolled over their whole range
// (FFV1_RangeCoder.cpp:206-236 allows e up to 31) and the lanes a step does not concern are masked, a step that concerns no lane is
// skipped by a uniform branch -- so a state is a bit-field extract, and the one LDS access left, the transition pair of that state
// (t16[s] = zero_state[s] | one_state[s] << 8), is issued before the arithmetic that decides which half is wanted.
template <int K> __device__ __forceinline__ uint32_t st_get(const uint32_t (&w)[8]) { return (w[K >> 2] >> (8 * (K & 3))) & 0xFFu; }
template <int K> __device__ __forceinline__ void st_put(uint32_t (&w)[8], uint32_t v)
{
    w[K >> 2] = (w[K >> 2] & ~(0xFFu << (8 * (K & 3)))) | (v << (8 * (K & 3)));
}
// one decision of the lanes in `active` against state s; ns = the state afterwards (s itself for the other lanes).  The FAST form: a
// renormalisation takes the window's top byte and shifts the window, whatever is in it -- no look at how much is left, no counters (the
// sentinel keeps the count).  In lock-step some lane renormalises at almost every decision; the three instructions and the branch the
// look cost were paid by all 64 slices every time.  A lane that runs dry reads its sentinel and then zeros: rd_underflow() says so
// after the sample, and the sample is decoded again carefully.
__device__ __forceinline__ uint32_t rd_core(rd_lane& r, uint32_t s, const uint16_t* t16, bool active, uint32_t& ns)
{
    const uint32_t t2 = t16[s];
    uint32_t bit = 0;
    ns = s;
    if (active) {
        const bool need = r.mask < 0x100;
        const uint32_t cur2 = __builtin_amdgcn_alignbit(r.current, r.win_hi, 24);     // (current << 8) | top byte of the window
        r.current = need ? cur2 : r.current;
        r.mask = need ? r.mask << 8 : r.mask;
        const uint32_t nhi = __builtin_amdgcn_alignbit(r.win_hi, r.win_lo, 24);       // (hi:lo) << 8
        r.win_hi = need ? nhi : r.win_hi;
        r.win_lo = need ? r.win_lo << 8 : r.win_lo;
        const uint32_t m2 = __umul24(r.mask, s) >> 8;                 // mask < 2^16 after the renormalisation, s < 2^8
        const uint32_t nm = r.mask - m2;
        const bool one = r.current >= nm;
        r.current -= one ? nm : 0u;
        r.mask = one ? m2 : nm;
        ns = one ? t2 >> 8 : t2 & 0xFFu;
        bit = one ? 1u : 0u;
    }
    return bit;
}
template <int K> __device__ __forceinline__ uint32_t rd_bit_k(rd_lane& r, uint32_t (&w)[8], const uint16_t* t16, bool active)
{
    uint32_t ns;
    const uint32_t bit = rd_core(r, st_get<K>(w), t16, active, ns);
    st_put<K>(w, ns);
    return bit;
}
template <int J> struct rd_steps {
    // exponent: step J reads state 1 + min(J, 9); `going` = the lanes whose ones have not ended yet.  Returns the number of steps that ran (a
    // uniform value: every branch here is taken by the whole wavefront)
    static __device__ __forceinline__ int unary(rd_lane& r, uint32_t (&w)[8], const uint16_t* t16, bool& going, uint32_t& e, bool& over)
    {
        if (__builtin_amdgcn_ballot_w64(going) == 0) return J;
        const uint32_t b = rd_bit_k<1 + (J < 9 ? J : 9)>(r, w, t16, going);
        going = going && b != 0;
        e += going ? 1u : 0u;
        if (J == 31) { over = going; going = false; return 32; }     // a 32nd one: the value is 0 (`if (++e > 31) return 0`)
        else return rd_steps<J + 1>::unary(r, w, t16, going, e, over);
    }
};
template <> struct rd_steps<32> {
    static __device__ __forceinline__ int unary(rd_lane&, uint32_t (&)[8], const uint16_t*, bool&, uint32_t&, bool&) { return 32; }
};
// mantissa: bit I (from e - 1 down to 0) reads state 22 + min(I, 9).  The exponent chain ran `steps` steps, so the largest exponent of the
// wavefront is steps - 1 and its first mantissa bit is I = steps - 2: steps above that are skipped on a SCALAR comparison (two SALU
// instructions; asking the lanes -- compare, ballot, branch -- was five, for each of 32 steps of every sample: a ninth of the kernel's
// instructions), and from there down every step concerns at least that lane.
template <int I> struct rd_man {
    static __device__ __forceinline__ void run(rd_lane& r, uint32_t (&w)[8], const uint16_t* t16, bool live, uint32_t e, int32_t& a, int top)
    {
        if (I <= top) {
            const bool act = live && uint32_t(I) < e;
            const uint32_t b = rd_bit_k<22 + (I < 9 ? I : 9)>(r, w, t16, act);
            a = act ? (a << 1) | int32_t(b) : a;
        }
        rd_man<I - 1>::run(r, w, t16, live, e, a, top);
    }
};
template <> struct rd_man<-1> { static __device__ __forceinline__ void run(rd_lane&, uint32_t (&)[8], const uint16_t*, bool, uint32_t, int32_t&, int) {} };
__device__ __forceinline__ void rd_mantissa(rd_lane& r, uint32_t (&w)[8], const uint16_t* t16, bool live, uint32_t e, int32_t& a, int steps)
{
    rd_man<30>::run(r, w, t16, live, e, a, steps - 2);
}
// rangecoder::s (FFV1_RangeCoder.cpp:206-236) for all lanes of the wavefront
__device__ __forceinline__ int32_t rd_s_regs(rd_lane& r, uint32_t (&w)[8], const uint16_t* t16)
{
    const bool zero = rd_bit_k<0>(r, w, t16, true) != 0;
    bool going = !zero, over = false;
    uint32_t e = 0;
    const int steps = __builtin_amdgcn_readfirstlane(rd_steps<0>::unary(r, w, t16, going, e, over));
    const bool live = !zero && !over;
    int32_t a = 1;
    rd_mantissa(r, w, t16, live, e, a, steps);
    // sign: state 11 + min(e, 10) -- the one index that differs from lane to lane: its dword is picked by selects, its byte by a shift
    const uint32_t k = 11 + (e < 10 ? e : 10), d = k >> 2, sh = (k & 3) * 8;
    uint32_t c2 = w[2], c3 = w[3], c4 = w[4], c5 = w[5];
    asm("" : "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5));                 // opaque copies: left alone, the selects below become an indexed load of a stack copy of w
    const uint32_t word = d == 2 ? c2 : d == 3 ? c3 : d == 4 ? c4 : c5;
    uint32_t ns;
    const bool neg = rd_core(r, (word >> sh) & 0xFFu, t16, live, ns) != 0;
    const uint32_t put = (word & ~(0xFFu << sh)) | (ns << sh);
    w[2] = d == 2 ? put : w[2]; w[3] = d == 3 ? put : w[3]; w[4] = d == 4 ? put : w[4]; w[5] = d == 5 ? put : w[5];
    return live ? (neg ? -a : a) : 0;
}

__device__ __forceinline__ int32_t med3(int32_t a, int32_t b, int32_t c) { return max(min(a, b), min(max(a, b), c)); }

// inverse of k_unpack: JPEG2000RCT (Transform.cpp:29-37) + packers; whole lines incl. DPX padding are written
__device__ __forceinline__ void st16(uint8_t* p, uint32_t v, bool be) { *reinterpret_cast<uint16_t*>(p) = uint16_t(be ? ((v >> 8) & 0xFF) | ((v & 0xFF) << 8) : v); }   // one store; global memory takes any alignment

// One pixel: JPEG2000RCT (Transform.cpp:29-37) on the four plane values, then the packer of its flavor.  `line` is the start of
// the payload line the pixel belongs to, `y` the picture line (EXR stores it in the line header).
__device__ __forceinline__ void pack_px(const dec_const* C, int32_t v0, int32_t v1, int32_t v2, int32_t v3, uint8_t* line, uint32_t x, uint32_t y)
{
    const uint32_t W = C->W;
    uint8_t* p = line + size_t(x) * C->bytes_pp;
    const bool be = C->big_endian;
    uint32_t c0, c1 = 0, c2 = 0, c3 = 0;
    if (!C->rgb) c0 = uint32_t(v0);
    else {
        int32_t g = v0, b = v1, r = v2;
        const int32_t off = int32_t(1) << C->bps;
        b -= off; r -= off; g -= (b + r) >> 2; b += g; r += g;
        if (C->gb_swap) { const int32_t t = g; g = b; b = t; }
        c0 = uint32_t(r); c1 = uint32_t(g); c2 = uint32_t(b);
        if (C->planes == 4) c3 = uint32_t(v3);
    }
    switch (C->pixfmt) {
    case RCGPU_PIX_EXR_RGB16: {                            // Transform.cpp:1062-1127: line header (y, byte count) then B, G, R runs
        uint16_t* l16 = reinterpret_cast<uint16_t*>(line + 8);
        l16[x] = uint16_t(c2); l16[W + x] = uint16_t(c1); l16[2 * W + x] = uint16_t(c0);
        if (x == 0) { uint32_t* h32 = reinterpret_cast<uint32_t*>(line); h32[0] = y; h32[1] = 6 * W; }
        return; }
    case RCGPU_PIX_RGB8: p[0] = uint8_t(c0); p[1] = uint8_t(c1); p[2] = uint8_t(c2); break;
    case RCGPU_PIX_RGBA8: p[0] = uint8_t(c0); p[1] = uint8_t(c1); p[2] = uint8_t(c2); p[3] = uint8_t(c3); break;
    case RCGPU_PIX_RGB10_FILLEDA_BE: case RCGPU_PIX_RGB10_FILLEDA_LE: {
        uint32_t w = ((c0 & 0x3FF) << 22) | ((c1 & 0x3FF) << 12) | ((c2 & 0x3FF) << 2);
        if (be) w = __builtin_bswap32(w);
        *reinterpret_cast<uint32_t*>(p) = w; break; }
    case RCGPU_PIX_RGB12_FILLEDA_BE: case RCGPU_PIX_RGB12_FILLEDA_LE:
        st16(p, (c0 << 4) & 0xFFFF, be); st16(p + 2, (c1 << 4) & 0xFFFF, be); st16(p + 4, (c2 << 4) & 0xFFFF, be); break;
    case RCGPU_PIX_RGB16_BE: case RCGPU_PIX_RGB16_LE:
        st16(p, c0 & 0xFFFF, be); st16(p + 2, c1 & 0xFFFF, be); st16(p + 4, c2 & 0xFFFF, be); break;
    case RCGPU_PIX_RGBA16_BE: case RCGPU_PIX_RGBA16_LE:
        st16(p, c0 & 0xFFFF, be); st16(p + 2, c1 & 0xFFFF, be); st16(p + 4, c2 & 0xFFFF, be); st16(p + 6, c3 & 0xFFFF, be); break;
    case RCGPU_PIX_RGBA12_FILLEDA_BE: case RCGPU_PIX_RGBA12_FILLEDA_LE:
        st16(p, (c0 << 4) & 0xFFFF, be); st16(p + 2, (c1 << 4) & 0xFFFF, be); st16(p + 4, (c2 << 4) & 0xFFFF, be); st16(p + 6, (c3 << 4) & 0xFFFF, be); break;
    case RCGPU_PIX_Y8: p[0] = uint8_t(c0); break;
    default: st16(p, c0 & 0xFFFF, be); break;
    }
    if (x == W - 1)                                   // DPX lines are padded to 32 bit (RawFrame.cpp:109): zero the padding
        for (uint32_t i = W * C->bytes_pp; i < C->line_bytes; i++) line[i] = 0;
}

template <bool RING>
__global__ __launch_bounds__(64) void k_dec_slices(const dec_const* __restrict__ C, const uint8_t* const* __restrict__ packets,
                                                   const unsigned long long* __restrict__ slice_start, const uint32_t* __restrict__ slice_len,
                                                   uint32_t nchains, uint8_t* __restrict__ states, uint32_t nkeys,
                                                   int32_t* __restrict__ planes, uint8_t* const* __restrict__ payloads, uint32_t ring_w,
                                                   uint32_t* __restrict__ err, const uint16_t* __restrict__ hdr)
{
    __shared__ uint8_t trans[512];
    __shared__ uint16_t t16[256];
    extern __shared__ __attribute__((aligned(16))) int16_t qs_dyn[];          // [nqslots][5][256]
    __shared__ __attribute__((aligned(16))) uint8_t slot[64 * 32];
    const int lane = threadIdx.x;
    // Every wavefront of this kernel has the same work and the kernel ends with its slowest one.  A wavefront of another kernel on the same
    // SIMD -- k_md5 hashing the previous batch is one dependent VALU chain per wavefront, issuing all the time -- takes issue slots from
    // the one or two decoder wavefronts beside it and from nobody else: those few finish late and everybody waits (measured: 2.22 s alone,
    // 2.64 s beside k_md5 for 1600 frames of 64 slices; 0.46 and 0.76-0.80 s for 336 frames of 576).  With the higher priority the
    // decoder's instructions go first and the hash takes the slots the decoder leaves while it waits for memory.
    __builtin_amdgcn_s_setprio(3);
    for (int i = lane; i < 256; i += 64) { trans[i] = C->zero_state[i]; trans[256 + i] = C->one_state[i]; t16[i] = uint16_t(C->zero_state[i] | C->one_state[i] << 8); }
    for (int i = lane; i < int(C->nqslots) * 5 * 256; i += 64) qs_dyn[i] = (&C->q[0][0][0])[i];
    __syncthreads();
    const uint32_t chain = blockIdx.x * 64 + lane;
    if (chain >= nchains) return;
    const uint32_t S = C->S, f = chain / S;
    const uint32_t tail = C->v1 ? 0 : C->ec ? 8 : 3;
    const uint32_t len = slice_len[chain];
    if (len < tail) { atomicOr(err, 8u); return; }
    const uint8_t* buf = packets[f] + slice_start[chain];
    rd_lane r;
    r.cap = C->win_cap;
    r.n = len - tail; r.pos = 0; r.win_hi = 0x80000000u; r.win_lo = 0; r.have = 0; r.pend = 0; r.npend = r.nraw = 0; r.raw[0] = r.raw[1] = r.raw[2] = r.raw[3] = 0; r.next = buf; r.end = buf + r.n;
    rd_refill(r); rd_refill(r);                                      // the first call fetches, the second fills the window
    r.current = r.win_hi >> 24; r.win_hi = __builtin_amdgcn_alignbit(r.win_hi, r.win_lo, 24); r.win_lo <<= 8;   // AssignBuffer, FFV1_RangeCoder.cpp:22-33 (the byte is counted at the next refill)
    r.mask = 0xFF;
    uint8_t* my = slot + lane * 4;
    uint32_t* myw = reinterpret_cast<uint32_t*>(my);               // dword k of this lane's states: myw[k * 64]
    auto fresh = [&]() { for (int k = 0; k < 8; k++) myw[k * 64] = 0x80808080u; };
    uint32_t sx = 0, sy = 0;
    bool bad = false;
    if (C->v1) {
        // version 1: keyframe bit and the stream header (parameters::Parse(E, false), FFV1_Parameters.cpp:23-104) sit in front of the
        // one slice, in the same coder.  The header of a stream this decoder is configured for is fully determined: replay its
        // decisions (state | bit << 8, ffv1_host.cpp v1_frame_header_decisions) and insist on every bit.
        for (uint32_t i = 0; i < C->hdr_n; i++) {
            const uint32_t d = hdr[i];
            my[0] = uint8_t(d);
            bad |= uint32_t(rd_bit(r, my, 0, trans)) != (d >> 8);
        }
    } else {
    if (slice_start[chain] == 0) { fresh(); if (!rd_bit(r, my, 0, trans)) atomicOr(err, 16u); }     // keyframe bit of the first slice in the packet
    // slice header, FFV1_Slice.cpp:113-177
    fresh();
    sx = rd_u(r, my, trans); sy = rd_u(r, my, trans);
    const uint32_t sw1 = rd_u(r, my, trans), sh1 = rd_u(r, my, trans);
    bad = sx >= C->num_h || sy >= C->num_v || sw1 || sh1;
    for (uint32_t i = 0; i < C->index_count; i++) bad |= rd_u(r, my, trans) != C->idx[i];      // per-slice fields (FFV1_Slice.cpp:158-168); one tuple per stream here
    (void)rd_u(r, my, trans); (void)rd_u(r, my, trans); (void)rd_u(r, my, trans);
    }
    if (bad) { atomicOr(err, 32u); return; }
    const uint32_t W = C->W, H = C->H, np = C->planes;
    const uint32_t x0 = uint32_t((unsigned long long)sx * W / C->num_h), y0 = uint32_t((unsigned long long)sy * H / C->num_v);
    const uint32_t w = uint32_t((unsigned long long)(sx + 1) * W / C->num_h) - x0, h = uint32_t((unsigned long long)(sy + 1) * H / C->num_v) - y0;
    // RING: no picture-sized planes.  A lane keeps the last three lines of each plane of its slice and packs a picture line into the
    // payload as soon as its last plane is decoded: 125 instead of 230 MB per frame in flight.  The rings of the 64 slices of a wavefront
    // are INTERLEAVED: element (plane p, ring row r, column x) of lane l lies at ((p * 3 + r) * ring_w + x) * 64 + l of the wavefront's
    // block.  All lanes of a wavefront are at the same p, y and x (one instruction stream), so the store of cur[x], the fetches of
    // prev[x + 2] and pp[x + 1] and the packer's reads are ONE 256-byte access per wave instruction; with a ring of its own per lane each of
    // them touched 64 lines for 4 bytes apiece -- 5.5 of the kernel's 7.4 GB of HBM traffic per 4K frame (profiles/r03g_check_pmc_*).
    const size_t plane_sz = RING ? size_t(3) * ring_w : size_t(W) * H;          // elements of one plane of one lane / of one picture plane
    const uint32_t pitch = RING ? ring_w : W;
    constexpr size_t xs = RING ? 64 : 1;                                         // distance between neighbouring columns, in elements
    const size_t pstride = plane_sz * xs;                                        // distance between planes
    int32_t* fp = RING ? planes + size_t(blockIdx.x) * 64 * np * plane_sz + lane : planes + size_t(f) * np * plane_sz + size_t(y0) * W + x0;
    uint8_t* st_base = states + size_t(chain) * nkeys * 32;          // pre-set by the host: 128, or the set's coded initial states (k_dec_preset)
    const bool ov16 = C->overflow16, rgb = C->rgb;
    const int32_t bitmask = int32_t((1u << C->bits) - 1);
    const uint32_t five = C->is5[0] | C->is5[1] << 1 | C->is5[2] << 2, kb0 = C->kbase[0], kb1 = C->kbase[1], kb2 = C->kbase[2];
    const uint32_t qs0 = C->qslot[0] * 1280, qs1 = C->qslot[1] * 1280, qs2 = C->qslot[2] * 1280;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t p = 0; p < np; p++) {
            // the plane's group: its table set, its contexts (uniform over the wavefront -- every lane is at the same plane)
            const uint32_t g = rgb ? (p + 1) >> 1 : 0;
            const int16_t (*q)[256] = reinterpret_cast<const int16_t (*)[256]>(qs_dyn + (g == 0 ? qs0 : g == 1 ? qs1 : qs2));
            const bool is5 = (five >> g) & 1;
            const uint32_t kbase = g == 0 ? kb0 : g == 1 ? kb1 : kb2;
            int32_t* cur = fp + p * pstride + size_t(RING ? y % 3 : y) * pitch * xs;
            const int32_t* prev = RING ? fp + p * pstride + size_t((y + 2) % 3) * pitch * xs : cur - W;                  // valid when y >= 1
            const int32_t* pp = RING ? fp + p * pstride + size_t((y + 1) % 3) * pitch * xs : cur - 2 * size_t(W);        // valid when y >= 2
            // edge rules of SliceContent_LineThenPlane (FFV1_Slice.cpp:427-441): cur[-1] = prev[0], prev[w] = prev[w-1],
            // everything above the slice is 0, cur[-2] is 0
            int32_t L = y ? prev[0] : 0, LL = 0;
            int32_t LT = y >= 2 ? pp[0] : 0;
            int32_t T = y ? prev[0] : 0;
            // the two neighbours that come from memory are fetched one sample ahead
            int32_t RTn = y ? (1 < w ? prev[xs] : T) : 0, TTn = y >= 2 ? pp[0] : 0;
            asm volatile("" :: "v"(L), "v"(LT), "v"(T), "v"(RTn), "v"(TTn));      // (arrived before the loop is entered: see the end of its body)
            for (uint32_t x = 0; x < w; x++) {
                // The context's state record is what a sample waits for longest (a random 32-byte gather that depends on the sample decoded
                // just before): its address comes first, and everything that does not need it -- the window's top-up, the next sample's
                // neighbours, the prediction -- is done while it is on its way.
                const int32_t RT = RTn, TT = TTn;
                int32_t ctx = q[0][(L - LT) & 0xFF] + q[1][(LT - T) & 0xFF] + q[2][(T - RT) & 0xFF];
                if (is5) ctx += q[3][(LL - L) & 0xFF] + q[4][(TT - T) & 0xFF];
                const uint32_t key = kbase + uint32_t(ctx < 0 ? -ctx : ctx);
                uint4* gp = reinterpret_cast<uint4*>(st_base + size_t(key) * 32);
                const uint4 a0 = gp[0], a1 = gp[1];
                asm volatile("" ::: "memory");                        // the gather is issued before the loads below, not behind them
                RTn = y ? (x + 2 < w ? prev[size_t(x + 2) * xs] : RT) : 0;
                TTn = y >= 2 && x + 1 < w ? pp[size_t(x + 1) * xs] : 0;
                rd_refill(r);
                int32_t v = ov16 ? med3(int16_t(L), int16_t(L) + int16_t(T) - int16_t(LT), int16_t(T)) : med3(L, L + T - LT, T);
                uint32_t sw[8];
                sw[0] = a0.x; sw[1] = a0.y; sw[2] = a0.z; sw[3] = a0.w; sw[4] = a1.x; sw[5] = a1.y; sw[6] = a1.z; sw[7] = a1.w;
                const uint32_t c_cur = r.current, c_mask = r.mask, c_hi = r.win_hi, c_lo = r.win_lo;      // what the fast decoder changes
                int32_t delta = rd_s_regs(r, sw, t16);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(rd_underflow(r)) != 0, 0)) {
                    // some slice's sample took more bytes than its window held (up to 35 decisions may each take one; a window holds 7):
                    // everybody back to the start of the sample, and once more with the decoder that looks before it takes
                    if (lane == 0) atomicAdd(err + 2, 1u);                                        // (counted: rcgpu_ffv1_decoder_debug_careful)
                    r.current = c_cur; r.mask = c_mask; r.win_hi = c_hi; r.win_lo = c_lo;
                    { const uint4 a0 = gp[0], a1 = gp[1];
                      myw[0] = a0.x; myw[64] = a0.y; myw[128] = a0.z; myw[192] = a0.w; myw[256] = a1.x; myw[320] = a1.y; myw[384] = a1.z; myw[448] = a1.w; }
                    delta = rd_s_careful(r, my, trans);
                    for (int k = 0; k < 8; k++) sw[k] = myw[k * 64];
                }
                // Everything this sample LOADED for the next one (its neighbours, the window's next bytes) is asked for here, before the stores
                // below are issued: the loop's next turn then finds nothing it has to wait for, where a wait at its top would also wait for
                // the stores -- a write's acknowledgement under this kernel's load is thousands of cycles (the counter is one for both)
                asm volatile("" :: "v"(RTn), "v"(TTn), "v"(r.raw[0]), "v"(r.raw[1]), "v"(r.raw[2]), "v"(r.raw[3]));
                gp[0] = make_uint4(sw[0], sw[1], sw[2], sw[3]); gp[1] = make_uint4(sw[4], sw[5], sw[6], sw[7]);
                v = (ctx >= 0 ? v + delta : v - delta) & bitmask;
                cur[size_t(x) * xs] = v;
                LL = L; L = v; LT = T; T = RT;
            }
            if (RING && p + 1 == np) {                                // the picture line is complete: inverse RCT + pack, Transform.cpp From()
                const uint32_t gy = y0 + y;
                uint8_t* line = payloads[f] + size_t(C->vflip ? H - 1 - gy : gy) * C->line_bytes;
                const int32_t* r0 = fp + size_t(y % 3) * pitch * xs;
                for (uint32_t xb = 0; xb < w; xb += 8) {               // eight pixels' loads in flight together, then their stores
                    int32_t a0[8], a1[8], a2[8], a3[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const size_t x = size_t(min(xb + uint32_t(j), w - 1)) * xs;
                        a0[j] = r0[x]; a1[j] = np > 1 ? r0[pstride + x] : 0; a2[j] = np > 1 ? r0[2 * pstride + x] : 0; a3[j] = np > 3 ? r0[3 * pstride + x] : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (xb + uint32_t(j) < w) pack_px(C, a0[j], a1[j], a2[j], a3[j], line, x0 + xb + uint32_t(j), gy);
                }
            }
        }
    // end-of-slice bit, underrun and junk checks (FFV1_Slice.cpp:286-299,336-340)
    my[0] = 129; rd_bit(r, my, 0, trans);
    const uint32_t pos = rd_pos(r);
    const bool underrun = pos - (r.mask < 0x100 ? 0 : 1) > r.n;
    const size_t used = pos > r.n ? size_t(r.n) : size_t(pos) - (r.mask < 0x100 ? 0 : 1);
    if (underrun) atomicOr(err, 64u);
    if (used < len - tail) atomicOr(err, 128u);
    if (C->ec && buf[len - 5]) atomicOr(err, 256u);                  // error_status
}


// states_coded = 1: every chain's context states start as the stream's initial states (coder_rangecoder::GOP_Init copies them at every key frame,
// Coder/FFV1_Coder_RangeCoder.cpp:34-57) -- `init` holds one chain's worth (nkeys x 32 bytes = n16 uint4), written to every chain
__global__ __launch_bounds__(256) void k_dec_preset(uint4* __restrict__ states, const uint4* __restrict__ init, uint32_t n16, unsigned long long total)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256)
        states[i] = init[i % n16];
}

// k_pack for the word-stream layouts (rc_common.h kFields*): one thread per 32-bit word of the payload assembles every field that
// touches it, so no two threads write the same word.  Restates the From() loops of Transform.cpp:214-322 (RGB 12-bit packed),
// :445-550 (RGBA 10/12-bit), :781-796 (Y 10-bit), :905-990 (Y 12-bit packed); padding bits are written as zero.
__device__ __forceinline__ uint32_t field_value(const dec_const* C, const int32_t* fp, size_t plane_sz, uint32_t pix, uint32_t comp)
{
    if (!C->rgb) return uint32_t(fp[pix]);
    if (comp == 3) return uint32_t(fp[3 * plane_sz + pix]);
    int32_t g = fp[pix], b = fp[plane_sz + pix], r = fp[2 * plane_sz + pix];
    const int32_t off = int32_t(1) << C->bps;
    b -= off; r -= off; g -= (b + r) >> 2; b += g; r += g;
    if (C->gb_swap) { const int32_t t = g; g = b; b = t; }
    return uint32_t(comp == 0 ? r : comp == 1 ? g : b);
}

__global__ __launch_bounds__(256) void k_pack_words(const dec_const* __restrict__ C, const int32_t* __restrict__ planes, uint8_t* const* __restrict__ payloads)
{
    const uint32_t W = C->W, H = C->H, np = C->planes, fields = C->fields;
    const uint32_t words_per_line = C->altern ? (W * H + 2) / 3 : C->line_bytes / 4, nlines = C->altern ? 1 : H;
    const uint32_t widx = blockIdx.x * 256 + threadIdx.x;
    if (widx >= words_per_line * nlines) return;
    const uint32_t f = blockIdx.y, fy = widx / words_per_line, k = widx - fy * words_per_line;
    const uint32_t y = C->vflip ? H - 1 - fy : fy;
    const size_t plane_sz = size_t(W) * H;
    const int32_t* fp = planes + size_t(f) * np * plane_sz;
    const uint32_t nfields = C->altern ? W * H : W * np;             // fields in this line (altern: in the frame)
    const uint32_t pix0 = C->altern ? 0 : y * W;
    uint32_t word = 0;
    if (fields == kFieldsPacked) {
        const uint32_t lo = k * 32, first = lo / 12, last = min((lo + 31) / 12, nfields - 1);
        for (uint32_t i = first; i <= last && i < nfields; i++) {
            const uint32_t v = field_value(C, fp, plane_sz, pix0 + i / np, i % np) & 0xFFF, bit = i * 12;
            word |= bit >= lo ? v << (bit - lo) : v >> (lo - bit);
        }
        word = __builtin_bswap32(word);
    } else {
        for (uint32_t slot = 0; slot < 3; slot++) {
            const uint32_t i = k * 3 + slot;
            if (i >= nfields) break;
            const uint32_t v = field_value(C, fp, plane_sz, pix0 + i / np, i % np) & 0x3FF;
            word |= v << (fields == kFieldsTop ? 22 - 10 * slot : 10 * slot + C->fill);
        }
        if (C->big_endian) word = __builtin_bswap32(word);
    }
    reinterpret_cast<uint32_t*>(payloads[f] + (C->altern ? size_t(0) : size_t(fy) * C->line_bytes))[k] = word;
}

__global__ __launch_bounds__(256) void k_compare(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, unsigned long long n,
                                                 unsigned long long* __restrict__ first_diff)
{
    unsigned long long best = ~0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256)
        if (a[i] != b[i]) { best = i; break; }
    if (best != ~0ull) atomicMin(first_diff, best);
}

// The same for n pairs of buffers in one launch (grid.y = pair), 16 bytes per thread and turn where both pointers allow it: what
// frame_writer's compare (FileWriter.cpp:448-463) does for every decoded frame of a batch.  first_diff[pair] starts at ~0.
__global__ __launch_bounds__(256) void k_compare_batch(const uint8_t* const* __restrict__ as, const uint8_t* const* __restrict__ bs,
                                                       const unsigned long long* __restrict__ sizes, unsigned long long* __restrict__ first_diff)
{
    const uint8_t* a = as[blockIdx.y]; const uint8_t* b = bs[blockIdx.y];
    const unsigned long long n = sizes[blockIdx.y];
    unsigned long long best = ~0ull;
    const bool wide = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    const unsigned long long nw = wide ? n / 16 : 0;
    const uint4* a4 = reinterpret_cast<const uint4*>(a); const uint4* b4 = reinterpret_cast<const uint4*>(b);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nw; i += (unsigned long long)gridDim.x * 256) {
        const uint4 x = a4[i], y = b4[i];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) {
            for (unsigned long long k = i * 16; k < i * 16 + 16; k++) if (a[k] != b[k]) { best = k; break; }
            break;
        }
    }
    for (unsigned long long i = nw * 16 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n && best == ~0ull; i += (unsigned long long)gridDim.x * 256)
        if (a[i] != b[i]) best = i;
    if (best != ~0ull) atomicMin(&first_diff[blockIdx.y], best);
}

// RFC 1321, one buffer per lane (the hash is serial per buffer; many buffers are in flight).  A lane's chain is 64 steps of four
// dependent instructions per 64-byte block and nothing else shares its wavefront's issue slot, so what the kernel can hide is the
// latency of the block's own load: the next block is fetched while this one is hashed.
__device__ __forceinline__ uint32_t rol(uint32_t x, int s) { return __builtin_amdgcn_alignbit(x, x, 32 - s); }
__device__ __forceinline__ void md5_block(const uint32_t (&m)[16], uint32_t& h0, uint32_t& h1, uint32_t& h2, uint32_t& h3)
{
    const uint32_t K[64] = {
        0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
        0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
        0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
        0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
    const int S[16] = { 7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21 };
    uint32_t a = h0, b = h1, c = h2, d = h3;
#pragma unroll
    for (int r = 0; r < 64; r++) {
        uint32_t fn; int g;
        if (r < 16) { fn = (b & c) | (~b & d); g = r; }
        else if (r < 32) { fn = (d & b) | (~d & c); g = (5 * r + 1) & 15; }
        else if (r < 48) { fn = b ^ c ^ d; g = (3 * r + 5) & 15; }
        else { fn = c ^ (b | ~d); g = (7 * r) & 15; }
        const uint32_t t = d; d = c; c = b; b = b + rol(a + fn + (K[r] + m[g]), S[(r >> 4) * 4 + (r & 3)]); a = t;
    }
    h0 += a; h1 += b; h2 += c; h3 += d;
}
// MODE 0: 16-byte aligned, 1: 4-byte aligned, 2: any address -- aligned words, funnel-shifted by `sh` bits (reads the word after the block)
template <int MODE> __device__ __forceinline__ void md5_load(uint32_t (&m)[16], const uint8_t* p, uint32_t sh)
{
    if (MODE == 0) {
        const uint4* w = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint4 v = w[k]; m[4 * k] = v.x; m[4 * k + 1] = v.y; m[4 * k + 2] = v.z; m[4 * k + 3] = v.w; }
    } else if (MODE == 1) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (int k = 0; k < 16; k++) m[k] = w[k];
    } else {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
        uint32_t lo = w[0];
#pragma unroll
        for (int k = 0; k < 16; k++) { const uint32_t hi = w[k + 1]; m[k] = __builtin_amdgcn_alignbit(hi, lo, sh); lo = hi; }
    }
}
template <int MODE> __device__ __forceinline__ void md5_whole_blocks(const uint8_t* p, uint32_t sh, unsigned long long nblk, uint32_t& h0, uint32_t& h1, uint32_t& h2, uint32_t& h3)
{
    uint32_t m[16], x[16];
    if (nblk) md5_load<MODE>(m, p, sh);
    unsigned long long k = 0;
    for (; k + 2 <= nblk; k += 2) {                             // m holds block k
        md5_load<MODE>(x, p + (k + 1) * 64, sh);
        md5_block(m, h0, h1, h2, h3);
        if (k + 2 < nblk) md5_load<MODE>(m, p + (k + 2) * 64, sh);
        md5_block(x, h0, h1, h2, h3);
    }
    if (k < nblk) md5_block(m, h0, h1, h2, h3);
}
__global__ __launch_bounds__(64) void k_md5(const uint8_t* const* __restrict__ bufs, const unsigned long long* __restrict__ sizes, uint32_t n,
                                            uint8_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = bufs[i];
    const unsigned long long size = sizes[i];
    uint32_t h0 = 0x67452301, h1 = 0xefcdab89, h2 = 0x98badcfe, h3 = 0x10325476;
    const unsigned long long total = ((size + 8) / 64 + 1) * 64;
    const uintptr_t mis = reinterpret_cast<uintptr_t>(p) & 15;
    // blocks that are read as words; the rest, and the padding, byte by byte
    const unsigned long long whole = (mis & 3) ? (size >= 68 ? (size - 4) / 64 : 0) : size / 64;
    if (mis == 0) md5_whole_blocks<0>(p, 0, whole, h0, h1, h2, h3);
    else if (!(mis & 3)) md5_whole_blocks<1>(p, 0, whole, h0, h1, h2, h3);
    else md5_whole_blocks<2>(p - (mis & 3), uint32_t(mis & 3) * 8, whole, h0, h1, h2, h3);
    for (unsigned long long off = whole * 64; off < total; off += 64) {
        uint32_t m[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t v = 0;
            for (int b = 0; b < 4; b++) {
                const unsigned long long q = off + 4 * k + b;
                uint32_t byte = q < size ? p[q] : (q == size ? 0x80u : 0u);
                if (off + 64 == total && 4 * k + b >= 56) byte = uint32_t(((size * 8) >> (8 * (4 * k + b - 56))) & 0xFF);
                v |= byte << (8 * b);
            }
            m[k] = v;
        }
        md5_block(m, h0, h1, h2, h3);
    }
    const uint32_t hh[4] = { h0, h1, h2, h3 };
    for (int k = 0; k < 16; k++) out[size_t(i) * 16 + k] = uint8_t(hh[k / 4] >> (8 * (k % 4)));
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// CUs of its own for the hash.  k_md5 is one dependent VALU chain per wavefront, ready to issue all the time for the second a batch of 4K
// files takes; k_dec_slices ends with its slowest wavefront, and the ones that share a SIMD with a hash wavefront are the slowest (2.64 s
// instead of 2.19 for 1600 frames; 2.41-2.45 with the decoder at the higher wave priority: DESIGN.md section 5).  So the two get
// disjoint CUs through CU-masked streams: one CU of every XCD for the hash (32 SIMDs: 2048 files a wavefront apiece), the other 248 for
// the decoder.  What the mask means was measured (tools/cu_mask_probe.hip, profiles/r04_cu_mask_probe.txt): bit i stands for a CU of XCD
// i % 8, and a mask that leaves an XCD without a CU is ignored as a whole.  Devices that are not 8 x 32 CUs get plain streams.
constexpr uint32_t kHashCuBits = 0xFFu;                    // bits 0..7: one CU of each XCD
static bool partition_streams(int device, hipStream_t* decode, hipStream_t* hash)
{
    // RCGPU_NO_CU_PARTITION: plain streams (rocprofv3 --kernel-trace of ROCm 7.2 dies of a segmentation fault in a process that has
    // made a CU-masked stream; tools/profile_check.sh sets it).  Same bytes either way.
    if (const char* e = getenv("RCGPU_NO_CU_PARTITION")) if (*e && *e != '0') return false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount != 256) return false;
    uint32_t mh[8] = { kHashCuBits, 0, 0, 0, 0, 0, 0, 0 };
    uint32_t md[8] = { ~kHashCuBits, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u };
    hipStream_t a = nullptr, b = nullptr;
    if (decode && hipExtStreamCreateWithCUMask(&a, 8, md) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hash && hipExtStreamCreateWithCUMask(&b, 8, mh) != hipSuccess) { (void)hipGetLastError(); if (a) (void)hipStreamDestroy(a); return false; }
    if (decode) *decode = a;
    if (hash) *hash = b;
    return true;
}
// the hash stream of the free-standing rcgpu_md5_device, one per device, made on first use and kept
// CU-masked streams are POOLED, never destroyed while the library is in use: a decoder borrows a (decode, hash) pair and hands it back.  The HIP
// runtime torch 2.10 bundles (ROCm 7.0.5) dies of a segmentation fault in an out-of-memory hipMalloc once such a stream has been DESTROYED in
// the process (alive: fine; ROCm 7.2's own runtime: fine either way -- tools/oom_after_decoder.py, tools/oom_after_decoder.cpp), and callers rely on
// that error code (route C halves its batch on it, rcgpu_ffv1_set_run_on falls back to one batch at a time).
struct masked_pair { hipStream_t decode, hash; };
static std::mutex g_pool_mu; static std::vector<masked_pair> g_pool[16];
// set by rcgpu_release_device_streams: from then on nothing CU-masked is made or handed out (plain streams: same bytes, the hash beside the decoder
// on shared CUs) -- a masked stream made after the release would be destroyed by nobody, and one handed out across it by somebody else
static std::atomic<bool> g_streams_released{ false };
static bool acquire_masked_pair(int device, hipStream_t* decode, hipStream_t* hash)
{
    if (g_streams_released.load()) return false;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        std::vector<masked_pair>& P = g_pool[device & 15];
        if (!P.empty()) { *decode = P.back().decode; *hash = P.back().hash; P.pop_back(); return true; }
    }
    return partition_streams(device, decode, hash);
}
static void release_masked_pair(int device, hipStream_t decode, hipStream_t hash)
{
    if (!decode || !hash) return;
    (void)hipStreamSynchronize(decode); (void)hipStreamSynchronize(hash);
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (g_streams_released.load()) { (void)hipStreamDestroy(decode); (void)hipStreamDestroy(hash); return; }      // a decoder that outlived the release
    g_pool[device & 15].push_back({ decode, hash });
}
static std::mutex g_hash_mu; static hipStream_t g_hash_st[16]; static bool g_hash_tried[16];
static hipStream_t device_hash_stream(int device)
{
    std::lock_guard<std::mutex> g(g_hash_mu);
    const int k = device & 15;
    if (g_streams_released.load()) return nullptr;
    if (!g_hash_tried[k]) { g_hash_tried[k] = true; g_hash_st[k] = nullptr; if (!partition_streams(device, nullptr, &g_hash_st[k])) g_hash_st[k] = nullptr; }
    return g_hash_st[k];
}
// The per-device hash streams live as long as the library; a caller that wants them gone before the process ends (a profiler's finalisation
// after this library's streams, say) gives them back itself.
extern "C" void rcgpu_release_device_streams(void)
{
    g_streams_released.store(true);
    {
        std::lock_guard<std::mutex> g(g_hash_mu);
        for (int k = 0; k < 16; k++)
            if (g_hash_st[k]) { (void)hipStreamSynchronize(g_hash_st[k]); (void)hipStreamDestroy(g_hash_st[k]); g_hash_st[k] = nullptr; g_hash_tried[k] = false; }
    }
    std::lock_guard<std::mutex> g(g_pool_mu);                 // the pairs no decoder holds at the moment
    for (auto& P : g_pool) { for (masked_pair& m : P) { (void)hipStreamDestroy(m.decode); (void)hipStreamDestroy(m.hash); } P.clear(); }
}

// ------------------------------------------------------------------------------------------------------------
// Host memory that is not pinned -- the mapped MKV, mapped source files -- goes up through pinned staging buffers, several threads side
// by side (upload_side_by_side below): the streams, buffers and events of those threads, made on first use
struct stager {
    static constexpr unsigned kLanes = 8; static constexpr size_t kStage = size_t(8) << 20;
    struct lane { hipStream_t st = nullptr; uint8_t* stage[2] = { nullptr, nullptr }; hipEvent_t ev[2] = { nullptr, nullptr }; } lanes[kLanes];
    void release() {
        for (auto& l : lanes) {
            for (auto& p : l.stage) if (p) { (void)hipHostFree(p); p = nullptr; }
            for (auto& e : l.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            if (l.st) { (void)hipStreamDestroy(l.st); l.st = nullptr; }
        }
    }
};

struct rcgpu_ffv1_decoder {
    rcgpu_ffv1_config cfg{};
    dec_const hc{};
    uint32_t nkeys = 0;
    bool ring = false; uint32_t ring_w = 0;       // line-ring mode: no picture-sized planes, lines are packed by the decoding lane
    size_t payload_bytes = 0;
    dec_const* d_const = nullptr;
    const uint8_t** d_pkt_ptrs = nullptr; uint8_t** d_out_ptrs = nullptr; unsigned long long* d_sizes = nullptr;
    unsigned long long* d_slice_start = nullptr; uint32_t* d_slice_len = nullptr; uint16_t* d_hdr = nullptr;
    uint8_t* d_states = nullptr; int32_t* d_planes = nullptr; uint32_t* d_err = nullptr;
    uint8_t* d_init = nullptr;                     // one chain's initial states when the stream codes them (states_coded), else null: all 128
    size_t states_off = 0, states_slack = 0;       // test hook (rcgpu_ffv1_decoder_debug_states_offset): the state arrays begin this far into their allocation
    void** h_ptrs = nullptr; unsigned long long* h_sizes = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t dec_stream = nullptr;              // k_dec_slices' stream when the hash has CUs of its own (partition_streams), else null
    hipEvent_t ev[8]{};
    bool ev_valid = false;
    // the payloads of the last decode_keep: slot i = d_kept + i * kept_stride, [room | payload | room] so that the bytes a file has
    // before and after its payload can be put next to it and the file hashed as one buffer
    // Two such sets: while the files of one batch are being hashed (verify_kept_begin ... _end) the next batch is decoded into the other.
    struct kept_set {
        uint8_t* d = nullptr; size_t cap = 0; uint32_t n = 0;
        uint64_t* h_tab = nullptr; uint64_t* d_tab = nullptr; uint32_t tab_cap = 0;     // tables of the verification kernels, pinned + device
        bool pending = false;                                                            // begun, not ended
        uint32_t pn = 0, pn_md5 = 0, pn_cmp = 0;
        std::vector<uint32_t> img_of, cmp_of;
        struct host_part { uint64_t mine, on_disk_size, before_size, head_diff, tail_diff; };   // what the host already knows of a comparison
        std::vector<host_part> part;
    } kept[3];
    int kept_cur = 0;
    // A batch decoded AHEAD of its decode_keep call (rcgpu_ffv1_decoder_decode_keep_hint): by a thread of its own, into the set of slots that
    // is neither the current one nor under verification, from a packet buffer and staging lanes of its own.  While it runs the caller hands
    // out the current batch's frames and has them verified; nothing else may decode with this decoder until it is joined.
    struct hint_t {
        std::thread th; bool active = false; int set = -1, rc = 0;
        std::string error;                          // rcgpu_last_error() of the hint thread (the text is thread-local: the adopting call reports it)
        std::vector<const uint8_t*> packets; std::vector<uint64_t> sizes;
    } hint;
    uint8_t* d_kept_in2 = nullptr; size_t kept_in2_cap = 0;
    // the library's own mapping of the file a batch is hinted from (decode_keep_hint_file): the reference maps its Matroska file anew every
    // megabyte it advances (Matroska.cpp:394-408, FileIO.cpp:258-283), so its pointers do not outlive the call they are passed in
    std::string map_path; const uint8_t* map_base = nullptr; size_t map_size = 0;
    dev_t map_dev = 0; ino_t map_ino = 0; struct timespec map_mtime = {};     // what was mapped: a file rewritten or replaced under the same name is mapped anew
    size_t kept_stride = 0;
    uint8_t* d_kept_in = nullptr; size_t kept_in_cap = 0;       // the packets of a batch
    uint8_t* d_disk = nullptr; size_t disk_cap = 0;
    hipStream_t side_stream = nullptr, md5_stream = nullptr; hipEvent_t ev_tab = nullptr, ev_side = nullptr;
    uint8_t* h_edges = nullptr; size_t edges_cap = 0;          // pinned: the bytes before and after the payloads of a batch on their way up
    stager up, up2;                                            // up2: the hinted batch's
};

extern "C" void rcgpu_ffv1_decoder_destroy(rcgpu_ffv1_decoder* d)
{
    if (!d) return;
    (void)hipSetDevice(d->cfg.device);
    if (d->hint.active) { d->hint.th.join(); d->hint.active = false; }
    void* bufs[] = { d->d_const, d->d_pkt_ptrs, d->d_out_ptrs, d->d_sizes, d->d_slice_start, d->d_slice_len, d->d_states, d->d_planes, d->d_err, d->d_hdr, d->d_init };
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (d->h_ptrs) (void)hipHostFree(d->h_ptrs);
    if (d->h_sizes) (void)hipHostFree(d->h_sizes);
    for (auto& e : d->ev) if (e) (void)hipEventDestroy(e);
    if (d->own_stream) (void)hipStreamDestroy(d->own_stream);
    if (d->dec_stream) (void)hipStreamSynchronize(d->dec_stream);
    if (d->md5_stream) (void)hipStreamSynchronize(d->md5_stream);     // a verification begun and never ended
    for (void* b : { (void*)d->d_kept_in, (void*)d->d_kept_in2, (void*)d->kept[0].d, (void*)d->kept[1].d, (void*)d->kept[2].d, (void*)d->d_disk,
                     (void*)d->kept[0].d_tab, (void*)d->kept[1].d_tab, (void*)d->kept[2].d_tab }) if (b) (void)hipFree(b);
    for (auto& k : d->kept) if (k.h_tab) (void)hipHostFree(k.h_tab);
    if (d->h_edges) (void)hipHostFree(d->h_edges);
    if (d->side_stream) (void)hipStreamDestroy(d->side_stream);
    release_masked_pair(d->cfg.device, d->dec_stream, d->md5_stream);       // back into the pool, not destroyed (see acquire_masked_pair)
    if (d->ev_tab) (void)hipEventDestroy(d->ev_tab);
    if (d->ev_side) (void)hipEventDestroy(d->ev_side);
    d->up.release(); d->up2.release();
    if (d->map_base) (void)munmap(const_cast<uint8_t*>(d->map_base), d->map_size);
    delete d;
}

// room behind the state arrays for rcgpu_ffv1_decoder_debug_states_offset -- only in a process that asks for the hook (RCGPU_DEC_STATES_SLACK=1,
// bench.py --check-offsets): a production decoder allocates what it uses
constexpr size_t kStatesSlack = size_t(64) << 20;

// The decoder of the stream `s` for pictures laid out as `files` says (width, height, pixfmt, line_bytes, flags; max_batch, device).
static int decoder_create(const rcgpu_ffv1_config* files, const ffv1::stream_desc& s, rcgpu_ffv1_decoder** out)
{
    rcgpu_ffv1_config cfg_ = *files;
    rcgpu_ffv1_config* cfg = &cfg_;
    cfg->num_h_slices = s.num_h_slices; cfg->num_v_slices = s.num_v_slices; cfg->slicecrc = s.ec; cfg->level = s.version <= 1 ? 1 : 3; cfg->coder = s.custom_transitions ? 2 : 1;
    if (cfg->pixfmt >= RCGPU_PIX_COUNT || !cfg->width || !cfg->height || !cfg->num_h_slices || !cfg->num_v_slices || !cfg->max_batch)
        return fail(2, "ffv1 decoder: bad configuration");
    if ((unsigned long long)cfg->width * cfg->height * 4 >= (1ull << 32)) return fail(2, "ffv1 decoder: %ux%u: sample indices are 32 bit (a picture may hold 2^30 pixels)", cfg->width, cfg->height);
    if (cfg->level == 1 && (cfg->num_h_slices * cfg->num_v_slices != 1 || cfg->slicecrc))
        return fail(2, "ffv1 decoder: FFV1 version 1 (-level 1) has one slice and no slice CRC");
    const pix_desc& px = pix(cfg->pixfmt);
    const bool altern = (cfg->flags & RCGPU_FLAG_ALTERN) != 0;
    if (altern && px.fields != kFieldsLow) return fail(2, "ffv1 decoder: RCGPU_FLAG_ALTERN is a layout of the Y 10-bit flavors only");
    if ((cfg->flags & RCGPU_FLAG_VFLIP) && altern) return fail(2, "ffv1 decoder: RCGPU_FLAG_VFLIP and RCGPU_FLAG_ALTERN exclude each other");
    if (!payload_line_bytes(cfg->pixfmt, cfg->width, true)) return fail(2, "ffv1 decoder: a line of %u pixels does not fit 32 bits", cfg->width);
    if (!altern && cfg->line_bytes < payload_line_bytes(cfg->pixfmt, cfg->width, false)) return fail(2, "ffv1 decoder: line_bytes smaller than a line");
    if (px.fields != kFieldsBytes && px.fields != kFieldsExr && !altern && cfg->line_bytes % 4) return fail(2, "ffv1 decoder: line_bytes of a word-stream layout must be a multiple of 4");
    // the stream against the files, and against what the device decodes
    const bool rgb = px.planes != 1;
    // Valid streams no pixel format here describes are the CALLER's decoder's (kUnsupported), a stream that describes other files than these is an
    // error (5).  With colorspace_type 1 the reference counts 3 or 4 planes whatever chroma_planes says (FFV1_Parameters.cpp:174); gray with an alpha
    // plane (:169: two planes) it decodes as well.
    if (s.colorspace_type == 0 && s.chroma_planes) return fail(ffv1::kUnsupported, "ffv1 decoder: YCbCr planes are not decoded on the device");
    if (s.colorspace_type == 0 && s.alpha_plane) return fail(ffv1::kUnsupported, "ffv1 decoder: gray with an alpha plane is not decoded on the device");
    if (s.colorspace_type != (rgb ? 1u : 0u) || s.bits_per_raw_sample != px.bits || (rgb && s.alpha_plane != (px.planes == 4)))
        return fail(5, "ffv1 decoder: stream (colorspace %u, %u bit%s) does not match the pixel format of the files", s.colorspace_type, s.bits_per_raw_sample, s.alpha_plane ? ", alpha" : "");
    if (s.version == 3 && s.intra != 1) return fail(ffv1::kUnsupported, "ffv1 decoder: inter frames (intra = 0) are not decoded on the device");
    if (s.version == 3 && (s.num_h_slices >= cfg->width || s.num_v_slices >= cfg->height || s.num_h_slices > 0xFFFF || s.num_v_slices > 0xFFFF))      // FFV1_Frame.cpp:161-164
        return fail(5, "ffv1 decoder: %u x %u slices do not fit the picture (FFV1-HEADER-num_h_slices)", s.num_h_slices, s.num_v_slices);
    const uint32_t ngroups = rgb ? (px.planes == 4 ? 3u : 2u) : 1u;
    for (uint32_t g = 0; g < ngroups; g++) if (s.set_index[g] >= s.set_count) return fail(2, "ffv1 decoder: quant_table_set_index %u of %u sets", s.set_index[g], s.set_count);
    if (ffv1::reaches_state_zero(s)) return fail(ffv1::kUnsupported, "ffv1 decoder: state 0 is within reach of the stream's initial states and transitions: not decoded on the device");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(3, "ffv1 decoder: no HIP device available -- there is no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(3, "ffv1 decoder: device %d out of range", cfg->device);
    HIP_TRY(hipSetDevice(cfg->device));
    rcgpu_ffv1_decoder* d = new rcgpu_ffv1_decoder;
    d->cfg = *cfg;
    dec_const& c = d->hc;
    c.W = cfg->width; c.H = cfg->height; c.line_bytes = cfg->line_bytes; c.pixfmt = cfg->pixfmt;
    c.planes = px.planes; c.bps = px.bits; c.rgb = rgb; c.gb_swap = px.gb_swap; c.big_endian = px.big_endian; c.bytes_pp = px.bytes_pp;
    c.fields = px.fields; c.fill = px.fill; c.vflip = (cfg->flags & RCGPU_FLAG_VFLIP) != 0; c.altern = altern;
    c.bits = c.rgb ? px.bits + 1 : (px.bits <= 8 ? 8 : px.bits);
    c.overflow16 = (!c.rgb && px.bits == 16);
    c.num_h = cfg->num_h_slices; c.num_v = cfg->num_v_slices; c.S = c.num_h * c.num_v;
    c.ngroups = ngroups; c.ec = cfg->slicecrc ? 1 : 0; c.index_count = s.index_count;
    bool coded = false;
    uint32_t nkeys = 0;
    for (uint32_t g = 0; g < 3; g++) {
        const ffv1::quant_model& Q = s.sets[g < ngroups ? s.set_index[g] : s.set_index[0]];
        c.idx[g] = s.set_index[g]; c.nctx[g] = Q.context_count; c.kbase[g] = nkeys;
        c.is5[g] = Q.q[3][127] != 0;                                  // FFV1_Slice.cpp:453
        {   // the group's table set: a slot of its own unless a group before it uses the same set
            auto eff = [&](uint32_t x) { return x < ngroups ? s.set_index[x] : s.set_index[0]; };      // (a group the picture does not have reads group 0's)
            uint32_t k = 0;
            while (k < g && eff(k) != eff(g)) k++;
            if (k < g) c.qslot[g] = c.qslot[k];
            else { c.qslot[g] = c.nqslots; memcpy(c.q[c.nqslots], Q.q, sizeof c.q[0]); c.nqslots++; }
        }
        if (g < ngroups) { nkeys += Q.context_count; coded |= !s.initial[s.set_index[g]].empty(); }
    }
    memcpy(c.one_state, s.one_state, 256);
    ffv1::make_zero_state(c.zero_state, c.one_state);
    const std::vector<uint16_t>& hdr = s.inband;
    c.v1 = cfg->level == 1; c.hdr_n = uint32_t(hdr.size()); c.win_cap = 7;
    d->nkeys = nkeys;
    d->payload_bytes = size_t(cfg->line_bytes) * cfg->height;
    const uint32_t F = cfg->max_batch; const size_t nchains = size_t(F) * c.S;
    if (const char* e = getenv("RCGPU_DEC_STATES_SLACK")) if (*e && *e != '0') d->states_slack = kStatesSlack;
    hipError_t he = hipSuccess;
#define DM(p, b) if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&(p)), (b))
    DM(d->d_const, sizeof c); DM(d->d_pkt_ptrs, sizeof(void*) * F); DM(d->d_out_ptrs, sizeof(void*) * F); DM(d->d_sizes, 8 * F);
    DM(d->d_slice_start, nchains * 8); DM(d->d_slice_len, nchains * 4); DM(d->d_states, nchains * d->nkeys * 32 + d->states_slack); DM(d->d_hdr, hdr.size() * 2 + 16);
    std::vector<uint8_t> init;
    if (coded) {                                                      // one chain's states as GOP_Init leaves them
        init.assign(size_t(nkeys) * 32, 128);
        for (uint32_t g = 0; g < ngroups; g++) {
            const std::vector<uint8_t>& v = s.initial[s.set_index[g]];
            if (!v.empty()) memcpy(init.data() + size_t(c.kbase[g]) * 32, v.data(), v.size());
        }
        DM(d->d_init, init.size());
    }
    // whole-byte layouts (and EXR) are packed inline by the decoding lanes, which then only need three lines per plane and slice;
    // the word-stream layouts share words between neighbouring slices and keep the planes + k_pack_words route
    d->ring = c.fields == kFieldsBytes || c.fields == kFieldsExr;
    d->ring_w = (c.W + c.num_h - 1) / c.num_h + 1;
    // a kept payload's slot: [room | payload | room] (set here once: decode_keep runs on two threads when a batch is decoded ahead)
    d->kept_stride = (size_t(RCGPU_KEPT_ROOM) * 2 + size_t(payload_bytes(cfg->pixfmt, cfg->width, cfg->height, cfg->line_bytes, cfg->flags)) + 255) & ~size_t(255);
    DM(d->d_planes, d->ring ? (nchains + 63) / 64 * 64 * c.planes * 3 * d->ring_w * 4 : size_t(F) * c.planes * c.W * c.H * 4); DM(d->d_err, 16);      // rings: interleaved per wavefront of 64 chains
#undef DM
    if (he == hipSuccess) he = hipHostMalloc(reinterpret_cast<void**>(&d->h_ptrs), sizeof(void*) * F * 2);
    if (he == hipSuccess) he = hipHostMalloc(reinterpret_cast<void**>(&d->h_sizes), 8 * F);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking);
    if (he == hipSuccess && !acquire_masked_pair(cfg->device, &d->dec_stream, &d->md5_stream)) d->dec_stream = d->md5_stream = nullptr;
    for (auto& e : d->ev) if (he == hipSuccess) he = hipEventCreate(&e);
    if (he == hipSuccess) he = hipMemcpy(d->d_const, &c, sizeof c, hipMemcpyHostToDevice);
    if (he == hipSuccess && !hdr.empty()) he = hipMemcpy(d->d_hdr, hdr.data(), hdr.size() * 2, hipMemcpyHostToDevice);
    if (he == hipSuccess && coded) he = hipMemcpy(d->d_init, init.data(), init.size(), hipMemcpyHostToDevice);
    if (he != hipSuccess) { (void)hipGetLastError(); const int r = fail(100, "ffv1 decoder: device setup failed: %s", hipGetErrorString(he)); rcgpu_ffv1_decoder_destroy(d); return r; }
    *out = d;
    return 0;
}

// the stream this library's encoder writes for `cfg`
extern "C" int rcgpu_ffv1_decoder_create(const rcgpu_ffv1_config* cfg, rcgpu_ffv1_decoder** out)
{
    clear_error();
    if (!cfg || !out) return fail(1, "ffv1 decoder: null argument");
    *out = nullptr;
    if (cfg->pixfmt >= RCGPU_PIX_COUNT || !cfg->num_h_slices || !cfg->num_v_slices) return fail(2, "ffv1 decoder: bad configuration");
    if (cfg->level == 1 && (cfg->num_h_slices * cfg->num_v_slices != 1 || cfg->slicecrc))
        return fail(2, "ffv1 decoder: FFV1 version 1 (-level 1) has one slice and no slice CRC");
    const pix_desc& px = pix(cfg->pixfmt);
    ffv1::stream_params sp{};
    sp.bits_per_raw_sample = px.bits; sp.rgb = px.planes != 1; sp.alpha = px.planes == 4; sp.num_h_slices = cfg->num_h_slices; sp.num_v_slices = cfg->num_v_slices;
    sp.ec = cfg->slicecrc ? 1 : 0; sp.context_model = cfg->context ? 1 : 0; sp.compact = cfg->context == 2; sp.coder = cfg->coder == 2 ? 2 : 1; sp.version = cfg->level == 1 ? 1 : 3;
    ffv1::stream_desc s;
    ffv1::stream_of_encoder(sp, s);
    return decoder_create(cfg, s, out);
}

extern "C" int rcgpu_ffv1_decoder_create_for_stream(const rcgpu_ffv1_config* files, const rcgpu_ffv1_stream* stream, rcgpu_ffv1_decoder** out)
{
    clear_error();
    if (!files || !stream || !out) return fail(1, "ffv1 decoder: null argument");
    *out = nullptr;
    return decoder_create(files, stream->d, out);
}

// Decode n packets (device memory) into n payload buffers (device memory, data_size bytes each).  Asynchronous on the
// stream; *d_err_out (device, optional) receives the error flags -- 0 means every slice parsed, CRCs matched, no junk.
extern "C" int rcgpu_ffv1_decoder_decode_device(rcgpu_ffv1_decoder* d, const void* const* d_packets, const uint64_t* packet_sizes, uint32_t n,
                                                void* const* d_payloads, uint32_t* h_err_flags, void* hip_stream)
{
    clear_error();
    if (!d || !d_packets || !packet_sizes || !d_payloads) return fail(1, "ffv1 decoder: null argument");
    if (!n || n > d->cfg.max_batch) return fail(2, "ffv1 decoder: batch of %u frames (max_batch %u)", n, d->cfg.max_batch);
    HIP_TRY(hipSetDevice(d->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);      // NULL = the default stream: ordered after the caller's earlier work
    const dec_const& c = d->hc;
    const uint32_t nchains = n * c.S;
    for (uint32_t i = 0; i < n; i++) { d->h_ptrs[i] = const_cast<void*>(d_packets[i]); d->h_ptrs[d->cfg.max_batch + i] = d_payloads[i]; d->h_sizes[i] = packet_sizes[i]; }
    HIP_TRY(hipMemcpyAsync(d->d_pkt_ptrs, d->h_ptrs, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->d_out_ptrs, d->h_ptrs + d->cfg.max_batch, sizeof(void*) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->d_sizes, d->h_sizes, 8 * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d->d_err, 0, 16, st));
    if (d->d_init) {                                                                          // states_coded = 1: the stream's own initial states
        const unsigned long long total = (unsigned long long)nchains * d->nkeys * 2;
        hipLaunchKernelGGL(k_dec_preset, dim3(uint32_t(std::min<unsigned long long>((total + 255) / 256, 65536))), dim3(256), 0, st,
                           reinterpret_cast<uint4*>(d->d_states + d->states_off), reinterpret_cast<const uint4*>(d->d_init), d->nkeys * 2, total);
    }
    else HIP_TRY(hipMemsetAsync(d->d_states + d->states_off, 128, size_t(nchains) * d->nkeys * 32, st));     // states_coded = 0: every state starts at 128
    HIP_TRY(hipEventRecord(d->ev[0], st));
    hipLaunchKernelGGL(k_dec_split, dim3((n + 63) / 64), dim3(64), 0, st, d->d_const, d->d_pkt_ptrs, d->d_sizes, n, d->d_slice_start, d->d_slice_len, d->d_err);
    if (c.ec) hipLaunchKernelGGL(k_dec_crc, dim3(nchains), dim3(256), 0, st, d->d_const, d->d_pkt_ptrs, d->d_slice_start, d->d_slice_len, d->d_err);
    HIP_TRY(hipEventRecord(d->ev[1], st));
    // the slices are decoded on the CUs the hash does not use (partition_streams): the kernel goes to a stream of the decoder's own, between
    // two events that keep it where it was in the caller's stream
    hipStream_t ks = d->dec_stream ? d->dec_stream : st;
    if (ks != st) HIP_TRY(hipStreamWaitEvent(ks, d->ev[1], 0));
    if (d->ring)
        hipLaunchKernelGGL(k_dec_slices<true>, dim3((nchains + 63) / 64), dim3(64), size_t(c.nqslots) * 2560, ks, d->d_const, d->d_pkt_ptrs, d->d_slice_start, d->d_slice_len, nchains,
                           d->d_states + d->states_off, d->nkeys, d->d_planes, d->d_out_ptrs, d->ring_w, d->d_err, d->d_hdr);
    else
        hipLaunchKernelGGL(k_dec_slices<false>, dim3((nchains + 63) / 64), dim3(64), size_t(c.nqslots) * 2560, ks, d->d_const, d->d_pkt_ptrs, d->d_slice_start, d->d_slice_len, nchains,
                           d->d_states + d->states_off, d->nkeys, d->d_planes, d->d_out_ptrs, 0u, d->d_err, d->d_hdr);
    HIP_TRY(hipEventRecord(d->ev[2], ks));
    if (ks != st) HIP_TRY(hipStreamWaitEvent(st, d->ev[2], 0));
    if (d->ring) { /* packed inline */ }
    else {
        const uint32_t nwords = c.altern ? (c.W * c.H + 2) / 3 : c.H * (c.line_bytes / 4);
        hipLaunchKernelGGL(k_pack_words, dim3((nwords + 255) / 256, n), dim3(256), 0, st, d->d_const, d->d_planes, d->d_out_ptrs);
    }
    HIP_TRY(hipEventRecord(d->ev[3], st));
    HIP_TRY(hipGetLastError());
    d->ev_valid = true;
    if (h_err_flags) {
        HIP_TRY(hipMemcpyAsync(h_err_flags, d->d_err, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (*h_err_flags) return fail(210, "ffv1 decoder: undecodable frame in the batch (flags 0x%x)", *h_err_flags);
    }
    return 0;
}

static void hint_join(rcgpu_ffv1_decoder* d, bool keep);

extern "C" int rcgpu_ffv1_decoder_decode_host(rcgpu_ffv1_decoder* d, const uint8_t* const* packets, const uint64_t* packet_sizes, uint32_t n,
                                              uint8_t* const* payloads)
{
    clear_error();
    if (!d || !packets || !packet_sizes || !payloads) return fail(1, "ffv1 decoder: null argument");
    hint_join(d, false);                                             // a batch decoded ahead uses this decoder's buffers: it is dropped
    if (!n || n > d->cfg.max_batch) return fail(2, "ffv1 decoder: batch of %u frames (max_batch %u)", n, d->cfg.max_batch);
    HIP_TRY(hipSetDevice(d->cfg.device));
    const size_t out_bytes = size_t(payload_bytes(d->cfg.pixfmt, d->cfg.width, d->cfg.height, d->cfg.line_bytes, d->cfg.flags));
    const size_t out_stride = (out_bytes + 255) & ~size_t(255);
    uint64_t in_total = 0;
    for (uint32_t i = 0; i < n; i++) in_total += (packet_sizes[i] + 255) & ~uint64_t(255);
    uint8_t* d_in = nullptr; uint8_t* d_out = nullptr;
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_in), size_t(in_total) + 256);
    if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&d_out), out_stride * n);
    std::vector<const void*> pk(n); std::vector<void*> out(n);
    uint64_t off = 0;
    for (uint32_t i = 0; i < n && he == hipSuccess; i++) {
        he = hipMemcpyAsync(d_in + off, packets[i], packet_sizes[i], hipMemcpyHostToDevice, d->own_stream);
        pk[i] = d_in + off; out[i] = d_out + size_t(i) * out_stride;
        off += (packet_sizes[i] + 255) & ~uint64_t(255);
    }
    int rc = 0;
    if (he == hipSuccess) {
        uint32_t flags = 0;
        rc = rcgpu_ffv1_decoder_decode_device(d, pk.data(), packet_sizes, n, out.data(), &flags, d->own_stream);
        for (uint32_t i = 0; i < n && !rc && he == hipSuccess; i++)
            he = hipMemcpyAsync(payloads[i], out[i], out_bytes, hipMemcpyDeviceToHost, d->own_stream);
        if (he == hipSuccess) he = hipStreamSynchronize(d->own_stream);
    }
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (he != hipSuccess) return fail(100, "ffv1 decoder: %s", hipGetErrorString(he));
    return rc;
}

// ---- `--check` with the payloads staying on the device (rcgpu.h: decode_keep / kept_to_host / verify_kept)
namespace {

// what goes up: `size` bytes at `src`, or -- when src is NULL -- at offset `off` of the open file `fd`, read with pread() straight into the
// staging buffer.  For a caller that has the file mapped anyway the address is the faster of the two (6 GB of packets: 0.3 s from the mapping,
// 0.35-0.7 s with pread(), and no better through private windows mapped per chunk); the mapping then costs ~0.4 s to tear down, once.
struct up_item { uint8_t* dst; const uint8_t* src; size_t size; int fd = -1; uint64_t off = 0; };

// One thread is bound by the page faults of a mapping and by its own copy into the staging buffer; eight of them reach the link's rate
// (13.6 GB of mapped files in 0.25 s).
hipError_t upload_side_by_side(stager& up, int device, const std::vector<up_item>& items)
{
    std::vector<up_item> chunks;
    for (const up_item& it : items)
        for (size_t o = 0; o < it.size; o += stager::kStage)
            chunks.push_back({ it.dst + o, it.src ? it.src + o : nullptr, std::min(stager::kStage, it.size - o), it.fd, it.off + o });
    if (chunks.empty()) return hipSuccess;
    unsigned nt = stager::kLanes;
    if (const char* e = getenv("RCGPU_UPLOAD_THREADS")) nt = unsigned(std::max(1, std::min(int(stager::kLanes), atoi(e))));
    nt = unsigned(std::min<size_t>(nt, chunks.size()));
    for (unsigned t = 0; t < nt; t++) {
        auto& l = up.lanes[t];
        hipError_t he = hipSuccess;
        if (!l.st) he = hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking);
        for (int k = 0; k < 2 && he == hipSuccess; k++) {
            if (!l.stage[k]) he = hipHostMalloc(reinterpret_cast<void**>(&l.stage[k]), stager::kStage);
            if (he == hipSuccess && !l.ev[k]) he = hipEventCreateWithFlags(&l.ev[k], hipEventDisableTiming);
        }
        if (he != hipSuccess) return he;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> err{int(hipSuccess)}, short_read{0};
    auto work = [&](unsigned t) {
        auto& l = up.lanes[t];
        hipError_t he = hipSetDevice(device);
        bool used[2] = { false, false };
        for (int k = 0; he == hipSuccess; k ^= 1) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size()) break;
            if (used[k]) he = hipEventSynchronize(l.ev[k]);
            if (he != hipSuccess) break;
            if (chunks[i].src) memcpy(l.stage[k], chunks[i].src, chunks[i].size);
            else {
                size_t got = 0;
                while (got < chunks[i].size) {
                    const ssize_t r = pread(chunks[i].fd, l.stage[k] + got, chunks[i].size - got, off_t(chunks[i].off + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) break;
                    got += size_t(r);
                }
                if (got < chunks[i].size) { short_read.store(1); memset(l.stage[k] + got, 0, chunks[i].size - got); }
            }
            he = hipMemcpyAsync(chunks[i].dst, l.stage[k], chunks[i].size, hipMemcpyHostToDevice, l.st);
            if (he == hipSuccess) he = hipEventRecord(l.ev[k], l.st);
            used[k] = true;
        }
        const hipError_t hs = hipStreamSynchronize(l.st);
        if (he == hipSuccess) he = hs;
        if (he != hipSuccess) err.store(int(he));
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (short_read.load() && err.load() == int(hipSuccess)) return hipErrorFileNotFound;      // a file ended before the bytes asked for
    return hipError_t(err.load());
}

// RCGPU_TRACE_KEPT=1: where the time of a batch goes, on stderr
const std::chrono::steady_clock::time_point g_loaded = std::chrono::steady_clock::now();      // when the library was loaded: the traces say where in the run they are
struct kept_clock {
    const bool on = getenv("RCGPU_TRACE_KEPT") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what, uint32_t n) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rcgpu kept: %-28s %4u files %8.1f ms   (at %.2f s)\n", what, n, std::chrono::duration<double, std::milli>(now - t).count(),
                std::chrono::duration<double>(now - g_loaded).count());
        t = now;
    }
};

hipError_t grow(uint8_t*& p, size_t& cap, size_t need)
{
    if (need <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const hipError_t he = hipMalloc(reinterpret_cast<void**>(&p), need);
    if (he == hipSuccess) cap = need;
    return he;
}

}  // namespace

// joins the batch decoded ahead, if any; `keep` says whether its slots are wanted
static void hint_join(rcgpu_ffv1_decoder* d, bool keep)
{
    if (!d->hint.active) return;
    d->hint.th.join();
    d->hint.active = false;
    if (!keep) d->kept[d->hint.set].n = 0;
}

static int decode_keep_into(rcgpu_ffv1_decoder* d, int set, uint8_t*& d_in, size_t& d_in_cap, stager& lanes,
                            const uint8_t* const* packets, int fd, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n);

static int decode_keep(rcgpu_ffv1_decoder* d, const uint8_t* const* packets, int fd, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n)
{
    if (!n || n > d->cfg.max_batch) return fail(2, "ffv1 decoder: batch of %u frames (max_batch %u)", n, d->cfg.max_batch);
    HIP_TRY(hipSetDevice(d->cfg.device));
    if (d->hint.active) {
        // was this the batch that is being decoded ahead?  Then its slots become the current ones; else it is dropped and decoded anew
        const bool same = packets && d->hint.packets.size() == n && std::equal(d->hint.packets.begin(), d->hint.packets.end(), packets) &&
                          std::equal(d->hint.sizes.begin(), d->hint.sizes.end(), packet_sizes);
        hint_join(d, same);
        if (same && d->hint.rc == 0) { d->kept_cur = d->hint.set; return 0; }
        d->kept[d->hint.set].n = 0;
    }
    if (d->kept[d->kept_cur].pending) {                              // its files are still being hashed: this batch goes into another set
        int other = -1;
        for (int k = 0; k < 3; k++) if (!d->kept[k].pending) { other = k; break; }
        if (other < 0) return fail(2, "ffv1 decoder: three batches wait for rcgpu_ffv1_decoder_verify_kept_end");
        d->kept_cur = other;
    }
    return decode_keep_into(d, d->kept_cur, d->d_kept_in, d->kept_in_cap, d->up, packets, fd, offsets, packet_sizes, n);
}

static int decode_keep_into(rcgpu_ffv1_decoder* d, int set, uint8_t*& d_in, size_t& d_in_cap, stager& lanes,
                            const uint8_t* const* packets, int fd, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n)
{
    rcgpu_ffv1_decoder::kept_set& K = d->kept[set];
    K.n = 0;
    uint64_t in_total = 0;
    for (uint32_t i = 0; i < n; i++) in_total += (packet_sizes[i] + 255) & ~uint64_t(255);
    kept_clock clk;
    // Sized for the decoder's largest batch at the first call: a buffer that grows later is freed and allocated anew, and an allocation that
    // follows a free of tens of GB waits for the driver to wipe them (measured: 2-3 s for the 39 GB of a 744-frame batch after a 256-frame
    // one, 1.4 ms for the same bytes on memory that was never used)
    const uint32_t most = std::max(n, d->cfg.max_batch);
    HIP_TRY(grow(d_in, d_in_cap, size_t(in_total / n + 1) * most * 17 / 16 + 256));
    HIP_TRY(grow(K.d, K.cap, d->kept_stride * most));
    clk.lap("decode_keep: device buffers", n);
    std::vector<up_item> up(n);
    std::vector<const void*> pk(n); std::vector<void*> out(n);
    uint64_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        up[i].dst = d_in + off; up[i].size = size_t(packet_sizes[i]);
        if (packets) up[i].src = packets[i];
        else { up[i].src = nullptr; up[i].fd = fd; up[i].off = offsets[i]; }
        pk[i] = d_in + off; out[i] = K.d + size_t(i) * d->kept_stride + RCGPU_KEPT_ROOM;
        off += (packet_sizes[i] + 255) & ~uint64_t(255);
    }
    {
        const hipError_t he = upload_side_by_side(lanes, d->cfg.device, up);
        if (he == hipErrorFileNotFound) return fail(21, "ffv1 decoder: the file ends before a packet does");
        HIP_TRY(he);
        clk.lap("decode_keep: packets up", n);
    }
    uint32_t flags = 0;
    const int rc = rcgpu_ffv1_decoder_decode_device(d, pk.data(), packet_sizes, n, out.data(), &flags, d->own_stream);
    clk.lap("decode_keep: decoded", n);
    if (!rc) K.n = n;
    return rc;
}

extern "C" int rcgpu_ffv1_decoder_decode_keep(rcgpu_ffv1_decoder* d, const uint8_t* const* packets, const uint64_t* packet_sizes, uint32_t n)
{
    clear_error();
    if (!d || !packets || !packet_sizes) return fail(1, "ffv1 decoder: null argument");
    return decode_keep(d, packets, -1, nullptr, packet_sizes, n);
}

extern "C" int rcgpu_ffv1_decoder_decode_keep_fd(rcgpu_ffv1_decoder* d, int fd, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n)
{
    clear_error();
    if (!d || fd < 0 || !offsets || !packet_sizes) return fail(1, "ffv1 decoder: null argument");
    return decode_keep(d, nullptr, fd, offsets, packet_sizes, n);
}

// The batch the caller will pass to its NEXT rcgpu_ffv1_decoder_decode_keep, started now on a thread of its own (rcgpu.h).
extern "C" int rcgpu_ffv1_decoder_decode_keep_hint(rcgpu_ffv1_decoder* d, const uint8_t* const* packets, const uint64_t* packet_sizes, uint32_t n)
{
    clear_error();
    if (!d || !packets || !packet_sizes) return fail(1, "ffv1 decoder: null argument");
    if (!n || n > d->cfg.max_batch) return fail(2, "ffv1 decoder: batch of %u frames (max_batch %u)", n, d->cfg.max_batch);
    if (d->hint.active) return 0;                                   // one batch ahead, not two
    int set = -1;
    for (int k = 0; k < 3; k++) if (k != d->kept_cur && !d->kept[k].pending) { set = k; break; }
    if (set < 0) return 0;                                          // no set of slots to spare: the batch is decoded when it is asked for
    d->hint.packets.assign(packets, packets + n); d->hint.sizes.assign(packet_sizes, packet_sizes + n);
    d->hint.set = set; d->hint.rc = 0; d->hint.active = true;
    d->hint.th = std::thread([d, n] {
        (void)hipSetDevice(d->cfg.device);
        d->hint.rc = decode_keep_into(d, d->hint.set, d->d_kept_in2, d->kept_in2_cap, d->up2, d->hint.packets.data(), -1, nullptr, d->hint.sizes.data(), n);
        d->hint.error = d->hint.rc ? rcgpu_last_error() : "";
    });
    return 0;
}

// The same with the packets named by their place in a file, which the library maps for itself (once per file, for the decoder's life).
extern "C" int rcgpu_ffv1_decoder_decode_keep_hint_file(rcgpu_ffv1_decoder* d, const char* path, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n)
{
    clear_error();
    if (!d || !path || !offsets || !packet_sizes) return fail(1, "ffv1 decoder: null argument");
    if (!n || n > d->cfg.max_batch) return fail(2, "ffv1 decoder: batch of %u frames (max_batch %u)", n, d->cfg.max_batch);
    if (d->hint.active) return 0;
    // the mapping is kept for the decoder's life, but only while it is still THAT file: same name, device, inode, size and modification time
    struct stat now;
    const bool have = stat(path, &now) == 0;
    const bool same = have && d->map_base && d->map_path == path && now.st_dev == d->map_dev && now.st_ino == d->map_ino && size_t(now.st_size) == d->map_size &&
                      now.st_mtim.tv_sec == d->map_mtime.tv_sec && now.st_mtim.tv_nsec == d->map_mtime.tv_nsec;
    if (!same) {
        if (d->map_base) { (void)munmap(const_cast<uint8_t*>(d->map_base), d->map_size); d->map_base = nullptr; d->map_size = 0; d->map_path.clear(); }
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return fail(21, "ffv1 decoder: cannot open %s", path);
        struct stat st;
        void* m = fstat(fd, &st) == 0 && st.st_size > 0 ? mmap(nullptr, size_t(st.st_size), PROT_READ, MAP_SHARED, fd, 0) : MAP_FAILED;
        close(fd);
        if (m == MAP_FAILED) return fail(21, "ffv1 decoder: cannot map %s", path);
        d->map_base = static_cast<const uint8_t*>(m); d->map_size = size_t(st.st_size); d->map_path = path;
        d->map_dev = st.st_dev; d->map_ino = st.st_ino; d->map_mtime = st.st_mtim;
    }
    std::vector<const uint8_t*> pk(n);
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i] > d->map_size || packet_sizes[i] > d->map_size - offsets[i]) return fail(21, "ffv1 decoder: the file ends before a packet does");
        pk[i] = d->map_base + offsets[i];
    }
    return rcgpu_ffv1_decoder_decode_keep_hint(d, pk.data(), packet_sizes, n);
}

// The batch decoded ahead becomes the current one (the caller has made sure it is the batch it wants); an error if there is none or it failed
// -- the caller then decodes the batch with decode_keep as if nothing had been hinted.
extern "C" int rcgpu_ffv1_decoder_decode_keep_adopt(rcgpu_ffv1_decoder* d)
{
    clear_error();
    if (!d) return fail(1, "ffv1 decoder: null argument");
    if (!d->hint.active) return fail(2, "ffv1 decoder: no batch was decoded ahead");
    kept_clock clk;
    hint_join(d, true);
    clk.lap("decode_keep: waited for the batch decoded ahead", uint32_t(d->hint.packets.size()));
    if (d->hint.rc) { d->kept[d->hint.set].n = 0; return fail(d->hint.rc, "ffv1 decoder: the batch decoded ahead failed: %s", d->hint.error.c_str()); }
    d->kept_cur = d->hint.set;
    return 0;
}

extern "C" int rcgpu_ffv1_decoder_kept_to_host(rcgpu_ffv1_decoder* d, uint32_t slot, uint8_t* payload)
{
    clear_error();
    if (!d || !payload) return fail(1, "ffv1 decoder: null argument");
    const rcgpu_ffv1_decoder::kept_set& K = d->kept[d->kept_cur];
    if (slot >= K.n) return fail(2, "ffv1 decoder: slot %u of %u kept payloads", slot, K.n);
    HIP_TRY(hipSetDevice(d->cfg.device));
    const size_t out_bytes = size_t(payload_bytes(d->cfg.pixfmt, d->cfg.width, d->cfg.height, d->cfg.line_bytes, d->cfg.flags));
    kept_clock clk;
    HIP_TRY(hipMemcpy(payload, K.d + size_t(slot) * d->kept_stride + RCGPU_KEPT_ROOM, out_bytes, hipMemcpyDeviceToHost));
    clk.lap("kept_to_host: slot", slot);
    return 0;
}

// The verification of a batch in two calls, so that the next batch can be decoded while this one is hashed (one lane per file: 0.8 s for
// 53 MB files however many).  _begin copies what it needs of the caller's memory, uploads and compares the files on disk and starts the
// hashes; _end waits for them.  Between the two, decode_keep fills the other set of slots.
extern "C" int rcgpu_ffv1_decoder_verify_kept_begin(rcgpu_ffv1_decoder* d, const rcgpu_kept_file* files, uint32_t n)
{
    clear_error();
    if (!d || !files || !n) return fail(1, "ffv1 decoder: null argument");
    if (d->kept[0].pending || d->kept[1].pending || d->kept[2].pending) return fail(2, "ffv1 decoder: a verification is waiting for rcgpu_ffv1_decoder_verify_kept_end");
    HIP_TRY(hipSetDevice(d->cfg.device));
    rcgpu_ffv1_decoder::kept_set& K = d->kept[d->kept_cur];
    const size_t P = size_t(payload_bytes(d->cfg.pixfmt, d->cfg.width, d->cfg.height, d->cfg.line_bytes, d->cfg.flags));
    const size_t disk_stride = (P + 255) & ~size_t(255);
    uint32_t n_md5 = 0, n_cmp = 0;
    for (uint32_t i = 0; i < n; i++) {
        const rcgpu_kept_file& f = files[i];
        if (f.slot >= K.n) return fail(2, "ffv1 decoder: slot %u of %u kept payloads", f.slot, K.n);
        if (f.before_size > RCGPU_KEPT_ROOM || f.after_size > RCGPU_KEPT_ROOM || (f.before_size && !f.before) || (f.after_size && !f.after))
            return fail(2, "ffv1 decoder: %llu bytes before and %llu after the payload (at most %u each)", (unsigned long long)f.before_size, (unsigned long long)f.after_size, RCGPU_KEPT_ROOM);
        for (uint32_t j = 0; j < i; j++) if (files[j].slot == f.slot) return fail(2, "ffv1 decoder: slot %u named twice", f.slot);
        n_md5 += (f.flags & RCGPU_KEPT_MD5) != 0; n_cmp += f.on_disk != nullptr || f.on_disk_path != nullptr;
    }
    if (!d->side_stream) HIP_TRY(hipStreamCreateWithFlags(&d->side_stream, hipStreamNonBlocking));
    if (!d->md5_stream) {
        // the hashes are the background work of the next batch's decoding: lowest priority, which also gives the stream a hardware queue of
        // its own (streams of one priority share a few queues, and whatever is queued behind the 0.8 s hash kernel there waits for it)
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&d->md5_stream, hipStreamNonBlocking, lo));
    }
    if (!d->ev_tab) HIP_TRY(hipEventCreateWithFlags(&d->ev_tab, hipEventDisableTiming));
    if (!d->ev_side) HIP_TRY(hipEventCreateWithFlags(&d->ev_side, hipEventDisableTiming));
    kept_clock clk;
    // the tables of both kernels -- pointers, lengths, results -- in one pinned block and its mirror on the device, 8 words per file:
    // [image, image bytes, payload, file's bytes, bytes to compare, first difference, md5 (2 words)] x n, array after array
    if (K.tab_cap < n) {
        if (K.h_tab) (void)hipHostFree(K.h_tab);
        if (K.d_tab) (void)hipFree(K.d_tab);
        K.h_tab = nullptr; K.d_tab = nullptr; K.tab_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&K.h_tab), size_t(n) * 64));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&K.d_tab), size_t(n) * 64));
        K.tab_cap = n;
    }
    uint64_t* h = K.h_tab; uint64_t* g = K.d_tab;
    K.img_of.clear(); K.cmp_of.clear(); K.part.clear();
    hipError_t he = hipSuccess;
    // 1. the hashes: the bytes around the payload go next to it (on the side stream, which this call waits for: the caller's buffers are
    // free when it returns), one lane hashes one file
    // (through one pinned block: a copy from pageable memory waits for the stream every time, ~1 ms each, two per file)
    size_t edge_bytes = 0;
    for (uint32_t i = 0; i < n; i++) if (files[i].flags & RCGPU_KEPT_MD5) edge_bytes += size_t(files[i].before_size + files[i].after_size);
    if (edge_bytes > d->edges_cap) {
        if (d->h_edges) (void)hipHostFree(d->h_edges);
        d->h_edges = nullptr; d->edges_cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&d->h_edges), edge_bytes + (edge_bytes >> 2)));
        d->edges_cap = edge_bytes + (edge_bytes >> 2);
    }
    size_t eo = 0;
    for (uint32_t i = 0; i < n && he == hipSuccess; i++) {
        const rcgpu_kept_file& f = files[i];
        if (!(f.flags & RCGPU_KEPT_MD5)) continue;
        uint8_t* pay = K.d + size_t(f.slot) * d->kept_stride + RCGPU_KEPT_ROOM;
        if (f.before_size) { memcpy(d->h_edges + eo, f.before, f.before_size); he = hipMemcpyAsync(pay - f.before_size, d->h_edges + eo, f.before_size, hipMemcpyHostToDevice, d->side_stream); eo += f.before_size; }
        if (f.after_size && he == hipSuccess) { memcpy(d->h_edges + eo, f.after, f.after_size); he = hipMemcpyAsync(pay + P, d->h_edges + eo, f.after_size, hipMemcpyHostToDevice, d->side_stream); eo += f.after_size; }
        h[K.img_of.size()] = reinterpret_cast<uintptr_t>(pay - f.before_size); h[n + K.img_of.size()] = f.before_size + P + f.after_size;
        K.img_of.push_back(i);
    }
    // 2. the comparison (CheckFile_Compare over Pre, plane, Post, FileWriter.cpp:448-463,581-589, then the length, :200): the bytes
    // around the payload are compared here, the payload with the file's bytes on the device
    // A file named by its path is read with pread(): its few bytes around the payload here, the rest straight into the staging buffers.
    std::vector<up_item> up;
    std::vector<int> fds;                                   // open while their bytes go up; closed group by group (hundreds of files)
    if (n_cmp && he == hipSuccess) he = grow(d->d_disk, d->disk_cap, disk_stride * std::max(n_cmp, d->cfg.max_batch));
    for (uint32_t i = 0; i < n && he == hipSuccess; i++) {
        const rcgpu_kept_file& f = files[i];
        if (!f.on_disk && !f.on_disk_path) continue;
        uint64_t disk_size = f.on_disk_size;
        int fd = -1;
        std::vector<uint8_t> head, tail;                    // what the file has where this file has `before` and `after`
        if (!f.on_disk) {
            struct stat st;
            fd = open(f.on_disk_path, O_RDONLY);
            if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); for (int x : fds) close(x); return fail(20, "ffv1 decoder: cannot open %s: %s", f.on_disk_path, strerror(errno)); }
            disk_size = uint64_t(st.st_size);
            head.resize(size_t(std::min<uint64_t>(f.before_size, disk_size)));
            const uint64_t t0 = f.before_size + P, t1 = std::min<uint64_t>(t0 + f.after_size, disk_size);
            if (t1 > t0) tail.resize(size_t(t1 - t0));
            if ((!head.empty() && pread(fd, head.data(), head.size(), 0) != ssize_t(head.size())) || (!tail.empty() && pread(fd, tail.data(), tail.size(), off_t(t0)) != ssize_t(tail.size()))) {
                close(fd); for (int x : fds) close(x);
                return fail(20, "ffv1 decoder: cannot read %s", f.on_disk_path);
            }
            fds.push_back(fd);
        }
        const uint64_t have = disk_size > f.before_size ? std::min<uint64_t>(disk_size - f.before_size, P) : 0;
        uint8_t* dst = d->d_disk + K.cmp_of.size() * disk_stride;
        if (have) { up_item it{ dst, f.on_disk ? f.on_disk + f.before_size : nullptr, size_t(have) }; it.fd = fd; it.off = f.before_size; up.push_back(it); }
        const size_t k = K.cmp_of.size();
        h[2 * n + k] = reinterpret_cast<uintptr_t>(K.d + size_t(f.slot) * d->kept_stride + RCGPU_KEPT_ROOM); h[3 * n + k] = reinterpret_cast<uintptr_t>(dst);
        h[4 * n + k] = have; h[5 * n + k] = ~uint64_t(0);
        rcgpu_ffv1_decoder::kept_set::host_part hp{ f.before_size + P + f.after_size, disk_size, f.before_size, ~uint64_t(0), ~uint64_t(0) };
        const uint64_t common = std::min<uint64_t>(hp.mine, disk_size), nb = std::min<uint64_t>(f.before_size, common);
        const uint8_t* dh = f.on_disk ? f.on_disk : head.data();
        for (uint64_t q = 0; q < nb && hp.head_diff == ~uint64_t(0); q++) if (f.before[q] != dh[q]) hp.head_diff = q;
        for (uint64_t q = f.before_size + P; q < common && hp.tail_diff == ~uint64_t(0); q++)
            if (f.after[q - f.before_size - P] != (f.on_disk ? f.on_disk[q] : tail[size_t(q - f.before_size - P)])) hp.tail_diff = q;
        K.part.push_back(hp);
        K.cmp_of.push_back(i);
        if (fds.size() >= 128 && he == hipSuccess) {       // this group's bytes go up now
            he = upload_side_by_side(d->up, d->cfg.device, up);
            up.clear();
            for (int x : fds) close(x);
            fds.clear();
        }
    }
    if (he == hipSuccess) he = hipMemcpyAsync(g, h, size_t(n) * 48, hipMemcpyHostToDevice, d->side_stream);
    if (he == hipSuccess) he = hipEventRecord(d->ev_tab, d->side_stream);
    // the hashes start now, on their own low-priority stream (a hardware queue of its own: the comparison queued below does not wait behind them)
    if (n_md5 && he == hipSuccess) {
        he = hipStreamWaitEvent(d->md5_stream, d->ev_tab, 0);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(k_md5, dim3((n_md5 + 63) / 64), dim3(64), 0, d->md5_stream, reinterpret_cast<const uint8_t* const*>(g), reinterpret_cast<const unsigned long long*>(g + n), n_md5,
                               reinterpret_cast<uint8_t*>(g + 6 * size_t(n)));
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipMemcpyAsync(h + 6 * size_t(n), g + 6 * size_t(n), size_t(n_md5) * 16, hipMemcpyDeviceToHost, d->md5_stream);
    }
    clk.lap("verify_kept: md5 launched", n_md5);
    if (n_cmp && he == hipSuccess) {
        he = upload_side_by_side(d->up, d->cfg.device, up);
        for (int x : fds) close(x);
        fds.clear();
        if (he == hipErrorFileNotFound) return fail(20, "ffv1 decoder: a file on disk ended while it was read");
        clk.lap("verify_kept: files up", n_cmp);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(k_compare_batch, dim3(64, n_cmp), dim3(256), 0, d->side_stream, reinterpret_cast<const uint8_t* const*>(g + 2 * size_t(n)), reinterpret_cast<const uint8_t* const*>(g + 3 * size_t(n)),
                               reinterpret_cast<const unsigned long long*>(g + 4 * size_t(n)), reinterpret_cast<unsigned long long*>(g + 5 * size_t(n)));
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipMemcpyAsync(h + 5 * size_t(n), g + 5 * size_t(n), size_t(n_cmp) * 8, hipMemcpyDeviceToHost, d->side_stream);
    }
    for (int x : fds) close(x);                              // (an error on the way here skipped the upload that closes them)
    fds.clear();
    // everything that reads the caller's memory is queued on the side stream: mark it and wait for the mark
    if (he == hipSuccess) he = hipEventRecord(d->ev_side, d->side_stream);
    const hipError_t hs = he == hipSuccess ? hipEventSynchronize(d->ev_side) : hipStreamSynchronize(d->side_stream);
    if (he == hipSuccess) he = hs;
    clk.lap("verify_kept: compared", n_cmp);
    if (he != hipSuccess) { (void)hipStreamSynchronize(d->md5_stream); return fail(100, "ffv1 decoder: verify: %s", hipGetErrorString(he)); }
    K.pending = true; K.pn = n; K.pn_md5 = n_md5; K.pn_cmp = n_cmp;
    return 0;
}

extern "C" int rcgpu_ffv1_decoder_verify_kept_end(rcgpu_ffv1_decoder* d, rcgpu_kept_verdict* verdicts)
{
    clear_error();
    if (!d || !verdicts) return fail(1, "ffv1 decoder: null argument");
    rcgpu_ffv1_decoder::kept_set* Kp = d->kept[0].pending ? &d->kept[0] : d->kept[1].pending ? &d->kept[1] : d->kept[2].pending ? &d->kept[2] : nullptr;
    if (!Kp) return fail(2, "ffv1 decoder: no verification was begun");
    rcgpu_ffv1_decoder::kept_set& K = *Kp;
    HIP_TRY(hipSetDevice(d->cfg.device));
    kept_clock clk;
    K.pending = false;
    const hipError_t he = hipStreamSynchronize(d->md5_stream);
    clk.lap("verify_kept: md5 done", K.pn_md5);
    if (he != hipSuccess) return fail(100, "ffv1 decoder: verify: %s", hipGetErrorString(he));
    const size_t n = K.pn;
    const uint8_t* md5s = reinterpret_cast<const uint8_t*>(K.h_tab + 6 * n);
    const uint64_t* diffs = K.h_tab + 5 * n;
    for (size_t i = 0; i < n; i++) { memset(verdicts[i].md5, 0, 16); verdicts[i].first_diff = ~uint64_t(0); }
    for (uint32_t k = 0; k < K.pn_md5; k++) memcpy(verdicts[K.img_of[k]].md5, &md5s[size_t(k) * 16], 16);
    for (uint32_t k = 0; k < K.pn_cmp; k++) {
        const auto& hp = K.part[k];
        uint64_t at = hp.head_diff;
        if (at == ~uint64_t(0) && diffs[k] != ~uint64_t(0)) at = hp.before_size + diffs[k];
        if (at == ~uint64_t(0)) at = hp.tail_diff;
        if (at == ~uint64_t(0) && hp.mine != hp.on_disk_size) at = std::min<uint64_t>(hp.mine, hp.on_disk_size);
        verdicts[K.cmp_of[k]].first_diff = at;
    }
    return 0;
}

extern "C" int rcgpu_ffv1_decoder_verify_kept(rcgpu_ffv1_decoder* d, const rcgpu_kept_file* files, uint32_t n, rcgpu_kept_verdict* verdicts)
{
    if (!verdicts) { clear_error(); return fail(1, "ffv1 decoder: null argument"); }
    if (const int rc = rcgpu_ffv1_decoder_verify_kept_begin(d, files, n)) return rc;
    return rcgpu_ffv1_decoder_verify_kept_end(d, verdicts);
}

// Debug taps for the tests (like rcgpu_ffv1_debug_fetch: exported, not part of include/rcgpu.h).  The sample decoder takes bytes out of a
// 7-byte window without looking and decodes a sample again, carefully, when it outran the window: with real pictures that almost never
// happens (1.9 bytes per 16-bit sample), so the tests make the window smaller -- same payloads, the careful path on most samples -- and
// count how often it ran in the last batch.
extern "C" int rcgpu_ffv1_decoder_debug_window(rcgpu_ffv1_decoder* d, uint32_t bytes)
{
    if (!d || bytes < 1 || bytes > 7) return 1;
    if (hipSetDevice(d->cfg.device) != hipSuccess) return 2;
    d->hc.win_cap = bytes;
    return hipMemcpy(d->d_const, &d->hc, sizeof d->hc, hipMemcpyHostToDevice) == hipSuccess ? 0 : 2;
}
// Test hook: the decoder's context-state arrays begin `bytes` (a multiple of 256, at most 64 MiB) into their allocation from the next batch on --
// one allocation, two base addresses: does the decoder's time depend on WHERE its 33 GB of states lie?  (tools/check_offsets.py)
extern "C" int rcgpu_ffv1_decoder_debug_states_offset(rcgpu_ffv1_decoder* d, uint64_t bytes)
{
    if (!d || (bytes & 255) || bytes > d->states_slack) return 1;
    d->states_off = size_t(bytes);
    return 0;
}

extern "C" long long rcgpu_ffv1_decoder_debug_careful(rcgpu_ffv1_decoder* d)
{
    if (!d || hipSetDevice(d->cfg.device) != hipSuccess) return -1;
    uint32_t v[4] = {};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(v, d->d_err, 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)v[2];
}

extern "C" int rcgpu_ffv1_decoder_last_kernel_times(const rcgpu_ffv1_decoder* d, float ms[3])
{
    if (!d || !d->ev_valid || hipEventSynchronize(d->ev[3]) != hipSuccess) return 1;
    for (int k = 0; k < 3; k++) (void)hipEventElapsedTime(&ms[k], d->ev[k], d->ev[k + 1]);
    return 0;
}

// First differing byte of two device buffers (frame_writer::CheckFile, FileWriter.cpp:448-463); *first_diff = ~0 when equal.
// Padding-bit scan of DPX payloads, the test at the end of dpx::ParseBuffer (DPX.cpp:501-608): the reference walks the payload in
// units of `step` bytes and tests one byte of each against `mask` (the 2 or 4 filler bits of the FilledA / FilledB layouts), except
// for the unit that holds the end-of-line word of a line whose used bits do not fill it (packed layouts; Y 10-bit), which is
// tested as a big-endian 32-bit word against `eol_mask`.  It stops at the first hit; In_FirstNonZero = min(i, EOL_i).
struct pad_scan {
    uint32_t step, start, byte_at, mask;   // for (i = start; i < total; i += step) test byte i & mask; byte_at = i's offset inside its unit
    uint32_t eol_kind;                     // 0 none, 1 one word per line, 2 one word at the very end (RCGPU_FLAG_ALTERN)
    uint32_t eol_off, line, eol_mask;      // kind 1: word at eol_off of every `line` bytes; kind 2: at total - 4
    unsigned long long total;              // payload bytes
};

static int make_pad_scan(uint32_t pixfmt, uint32_t W, uint32_t H, uint32_t flags, pad_scan* out)
{
    if (pixfmt >= RCGPU_PIX_COUNT || pixfmt == RCGPU_PIX_EXR_RGB16) return fail(2, "padding scan: not a DPX layout (%u)", pixfmt);
    const pix_desc& d = pix(pixfmt);
    const bool altern = (flags & RCGPU_FLAG_ALTERN) != 0;
    pad_scan s{};
    const uint32_t line = payload_line_bytes(pixfmt, W, true);
    if (!W || !H || !line) return fail(2, "padding scan: %u x %u pixels", W, H);
    s.total = payload_bytes(pixfmt, W, H, line, flags);
    s.line = line;
    const bool y10 = pixfmt == RCGPU_PIX_Y10_FILLEDA_BE || pixfmt == RCGPU_PIX_Y10_FILLEDB_BE;
    const bool filled = y10 || pixfmt == RCGPU_PIX_RGB10_FILLEDA_BE || pixfmt == RCGPU_PIX_RGB10_FILLEDA_LE || pixfmt == RCGPU_PIX_RGB12_FILLEDA_BE ||
                        pixfmt == RCGPU_PIX_RGB12_FILLEDA_LE || pixfmt == RCGPU_PIX_RGBA10_FILLEDA_BE || pixfmt == RCGPU_PIX_RGBA10_FILLEDA_LE ||
                        pixfmt == RCGPU_PIX_RGBA12_FILLEDA_BE || pixfmt == RCGPU_PIX_RGBA12_FILLEDA_LE;
    if (!filled) {                                                  // packing::Packed, DPX.cpp:507-521
        const unsigned long long used = (unsigned long long)W * d.bits * d.planes;
        const uint32_t rem = uint32_t(used % 32);
        if (rem) { s.eol_kind = 1; s.eol_off = uint32_t(used / 32) * 4; s.line = s.eol_off + 4; s.step = s.line; s.start = s.eol_off; s.byte_at = 0; s.eol_mask = 0xFFFFFFFFu << rem; }
        *out = s; return 0;
    }
    const bool filled_b = pixfmt == RCGPU_PIX_Y10_FILLEDB_BE;       // DPX.cpp:523-566
    s.step = d.bits == 10 ? 4 : 2;
    s.byte_at = (d.big_endian != filled_b) ? s.step - 1 : 0;
    s.start = s.byte_at;
    s.mask = d.bits == 10 ? 0x3 : 0xF;
    if (filled_b) s.mask <<= d.bits == 10 ? 6 : 4;
    if (y10) {
        uint32_t rem = altern ? uint32_t(((unsigned long long)W * H) % 3) : W % 3;
        if (rem) {
            if (altern) s.eol_kind = 2;
            else { s.eol_kind = 1; s.eol_off = (W / 3) * 4; s.line = s.eol_off + 4; }
            rem *= 10;
            if (!filled_b) rem += 2;
            s.eol_mask = 0xFFFFFFFFu << rem;
            if (!filled_b) s.eol_mask |= 0x3;
        }
    }
    *out = s; return 0;
}

__global__ __launch_bounds__(256) void k_padscan(const uint8_t* const* __restrict__ payloads, pad_scan S, unsigned long long* __restrict__ first)
{
    const uint8_t* p = payloads[blockIdx.y];
    const unsigned long long units = S.step && S.total > S.start ? (S.total - S.start + S.step - 1) / S.step : 0;
    unsigned long long best = ~0ull;
    for (unsigned long long u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) {
        const unsigned long long i = S.start + u * S.step;
        bool eol = false; unsigned long long at = 0;
        if (S.eol_kind == 1) { const unsigned long long ln = i / S.line; eol = i - ln * S.line == S.eol_off + S.byte_at; at = ln * S.line + S.eol_off; }
        else if (S.eol_kind == 2) { at = S.total - 4; eol = i >= at && i - S.step < at; }
        bool hit;
        if (eol) { const uint32_t w = (uint32_t(p[at]) << 24) | (uint32_t(p[at + 1]) << 16) | (uint32_t(p[at + 2]) << 8) | p[at + 3]; hit = (w & S.eol_mask) != 0; }
        else { hit = (p[i] & S.mask) != 0; at = i; }
        if (hit) { best = at; break; }            // a thread's units only grow: its first hit is its smallest
    }
    if (best != ~0ull) atomicMin(&first[blockIdx.y], best);
}

// First non-zero padding position of n device payloads of one layout (dpx::ParseBuffer's scan, DPX.cpp:501-608: In_FirstNonZero
// relative to the payload), ~0 where all padding is zero -- the case in which the reversibility data needs no "In" block.
extern "C" int rcgpu_dpx_padding_scan_device(const void* const* d_payloads, uint32_t n, uint32_t pixfmt, uint32_t width, uint32_t height, uint32_t flags,
                                             uint64_t* first_nonzero, void* hip_stream)
{
    clear_error();
    if (!d_payloads || !first_nonzero || !n) return fail(1, "padding scan: null argument");
    pad_scan S;
    if (int r = make_pad_scan(pixfmt, width, height, flags, &S)) return r;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    unsigned long long* d_res = nullptr; const uint8_t** d_ptrs = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_res), 8 * size_t(n)));
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_ptrs), sizeof(void*) * size_t(n));
    if (he == hipSuccess) he = hipMemcpyAsync(d_ptrs, d_payloads, sizeof(void*) * size_t(n), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemsetAsync(d_res, 0xFF, 8 * size_t(n), st);
    if (he == hipSuccess && S.step) {
        const unsigned long long units = S.total > S.start ? (S.total - S.start + S.step - 1) / S.step : 0;
        const uint32_t blocks = uint32_t(std::min<unsigned long long>((units + 255) / 256, 4096));
        hipLaunchKernelGGL(k_padscan, dim3(blocks ? blocks : 1, n), dim3(256), 0, st, d_ptrs, S, d_res);
        he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipMemcpyAsync(first_nonzero, d_res, 8 * size_t(n), hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    (void)hipFree(d_res); if (d_ptrs) (void)hipFree(d_ptrs);
    if (he != hipSuccess) return fail(100, "padding scan: %s", hipGetErrorString(he));
    return 0;
}

extern "C" int rcgpu_compare_device(const void* d_a, const void* d_b, uint64_t n, uint64_t* first_diff, void* hip_stream)
{
    clear_error();
    if (!d_a || !d_b || !first_diff) return fail(1, "compare: null argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    unsigned long long* d_res = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_res), 8));
    hipError_t he = hipMemsetAsync(d_res, 0xFF, 8, st);
    if (he == hipSuccess) { hipLaunchKernelGGL(k_compare, dim3(2048), dim3(256), 0, st, static_cast<const uint8_t*>(d_a), static_cast<const uint8_t*>(d_b), (unsigned long long)n, d_res); he = hipGetLastError(); }
    unsigned long long r = 0;
    if (he == hipSuccess) he = hipMemcpyAsync(&r, d_res, 8, hipMemcpyDeviceToHost, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    (void)hipFree(d_res);
    if (he != hipSuccess) return fail(100, "compare: %s", hipGetErrorString(he));
    *first_diff = r;
    return 0;
}

// n pairs of device buffers at once; everything is ordered on `st` alone (see rcgpu_md5_device), first_diff = n values on the host.
extern "C" int rcgpu_compare_device_batch(const void* const* d_a, const void* const* d_b, const uint64_t* sizes, uint32_t n, uint64_t* first_diff, void* hip_stream)
{
    clear_error();
    if (!d_a || !d_b || !sizes || !first_diff || !n) return fail(1, "compare: null argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint8_t** d_pa = nullptr; const uint8_t** d_pb = nullptr; unsigned long long* d_s = nullptr; unsigned long long* d_r = nullptr;
    hipError_t he = hipMallocAsync(reinterpret_cast<void**>(&d_pa), sizeof(void*) * n, st);
    if (he == hipSuccess) he = hipMallocAsync(reinterpret_cast<void**>(&d_pb), sizeof(void*) * n, st);
    if (he == hipSuccess) he = hipMallocAsync(reinterpret_cast<void**>(&d_s), 8 * size_t(n), st);
    if (he == hipSuccess) he = hipMallocAsync(reinterpret_cast<void**>(&d_r), 8 * size_t(n), st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_pa, d_a, sizeof(void*) * n, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_pb, d_b, sizeof(void*) * n, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_s, sizes, 8 * size_t(n), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemsetAsync(d_r, 0xFF, 8 * size_t(n), st);
    if (he == hipSuccess) { hipLaunchKernelGGL(k_compare_batch, dim3(64, n), dim3(256), 0, st, d_pa, d_pb, d_s, d_r); he = hipGetLastError(); }
    if (he == hipSuccess) he = hipMemcpyAsync(first_diff, d_r, 8 * size_t(n), hipMemcpyDeviceToHost, st);
    for (void* p : { (void*)d_pa, (void*)d_pb, (void*)d_s, (void*)d_r }) if (p) (void)hipFreeAsync(p, st);
    const hipError_t hs = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hs;
    if (he != hipSuccess) return fail(100, "compare: %s", hipGetErrorString(he));
    return 0;
}

// MD5 of n device buffers, one lane each (frame_writer::CheckMD5, FileWriter.cpp:596-727; Input_Base.cpp:54-81).
extern "C" int rcgpu_md5_device(const void* const* d_bufs, const uint64_t* sizes, uint32_t n, uint8_t* out_md5 /* n x 16, host */, void* hip_stream)
{
    clear_error();
    if (!d_bufs || !sizes || !out_md5 || !n) return fail(1, "md5: null argument");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    // everything is ordered on `st` alone (stream-ordered allocations, asynchronous copies): a caller may hash one batch on a side
    // stream while another stream decodes the next one
    const uint8_t** d_p = nullptr; unsigned long long* d_s = nullptr; uint8_t* d_o = nullptr;
    hipError_t he = hipMallocAsync(reinterpret_cast<void**>(&d_p), sizeof(void*) * n, st);
    if (he == hipSuccess) he = hipMallocAsync(reinterpret_cast<void**>(&d_s), 8 * size_t(n), st);
    if (he == hipSuccess) he = hipMallocAsync(reinterpret_cast<void**>(&d_o), 16 * size_t(n), st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_p, d_bufs, sizeof(void*) * n, hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipMemcpyAsync(d_s, sizes, 8 * size_t(n), hipMemcpyHostToDevice, st);
    // the kernel itself runs on the hash's own CUs when the device has them to give (partition_streams; up to 64 wavefronts: beyond that
    // the 32 SIMDs of the partition would be the slower place), in the caller's order
    int dev = 0; hipStream_t hs_ = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    if (he == hipSuccess && n <= 4096 && hipGetDevice(&dev) == hipSuccess) hs_ = device_hash_stream(dev);
    if (hs_ && (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess)) hs_ = nullptr;
    if (he == hipSuccess && hs_) { he = hipEventRecord(e0, st); if (he == hipSuccess) he = hipStreamWaitEvent(hs_, e0, 0); }
    if (he == hipSuccess) { hipLaunchKernelGGL(k_md5, dim3((n + 63) / 64), dim3(64), 0, hs_ ? hs_ : st, d_p, d_s, n, d_o); he = hipGetLastError(); }
    if (he == hipSuccess && hs_) { he = hipEventRecord(e1, hs_); if (he == hipSuccess) he = hipStreamWaitEvent(st, e1, 0); }
    if (he == hipSuccess) he = hipMemcpyAsync(out_md5, d_o, 16 * size_t(n), hipMemcpyDeviceToHost, st);
    for (void* p : { (void*)d_p, (void*)d_s, (void*)d_o }) if (p) (void)hipFreeAsync(p, st);
    const hipError_t hs = hipStreamSynchronize(st);
    if (hs_ && hs != hipSuccess) (void)hipStreamSynchronize(hs_);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (he == hipSuccess) he = hs;
    if (he != hipSuccess) return fail(100, "md5: %s", hipGetErrorString(he));
    return 0;
}

// Whole-file MD5 of n host buffers (what input_base::Hash computes one file at a time on one core, Input_Base.cpp:54-81): buffers go
// up once, one lane hashes one buffer: 0.78 s for a 53 MB file, and for hundreds of them.
// rcgpu_analysis_host_batch adds the other pass the analysis makes over every byte of a DPX file, the padding-bit test of
// dpx::ParseBuffer (DPX.cpp:501-608), for the files rcgpu_dpx_probe recognises.
static int analysis_host_batch(const uint8_t* const* bufs, const uint64_t* sizes, uint32_t n, uint8_t* out_md5, uint64_t* first_nonzero, uint8_t* scanned, int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(3, "analysis: no HIP device available");
    if (device < 0 || device >= ndev) return fail(3, "analysis: device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    uint64_t total = 0;
    std::vector<uint64_t> off(n);
    for (uint32_t i = 0; i < n; i++) { off[i] = total; total += (sizes[i] + 255) & ~uint64_t(255); }
    uint8_t* d_all = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_all), size_t(total) + 256));
    std::vector<const void*> ptrs(n);
    std::vector<up_item> items;
    for (uint32_t i = 0; i < n; i++) { ptrs[i] = d_all + off[i]; if (sizes[i]) items.push_back({ d_all + off[i], bufs[i], size_t(sizes[i]) }); }
    stager up;
    hipError_t he = upload_side_by_side(up, device, items);
    up.release();
    int rc = he == hipSuccess ? 0 : fail(100, "analysis: %s", hipGetErrorString(he));
    if (!rc && out_md5) rc = rcgpu_md5_device(ptrs.data(), sizes, n, out_md5, nullptr);
    if (!rc && first_nonzero && scanned) {
        // files of one layout and size are scanned in one launch
        std::vector<rcgpu_image_info> info(n);
        std::vector<uint8_t> is_dpx(n, 0);
        for (uint32_t i = 0; i < n; i++) {
            first_nonzero[i] = ~uint64_t(0); scanned[i] = 0;
            is_dpx[i] = sizes[i] >= 4 && rcgpu_dpx_probe(bufs[i], size_t(sizes[i]), &info[i]) == 0 && info[i].data_offset + info[i].data_size <= sizes[i];
        }
        clear_error();
        for (uint32_t i = 0; i < n && !rc; i++) {
            if (!is_dpx[i] || scanned[i]) continue;
            std::vector<const void*> group; std::vector<uint32_t> of;
            for (uint32_t j = i; j < n; j++)
                if (is_dpx[j] && !scanned[j] && info[j].pixfmt == info[i].pixfmt && info[j].width == info[i].width && info[j].height == info[i].height &&
                    info[j].flags == info[i].flags && info[j].data_size == info[i].data_size) {
                    group.push_back(d_all + off[j] + info[j].data_offset); of.push_back(j);
                }
            std::vector<uint64_t> res(group.size());
            pad_scan S;
            if (make_pad_scan(info[i].pixfmt, info[i].width, info[i].height, info[i].flags, &S) || S.total != info[i].data_size) { clear_error(); for (uint32_t j : of) is_dpx[j] = 0; continue; }
            rc = rcgpu_dpx_padding_scan_device(group.data(), uint32_t(group.size()), info[i].pixfmt, info[i].width, info[i].height, info[i].flags, res.data(), nullptr);
            for (size_t k = 0; k < of.size() && !rc; k++) { first_nonzero[of[k]] = res[k]; scanned[of[k]] = 1; }
        }
    }
    (void)hipFree(d_all);
    return rc;
}

extern "C" int rcgpu_md5_host_batch(const uint8_t* const* bufs, const uint64_t* sizes, uint32_t n, uint8_t* out_md5 /* n x 16 */, int device)
{
    clear_error();
    if (!bufs || !sizes || !out_md5 || !n) return fail(1, "md5: null argument");
    return analysis_host_batch(bufs, sizes, n, out_md5, nullptr, nullptr, device);
}

extern "C" int rcgpu_analysis_host_batch(const uint8_t* const* files, const uint64_t* sizes, uint32_t n, uint8_t* out_md5, uint64_t* first_nonzero, uint8_t* scanned, int device)
{
    clear_error();
    if (!files || !sizes || !n || (!out_md5 && !first_nonzero) || (!first_nonzero != !scanned)) return fail(1, "analysis: null argument");
    return analysis_host_batch(files, sizes, n, out_md5, first_nonzero, scanned, device);
}

