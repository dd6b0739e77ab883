// ffv1_gpu.hip -- FFV1 v3 intra encoder for MI355X (gfx950, wave64).  No CPU fallback: without a HIP device
// rcgpu_ffv1_create() fails.
//
// What it replaces: FFmpeg's ffv1enc inside the subprocess RAWcooked launches (CLI/Output.cpp:356).  What it mirrors:
// the in-tree decoder -- ffv1_frame::Process (Lib/CoDec/FFV1/FFV1_Frame.cpp:134-228), slice::Line
// (FFV1_Slice.cpp:447-472), rangecoder::b/s (FFV1_RangeCoder.cpp:71-305) and the packers of
// Lib/Transform/Transform.cpp, all inverted.
//
// Pipeline for a batch of F frames x S slices ("chains" = F*S independent range-coder chains):
//   K1 unpack_px    thread / pixel      payload bytes -> Y,Cb,Cr(,A) values             (inverse of Transform.cpp From()); runs inside
//   K2 k_model      thread / sample     K2, tile by tile through LDS; k_unpack writes whole planes only for the stage tests
//                                       neighbours -> context, folded residual, #decisions; symbols in coding order
//   K3 k_resolve    wave   / slice      adaptive-state resolution: walks the symbols 64 at a time, applies the state
//                                       transitions in coding order (same-context lanes serialised through LDS) and
//                                       emits (state, bit) decisions at 9 bits each, interleaved [piece][lane of chain][64 bytes = 56 decisions]
//   K4 k_rangecode  LANE   / slice      the serial low/range recurrence + carry-resolved byte emission; 64 slices per
//                                       wave advance in lock-step over the interleaved decision stream
//   K5 k_footer     wave   / slice      slice size, error_status, parallel CRC-32 (segment CRCs + GF(2) combine)
//   K6 k_scan       block  / frame      exclusive scan of slice sizes -> packet layout
//   K7 k_gather     blocks / slice      compaction into one contiguous FFV1 packet per frame
// The only serial recurrences are per-context state updates (K3) and low/range (K4); DESIGN.md explains why the
// second one is mapped lane-per-slice rather than wave-per-slice.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>
#include "ffv1_host.h"
#include "rc_common.h"
#include "crc_dev.h"
#include "ffv1_internal.h"

using namespace rc;

namespace {

// ---------------------------------------------------------------------------------------------------------
// Device-visible configuration
// ---------------------------------------------------------------------------------------------------------
struct enc_const {
    uint32_t W, H, line_bytes, pixfmt;
    uint32_t planes, bps, bits, rgb, gb_swap, big_endian, bytes_pp, overflow16;
    uint32_t fields, fill, vflip, altern;  // payload layout of the bit-packed DPX flavors (rc_common.h kFields*), RCGPU_FLAG_*
    uint32_t v1;                           // FFV1 version 1: no slice footer
    uint32_t num_h, num_v, S, nctx, nsets, ec, is5;
    uint32_t samples_per_frame;            // W*H*planes
    uint32_t nseg;                         // segments a slice is cut into for the k_resolve -> k_rangecode hand-over
    uint32_t rc_prio;                      // the whole-slice coder's wave priority (3; the timing build's RCGPU_RC_PRIO measures the others)
    uint32_t overlay;                      // != 0: the slice byte buffers lie inside the slices' own symbol areas (see rcgpu_ffv1::overlay); the value divides
                                           // the per-segment limit (1; rcgpu_ffv1_config::slice_buffer_div in the tests of the overflow report)
    int16_t  q[5][256];
    uint8_t  one_state[256], zero_state[256];
};

struct slice_geom {
    uint32_t x0, y0, w, h;
    uint32_t sym_off;     // first symbol of this slice inside a frame's symbol area
    uint32_t nsamp;       // w*h*planes
    uint32_t hdr_off, hdr_n;   // header decisions in d_hdr
    uint32_t cbuf_off_lo, cbuf_off_hi;   // byte offset of this slice's raw-byte buffer inside a frame's cbuf area
    uint32_t cbuf_cap;
    uint32_t seg_q;       // symbols per segment (multiple of 64)
};

// A piece = 64 bytes of the decision stream = 56 decisions at 9 bits: bytes 0..55 hold t of each decision (state for a coded 1,
// 256 - state for a coded 0), bytes 56..62 the 56 coded bits (decision k = bit k & 7 of byte 56 + k / 8), byte 63 is 0.  (Round 1 spent
// 16 bits per decision: t and c = 0 / 255; c is the coded bit said with eight.)
constexpr int kPieceEntries = 56;                 // decisions per 64-byte piece
constexpr int kPieceBytes = 64;
constexpr int kGroupPieceBytes = 64 * kPieceBytes; // one piece of each of the 64 chains of a group
constexpr int kMaxDecPerSample = 35;              // 2*16+3 for 17-bit residuals
constexpr int kStageEntries = kPieceEntries + 64 * kMaxDecPerSample + 24;   // carry + one chunk (+ slack); a multiple of 16
constexpr int kStageBitDwords = (kStageEntries + 31) / 32 + 7;              // the coded bits of the staged decisions (+ room for the widest OR)
constexpr int kStageMaxPieces = kStageEntries / kPieceEntries + 1;         // pieces one flush can emit (41) + 1
static_assert(kStageBitDwords <= 128, "the bit stage is cleared in two passes of 64 lanes");
static_assert(kStageEntries % 16 == 0 && (kStageBitDwords * 4) % 16 == 0, "LDS areas stay 16-byte aligned");
constexpr int kCandBytes = 512;                  // k_resolve: which lane of the previous chunk may hold a context, by the context's low bits
constexpr int kResolveFixedLds = kStageEntries + kStageBitDwords * 4 + kStageMaxPieces * 16 + 64 * 32 + 512 + 2 * 256 + kCandBytes;   // k_resolve: stage bytes | stage bits | piece tails | slots | transitions | powers | candidates


// (a failed call also leaves the runtime's sticky "last error" behind, which the NEXT user of the runtime in this thread -- torch, say -- would take for
// its own: it is read out here, the error travels in the return code)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); return fail(100, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)

// ---------------------------------------------------------------------------------------------------------
// K1: unpack + forward RCT.  One thread per pixel; 10-bit words and 8/16-bit triplets are read with the widest
// naturally aligned loads the layout allows.  Inverse of JPEG2000RCT (Transform.cpp:29-37) and of the packers.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld16(const uint8_t* p, bool be)
{
    const uint32_t v = *reinterpret_cast<const uint16_t*>(p);
    return be ? ((v >> 8) | ((v & 0xFF) << 8)) : v;
}

// Field `idx` of a word-stream line (rc_common.h kFields*): 12-bit fields fill big-endian words from the LSB up and may
// straddle two words; 10-bit fields sit three to a word.
__device__ __forceinline__ uint32_t ld_field(const uint8_t* line, uint32_t idx, uint32_t fields, uint32_t fill, bool be)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(line);
    if (fields == kFieldsPacked) {
        const uint32_t bit = idx * 12, k = bit >> 5, sh = bit & 31;
        uint32_t v = __builtin_bswap32(w[k]) >> sh;
        if (sh > 20) v |= __builtin_bswap32(w[k + 1]) << (32 - sh);
        return v & 0xFFF;
    }
    const uint32_t k = idx / 3, slot = idx - k * 3;
    uint32_t v = w[k];
    if (be) v = __builtin_bswap32(v);
    return (v >> (fields == kFieldsTop ? 22 - 10 * slot : 10 * slot + fill)) & 0x3FF;
}

// One pixel of the payload as FFV1 plane values: components in file order -> (g/b exchange) -> JPEG 2000 RCT with the offset
// FFV1 adds to Cb, Cr (inverse of Transform.cpp From()).  v[0..planes-1].
__device__ __forceinline__ void unpack_px(const enc_const* __restrict__ C, const uint8_t* __restrict__ frame, uint32_t x, uint32_t y, int32_t (&v)[4], bool file_components = false)
{
    const uint32_t W = C->W, H = C->H;
    // line in the file (Transform.cpp:181-185); k_rawvideo wants the lines as they are stored: `-vf vflip` belongs to the Matroska output only
    const uint32_t fy = (C->vflip && !file_components) ? H - 1 - y : y;
    const uint8_t* p = frame + size_t(fy) * C->line_bytes + size_t(x) * C->bytes_pp;
    const bool be = C->big_endian;
    uint32_t c0, c1 = 0, c2 = 0, c3 = 0;
    if (C->fields == kFieldsExr) {                                   // planar inside the line: B, G, R runs after an 8-byte line header
        const uint16_t* l16 = reinterpret_cast<const uint16_t*>(frame + size_t(fy) * C->line_bytes + 8);
        c2 = l16[x]; c1 = l16[W + x]; c0 = l16[2 * W + x];
    } else if (C->fields != kFieldsBytes) {
        const uint32_t fields = C->fields, fill = C->fill, np = C->planes;
        const uint8_t* line = frame + (C->altern ? size_t(0) : size_t(fy) * C->line_bytes);
        const uint32_t i0 = C->altern ? fy * W + x : x * np;
        c0 = ld_field(line, i0, fields, fill, be);
        if (np > 1) { c1 = ld_field(line, i0 + 1, fields, fill, be); c2 = ld_field(line, i0 + 2, fields, fill, be); }
        if (np > 3) c3 = ld_field(line, i0 + 3, fields, fill, be);
    } else
    switch (C->pixfmt) {
    case RCGPU_PIX_RGB8:  c0 = p[0]; c1 = p[1]; c2 = p[2]; break;
    case RCGPU_PIX_RGBA8: { const uint32_t w = *reinterpret_cast<const uint32_t*>(p); c0 = w & 0xFF; c1 = (w >> 8) & 0xFF; c2 = (w >> 16) & 0xFF; c3 = w >> 24; break; }
    case RCGPU_PIX_RGB10_FILLEDA_BE: case RCGPU_PIX_RGB10_FILLEDA_LE: {
        uint32_t w = *reinterpret_cast<const uint32_t*>(p);
        if (be) w = __builtin_bswap32(w);
        c0 = (w >> 22) & 0x3FF; c1 = (w >> 12) & 0x3FF; c2 = (w >> 2) & 0x3FF; break; }
    case RCGPU_PIX_RGB12_FILLEDA_BE: case RCGPU_PIX_RGB12_FILLEDA_LE:
        c0 = ld16(p, be) >> 4; c1 = ld16(p + 2, be) >> 4; c2 = ld16(p + 4, be) >> 4; break;
    case RCGPU_PIX_RGB16_BE: case RCGPU_PIX_RGB16_LE:
        c0 = ld16(p, be); c1 = ld16(p + 2, be); c2 = ld16(p + 4, be); break;
    case RCGPU_PIX_RGBA16_BE: case RCGPU_PIX_RGBA16_LE:
        c0 = ld16(p, be); c1 = ld16(p + 2, be); c2 = ld16(p + 4, be); c3 = ld16(p + 6, be); break;
    case RCGPU_PIX_RGBA12_FILLEDA_BE: case RCGPU_PIX_RGBA12_FILLEDA_LE:
        c0 = ld16(p, be) >> 4; c1 = ld16(p + 2, be) >> 4; c2 = ld16(p + 4, be) >> 4; c3 = ld16(p + 6, be) >> 4; break;
    case RCGPU_PIX_Y8: c0 = p[0]; break;
    default: c0 = ld16(p, be); break;
    }
    v[0] = int32_t(c0); v[1] = v[2] = 0; v[3] = int32_t(c3);
    if (file_components) { v[1] = int32_t(c1); v[2] = int32_t(c2); return; }      // k_rawvideo: the components as the file orders them
    if (!C->rgb) return;
    int32_t r = int32_t(c0), g = int32_t(c1), b = int32_t(c2);
    if (C->gb_swap) { const int32_t t = g; g = b; b = t; }
    b -= g; r -= g;
    g += (b + r) >> 2;
    const int32_t off = int32_t(1) << C->bps;
    v[0] = g; v[1] = b + off; v[2] = r + off;
}

// K1 as a kernel of its own: int32 planes of a sub-batch, for the stage tests (rcgpu_ffv1_debug_fetch(0)).  The encoder itself
// unpacks inside k_model.
__global__ __launch_bounds__(256) void k_unpack(const enc_const* __restrict__ C, const uint8_t* const* __restrict__ frames,
                                                int32_t* __restrict__ planes, uint32_t frame0)
{
    const uint32_t W = C->W, H = C->H;
    const uint32_t pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= W * H) return;
    const uint32_t f = blockIdx.y;           // frame inside this sub-batch; planes hold one sub-batch
    const uint32_t y = pix / W, x = pix - y * W;
    int32_t v[4];
    unpack_px(C, frames[frame0 + f], x, y, v);
    const size_t plane_sz = size_t(W) * H;
    int32_t* dst = planes + size_t(f) * C->planes * plane_sz + pix;
    for (uint32_t p = 0; p < C->planes; p++) dst[p * plane_sz] = v[p];
}

// The frame as the bytes FFmpeg's `-f framemd5` output hashes (CLI/Output.cpp:312-332): the picture its dpx/tiff decoder hands on,
// written by the rawvideo encoder without line padding [ffmpeg-knowledge].  8 bit: rgb24 / rgba / gray, bytes in file order;
// 16 bit: rgb48 / rgba64 / gray16 in the FILE's endianness; 10 and 12 bit: planar little-endian 16-bit words, planes G, B, R(, A)
// (gbrp10le, gbrap12le ...) or the one plane of gray10le / gray12le.  One thread per pixel.  Lines in FILE order: the reference puts `-vf vflip`
// (DPX stored bottom-up) in front of `-f matroska <out>`, and FFmpeg's output options belong to the output that follows them -- the framemd5
// output behind it sees the decoder's frames as they are [ffmpeg-knowledge].
__global__ __launch_bounds__(256) void k_rawvideo(const enc_const* __restrict__ C, const uint8_t* const* __restrict__ frames, uint8_t* __restrict__ out,
                                                  size_t out_stride)
{
    const uint32_t W = C->W, H = C->H;
    const uint32_t pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= W * H) return;
    const uint32_t y = pix / W, x = pix - y * W, np = C->planes;
    int32_t v[4];
    unpack_px(C, frames[blockIdx.y], x, y, v, true);
    uint8_t* dst = out + size_t(blockIdx.y) * out_stride;
    if (C->bps == 8) {
        for (uint32_t p = 0; p < np; p++) dst[size_t(pix) * np + p] = uint8_t(v[p]);
    } else if (C->bps == 16) {
        uint16_t* d16 = reinterpret_cast<uint16_t*>(dst) + size_t(pix) * np;
        for (uint32_t p = 0; p < np; p++) { const uint32_t c = uint32_t(v[p]); d16[p] = uint16_t(C->big_endian ? ((c >> 8) | (c << 8)) : c); }
    } else {
        uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
        const size_t plane = size_t(W) * H;
        if (np == 1) d16[pix] = uint16_t(v[0]);
        else {
            d16[pix] = uint16_t(v[1]); d16[plane + pix] = uint16_t(v[2]); d16[2 * plane + pix] = uint16_t(v[0]);
            if (np > 3) d16[3 * plane + pix] = uint16_t(v[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: context model.  For every sample: 5-tap context (get_context_5, FFV1_Slice.cpp:80-93), median prediction
// (:21-66), residual folded to `bits` (:469 inverse), and the number of binary decisions its symbol will take.
// Output symbol = set << 30 | |ctx| << 17 | (residual & 0x1FFFF), stored in CODING order of the slice
// (line-interleaved planes, SliceContent_LineThenPlane :427-441).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t median3(int32_t a, int32_t b, int32_t c)
{
    return max(min(a, b), min(max(a, b), c));
}

// ---------------------------------------------------------------------------------------------------------
// K1+K2 in one kernel: a block unpacks a tile of a slice -- kTileR picture rows x kTileW columns, plus the two rows above and the
// columns the neighbour rules reach -- from the payload straight into LDS and models it from there, so the int32 planes never
// exist in HBM (53 MB in + 106 MB out per 4K RGB16 frame instead of 371 MB through planes).
// grid = (max tiles of a slice, chains)
// ---------------------------------------------------------------------------------------------------------
constexpr int kTileW = 256, kTileR = 4, kTileCols = kTileW + 3, kTileRows = kTileR + 2;
__global__ __launch_bounds__(256) void k_model(const enc_const* __restrict__ C, const slice_geom* __restrict__ geom,
                                                     const uint8_t* const* __restrict__ frames, uint32_t* __restrict__ sym,
                                                     unsigned long long* __restrict__ chain_ndec)
{
    extern __shared__ int32_t tile[];                    // [planes][kTileRows][kTileCols]: rows y0-2 .. y0+R-1, columns x0-2 .. x0+TW
    __shared__ int16_t q[5][256];
    __shared__ unsigned long long segsum[64];
    const uint32_t S = C->S, f = blockIdx.y / S, s = blockIdx.y - f * S, chain = blockIdx.y;
    const slice_geom G = geom[s];
    const uint32_t tiles_x = (G.w + kTileW - 1) / kTileW, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = int(ty) * kTileR, x0 = int(tx) * kTileW;
    if (y0 >= int(G.h)) return;                          // whole block: this slice has fewer tiles than the largest one
    for (uint32_t i = threadIdx.x; i < 5 * 256; i += 256) (&q[0][0])[i] = (&C->q[0][0])[i];
    if (threadIdx.x < 64) segsum[threadIdx.x] = 0;
    const uint32_t np = C->planes;
    const uint8_t* frame = frames[f];
    for (int i = threadIdx.x; i < kTileRows * kTileCols; i += 256) {
        const int r = i / kTileCols, c = i - r * kTileCols, yy = y0 + r - 2, xx = x0 + c - 2;
        if (yy >= 0 && yy < int(G.h) && xx >= 0 && xx < int(G.w)) {
            int32_t v[4];
            unpack_px(C, frame, G.x0 + uint32_t(xx), G.y0 + uint32_t(yy), v);
            for (uint32_t p = 0; p < np; p++) tile[(p * kTileRows + r) * kTileCols + c] = v[p];
        } else
            for (uint32_t p = 0; p < np; p++) tile[(p * kTileRows + r) * kTileCols + c] = 0;     // above the slice: the decoder's zeros
    }
    __syncthreads();
    const int bits = int(C->bits);
    const bool is5 = C->is5, ov16 = C->overflow16, rgb = C->rgb;
    const uint32_t nseg_c = C->nseg;
    unsigned long long local = 0;
    uint32_t cur_seg = 0xFFFFFFFFu, seg_end = 0;
    uint32_t* out = sym + size_t(f) * C->samples_per_frame + G.sym_off;
    const int xi = x0 + int(threadIdx.x), wl = int(G.w) - 1;
    const int rows = min(kTileR, int(G.h) - y0);
    // neighbours with the decoder's edge rules (FFV1_Slice.cpp:386-387 of SURVEY appendix A; :432-433) as per-thread offsets inside the
    // tile, fixed once: left of the slice L is the sample above, LT the one two above, LL the one above-left (or 0 in column 0);
    // right of it RT is T; rows above the slice are zeros in the tile
    const int oL  = xi > 0 ? -1 : -kTileCols;
    const int oLT = xi > 0 ? -kTileCols - 1 : -2 * kTileCols;
    const int oRT = xi < wl ? -kTileCols + 1 : -kTileCols;
    const int oLL = xi > 1 ? -2 : -kTileCols - 1;
    const int32_t mLL = xi > 0 ? -1 : 0;
    if (xi <= wl)
    for (int r = 0; r < rows; r++) {
        const int yi = y0 + r;
        for (uint32_t p = 0; p < np; p++) {
            const int32_t* at = tile + (p * kTileRows + r + 2) * kTileCols + threadIdx.x + 2;
            const uint32_t set = rgb ? (p + 1) >> 1 : 0;
            const int32_t cur = at[0];
            const int32_t T  = at[-kTileCols];
            const int32_t L  = at[oL];
            const int32_t LT = at[oLT];
            const int32_t RT = at[oRT];
            int32_t ctx = q[0][(L - LT) & 0xFF] + q[1][(LT - T) & 0xFF] + q[2][(T - RT) & 0xFF];
            if (is5) {
                const int32_t LL = at[oLL] & mLL;
                const int32_t TT = at[-2 * kTileCols];
                ctx += q[3][(LL - L) & 0xFF] + q[4][(TT - T) & 0xFF];
            }
            int32_t pred;
            if (ov16) pred = median3(int16_t(L), int16_t(L) + int16_t(T) - int16_t(LT), int16_t(T));   // FFV1_Slice.cpp:52-57
            else pred = median3(L, L + T - LT, T);
            int32_t d = cur - pred;
            if (ctx < 0) { ctx = -ctx; d = -d; }
            d = int32_t(uint32_t(d) << (32 - bits)) >> (32 - bits);                  // fold: sign-extend to `bits`
            const uint32_t a = uint32_t(d < 0 ? -d : d);
            const uint32_t idx = (uint32_t(yi) * np + p) * G.w + uint32_t(xi);
            if (idx >= seg_end) {                 // a thread's symbol index only grows: one division at its first symbol, then increments
                if (local) atomicAdd(&segsum[cur_seg], local);
                local = 0;
                if (cur_seg == 0xFFFFFFFFu) { cur_seg = idx / G.seg_q; seg_end = (cur_seg + 1) * G.seg_q; }
                else do { cur_seg++; seg_end += G.seg_q; } while (idx >= seg_end);
            }
            local += a ? uint32_t(2 * (31 - __clz(int(a))) + 3) : 1u;
            out[size_t(idx)] = (set << 30) | (uint32_t(ctx) << 17) | (uint32_t(d) & 0x1FFFFu);
        }
    }
    if (local) atomicAdd(&segsum[cur_seg], local);
    __syncthreads();
    if (threadIdx.x < nseg_c && segsum[threadIdx.x]) atomicAdd(&chain_ndec[size_t(chain) * nseg_c + threadIdx.x], segsum[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------
// K3: adaptive-state resolution, one wavefront per slice.
// ---------------------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes with DPP moves (no LDS crossbar round trips): shifts inside each row of 16, then the
// row totals are broadcast into the following rows (row_bcast:15 for rows 1 and 3, row_bcast:31 for rows 2 and 3).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int)
{
    int x = int(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return uint32_t(x);
}

// OR over the 64 lanes, same DPP ladder; the result is uniform (read from lane 63).
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    int x = int(v);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return uint32_t(__builtin_amdgcn_readlane(x, 63));
}

// Which of a context's 32 states does a residual touch exactly once?  Row r = 0 for a zero residual, else exponent + 1; the row
// holds byte masks over the state registers S[0], S[1], S[2], S[5], S[6], S[7] (state k = byte k & 3 of S[k >> 2]): state 0 always,
// 1 + t for exponent slots t <= e (t < 9), 22 + t for mantissa bits t < e (t < 9).  States 10 and 31 (repeated) and the sign
// states are applied separately.
struct act_masks { uint32_t m[18][8]; };
constexpr act_masks make_act_masks()
{
    act_masks t{};
    for (int r = 0; r < 18; r++)
        for (int k = 0; k < 32; k++) {
            const int e = r - 1;
            const bool act = k == 0 || (r > 0 && ((k >= 1 && k <= 9 && k - 1 <= e) || (k >= 22 && k <= 30 && k - 22 < e)));
            const int j = k >> 2, col = j < 3 ? j : j - 2;                 // S[5..7] -> columns 3..5
            if (act) t.m[r][col] |= 0xFFu << (8 * (k & 3));
        }
    return t;
}
__device__ const act_masks kActMasks = make_act_masks();

// A k_resolve workgroup is ONE wavefront: its LDS operations execute in issue order, so phases that hand data from lane to lane
// through LDS need no barrier and, above all, no s_waitcnt vmcnt(0) -- which __syncthreads() implies and which would put the HBM
// latency of every prefetch and store on the critical path.  Only the compiler must keep the order.
#define WAVE_SYNC() asm volatile("" ::: "memory")
// Timing build only (make timing PROF=1: -DRCGPU_TIMING_BUILD -DRCGPU_TIMING_PROF, tools/prof_resolve.py): the shader clock around the phases
// of a chunk, summed over all wavefronts.  The shipped library holds none of it.
#if defined(RCGPU_TIMING_BUILD) && defined(RCGPU_TIMING_PROF)
#define RCGPU_PROF 1
__device__ unsigned long long g_prof[16];
#define PROF_T(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); prof[i] += now_ - tlast; tlast = now_; }
#else
#define PROF_T(i)
#endif
#define OPI(k) (k)
template <bool LDS_STATES>
__global__ __launch_bounds__(64) __attribute__((aligned(4096))) void k_resolve(const enc_const* __restrict__ C, const slice_geom* __restrict__ geom,
                                                const uint16_t* __restrict__ hdr, const uint32_t* __restrict__ sym,
                                                uint8_t* __restrict__ states, const unsigned long long* __restrict__ group_off,
                                                uint8_t* __restrict__ stream, uint32_t nkeys, uint32_t seg,
                                                uint8_t* __restrict__ resume, uint32_t resume_stride, uint32_t prio)
{
    // Issue priority against the range coder's wavefronts on the same SIMD (the split coder's spans run at 0, its serial pass at 3).
    if ((prio & 0xFF) == 1) __builtin_amdgcn_s_setprio(1); else if ((prio & 0xFF) == 2) __builtin_amdgcn_s_setprio(2);
    // One launch handles segment `seg` of every slice.  What must survive between launches -- the < 56 decisions that
    // did not fill a piece -- lives in `resume` (per chain: count, entries, their bits).  The context states of a batch are preset
    // to 128 in HBM by the host (states_coded = 0): 20 MB per 4K frame, against 3 GB of traffic the kernel itself causes.
    // The arrays of fixed size are static LDS: their addresses are compile-time constants that fold into the instructions'
    // offset fields.  What scales with the number of contexts follows as dynamic LDS.
    __shared__ __attribute__((aligned(16))) uint8_t fixed[kResolveFixedLds];
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t*  stage = fixed;                                                       // kStageEntries t-bytes
    uint32_t* sbits = reinterpret_cast<uint32_t*>(fixed + kStageEntries);          // their coded bits, decision i = bit i & 31 of dword i >> 5; zero beyond stage_count
    uint4*    ptail = reinterpret_cast<uint4*>(fixed + kStageEntries + kStageBitDwords * 4);   // last quarter of every piece about to leave: t 48..55 | 56 bits | 0
    uint8_t*  slot = fixed + kStageEntries + kStageBitDwords * 4 + kStageMaxPieces * 16;       // 64 x 32
    uint8_t*  trans = slot + 64 * 32;                                               // [256 +- state], see below
    uint8_t*  pw = trans + 512;                                                     // [2][256]: one_state applied 4 and 16 times
    uint8_t*  cand = pw + 512;                                                      // kCandBytes
    // Two bitmaps over the contexts, used by even and odd chunks in turn: a chunk marks its contexts in one (the atomic's return value
    // tells every lane but one of a context that it shares it), finds in the other what the chunk before it used, and clears that one.
    const uint32_t bmw = ((nkeys + 31) / 32 + 3) / 4 * 4;
    uint32_t* cbm = reinterpret_cast<uint32_t*>(smem);                              // 2 x bmw dwords
    // LDS_STATES (compact context model): every context's 32 states live here for the whole slice -- no HBM traffic per sample
    uint8_t*  lstates = reinterpret_cast<uint8_t*>(cbm + 2 * bmw);                  // nkeys x 32

    const int lane = threadIdx.x;
    const uint32_t chain = blockIdx.x;
    const uint32_t S = C->S, f = chain / S, s = chain - f * S;
    const slice_geom G = geom[s];
    const uint32_t nctx = C->nctx;
    const uint32_t* in = sym + size_t(f) * C->samples_per_frame + G.sym_off;
#ifdef RCGPU_TIMING_BUILD
    // RCGPU_EXP_STATES_L2 (timing build only; wrong bytes, right work): every state record this wavefront gathers and writes back lies in
    // a 1 MB region that stays in the L2 -- the kernel with everything but the HBM latency of its state traffic: its floor
    const bool l2res = (prio & 0x100) != 0;
    uint8_t* st_base = l2res ? states + size_t(chain & 1023) * 1024 : states + size_t(chain) * nkeys * 32;
#define ST_KEY(k) (l2res ? ((k) & 31u) : (k))
#else
    uint8_t* st_base = states + size_t(chain) * nkeys * 32;
#define ST_KEY(k) (k)
#endif
    // interleaved decision stream: piece k of this chain lives at group base + (k*64 + lane_of_chain)*64
    uint32_t* out32 = reinterpret_cast<uint32_t*>(stream + group_off[chain >> 6]) + (chain & 63) * 16;

    // transition table indexed by the signed decision: trans[256 + s] = next state after coding bit 1 in state s,
    // trans[256 - s] = after coding bit 0 (states are 1..255)
    for (int i = lane; i < 256; i += 64) { trans[(256 - i) & 255] = C->zero_state[i]; trans[256 + i] = C->one_state[i]; }
    // powers of the "coded a 1" transition, for runs of zero residuals in one context (see below)
    WAVE_SYNC();
    for (int i = lane; i < 256; i += 64) { const uint8_t* o = trans + 256; pw[i] = o[o[o[o[i]]]]; }
    WAVE_SYNC();
    for (int i = lane; i < 256; i += 64) pw[256 + i] = pw[pw[pw[pw[i]]]];
    uint8_t* rs = resume + size_t(chain) * resume_stride;
    uint4* rs_states = reinterpret_cast<uint4*>(rs + 80);             // LDS_STATES: the state table itself is parked here
    uint32_t stage_count;
    if (seg == 0) {
        if (LDS_STATES) for (uint32_t i = lane; i < nkeys * 2; i += 64) reinterpret_cast<uint4*>(lstates)[i] = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
        for (uint32_t i = lane; i < uint32_t(kStageBitDwords); i += 64) sbits[i] = 0;
        WAVE_SYNC();
        for (uint32_t i = lane; i < G.hdr_n; i += 64) {                 // header decisions from the host: t | coded bit << 8
            const uint32_t hd = hdr[G.hdr_off + i];
            stage[i] = uint8_t(hd);
            if (hd >> 8) atomicOr(&sbits[i >> 5], 1u << (i & 31));
        }
        stage_count = G.hdr_n;
    } else {
        if (LDS_STATES) for (uint32_t i = lane; i < nkeys * 2; i += 64) reinterpret_cast<uint4*>(lstates)[i] = rs_states[i];
        for (uint32_t i = lane; i < uint32_t(kStageBitDwords); i += 64) sbits[i] = i < 2 ? reinterpret_cast<const uint32_t*>(rs + 72)[i] : 0u;
        if (lane < 14) reinterpret_cast<uint32_t*>(stage)[lane] = reinterpret_cast<const uint32_t*>(rs + 16)[lane];
        stage_count = uint32_t(__builtin_amdgcn_readfirstlane(int(*reinterpret_cast<const uint32_t*>(rs))));
    }
    for (uint32_t i = lane; i < 2 * bmw; i += 64) cbm[i] = 0;
    uint32_t piece_base = 0;              // piece index inside this segment's window
    const uint32_t sym_begin = min(G.nsamp, seg * G.seg_q), sym_end = min(G.nsamp, (seg + 1) * G.seg_q);
    const bool last_seg = seg + 1 == C->nseg;
    WAVE_SYNC();

    // bits [7 * byte .. 7 * byte + 56) of the bit stage as two dwords (the second one's top byte is 0): byte-aligned, not dword-aligned
    auto bits56 = [&](uint32_t byte) -> uint2 {
        const uint32_t w = byte >> 2, sh = (byte & 3) * 8;
        const uint32_t a = sbits[w], b = sbits[w + 1], c = sbits[w + 2];
        return make_uint2(__builtin_amdgcn_alignbit(b, a, sh), __builtin_amdgcn_alignbit(c, b, sh) & 0x00FFFFFFu);     // a shift of 0 hands back the low dword
    };
    // Full pieces leave for HBM, what did not fill one moves to the front.  Every loop here is written out with its trip structure (a
    // piece index per lane, at most three passes): left to itself the compiler unrolls such loops eightfold with 64-bit indices and
    // splits the 16-byte stores -- three hundred instructions per chunk for forty stores.
    uint8_t* const out_bytes = reinterpret_cast<uint8_t*>(out32);
    auto flush_full = [&]() {
        WAVE_SYNC();
        const uint32_t np = stage_count / kPieceEntries;                  // uniform, < kStageMaxPieces
        if (np) {
            if (uint32_t(lane) < np) {                                    // lane p puts piece p's last quarter together: t 48..55, then its 56 bits
                const uint2 t = *reinterpret_cast<const uint2*>(stage + lane * kPieceEntries + 48), b = bits56(uint32_t(lane) * 7);
                ptail[lane] = make_uint4(t.x, t.y, b.x, b.y);
            }
            WAVE_SYNC();
            // 16 bytes per lane, four lanes cover one 64-byte piece, sixteen pieces per pass
            const uint32_t q = lane & 3;
            uint32_t pc = uint32_t(lane) >> 2;
            const uint8_t* lsrc = q < 3 ? stage + pc * kPieceEntries + q * 16                                  // 8-byte aligned: 56 = 7 x 8
                                        : reinterpret_cast<const uint8_t*>(ptail + pc);
            const uint32_t lstep = q < 3 ? 16u * kPieceEntries : 16u * 16u;
            size_t goff = size_t(piece_base + pc) * kGroupPieceBytes + q * 16;
#pragma clang loop unroll(disable) vectorize(disable)
            for (; pc < np; pc += 16, lsrc += lstep, goff += 16u * size_t(kGroupPieceBytes)) {
                const uint2 t0 = reinterpret_cast<const uint2*>(lsrc)[0], t1 = reinterpret_cast<const uint2*>(lsrc)[1];
                *reinterpret_cast<uint4*>(out_bytes + goff) = make_uint4(t0.x, t0.y, t1.x, t1.y);
            }
            // what did not fill a piece (< 56 decisions: 14 dwords of t, 7 bytes of bits) moves to the front; the bit stage is zero behind
            // the staged decisions at all times, so the 56 bits read here need no mask
            uint32_t keep = 0;
            if (lane < 14) keep = reinterpret_cast<const uint32_t*>(stage)[np * 14 + lane];
            const uint2 kb = bits56(np * 7);
            const uint32_t ndw = (stage_count + 31) / 32 + 1;             // bit-stage dwords that may hold a bit: at most kStageBitDwords - 6
            WAVE_SYNC();
            if (lane < 14) reinterpret_cast<uint32_t*>(stage)[lane] = keep;
            if (uint32_t(lane) < ndw) sbits[lane] = lane == 0 ? kb.x : lane == 1 ? kb.y : 0u;
            if (uint32_t(lane) + 64 < ndw) sbits[lane + 64] = 0u;
            piece_base += np;
            stage_count -= np * kPieceEntries;
        }
        WAVE_SYNC();
    };
    flush_full();

    const unsigned long long lane_bit = 1ull << lane;
#ifdef RCGPU_PROF
    unsigned long long prof[12] = {}; unsigned long long tlast = __builtin_readcyclecounter();
#endif
    // Software pipeline over chunks of 64 symbols.  While chunk k is binarised, the symbols of chunk k+2 and the context states of
    // chunk k+1 are already on their way from HBM, so neither latency sits on the chunk's critical path:
    //   sv_cur / sv_nxt   symbols of chunk k (in registers) and k+1 (loaded one chunk ago)
    //   P0, P1            every lane's own context states for chunk k, loaded one chunk ago -- possibly stale if chunk k-1 updated
    //                     that context, in which case the states are forwarded from chunk k-1's LDS slot instead
    auto key_of = [&](uint32_t v) { return (v >> 30) * nctx + ((v >> 17) & 0x1FFF); };
    uint32_t sv_cur = sym_begin + lane < sym_end ? in[sym_begin + lane] : 0;
    uint32_t sv_nxt = sym_begin + 64 + lane < sym_end ? in[sym_begin + 64 + lane] : 0;
    uint4 P0 = make_uint4(0, 0, 0, 0), P1 = P0;
    if (!LDS_STATES && sym_begin + lane < sym_end) {
        const uint4* gp = reinterpret_cast<const uint4*>(st_base + size_t(ST_KEY(key_of(sv_cur))) * 32);
        P0 = gp[0]; P1 = gp[1];
    }
    uint32_t ppack = 0xFFFFu;             // previous chunk: this lane's key (keys stay below 2^14; 0xFFFF = none yet) | its group's slot << 16
    for (uint32_t base = sym_begin; base < sym_end; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < sym_end;
        const uint32_t sv = sv_cur;
        const int32_t d = int32_t(sv << 15) >> 15;                    // 17-bit signed residual
        const uint32_t key = (sv >> 30) * nctx + ((sv >> 17) & 0x1FFF);
        const uint32_t a = uint32_t(d < 0 ? -d : d);
        const int e = a ? 31 - __clz(int(a)) : 0;
        const uint32_t ndec = valid ? (a ? uint32_t(2 * e + 3) : 1u) : 0u;
        const uint32_t amax = wave_or(valid ? a : 0u);
        const int emax = amax ? 31 - __clz(int(amax)) : -1;           // uniform: no lane of this chunk has a larger exponent
        int sg_e[9], sg_m[9];                                          // per slot: +1 codes a 1, -1 codes a 0
#pragma unroll
        for (int t = 0; t < 9; t++) { sg_e[t] = t < e ? 1 : -1; sg_m[t] = (a >> t) & 1u ? 1 : -1; }
        const uint4* mrow = reinterpret_cast<const uint4*>(kActMasks.m[a ? e + 1 : 0]);
        const uint4 Ma = mrow[0];                                      // S[0], S[1], S[2], S[5]
        const uint2 Mb = *reinterpret_cast<const uint2*>(mrow + 1);    // S[6], S[7]
        const uint32_t incl = wave_incl_scan(ndec, lane);
        const uint32_t excl = incl - ndec;
        const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));        // a scalar: stage_count and what the flush derives from it stay in SGPRs
        PROF_T(0)

        // --- which lanes share a context?  Marking the contexts in this chunk's bitmap tells every lane of a context but one that it is
        // not alone; a scalar loop over the distinct shared contexts then gives every lane its predecessor, its group leader and whether it
        // is the last.  Did the previous chunk use this lane's context?  Its bitmap says so exactly; the candidate table, indexed by the
        // context's low bits, names the lane that most probably did (checked against that lane's key), and the few lanes whose candidate
        // was overwritten by another context find theirs by ballot.
        const uint32_t kbit = 1u << (key & 31), kw = key >> 5;
        uint32_t* const cur_bm = cbm + ((base >> 6) & 1 ? bmw : 0u);
        uint32_t* const prv_bm = cbm + ((base >> 6) & 1 ? 0u : bmw);
        const bool pbit = !LDS_STATES && valid && (prv_bm[kw] & kbit);
        const uint32_t hp = valid ? uint32_t(cand[key & (kCandBytes - 1)]) & 63u : 0u;
        const uint32_t hp_pack = uint32_t(__shfl(int(ppack), int(hp)));    // all lanes take part: the source lane may be past the end of this chunk
        bool fwd = pbit && (hp_pack & 0xFFFFu) == key;
        int fwd_slot = int(hp_pack >> 16);
        WAVE_SYNC();
        if (valid) cand[key & (kCandBytes - 1)] = uint8_t(lane);
        const uint32_t old = valid ? atomicOr(&cur_bm[kw], kbit) : 0u;
        if ((ppack & 0xFFFFu) != 0xFFFFu) atomicAnd(&prv_bm[(ppack & 0xFFFFu) >> 5], ~(1u << (ppack & 31)));     // the next chunk's bitmap is clean again
        WAVE_SYNC();
        for (unsigned long long fm = __ballot(pbit && !fwd); fm; ) {
            const int src = __ffsll((long long)fm) - 1;
            const uint32_t k = uint32_t(__builtin_amdgcn_readlane(int(key), src));
            const unsigned long long pm = __ballot((ppack & 0xFFFFu) == k);          // not empty: the bitmap said so
            const uint32_t pslot = uint32_t(__builtin_amdgcn_readlane(int(ppack), __ffsll((long long)pm) - 1)) >> 16;
            const bool mine = valid && key == k;
            if (mine) { fwd = true; fwd_slot = int(pslot); }
            fm &= ~__ballot(mine);
        }
        int leader = lane, pred = -1;
        bool last = true, zrun = false;
        uint32_t rank = 0;
        unsigned long long lm = __ballot(valid && (old & kbit));
        while (lm) {
            const int src = __ffsll((long long)lm) - 1;
            const uint32_t k = uint32_t(__builtin_amdgcn_readlane(int(key), src));      // src is uniform: no trip through the LDS crossbar
            const bool mine = valid && key == k;
            const unsigned long long m = __ballot(mine);
            const bool allzero = __ballot(mine && a == 0) == m;        // a run of zero residuals: only state 0 moves, always by "coded a 1"
            if (mine) {
                leader = __ffsll((long long)m) - 1;
                const unsigned long long lower = m & (lane_bit - 1);
                pred = lower ? 63 - __clzll((long long)lower) : -1;
                last = ((m >> lane) >> 1) == 0;
                zrun = allzero;
                rank = uint32_t(__popcll(lower));
            }
            lm &= ~m;
        }

        PROF_T(1)
        // --- group leaders install the context's 32 states in their LDS slot: all 128 on first use in this slice (states_coded = 0),
        // the previous chunk's result if it used the same context, else what was prefetched from the slice's state array in HBM.
        // Everything this wavefront has in flight is at least most of a chunk old: the wait is (almost) free, and it also makes
        // the write-backs of the previous chunk visible before the next prefetch is issued.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PROF_T(2)
        const bool first = !LDS_STATES && valid && pred < 0;
        uint4 s0 = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u), s1 = s0;
        if (first) {
            if (fwd) { const uint4* fp = reinterpret_cast<const uint4*>(slot + fwd_slot * 32); s0 = fp[0]; s1 = fp[1]; }
            else { s0 = P0; s1 = P1; }
        }
        WAVE_SYNC();                      // forwarded reads come before this chunk's leaders overwrite the slots
        if (first) {
            uint4* sp = reinterpret_cast<uint4*>(slot + lane * 32);
            sp[0] = s0; sp[1] = s1;
        }
        // --- next chunk's states and the symbols after it leave for HBM now
        sv_cur = sv_nxt;
        if (!LDS_STATES && i + 64 < sym_end) {
            const uint4* gp = reinterpret_cast<const uint4*>(st_base + size_t(ST_KEY(key_of(sv_nxt))) * 32);
            P0 = gp[0]; P1 = gp[1];
        }
        sv_nxt = i + 128 < sym_end ? in[i + 128] : 0;
        WAVE_SYNC();

        PROF_T(3)
        // --- rounds: a lane runs once its predecessor (same context, earlier in coding order) is done.
        unsigned long long done = 0;
        bool pending = valid;
        uint8_t* sl = LDS_STATES ? lstates + size_t(key) * 32 : slot + leader * 32;
        uint8_t* op = stage + stage_count + excl;
        {   // the coded bits of this lane's decisions do not depend on any state: zero flag | e ones, a zero | mantissa from the top | sign.
            // 2e + 3 bits: one dword unless some lane of the chunk has e >= 15 (emax is uniform)
            const uint32_t o = stage_count + excl, w = o >> 5, sh = o & 31;
            if (emax < 15) {
                const uint32_t mant = a & ((1u << e) - 1), rev = __brev(mant) >> ((32 - e) & 31);            // mantissa bit e-1 first (e == 0: mant == 0)
                uint32_t b = a ? ((((1u << e) - 1) << 1) | (rev << (e + 2)) | (uint32_t(d < 0) << (2 * e + 2))) : 1u;
                b = valid ? b : 0u;
                atomicOr(&sbits[w], b << sh);
                atomicOr(&sbits[w + 1], (b >> 1) >> (31 - sh));                                               // b >> (32 - sh), also right for sh == 0
            } else {
                uint32_t blo = 1, bhi = 0;
                if (a) {
                    const uint32_t mant = a & ((1u << e) - 1), rev = e ? __brev(mant) >> (32 - e) : 0u;
                    const unsigned long long v = ((unsigned long long)(((1u << e) - 1) << 1)) | ((unsigned long long)rev << (e + 2)) | ((unsigned long long)(d < 0) << (2 * e + 2));
                    blo = uint32_t(v); bhi = uint32_t(v >> 32);
                }
                if (!valid) blo = bhi = 0;
                const unsigned long long v = ((unsigned long long)bhi << 32) | blo;
                atomicOr(&sbits[w], blo << sh);
                atomicOr(&sbits[w + 1], uint32_t((v >> 1) >> (31 - sh)));
                atomicOr(&sbits[w + 2], (bhi >> 1) >> (31 - sh));
            }
        }
        PROF_T(4)
        // Lanes of one context that all carry a zero residual (flat picture areas, letterbox bars) need no rounds: the r-th of them
        // sees state 0 after r "coded a 1" transitions, which the power tables give in a few look-ups.
        if (__ballot(zrun)) {
            if (zrun) {
                uint32_t st = sl[0];
                for (uint32_t r = rank & 3; r; r--) st = trans[256 + st];          // rank = 16 a + 4 b + c: at most 3 + 3 + 3 look-ups
                for (uint32_t r = (rank >> 2) & 3; r; r--) st = pw[st];
                for (uint32_t r = rank >> 4; r; r--) st = pw[256 + st];
                op[OPI(0)] = uint8_t(st);                                   // a coded 1: t = state
                if (last) sl[0] = trans[256 + st];
            }
            done |= __ballot(zrun);
            pending = pending && !zrun;
            WAVE_SYNC();
        }
        while (__ballot(pending)) {
            const bool ready = pending && (pred < 0 || ((done >> pred) & 1));
            if (ready) {
                // Symbol binarisation, inverse of rangecoder::s (FFV1_RangeCoder.cpp:135-305).  The context's 32 states sit in 8
                // registers and every state index below is a compile-time constant.  A symbol touches each state at most once,
                // except state 10 (exponent bits 9 and up) and state 31 (mantissa bits 9 and up): so the transition look-ups of
                // all other decisions are independent.  Phase 1 issues them back to back (nothing waits on LDS), phase 2 applies
                // them; only the two short chains are walked with dependent LDS round trips.
                // Slots are grouped by the register their states live in, and the body exists once per LEVEL = how many groups
                // some lane of the chunk reaches (emax is uniform): 0 only the zero flag, 1 + S[0] and S[5], 2 + S[1] and S[6],
                // 3 + S[2], S[7] and the chains.
                auto binarise = [&](auto level) {
                    constexpr int L = decltype(level)::value;
                    uint32_t S[8];
                    { const uint4 v0 = reinterpret_cast<const uint4*>(sl)[0], v1 = reinterpret_cast<const uint4*>(sl)[1];
                      S[0] = v0.x; S[1] = v0.y; S[2] = v0.z; S[3] = v0.w; S[4] = v1.x; S[5] = v1.y; S[6] = v1.z; S[7] = v1.w; }
#define ST_GET(k) ((S[(k) >> 2] >> (8 * ((k) & 3))) & 0xFFu)
#define ST_PUT_IF(c, k, v) (S[(k) >> 2] = (c) ? ((S[(k) >> 2] & ~(0xFFu << (8 * ((k) & 3)))) | ((v) << (8 * ((k) & 3)))) : S[(k) >> 2])
#define NEXT(en) uint32_t(trans[256 + (en)])                            /* en = +state for bit 1, -state for bit 0 */
#define EXP_SLOT(t) { const int en = __mul24(int(ST_GET(1 + (t))), sg_e[t]); nxe[t] = NEXT(en); if (nz && (t) <= e) op[OPI(1 + (t))] = uint8_t(en); }
#define MAN_SLOT(t) { const int en = __mul24(int(ST_GET(22 + (t))), sg_m[t]); nxm[t] = NEXT(en); if (nz && (t) < e) op[OPI(2 * e + 1 - (t))] = uint8_t(en); }
#define MERGE(x, n, m) x = ((n) & (m)) | ((x) & ~(m))
                    const bool nz = a != 0;
                    const int ks = 11 + (e < 10 ? e : 10);              // sign state: the only index that depends on the symbol
                    // ---- phase 1: read states, emit decisions, issue the look-ups.  A decision is (state, bit); en = +state for
                    // a coded 1, -state for a coded 0: its low byte is the stream's t (state or 256 - state), and the same signed
                    // number indexes the transition table.  (The coded bits went into the bit stage before the rounds.)
                    const int en_z = nz ? -int(ST_GET(0)) : int(ST_GET(0));
                    const uint32_t nx_z = NEXT(en_z);
                    op[OPI(0)] = uint8_t(en_z);
                    uint32_t nxe[9] = {}, nxm[9] = {}, nx_s = 0;
                    if constexpr (L >= 1) {
                        const int st_s = nz ? int(sl[ks]) : 128;       // sign states 11..21 are touched by nothing else: straight from LDS
                        EXP_SLOT(0) EXP_SLOT(1) EXP_SLOT(2) MAN_SLOT(0) MAN_SLOT(1)
                        if constexpr (L >= 2) { EXP_SLOT(3) EXP_SLOT(4) EXP_SLOT(5) EXP_SLOT(6) MAN_SLOT(2) MAN_SLOT(3) MAN_SLOT(4) MAN_SLOT(5) }
                        if constexpr (L >= 3) { EXP_SLOT(7) EXP_SLOT(8) MAN_SLOT(6) MAN_SLOT(7) MAN_SLOT(8) }
                        const int en_s = d < 0 ? st_s : -st_s;
                        nx_s = NEXT(en_s);
                        if (nz) op[OPI(2 * e + 2)] = uint8_t(en_s);
                    }
                    PROF_T(9)
                    // ---- the two chains: state 10 for exponent bits 9.., state 31 for mantissa bits e-1 .. 9
                    if constexpr (L >= 3) {
#pragma unroll 1
                        for (int t = 9; t <= emax; t++) {
                            const bool act = nz && t <= e;
                            const int en = t < e ? int(ST_GET(10)) : -int(ST_GET(10));
                            if (act) op[OPI(1 + t)] = uint8_t(en);
                            const uint32_t nx = NEXT(en);
                            ST_PUT_IF(act, 10, nx);
                        }
#pragma unroll 1
                        for (int t = emax - 1; t >= 9; t--) {
                            const bool act = nz && t < e;
                            const int en = (a >> t) & 1u ? int(ST_GET(31)) : -int(ST_GET(31));
                            if (act) op[OPI(2 * e + 1 - t)] = uint8_t(en);
                            const uint32_t nx = NEXT(en);
                            ST_PUT_IF(act, 31, nx);
                        }
                    }
                    PROF_T(10)
                    // ---- phase 2: apply the independent transitions: pack the looked-up bytes, merge under the lane's mask
                    if constexpr (L == 0) MERGE(S[0], nx_z, 0xFFu);
                    if constexpr (L >= 1) {
                        MERGE(S[0], nx_z | (nxe[0] << 8) | (nxe[1] << 16) | (nxe[2] << 24), Ma.x);
                        MERGE(S[5], (nxm[0] << 16) | (nxm[1] << 24), Ma.w);
                    }
                    if constexpr (L >= 2) {
                        MERGE(S[1], nxe[3] | (nxe[4] << 8) | (nxe[5] << 16) | (nxe[6] << 24), Ma.y);
                        MERGE(S[6], nxm[2] | (nxm[3] << 8) | (nxm[4] << 16) | (nxm[5] << 24), Mb.x);
                    }
                    if constexpr (L >= 3) {
                        MERGE(S[2], nxe[7] | (nxe[8] << 8), Ma.z);
                        MERGE(S[7], nxm[6] | (nxm[7] << 8) | (nxm[8] << 16), Mb.y);
                    }
#undef MERGE
#undef EXP_SLOT
#undef MAN_SLOT
#undef ST_GET
#undef ST_PUT_IF
#undef NEXT
                    reinterpret_cast<uint4*>(sl)[0] = make_uint4(S[0], S[1], S[2], S[3]);
                    reinterpret_cast<uint4*>(sl)[1] = make_uint4(S[4], S[5], S[6], S[7]);
                    PROF_T(11)
                    if (L >= 1 && nz) sl[ks] = uint8_t(nx_s);             // after the bulk write-back (LDS operations of a wave stay in order)
                };
                PROF_T(5)
                if (emax >= 7) binarise(std::integral_constant<int, 3>());
                else if (emax >= 3) binarise(std::integral_constant<int, 2>());
                else if (emax >= 0) binarise(std::integral_constant<int, 1>());
                else binarise(std::integral_constant<int, 0>());
            }
            done |= __ballot(ready);
            pending = pending && !ready;
            WAVE_SYNC();
        }

        PROF_T(5)
        // --- the states go back to HBM, once per group.  A record is 32 bytes and a store carries 16 per lane: written by its own lane in two
        // stores, every record costs two write requests to the L2 -- and k_resolve is bound by the requests its CU's L1 has in flight
        // (TCP_PENDING_STALL_CYCLES 85 % of the L1's cycles; 0.98 read + 2.29 write requests per sample, profiles/r03_tcp_counters.txt).
        // So a PAIR of lanes writes one record, its halves side by side in one store instruction: one request per record.
        if (!LDS_STATES) {
            const uint32_t wb = (valid && last) ? (key | uint32_t(leader) << 16) : 0xFFFFFFFFu;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t info = uint32_t(__shfl(int(wb), h * 32 + (lane >> 1)));
                if (info != 0xFFFFFFFFu) {
                    const uint32_t half = uint32_t(lane & 1) * 16u;
                    *reinterpret_cast<uint4*>(st_base + size_t(ST_KEY(info & 0xFFFFu)) * 32 + half) = *reinterpret_cast<const uint4*>(slot + (info >> 16) * 32 + half);
                }
            }
        }
        PROF_T(6)
        stage_count += total;
        flush_full();
        PROF_T(7)
#ifdef RCGPU_PROF
        prof[8] += 1;
#endif
        ppack = (valid ? key : 0xFFFFu) | uint32_t(leader) << 16;
    }
#ifdef RCGPU_PROF
    if (lane == 0) for (int i = 0; i < 12; i++) atomicAdd(&g_prof[i], prof[i]);
#endif
    if (!last_seg) {      // park the unfinished piece and the bitmap (or the state table) for the next segment
        if (LDS_STATES) for (uint32_t i = lane; i < nkeys * 2; i += 64) rs_states[i] = reinterpret_cast<const uint4*>(lstates)[i];
        if (lane < 14) reinterpret_cast<uint32_t*>(rs + 16)[lane] = reinterpret_cast<const uint32_t*>(stage)[lane];
        if (lane < 2) reinterpret_cast<uint32_t*>(rs + 72)[lane] = sbits[lane];
        if (lane == 0) *reinterpret_cast<uint32_t*>(rs) = stage_count;
        return;
    }
    // end-of-slice bit (state 129, FFV1_Slice.cpp:336-340), then pad the last piece
    if (lane == 0) stage[stage_count] = uint8_t(256 - 129);                  // state 129, a coded 0 (its bit in the bit stage is already 0)
    stage_count += 1;
    WAVE_SYNC();
    const uint32_t padded = (stage_count + kPieceEntries - 1) / kPieceEntries * kPieceEntries;
    for (uint32_t i = stage_count + lane; i < padded; i += 64) stage[i] = 0x80;     // never coded: k_rangecode knows the slice's decision count
    stage_count = padded;
    flush_full();
}

// ---------------------------------------------------------------------------------------------------------
#undef WAVE_SYNC
#undef ST_KEY

// K4: range coder, one LANE per slice (chain).  The 64 chains of a wavefront read the same piece index of their
// interleaved streams each iteration and run the low/range recurrence of RFC 9043 3.8.1 in lock-step.
//
// Decisions arrive pre-digested by k_resolve as t = bit ? state : 256 - state plus the coded bit; with c = bit ? 0 : 255
//     new_range = (range * t + c) >> 8          [ == bit ? range*state>>8 : range - (range*state>>8) ]
//     low      += bit ? range - new_range : 0   [ the bit as an all-ones / zero mask: one v_bfe_i32, one v_and ]
// and the dependent chain per decision is mad -> shift -> (compare || shift) -> select.  A single wavefront issues
// in order, so the length of that chain -- not the instruction count -- sets the time per decision; everything else
// (low, byte bookkeeping) is independent work that fills the gaps.  There is no branch in the steady state:
//   * `low` is a 64-bit big number: 16 live bits, up to five finished bytes above them, carry above those; carries
//     ripple by plain integer addition;
//   * every second decision a predicated flush moves four finished bytes to `pd`, a one-dword second stage that
//     absorbs late carries, and parks the previous `pd` in LDS;
//   * parked dwords leave for HBM at the next piece boundary, before the next prefetch is issued, so the loads the
//     loop waits on are always younger than every store;
//   * a carry out of `pd` (probability ~2^-32 per flush) is flagged per parked dword and handed to k_footer as an
//     event, which ripples it through the bytes already in HBM.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kMaxCarryEvents = 4096;
// Finished dwords are parked in a per-lane ring in LDS and leave for HBM SIXTEEN BYTES at a time: a dword per lane and store instruction is
// a write request per four bytes of output -- as many requests as the coder's stream reads cause, on CUs whose L1 request queues are what
// k_resolve waits for (profiles/r03_tcp_counters.txt).  Between two drains the decisions renormalise at most once each, a flush per four
// bytes (+1 carried in), and up to three rows stay behind for the next group: the whole-slice coder drains twice per piece (always before
// a prefetch is issued or in the middle of a piece), the span coder of the split mapping every 14 decisions into half the rows.
// The whole-slice coder keeps the one-dword stores of its first version, all of them at the piece boundary before the next prefetch is
// issued: its wavefronts are serial chains, and stores in the middle of a piece made them 7 % longer (measured, round 3).
constexpr int kDrainWhole = kPieceEntries, kDrainSpan = 14;
constexpr int ring_rows(int drain_every) { return drain_every / 4 + 1 + (drain_every % 4 ? 1 : 0) + 3 <= 8 ? 8 : drain_every / 4 + 1 <= 16 ? 16 : 32; }
static_assert(kDrainWhole / 4 + 1 <= 16 && kDrainSpan / 4 + 2 + 3 <= 8, "the ring holds what two drains leave and add");

struct rc_resume { uint32_t range, nb, pd; int pos; unsigned long long low; unsigned long long pad; };   // per chain, between segments

struct rc_lane {
    uint32_t range; unsigned long long low; uint32_t nb; uint32_t pd;
    uint32_t wr, rd;                  // dwords parked / drained so far (ring positions modulo its size)
    uint32_t* obuf;                   // LDS [rows][64], this lane's column
    int pos; uint8_t* out; int cap; uint32_t chain;
    int lo;                           // first byte position this lane codes (0 for a whole slice, the checkpoint's for a span)
};

// m = all ones for a coded 1, zero for a coded 0 (one v_bfe_i32 off the piece's bit words); c = 255 for a coded 0
__device__ __forceinline__ void rc_step(rc_lane& r, uint32_t t, uint32_t m)
{
    const uint32_t c = 255u & ~m;
    const uint32_t nr = (__umul24(r.range, t) + c) >> 8;
    const uint32_t inc = (r.range - nr) & m;
    const uint32_t sh = nr < 0x100 ? 8u : 0u;                 // renormalise: one byte at most, since t >= 1 keeps nr >= range >> 8
    r.range = nr << sh;
    r.low = (r.low + inc) << sh;
    r.nb += sh;                                               // nb counts finished BITS (8 per byte)
}

// Predicated flush: when >= 4 finished bytes are pending, the four oldest move to pd and the old pd is parked.  The parked slot is
// written unconditionally (a lane that does not flush rewrites it at its next real flush), and the ~2^-32 carry out of pd is the
// only branch: taken by a whole wavefront almost never.
template <int R>
__device__ __forceinline__ void rc_check(rc_lane& r, uint32_t* ev_count, uint2* ev)
{
    const bool f = r.nb >= 32;                                // nb <= 40 here
    const uint32_t sh16 = r.nb - 16;                          // 16 or 24 when f: the four oldest bytes start at this bit
    const uint32_t four = uint32_t(r.low >> (sh16 & 63));
    const uint32_t carry = f ? uint32_t(r.low >> 32) >> (sh16 & 31) : 0u;   // whatever sits above those four bytes
    const uint32_t npd = r.pd + carry;
    r.obuf[(r.wr & uint32_t(R - 1)) * 64] = __builtin_bswap32(npd);
    if (__builtin_expect(__ballot(npd < carry) != 0, 0)) {    // pd wrapped: +1 belongs to the bytes below that dword, already in HBM
        if (npd < carry) {
            const uint32_t slot = atomicAdd(ev_count, 1u);
            if (slot < kMaxCarryEvents) ev[slot] = make_uint2(r.chain, uint32_t(r.pos + 4 * int(r.wr - r.rd)));
        }
    }
    r.wr += r.nb >> 5;                                        // bit 5 of nb <=> f
    r.pd = f ? four : r.pd;
    r.low = f ? (unsigned long long)__builtin_amdgcn_ubfe(uint32_t(r.low), 0u, sh16) : r.low;
    r.nb &= 31;                                               // -32 when f
}

// Groups of four parked dwords leave as one 16-byte store (WIDE); ALL = everything goes, dword by dword: at the end of a launch or span the
// one to three rows the groups left, in the whole-slice coder every row.
template <int R, bool WIDE, bool ALL = false>
__device__ __forceinline__ void rc_drain(rc_lane& r)
{
    // the first dword a lane parks is its still empty second stage (r.pos == r.lo - 4): it is never stored
    if (r.pos < r.lo && r.wr != r.rd) { r.rd++; r.pos += 4; }
    if (WIDE)
#pragma unroll
    for (int g = 0; g < R / 4; g++) {
        if (r.wr - r.rd >= 4u) {
            uint4 v;
            v.x = r.obuf[(r.rd & uint32_t(R - 1)) * 64]; v.y = r.obuf[((r.rd + 1) & uint32_t(R - 1)) * 64];
            v.z = r.obuf[((r.rd + 2) & uint32_t(R - 1)) * 64]; v.w = r.obuf[((r.rd + 3) & uint32_t(R - 1)) * 64];
            const int at = r.pos + 16 <= r.cap ? r.pos : r.cap - 16;       // an overflowing slice is reported at its end, its stores stay inside
            *reinterpret_cast<uint4*>(r.out + at) = v;
            r.pos += 16; r.rd += 4;
        }
    }
    if (ALL) {
#pragma unroll
        for (int k = 0; k < (WIDE ? 3 : R); k++) {
            if (r.wr != r.rd) {
                const int at = r.pos + 4 <= r.cap ? r.pos : r.cap - 4;
                *reinterpret_cast<uint32_t*>(r.out + at) = r.obuf[(r.rd & uint32_t(R - 1)) * 64];
                r.pos += 4; r.rd++;
            }
        }
    }
}

template <int DRAIN>
__device__ __forceinline__ void rc_piece(rc_lane& r, const uint4 (&q)[4], uint32_t cnt, uint32_t* ev_count, uint2* ev)
{
    const uint32_t w[16] = { q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w,
                             q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w };
    if (cnt == kPieceEntries) {
#pragma unroll
        for (int j = 0; j < kPieceEntries; j++) {
            const uint32_t t = (w[j >> 2] >> (8 * (j & 3))) & 0xFF;
            const uint32_t m = uint32_t(__builtin_amdgcn_sbfe(int(w[14 + (j >> 5)]), uint32_t(j & 31), 1u));
            rc_step(r, t, m);
            if (j & 1) rc_check<ring_rows(DRAIN)>(r, ev_count, ev);
            if (DRAIN < kPieceEntries && (j + 1) % DRAIN == 0 && j + 1 < kPieceEntries) rc_drain<ring_rows(DRAIN), true>(r);
        }
    } else {
#pragma unroll 1
        for (uint32_t j = 0; j < cnt; j++) {      // last (partial) piece only: pick dword j/4 without indexing registers dynamically
            uint32_t ww = 0;
#pragma unroll
            for (int k = 0; k < 14; k++) ww = (j >> 2) == uint32_t(k) ? w[k] : ww;
            const uint32_t bw = j < 32 ? w[14] : w[15];
            rc_step(r, (ww >> (8 * (j & 3))) & 0xFF, 0u - ((bw >> (j & 31)) & 1u));
            rc_check<ring_rows(DRAIN)>(r, ev_count, ev);
            if ((j & 3) == 3) rc_drain<ring_rows(DRAIN), DRAIN < kPieceEntries, DRAIN == kPieceEntries>(r);       // a check per decision here: at most four rows between drains
        }
    }
}

// A checkpoint of the split coder, per (span, chain): k_rc_range leaves the chain's `range` and byte position at the start of the span,
// k_rangecode<true> replaces the range by the span's tail -- the 16 live bits of `low` its bytes do not hold yet.
struct rc_ckpt { uint32_t v, pos; };

// SPAN = false: one lane codes its slice's pieces of this segment, state parked in `resume` between segments.
// SPAN = true (split coder): blockIdx.y = span; the lane codes pieces [span * span_pieces, + span_pieces) of its slice's segment from the
// checkpoint k_rc_range left, with low = 0: all operations on `low` are additions and shifts, so the true stream is the sum of the spans'
// streams, each at its own byte position -- the bytes of a span plus its tail, which k_rc_tails adds into the bytes that follow.  A span
// started at low = 0 never carries out of its own first byte (low + range <= the checkpoint's range < 2^16 throughout).
template <bool SPAN>
__global__ __launch_bounds__(64) __attribute__((aligned(4096))) void k_rangecode(const enc_const* __restrict__ C, const slice_geom* __restrict__ geom,
                                                  const unsigned long long* __restrict__ total_n, const uint32_t* __restrict__ seg_pieces,
                                                  uint32_t seg, rc_resume* __restrict__ resume,
                                                  const unsigned long long* __restrict__ group_off,
                                                  const uint8_t* __restrict__ stream, uint8_t* __restrict__ cbuf,
                                                  unsigned long long cbuf_frame_stride, uint32_t nchains,
                                                  uint32_t* __restrict__ out_len, uint32_t* __restrict__ err, uint2* __restrict__ events,
                                                  rc_ckpt* __restrict__ ckpt, uint32_t span_pieces)
{
    constexpr int kDrain = SPAN ? kDrainSpan : kDrainWhole, kRows = ring_rows(kDrain);
    __shared__ uint32_t obuf[kRows * 64];
    // The whole-slice coder is latency-bound and shares SIMDs with throughput-bound k_resolve wavefronts: take issue priority.
    if (!SPAN) { const uint32_t pr = C->rc_prio; if (pr == 3) __builtin_amdgcn_s_setprio(3); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 1) __builtin_amdgcn_s_setprio(1); }
    const int lane = threadIdx.x;
    const uint32_t chain = blockIdx.x * 64 + lane;
    const bool active = chain < nchains;
    const uint32_t S = C->S;
    const uint32_t cc = active ? chain : nchains - 1;
    const uint32_t f = cc / S, s = cc - f * S;
    const slice_geom G = geom[s];
    // this launch codes the pieces k_resolve produced for segment `seg`; only the last piece of the last segment is partial
    const bool last_seg = seg + 1 == C->nseg;
    const unsigned long long seg_np = active ? seg_pieces[cc] : 0;
    unsigned long long n = seg_np * kPieceEntries;
    if (active && last_seg && seg_np) n -= (kPieceEntries - 1) - ((total_n[cc] - 1) % kPieceEntries);
    // the pieces this lane codes: [p0, p0 + npieces) of the segment
    const unsigned long long p0 = SPAN ? (unsigned long long)blockIdx.y * span_pieces : 0ull;
    const unsigned long long npieces = SPAN ? (seg_np > p0 ? (seg_np - p0 < span_pieces ? seg_np - p0 : span_pieces) : 0ull) : seg_np;
    const bool ends_chain = last_seg && npieces && p0 + npieces == seg_np;       // this lane codes the slice's last decision
    const uint4* src = reinterpret_cast<const uint4*>(stream + group_off[blockIdx.x]) + p0 * (kGroupPieceBytes / 16) + lane * 4;
    rc_ckpt* const ck = SPAN ? ckpt + size_t(blockIdx.y) * nchains + cc : nullptr;

    rc_lane r;
    r.obuf = obuf + lane; r.wr = r.rd = 0;
    r.range = 0xFF00; r.low = 0; r.nb = 0; r.pd = 0; r.pos = -4; r.chain = cc; r.lo = 0;
    if (SPAN) { const rc_ckpt v = *ck; r.range = v.v; r.lo = int(v.pos); r.pos = r.lo - 4; }
    else if (seg) { const rc_resume v = resume[cc]; r.range = v.range; r.low = v.low; r.nb = v.nb; r.pd = v.pd; r.pos = v.pos; }   // nb in bits
    r.out = cbuf + size_t(f) * cbuf_frame_stride + (size_t(G.cbuf_off_hi) << 32 | G.cbuf_off_lo);
    r.cap = int(G.cbuf_cap);
    // Overlay: the slice's bytes go where its symbols lay.  While segment `seg` is coded k_resolve may be reading the symbols of segment
    // seg + 1 onwards, which begin 4 * seg_q * (seg + 1) bytes into the area (16 of them in front of the byte buffer): the bytes must stay below
    // that.  A symbol is four bytes and codes to 3.4 at the very worst (16-bit noise, untrained states), so they do; a slice that does not
    // is reported as overflowing, like one that outgrows its buffer.
    const bool seg_limited = C->overlay && !last_seg;
    if (seg_limited) { const unsigned long long lim = 4ull * G.seg_q * (seg + 1) / C->overlay - 48; if (lim < (unsigned long long)r.cap) r.cap = int(lim); }

    unsigned long long maxp = npieces;
    for (int o = 32; o; o >>= 1) { const unsigned long long t = __shfl_xor(maxp, o); maxp = t > maxp ? t : maxp; }

    // Loads are unconditional (the index is clamped to a valid piece) so that they stay plain global_load_dwordx4
    // (a select against a zero constant turns them into flat loads of an unknown address space).
    uint4 cur[4], nxt[4];
    if (SPAN && !maxp) { if (active) ck->v = 0; return; }               // a span past the end of every slice of this group (uniform)
    if (SPAN) {
        // The span coder is a throughput kernel: its wavefronts fill the issue slots k_resolve's leave while they wait for the LDS, and
        // what counts is how many of them fit beside those -- 64 registers instead of 80 without the second piece buffer.  The load's
        // latency is covered by the other wavefronts of the SIMD.
        for (unsigned long long pc = 0; pc < maxp; pc++) {
            const unsigned long long pi = pc < npieces ? pc : (npieces ? npieces - 1 : 0);
            const uint4* p = src + pi * (kGroupPieceBytes / 16);
#pragma unroll
            for (int k = 0; k < 4; k++) cur[k] = p[k];
            rc_drain<kRows, SPAN, !SPAN>(r);
            if (pc < npieces) {
                const unsigned long long left = n - (p0 + pc) * kPieceEntries;
                rc_piece<kDrain>(r, cur, left < kPieceEntries ? uint32_t(left) : uint32_t(kPieceEntries), err + 1, events);
            }
        }
    } else {
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = src[k];
    for (unsigned long long pc = 0; pc < maxp; pc++) {
        // Pin the current piece in VGPRs *before* anything new is issued: the compiler's s_waitcnt for these registers then
        // sits here (covering only loads issued a whole piece ago), not behind the next prefetch.
#pragma unroll
        for (int k = 0; k < 4; k++) { asm volatile("" : "+v"(cur[k].x), "+v"(cur[k].y), "+v"(cur[k].z), "+v"(cur[k].w)); }
        rc_drain<kRows, SPAN, !SPAN>(r);
        const unsigned long long pn = pc + 1 < npieces ? pc + 1 : (npieces ? npieces - 1 : 0);
        const uint4* p = src + pn * (kGroupPieceBytes / 16);
#pragma unroll
        for (int k = 0; k < 4; k++) nxt[k] = p[k];                      // prefetch: in flight while this piece is coded
        if (pc < npieces) {
            const unsigned long long left = n - (p0 + pc) * kPieceEntries;
            rc_piece<kDrain>(r, cur, left < kPieceEntries ? uint32_t(left) : uint32_t(kPieceEntries), err + 1, events);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    }
    rc_drain<kRows, SPAN, true>(r);
    if (seg_limited && active && r.pos > r.cap) atomicOr(err, 1u);       // the bytes ran into symbols not read yet: their stores were held back, the batch is void
    if (!SPAN && active && !last_seg) { rc_resume v; v.range = r.range; v.low = r.low; v.nb = r.nb; v.pd = r.pd; v.pos = r.pos; v.pad = 0; resume[cc] = v; }
    if (SPAN && active && !ends_chain) {
        // End of a span: the second stage and the finished bytes leave one by one (the next span's first byte follows directly), the 16
        // live bits are the tail.  A lane without pieces leaves a zero tail.
        uint32_t tail = 0;
        if (npieces) {
            const uint32_t nby = r.nb >> 3;                                         // <= 3 finished bytes above the live bits (a span ends with a full piece: checked)
            const uint32_t carry = uint32_t(r.low >> (16 + 8 * nby));
            const uint32_t npd = r.pd + carry;
            if (npd < carry) { const uint32_t slot = atomicAdd(err + 1, 1u); if (slot < kMaxCarryEvents) events[slot] = make_uint2(r.chain, uint32_t(r.pos)); }
            if (r.pos >= r.lo) for (int t = 0; t < 4; t++) if (r.pos + t < r.cap) r.out[r.pos + t] = uint8_t(npd >> (24 - 8 * t));
            r.pos += 4;
            for (int t = int(nby) - 1; t >= 0; t--) { if (r.pos < r.cap) r.out[r.pos] = uint8_t(r.low >> (16 + 8 * t)); r.pos++; }
            tail = uint32_t(r.low) & 0xFFFFu;
        }
        ck->v = tail;
    }
    if (active && ends_chain) {
        // terminate (the state-129 end bit is already the last decision): two forced renormalisations; the last
        // latched byte is not part of the slice -- the decoder reads bytes past the end as zero (FFV1_RangeCoder.cpp:79-85).
        r.nb >>= 3;                            // back to bytes for the tail
        r.low += 0xFF;                         // nb <= 3 here: a flush check follows the last decision of a piece, full or partial
        r.low <<= 8; r.nb++;
        r.low <<= 8; r.nb++;                   // nb <= 5: 16 + 40 + carry bits still fit
        const uint32_t carry = uint32_t(r.low >> (16 + 8 * r.nb));
        const uint32_t npd = r.pd + carry;
        if (npd < carry) { const uint32_t slot = atomicAdd(err + 1, 1u); if (slot < kMaxCarryEvents) events[slot] = make_uint2(r.chain, uint32_t(r.pos)); }
        bool overflow = false;
        if (r.pos >= r.lo) { if (r.pos + 4 <= r.cap) *reinterpret_cast<uint32_t*>(r.out + r.pos) = __builtin_bswap32(npd); else overflow = true; }
        r.pos += 4;
        // the split coder also stores the latched byte behind the slice's last one: a tail added there may still carry into the slice
        for (int t = int(r.nb) - 1; t >= (SPAN ? 0 : 1); t--) {
            if (r.pos < r.cap) r.out[r.pos] = uint8_t(r.low >> (16 + 8 * t)); else overflow = true;
            r.pos++;
        }
        out_len[chain] = uint32_t(SPAN ? r.pos - 1 : r.pos);
        if (overflow) atomicOr(err, 1u);
        if (SPAN) ck->v = 0;
    }
}

// Split coder, first pass: the serial part of the range coder and nothing else.  One lane per slice walks the decisions of this segment
// with  range' = renormalised((range * t + c) >> 8)  and counts the renormalisations (= bytes); at the start of every span of
// `span_pieces` pieces it leaves (range, byte position) for the lane of k_rangecode<true> that will code the span.  Nine VALU
// instructions per decision instead of twenty-one: this chain is what a batch's latency is made of.
__device__ __forceinline__ void rr_step(uint32_t& range, uint32_t& cnt, uint32_t t, uint32_t m)
{
    const uint32_t x = __umul24(range, t) + (255u & ~m);
    const uint32_t nr = x >> 8;
    const bool ren = x < 0x10000u;                            // nr < 0x100
    range = ren ? (x & 0xFFFFFF00u) : nr;                     // nr << 8
    cnt += ren ? 1u : 0u;
}

// The same step for the decisions of a full piece, written out (tools/gen_rc_range_asm.py): this wavefront is alone on its SIMD's issue port for most of its life, every
// VALU instruction costs it ~4.7 cycles and a dependent one 8.1 (tools/valu_peak), so the count is nine and the order keeps two
// independent instructions between every pair on the range -> range path.  t and c = bit ? 0 : 255 of decision j arrive in registers, the
// step extracts those of decision j + 1 in the gaps -- which is also what keeps two instructions between the compare and the select that
// reads VCC (the hazard the compiler covers with s_nop 1).  One asm block: between two blocks the compiler puts an s_nop.
// A full piece: tA / cA hold t and c = bit ? 0 : 255 of decision 0 on entry.
__device__ __forceinline__ void rr_piece_asm(uint32_t& range, uint32_t& cnt, const uint32_t (&w)[16], uint32_t k255)
{
    uint32_t tA = w[0] & 0xFF, cA = (w[14] & 1u) ? 0u : 255u, tB, cB, x, nr, hi, m;
    asm volatile(
#include "rc_range_asm.inc"
                 : [range] "+v"(range), [cnt] "+v"(cnt), [tA] "+v"(tA), [cA] "+v"(cA), [tB] "=&v"(tB), [cB] "=&v"(cB),
                   [x] "=&v"(x), [nr] "=&v"(nr), [hi] "=&v"(hi), [m] "=&v"(m)
                 : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7]),
                   [w8] "v"(w[8]), [w9] "v"(w[9]), [w10] "v"(w[10]), [w11] "v"(w[11]), [w12] "v"(w[12]), [w13] "v"(w[13]),
                   [b0] "v"(w[14]), [b1] "v"(w[15]), [k255] "s"(k255)
                 : "vcc");
}

__global__ __launch_bounds__(64) __attribute__((aligned(4096))) void k_rc_range(const enc_const* __restrict__ C,
                                                  const unsigned long long* __restrict__ total_n, const uint32_t* __restrict__ seg_pieces,
                                                  uint32_t seg, rc_resume* __restrict__ resume,
                                                  const unsigned long long* __restrict__ group_off,
                                                  const uint8_t* __restrict__ stream, uint32_t nchains,
                                                  rc_ckpt* __restrict__ ckpt, uint32_t span_pieces, uint32_t nspans)
{
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const uint32_t chain = blockIdx.x * 64 + lane;
    const bool active = chain < nchains;
    const uint32_t cc = active ? chain : nchains - 1;
    const bool last_seg = seg + 1 == C->nseg;
    const unsigned long long npieces = active ? seg_pieces[cc] : 0;
    unsigned long long n = npieces * kPieceEntries;
    if (active && last_seg && npieces) n -= (kPieceEntries - 1) - ((total_n[cc] - 1) % kPieceEntries);
    const uint4* src = reinterpret_cast<const uint4*>(stream + group_off[blockIdx.x]) + lane * 4;
    uint32_t range = 0xFF00, cnt = 0;
    if (seg) { const rc_resume v = resume[cc]; range = v.range; cnt = uint32_t(v.pos); }
    rc_ckpt* ck = ckpt + cc;

    unsigned long long maxp = npieces;
    for (int o = 32; o; o >>= 1) { const unsigned long long t = __shfl_xor(maxp, o); maxp = t > maxp ? t : maxp; }
    uint4 cur[4], nxt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = src[k];
    uint32_t to_ckpt = 0, spans_done = 0;
    for (unsigned long long pc = 0; pc < maxp; pc++) {
#pragma unroll
        for (int k = 0; k < 4; k++) { asm volatile("" : "+v"(cur[k].x), "+v"(cur[k].y), "+v"(cur[k].z), "+v"(cur[k].w)); }
        if (to_ckpt == 0) {                                             // uniform: a span starts here (a slice that has run out repeats its state)
            if (active) { rc_ckpt v; v.v = range; v.pos = cnt; *ck = v; }
            ck += nchains; to_ckpt = span_pieces; spans_done++;
        }
        to_ckpt--;
        const unsigned long long pn = pc + 1 < npieces ? pc + 1 : (npieces ? npieces - 1 : 0);
        const uint4* p = src + pn * (kGroupPieceBytes / 16);
#pragma unroll
        for (int k = 0; k < 4; k++) nxt[k] = p[k];
        if (pc < npieces) {
            const uint32_t w[16] = { cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w,
                                     cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w };
            const unsigned long long left = n - pc * kPieceEntries;
            if (left >= kPieceEntries) {
                rr_piece_asm(range, cnt, w, 255u);
            } else {
#pragma unroll 1
                for (uint32_t j = 0; j < uint32_t(left); j++) {
                    uint32_t ww = 0;
#pragma unroll
                    for (int k = 0; k < 14; k++) ww = (j >> 2) == uint32_t(k) ? w[k] : ww;
                    const uint32_t bw = j < 32 ? w[14] : w[15];
                    rr_step(range, cnt, (ww >> (8 * (j & 3))) & 0xFF, 0u - ((bw >> (j & 31)) & 1u));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    for (; spans_done < nspans; spans_done++, ck += nchains) if (active) { rc_ckpt v; v.v = range; v.pos = cnt; *ck = v; }
    if (active && !last_seg) { rc_resume v; v.range = range; v.low = 0; v.nb = 0; v.pd = 0; v.pos = int(cnt); v.pad = 0; resume[cc] = v; }
}

// Split coder, last pass: the tail of span g -- the 16 live bits of `low` its coder ended with -- belongs to the two bytes that follow the
// span, i.e. to the start of span g + 1; carries ripple towards the front of the slice.  One lane per slice adds its tails in order; the
// checkpoints of all segments of the batch are laid out [span][chain].
__global__ __launch_bounds__(64) void k_rc_tails(const enc_const* __restrict__ C, const slice_geom* __restrict__ geom,
                                                 const rc_ckpt* __restrict__ ckpt, uint32_t total_spans, uint32_t nchains,
                                                 uint8_t* __restrict__ cbuf, unsigned long long cbuf_frame_stride)
{
    const uint32_t chain = blockIdx.x * 64 + threadIdx.x;
    if (chain >= nchains) return;
    const uint32_t S = C->S, f = chain / S, s = chain - f * S;
    const slice_geom G = geom[s];
    uint8_t* out = cbuf + size_t(f) * cbuf_frame_stride + (size_t(G.cbuf_off_hi) << 32 | G.cbuf_off_lo);
    const int cap = int(G.cbuf_cap);
    const rc_ckpt* ck = ckpt + chain;
    uint32_t tail = total_spans ? ck->v : 0;
    for (uint32_t g = 1; g < total_spans; g++) {
        ck += nchains;
        const rc_ckpt v = *ck;
        const int q = int(v.pos);
        if (tail && q + 2 <= cap) {
            const uint32_t sum = (uint32_t(out[q]) << 8 | out[q + 1]) + tail;
            out[q + 1] = uint8_t(sum); out[q] = uint8_t(sum >> 8);
            if (sum >> 16) { int p = q; while (p > 0) { p--; const uint8_t b = uint8_t(out[p] + 1); out[p] = b; if (b) break; } }
        }
        tail = v.v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K5: slice footer (size24, error_status, CRC-32), one wavefront per slice.
// CRC: poly 0x04C11DB7, MSB first, init 0, no final xor (ZenCRC32.cpp:1097-1135).  Each lane takes one contiguous
// segment; crc(A||B) = crc(A) * x^(8|B|) + crc(B) in GF(2)[x]/P combines them.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_footer(const enc_const* __restrict__ C, const slice_geom* __restrict__ geom,
                                                uint8_t* __restrict__ cbuf, unsigned long long cbuf_frame_stride,
                                                const uint32_t* __restrict__ out_len, uint32_t* __restrict__ tot_len,
                                                uint32_t* __restrict__ err, const uint2* __restrict__ events)
{
    __shared__ uint32_t T[4][256], TM[4][256];          // slicing-by-4 tables and the tile multiplier's (crc_dev.h)
    __shared__ uint32_t part[4];
    const int tid = threadIdx.x;
    crc_tables(T, tid);
    const uint32_t chain = blockIdx.x;
    const uint32_t S = C->S, f = chain / S, s = chain - f * S;
    const slice_geom G = geom[s];
    uint8_t* out = cbuf + size_t(f) * cbuf_frame_stride + (size_t(G.cbuf_off_hi) << 32 | G.cbuf_off_lo);
    const uint32_t len = out_len[chain];
    const uint32_t tail = C->v1 ? 0 : C->ec ? 8 : 3;
    if (len + tail > G.cbuf_cap || (!C->v1 && len > 0xFFFFFF)) { if (tid == 0) { atomicOr(err, 2u); tot_len[chain] = 0; } return; }
    if (tid == 0) {
        // carries that left k_rangecode's second stage after the bytes below them were already stored
        const uint32_t nev = err[1];
        if (nev > kMaxCarryEvents) atomicOr(err, 4u);
        for (uint32_t i = 0; i < nev && i < kMaxCarryEvents; i++)
            if (events[i].x == chain) { uint32_t p = events[i].y; while (p) { p--; const uint8_t b = uint8_t(out[p] + 1); out[p] = b; if (b) break; } }
        if (!C->v1) {
            out[len] = uint8_t(len >> 16); out[len + 1] = uint8_t(len >> 8); out[len + 2] = uint8_t(len);
            if (C->ec) out[len + 3] = 0;                  // error_status
        }
    }
    __syncthreads();
    if (C->v1) { if (tid == 0) tot_len[chain] = len; return; }      // version 1: the frame is the coder's bytes
    if (!C->ec) { if (tid == 0) tot_len[chain] = len + 3; return; }
    uint32_t c = block_crc_share(out, len + 4, T, TM, tid);       // the tiled CRC the decoder's k_dec_crc uses too
    // xor-reduce over the block
    for (int o = 32; o; o >>= 1) c ^= __shfl_xor(c, o);
    if ((tid & 63) == 0) part[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        c = part[0] ^ part[1] ^ part[2] ^ part[3];
        out[len + 4] = uint8_t(c >> 24); out[len + 5] = uint8_t(c >> 16); out[len + 6] = uint8_t(c >> 8); out[len + 7] = uint8_t(c);
        tot_len[chain] = len + 8;
    }
}

// K6: per-frame exclusive scan of slice sizes (slices are stored in raster order sy*num_h + sx, the order FFmpeg
// emits them; the decoder walks them from the tail, FFV1_Frame.cpp:177-198).
__global__ __launch_bounds__(64) void k_scan(const enc_const* __restrict__ C, const uint32_t* __restrict__ tot_len,
                                             unsigned long long* __restrict__ slice_dst, unsigned long long* __restrict__ packet_sizes)
{
    const int lane = threadIdx.x;
    const uint32_t f = blockIdx.x, S = C->S;
    unsigned long long run = 0;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t s = base + lane;
        const uint32_t v = s < S ? tot_len[f * S + s] : 0;
        const uint32_t incl = wave_incl_scan(v, lane);
        if (s < S) slice_dst[f * S + s] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0) packet_sizes[f] = run;
}

// K7: gather slices into the packet.  Destination dwords are written aligned; the source is re-aligned with a
// funnel shift.  grid = (chains, 8).
__global__ __launch_bounds__(256) void k_gather(const enc_const* __restrict__ C, const slice_geom* __restrict__ geom,
                                                const uint8_t* __restrict__ cbuf, unsigned long long cbuf_frame_stride,
                                                const uint32_t* __restrict__ tot_len, const unsigned long long* __restrict__ slice_dst,
                                                uint8_t* __restrict__ packets, unsigned long long packet_stride)
{
    const uint32_t chain = blockIdx.x;           // chains on x, the eight blocks of a slice on y
    const uint32_t S = C->S, f = chain / S, s = chain - f * S;
    const slice_geom G = geom[s];
    const uint8_t* src = cbuf + size_t(f) * cbuf_frame_stride + (size_t(G.cbuf_off_hi) << 32 | G.cbuf_off_lo);
    uint8_t* dst = packets + size_t(f) * packet_stride + slice_dst[chain];
    const uint32_t len = tot_len[chain];
    const uint32_t tid = blockIdx.y * 256 + threadIdx.x, nthreads = gridDim.y * 256;
    const uint32_t head = min(len, uint32_t((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3));
    if (tid < head) dst[tid] = src[tid];
    const uint32_t ndw = (len - head) / 4;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);      // src is 16-byte aligned
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
    const uint32_t sh = head * 8;                                        // source byte offset of dst dword 0 is `head` (0..3)
    for (uint32_t t = tid; t < ndw; t += nthreads) {
        const uint32_t lo = s32[t], hi = s32[t + 1];
        d32[t] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    }
    const uint32_t done = head + ndw * 4;
    if (tid < len - done) dst[done + tid] = src[done + tid];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kMaxSeg = 64;
constexpr uint32_t kMaxWindows = 8, kWindows = 2;      // hand-over windows between k_resolve and k_rangecode
constexpr uint32_t kSubBatch = 16;           // frames whose int32 planes rcgpu_ffv1_debug_fetch(0) returns (k_unpack on demand)

// What an encoder of this configuration writes into its configuration record.
static ffv1::stream_params stream_params_of(const rcgpu_ffv1_config& cfg)
{
    const pix_desc& d = pix(cfg.pixfmt);
    ffv1::stream_params sp{};
    sp.bits_per_raw_sample = d.bits; sp.rgb = d.planes != 1; sp.alpha = d.planes == 4;
    sp.num_h_slices = cfg.num_h_slices; sp.num_v_slices = cfg.num_v_slices;
    sp.ec = cfg.slicecrc ? 1 : 0; sp.context_model = cfg.context ? 1 : 0; sp.compact = cfg.context == 2; sp.coder = cfg.coder == 2 ? 2 : 1; sp.version = cfg.level == 1 ? 1 : 3;
    return sp;
}

// Byte buffer of one slice: room for 1.5x its raw payload (incompressible 16-bit noise codes to ~1.1x once the contexts have adapted)
// + what ~10^4 contexts x 32 states can cost before they have (tiny slices reach 1.7x), + header/footer.
static size_t slice_buffer_bytes(const pix_desc& d, uint32_t w, uint32_t h, uint32_t version, uint32_t div)
{
    const size_t raw15 = size_t(w) * h * (d.bytes_pp ? d.bytes_pp : d.planes * 2u) * 3 / 2;
    size_t cap = (raw15 + std::min<size_t>(raw15, 256u << 10) + 4096 + 15) & ~size_t(15);
    if (div > 1) cap = std::max<size_t>(64, (cap / div) & ~size_t(15));      // rcgpu_ffv1_config::slice_buffer_div: the overflow report can be exercised
    if (cap > 0xFFFFFF + 64 && version != 1) cap = 0xFFFFFF + 64;          // slice size field is 24 bit (version 1 has none)
    return cap;
}

// Do the slice byte buffers of this configuration lie inside the slices' symbol areas (rcgpu_ffv1::overlay)?  nseg as the encoder chooses it.
constexpr size_t kOverlayMinSamples = size_t(64) << 10;
static size_t overlay_cap(size_t cap, size_t nsamp) { return std::min(cap, (nsamp * 4 - 48) & ~size_t(15)); }      // 16 bytes in front, 16 behind, inside nsamp * 4
static bool overlay_fits(const rcgpu_ffv1_config& cfg)
{
    if (cfg.pixfmt >= RCGPU_PIX_COUNT || !cfg.num_h_slices || !cfg.num_v_slices || cfg.level == 1 || (cfg.flags & RCGPU_FLAG_OWN_SLICE_BUFFERS)) return false;
    const pix_desc& d = pix(cfg.pixfmt);
    uint32_t min_nsamp = ~0u;
    for (uint32_t sy = 0; sy < cfg.num_v_slices; sy++)
        for (uint32_t sx = 0; sx < cfg.num_h_slices; sx++) {
            const uint32_t w = uint32_t(uint64_t(sx + 1) * cfg.width / cfg.num_h_slices) - uint32_t(uint64_t(sx) * cfg.width / cfg.num_h_slices);
            const uint32_t h = uint32_t(uint64_t(sy + 1) * cfg.height / cfg.num_v_slices) - uint32_t(uint64_t(sy) * cfg.height / cfg.num_v_slices);
            const size_t nsamp = size_t(w) * h * d.planes;
            // a slice of 64 K samples and more: 16-bit noise codes to 3.1 bytes per sample there (3.4 at 12 K, 3.9 in the first hundreds, where
            // every context is untrained: small slices keep buffers of their own); its cap inside the overlay is its symbol area (kOverlayCap)
            if (nsamp < kOverlayMinSamples) return false;
            min_nsamp = std::min<uint32_t>(min_nsamp, uint32_t(nsamp));
        }
    const uint32_t nseg = cfg.segments ? cfg.segments : std::max(1u, std::min(32u, min_nsamp / 1024));
    return (((size_t(min_nsamp) + nseg - 1) / nseg + 63) & ~size_t(63)) * 4 / std::max(1u, cfg.slice_buffer_div) > 4096;      // a segment's symbols: room for the per-segment limit to mean something
}

namespace rc {
bool ffv1_overlays_slice_buffers(const rcgpu_ffv1_config& cfg) { return overlay_fits(cfg); }
std::vector<uint8_t> ffv1_config_record_for(const rcgpu_ffv1_config& cfg)
{
    if (cfg.pixfmt >= RCGPU_PIX_COUNT || !cfg.num_h_slices || !cfg.num_v_slices) return {};
    return ffv1::config_record(stream_params_of(cfg));
}

size_t ffv1_max_packet_bytes_for(const rcgpu_ffv1_config& cfg)
{
    if (cfg.pixfmt >= RCGPU_PIX_COUNT || !cfg.num_h_slices || !cfg.num_v_slices) return 0;
    const pix_desc& d = pix(cfg.pixfmt);
    size_t cb = 0;
    for (uint32_t sy = 0; sy < cfg.num_v_slices; sy++)
        for (uint32_t sx = 0; sx < cfg.num_h_slices; sx++) {
            const uint32_t w = uint32_t(uint64_t(sx + 1) * cfg.width / cfg.num_h_slices) - uint32_t(uint64_t(sx) * cfg.width / cfg.num_h_slices);
            const uint32_t h = uint32_t(uint64_t(sy + 1) * cfg.height / cfg.num_v_slices) - uint32_t(uint64_t(sy) * cfg.height / cfg.num_v_slices);
            cb += 16 + slice_buffer_bytes(d, w, h, cfg.level == 1 ? 1 : 3, cfg.slice_buffer_div);
        }
    return (cb + 15) & ~size_t(15);
}
}  // namespace rc

// Small control transfers (decision counts to the host, window offsets back, packet sizes, the error word) go through a kernel on the
// encoder's stream, not through hipMemcpyAsync: a copy is a command for the copy engines, which the pipeline keeps busy with batches of
// 50 MB payload and packet copies; a kernel keeps the control path independent of their queues.  (No measurable difference on the
// host pipeline's rate, 530 against 534 frames/s; what that rate was waiting for were more copy streams, pipeline.hip.)
// Pinned host memory is mapped into the device's address space: a kernel reads and writes it directly.
namespace {
__global__ __launch_bounds__(256) void k_copy8(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, size_t n)
{
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}
}  // namespace
namespace rc {
hipError_t copy_by_kernel(void* dst, const void* src, size_t bytes, hipStream_t stream)       // bytes: a multiple of 8, both 8-byte aligned
{
    if (!bytes) return hipSuccess;
    const size_t n = bytes / 8;
    hipLaunchKernelGGL(k_copy8, dim3(unsigned(std::min<size_t>(1024, (n + 255) / 256))), dim3(256), 0, stream,
                       static_cast<unsigned long long*>(dst), static_cast<const unsigned long long*>(src), n);
    return hipGetLastError();
}
int copy_by_kernel_on(void* dst, const void* src, size_t bytes, void* hip_stream) { return int(copy_by_kernel(dst, src, bytes, static_cast<hipStream_t>(hip_stream))); }
}  // namespace rc

struct rcgpu_ffv1 {
    rcgpu_ffv1_config cfg{};
    ffv1::stream_params sp{};
    enc_const hc{};
    std::vector<slice_geom> geom;
    std::vector<uint8_t> record;
    uint32_t nkeys = 0, nseg = 1, resume_stride = 0;
    bool lds_states = false;            // compact context model: k_resolve keeps the slice's states in LDS
    size_t resolve_lds = 0;
    size_t frame_payload = 0, cbuf_frame_stride = 0, max_packet = 0;
    // OVERLAY (round 6): a slice's byte buffer lies inside its own symbol area.  The coder consumes segment j's decisions after k_resolve has read
    // segment j's symbols, and it writes at most 3.4 bytes where four bytes of symbol lay: the 96 MB of slice buffers per 4K frame (and per bank
    // in run-on mode) are the symbol buffer's dead front.  On when every slice's buffer fits its symbol area (large slices; small ones keep
    // buffers of their own: their caps are dominated by the constant room for untrained contexts).
    bool overlay = false;
    uint8_t* cbuf_base() const { return overlay ? reinterpret_cast<uint8_t*>(d_sym) : d_cbuf; }
    hipStream_t own_stream = nullptr, rc_stream = nullptr;      // rc_stream: k_rangecode runs beside k_resolve
    hipStream_t rr_stream = nullptr;                            // split coder: k_rc_range's stream (the spans are coded on rc_stream)
    uint32_t span_pieces = 0;                                   // split coder: pieces per span; 0 = one lane codes a whole slice
    uint32_t resolve_prio = 0;                                  // k_resolve's s_setprio
    bool exp_skip_rc = false;                                   // timing build only: no range coder at all (k_resolve alone; no valid output)
    rc_ckpt* d_ckpt = nullptr; size_t ckpt_cap = 0;             // [span of the batch][chain]
    std::vector<uint32_t> seg_spans, seg_span_off;
    // device buffers
    enc_const* d_const = nullptr; slice_geom* d_geom = nullptr; uint16_t* d_hdr = nullptr;
    const uint8_t** d_frame_ptrs = nullptr;
    int32_t* d_planes = nullptr; uint32_t* d_sym = nullptr; uint8_t* d_states = nullptr;
    unsigned long long* d_ndec = nullptr;          // [chain][seg] decisions coded by the samples of a segment
    unsigned long long* d_total_n = nullptr;       // [chain] all decisions incl. header and end bit
    uint32_t* d_seg_pieces = nullptr;              // [seg][chain] 64-byte pieces produced in a segment
    unsigned long long* d_group_off = nullptr;     // [seg][group] byte offset inside the segment's window
    uint8_t* d_k3_resume = nullptr; rc_resume* d_k4_resume = nullptr;
    uint8_t* d_window[kMaxWindows] = {}; size_t window_cap = 0; uint32_t nwin = 2;
    uint8_t* d_cbuf = nullptr; uint32_t* d_out_len = nullptr; uint32_t* d_tot_len = nullptr;
    unsigned long long* d_slice_dst = nullptr; uint32_t* d_err = nullptr; uint2* d_events = nullptr;
    // host staging for the convenience path
    uint8_t* d_in = nullptr; uint8_t* d_packets = nullptr; unsigned long long* d_psizes = nullptr;
    unsigned long long* h_psizes = nullptr;
    unsigned long long* h_ndec_pinned = nullptr; const void** h_frame_ptrs = nullptr;
    unsigned long long* h_total_n = nullptr; uint32_t* h_seg_pieces = nullptr; unsigned long long* h_group_off = nullptr;
    // instrumentation: start/stop event pairs on the stream each kernel is launched on
    static constexpr int kNumK = 9;
    std::vector<hipEvent_t> ev;                    // pairs, in launch order
    std::vector<int> ev_kernel;                    // kernel index of each pair
    size_t ev_used = 0;
    std::vector<hipEvent_t> ev_prev; std::vector<int> ev_kernel_prev; size_t ev_used_prev = 0;   // the call before: still readable while the next batch runs
    hipEvent_t ev_k3[2 * kMaxSeg]{}, ev_k4[2 * kMaxSeg]{}, ev_rr[kMaxSeg]{}, ev_fork = nullptr;     // k3 / k4: a ring over the segments of two batches
    unsigned long long gseg = 0;                   // segments issued so far: segment G uses window G % nwin and the events G % (2 * nseg)
    bool ev_valid = false;
    // RUN-ON mode (rcgpu_ffv1_set_run_on): the device does not go idle between two batches.  Today's step is k_model (29 ms, both
    // entropy kernels waiting), the first k_resolve segment (13 ms, the coder waiting), 32 range-coder segments, and footer / scan / gather
    // (14 ms, nothing else running): 56 ms of a 490 ms step in which the coder's chain -- the step's critical path -- stands still.  With
    // two BANKS of everything a batch owns (symbols, states, tables, coder output) batch k+1 is modelled on a stream of its own while
    // batch k is in flight, its k_resolve segments follow batch k's on the front stream, its coder segments follow batch k's on the coder's
    // stream, and batch k's footer / scan / gather run beside them on a tail stream.  The windows' ring and its events run through.
    // The d_* members above are the CURRENT bank (the batch issued last); `alt` is the other one.
    struct bank_t {
        const uint8_t** d_frame_ptrs = nullptr; uint32_t* d_sym = nullptr; uint8_t* d_states = nullptr;
        unsigned long long* d_ndec = nullptr; unsigned long long* d_total_n = nullptr; uint32_t* d_seg_pieces = nullptr; unsigned long long* d_group_off = nullptr;
        uint8_t* d_k3_resume = nullptr; rc_resume* d_k4_resume = nullptr; uint8_t* d_cbuf = nullptr; uint32_t* d_out_len = nullptr; uint32_t* d_tot_len = nullptr;
        unsigned long long* d_slice_dst = nullptr; uint32_t* d_err = nullptr; uint2* d_events = nullptr;
        rc_ckpt* d_ckpt = nullptr; size_t ckpt_cap = 0;       // split coder: a batch's checkpoints are read until its k_rc_tails has run
        hipEvent_t ev_done = nullptr;              // behind the last kernel of the bank's batch
        bool used = false, joined = true;          // a batch has run in it; a caller's stream has been made to wait for it
    } alt;
    hipEvent_t ev_done = nullptr; bool used = false, joined = true;       // the current bank's
    bool run_on = false, alt_allocated = false;
    hipStream_t model_stream = nullptr, front_stream = nullptr, tail_stream = nullptr;
    hipStream_t front_own = nullptr, chain_stream = nullptr;     // run-on mode with the split coder: k_resolve's own stream, k_rc_range's (high priority)
    hipEvent_t ev_in = nullptr, ev_model = nullptr, ev_tails = nullptr;
    static constexpr int kBatchEvents = 64;        // a timing event behind every batch's last kernel, on the stream that kernel ran on: rcgpu_ffv1_batch_intervals
    hipEvent_t ev_batch[kBatchEvents] = {}; unsigned long long nbatch = 0;
    uint64_t last_decisions = 0, last_packet_bytes = 0;
    uint32_t last_n = 0;
    hipEvent_t input_event = nullptr;              // pipeline, run-on mode: the next batch's frames are on the device when this event has happened
    hipEvent_t gather_wait = nullptr;              // pipeline: k_gather of the next batch waits for the previous batch's download
    bool defer_gather = false;                     // pipeline: encode_device stops after k_scan, ffv1_gather() follows later
    size_t in_stride = 0;
    uint32_t* h_err = nullptr;                     // pinned copy of d_err
};

static const char* const kKernelNames[rcgpu_ffv1::kNumK] = { "k_unpack", "k_model", "k_resolve", "k_rangecode", "k_footer", "k_scan", "k_gather", "k_rc_range", "k_rc_tails" };

extern "C" int rcgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void rcgpu_ffv1_destroy(rcgpu_ffv1* e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    void* bufs[] = { e->d_const, e->d_geom, e->d_hdr, e->d_frame_ptrs, e->d_planes, e->d_sym, e->d_states, e->d_ndec, e->d_total_n, e->d_seg_pieces,
                     e->d_group_off, e->d_k3_resume, e->d_k4_resume, e->d_cbuf, e->d_out_len, e->d_tot_len,
                     e->d_slice_dst, e->d_err, e->d_events, e->d_in, e->d_packets, e->d_psizes, e->d_ckpt };
    for (hipStream_t q : { e->model_stream, e->front_stream, e->tail_stream, e->rc_stream, e->rr_stream, e->chain_stream }) if (q) (void)hipStreamSynchronize(q);
    for (void* b : bufs) if (b) (void)hipFree(b);
    void* abufs[] = { e->alt.d_frame_ptrs, e->alt.d_sym, e->alt.d_states, e->alt.d_ndec, e->alt.d_total_n, e->alt.d_seg_pieces, e->alt.d_group_off, e->alt.d_k3_resume,
                      e->alt.d_k4_resume, e->alt.d_cbuf, e->alt.d_out_len, e->alt.d_tot_len, e->alt.d_slice_dst, e->alt.d_err, e->alt.d_events };
    for (void* b : abufs) if (b) (void)hipFree(b);
    if (e->alt.d_ckpt) (void)hipFree(e->alt.d_ckpt);
    for (hipEvent_t q : { e->ev_done, e->alt.ev_done, e->ev_in, e->ev_model, e->ev_tails }) if (q) (void)hipEventDestroy(q);
    for (hipStream_t q : { e->model_stream, e->tail_stream, e->front_own, e->chain_stream }) if (q) (void)hipStreamDestroy(q);       // (front_stream is rr_stream or front_own)
    for (uint8_t* w : e->d_window) if (w) (void)hipFree(w);
    void* hosts[] = { e->h_psizes, e->h_ndec_pinned, e->h_frame_ptrs, e->h_total_n, e->h_seg_pieces, e->h_group_off, e->h_err };
    for (void* h : hosts) if (h) (void)hipHostFree(h);
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_prev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_k3) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_k4) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : e->ev_rr) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (auto& ev : e->ev_batch) if (ev) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    if (e->rc_stream) (void)hipStreamDestroy(e->rc_stream);
    if (e->rr_stream) (void)hipStreamDestroy(e->rr_stream);
    delete e;
}

extern "C" int rcgpu_ffv1_create(const rcgpu_ffv1_config* cfg, rcgpu_ffv1** out)
{
    clear_error();
    if (!cfg || !out) return fail(1, "ffv1: null argument");
    *out = nullptr;
    if (cfg->pixfmt >= RCGPU_PIX_COUNT) return fail(2, "ffv1: unknown pixel format %u", cfg->pixfmt);
    if (!cfg->width || !cfg->height) return fail(2, "ffv1: empty picture");
    if ((unsigned long long)cfg->width * cfg->height * 4 >= (1ull << 32)) return fail(2, "ffv1: %ux%u: sample indices are 32 bit (a picture may hold 2^30 pixels)", cfg->width, cfg->height);
    if (!cfg->num_h_slices || !cfg->num_v_slices || cfg->num_h_slices < cfg->num_v_slices)
        return fail(2, "ffv1: slice grid %ux%u needs num_h >= num_v >= 1 (reference decoder limit, FFV1_Slice.cpp:127)", cfg->num_h_slices, cfg->num_v_slices);
    const uint32_t S = cfg->num_h_slices * cfg->num_v_slices;
    if (S > 1 && (cfg->num_h_slices >= cfg->width || cfg->num_v_slices >= cfg->height))
        return fail(2, "ffv1: more slices than pixels (FFV1_Frame.cpp:161-164)");
    if (!cfg->max_batch) return fail(2, "ffv1: max_batch is 0");
    if (cfg->segments > kMaxSeg) return fail(2, "ffv1: at most %u segments", kMaxSeg);
    if (cfg->coder > 2) return fail(2, "ffv1: coder %u (0/1 default transitions, 2 transmitted table)", cfg->coder);
    if (cfg->level != 0 && cfg->level != 1 && cfg->level != 3) return fail(2, "ffv1: level %u (1 or 3)", cfg->level);
    if (cfg->level == 1 && (S != 1 || cfg->slicecrc)) return fail(2, "ffv1: level 1 (FFV1 version 1) means one slice and no slice CRC");
    if (cfg->rc_span > RCGPU_RC_WHOLE && cfg->rc_span < 8) return fail(2, "ffv1: rc_span %u (0 automatic, %u whole slices, or at least 8 pieces per span)", cfg->rc_span, RCGPU_RC_WHOLE);
    // version 1 frames are one slice with no footer and a header replayed in front of the samples: coded by the whole-slice mapping only
    // (the automatic choice never splits them; the split mapping has no bit-exactness test at level 1)
    if (cfg->level == 1 && cfg->rc_span > RCGPU_RC_WHOLE) return fail(2, "ffv1: rc_span %u with level 1: FFV1 version 1 frames are coded by the whole-slice mapping", cfg->rc_span);
    const pix_desc& d = pix(cfg->pixfmt);
    const bool altern = (cfg->flags & RCGPU_FLAG_ALTERN) != 0;
    if (altern && d.fields != kFieldsLow) return fail(2, "ffv1: RCGPU_FLAG_ALTERN is a layout of the Y 10-bit flavors only (DPX.cpp:363-368)");
    if ((cfg->flags & RCGPU_FLAG_VFLIP) && altern) return fail(2, "ffv1: RCGPU_FLAG_VFLIP and RCGPU_FLAG_ALTERN exclude each other");
    if (!payload_line_bytes(cfg->pixfmt, cfg->width, true)) return fail(2, "ffv1: a line of %u pixels does not fit 32 bits", cfg->width);
    if (!altern && cfg->line_bytes < payload_line_bytes(cfg->pixfmt, cfg->width, false)) return fail(2, "ffv1: line_bytes smaller than a line");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(3, "ffv1: no HIP device available -- this encoder has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(3, "ffv1: device %d out of range (%d visible)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    rcgpu_ffv1* e = new rcgpu_ffv1;
    e->cfg = *cfg;
    e->sp = stream_params_of(*cfg);
    e->record = ffv1::config_record(e->sp);

    ffv1::quant_model qm[2];
    ffv1::build_quant_models(d.bits, qm, e->sp.compact);
    const ffv1::quant_model& Q = qm[e->sp.context_model];
    enc_const& c = e->hc;
    c.W = cfg->width; c.H = cfg->height; c.line_bytes = cfg->line_bytes; c.pixfmt = cfg->pixfmt;
    c.planes = d.planes; c.bps = d.bits; c.rgb = d.planes != 1; c.gb_swap = d.gb_swap; c.big_endian = d.big_endian; c.bytes_pp = d.bytes_pp;
    c.fields = d.fields; c.fill = d.fill; c.vflip = (cfg->flags & RCGPU_FLAG_VFLIP) != 0; c.altern = altern; c.v1 = cfg->level == 1;
    c.bits = c.rgb ? d.bits + 1 : (d.bits <= 8 ? 8 : d.bits);                  // FFV1_Parameters.cpp:164-181
    c.overflow16 = (!c.rgb && d.bits == 16);                                   // FFV1_Parameters.cpp:160
    c.num_h = cfg->num_h_slices; c.num_v = cfg->num_v_slices; c.S = S; c.nctx = Q.context_count;
    c.nsets = c.rgb ? (d.planes == 4 ? 3 : 2) : 1; c.ec = e->sp.ec; c.is5 = Q.q[3][127] != 0;
    c.samples_per_frame = cfg->width * cfg->height * d.planes;
    memcpy(c.q, Q.q, sizeof c.q);
    memcpy(c.one_state, ffv1::one_state_table(cfg->coder), 256);
    ffv1::make_zero_state(c.zero_state, c.one_state);
    if (c.nctx > 8191) { delete e; return fail(2, "ffv1: context count %u does not fit the symbol format", c.nctx); }
    e->nkeys = c.nsets * c.nctx;
    e->resolve_lds = ((e->nkeys + 31) / 32 + 3) / 4 * 16 * 2;                              // dynamic part (two bitmaps over the contexts); kResolveFixedLds is static
    e->resume_stride = 80;
    e->lds_states = size_t(e->nkeys) * 32 <= (48u << 10);                        // 338 contexts x 2 (3) sets x 32 B = 21.6 (32.4) KB
    // Workgroup LDS = kResolveFixedLds + two bitmaps: 9.4 KB with the 10 126 contexts of the default model, so that the registers (four
    // wavefronts per SIMD) and not the LDS bound k_resolve's occupancy.  A total can be forced to re-measure what occupancy is worth.
    // Range coder mapping.  One lane per slice is a serial chain of 21 VALU instructions per decision, 64 chains to a wavefront, each
    // wavefront alone on its SIMD's issue port: 0.41 s for a 4K slice whatever the batch.  With 300 and more such wavefronts in flight that
    // chain hides behind k_resolve (4096x2160 x 336 frames: 684-696 frames/s either way); with fewer, the chain sets the batch time and
    // the coder is split (k_rc_range / k_rangecode<true> / k_rc_tails: 168 frames per step 565 instead of 376 frames/s, 112: 459 / 259).
    {
        const size_t waves = (size_t(cfg->max_batch) * S + 63) / 64;
        e->span_pieces = cfg->rc_span == RCGPU_RC_WHOLE ? 0u : cfg->rc_span ? cfg->rc_span : (waves < 300 && e->sp.version != 1 ? 64u : 0u);
        e->exp_skip_rc = TIMING_ENV("RCGPU_EXP_SKIP_RC") != nullptr;
        if (const char* x = TIMING_ENV("RCGPU_RESOLVE_PRIO")) e->resolve_prio = uint32_t(atoi(x));        // for measuring
        if (TIMING_ENV("RCGPU_EXP_STATES_L2")) e->resolve_prio |= 0x100;                                  // k_resolve's floor without HBM latency (wrong bytes)
        if (const char* x = TIMING_ENV("RCGPU_RC_SPAN")) { const int v = atoi(x); e->span_pieces = v >= 8 ? uint32_t(v) : 0u; }      // for measuring
    }
    // Measured (4096x2160, 336 frames, round 3): k_resolve alone 390 / 372 / 408 / 431 / 509 / 617 ms per step at 16 / 12 / 10 / 8 / 6 / 4
    // wavefronts per CU; beside the whole-slice range coder 10 per CU (16 000 B) is best, beside the split coder 11-12 (13 600 B): the
    // other kernels' wavefronts need room in the CU too.
    {
        const size_t have = size_t(kResolveFixedLds) + e->resolve_lds;
        size_t want = e->lds_states ? 0 : e->span_pieces ? 13600 : 16000;
        if (const char* x = TIMING_ENV("RCGPU_RESOLVE_LDS_TOTAL")) want = size_t(atoi(x));
        if (want > have) e->resolve_lds += want - have;
    }
    if (e->lds_states) { e->resolve_lds += size_t(e->nkeys) * 32; e->resume_stride = uint32_t(80 + e->nkeys * 32); }
    e->frame_payload = size_t(payload_bytes(cfg->pixfmt, cfg->width, cfg->height, cfg->line_bytes, cfg->flags));

    // segments: the decision stream of a slice is produced and consumed in nseg windows (double buffered) instead of
    // being resident as a whole; auto = up to 32, at least 1024 symbols (16 chunks) of every slice per segment -- with 576 slices per
    // 4K frame that is what keeps the two windows of a 256-frame batch at 2 x 17 GB instead of 2 x 70 GB
    uint32_t min_nsamp = ~0u;
    for (uint32_t sy = 0; sy < c.num_v; sy++)
        for (uint32_t sx = 0; sx < c.num_h; sx++) {
            const uint32_t w = uint32_t(uint64_t(sx + 1) * c.W / c.num_h) - uint32_t(uint64_t(sx) * c.W / c.num_h);
            const uint32_t h = uint32_t(uint64_t(sy + 1) * c.H / c.num_v) - uint32_t(uint64_t(sy) * c.H / c.num_v);
            min_nsamp = std::min(min_nsamp, w * h * c.planes);
        }
    e->nseg = cfg->segments ? cfg->segments : std::max(1u, std::min(32u, min_nsamp / 1024));
    c.nseg = e->nseg;
    c.rc_prio = 3;
    if (const char* x = TIMING_ENV("RCGPU_RC_PRIO")) c.rc_prio = uint32_t(atoi(x));                          // for measuring
    // the split coder keeps a window until its spans are coded, two kernels behind k_resolve: a third window keeps k_resolve from waiting
    // (4096x2160, 168 frames per step: 476 frames/s with two windows, 565 with three or four)
    e->nwin = e->span_pieces ? kWindows + 1 : kWindows;
    if (const char* x = TIMING_ENV("RCGPU_WINDOWS")) e->nwin = uint32_t(std::min<int>(kMaxWindows, std::max(2, atoi(x))));       // for measuring

    // slice geometry (FFV1_Slice.cpp:153-156), header decisions, raw-byte buffers
    std::vector<uint16_t> hdr;
    uint32_t sym_off = 0; size_t cb = 0;
    for (uint32_t sy = 0; sy < c.num_v; sy++)
        for (uint32_t sx = 0; sx < c.num_h; sx++) {
            slice_geom g{};
            g.x0 = uint32_t(uint64_t(sx) * c.W / c.num_h); g.y0 = uint32_t(uint64_t(sy) * c.H / c.num_v);
            g.w = uint32_t(uint64_t(sx + 1) * c.W / c.num_h) - g.x0; g.h = uint32_t(uint64_t(sy + 1) * c.H / c.num_v) - g.y0;
            g.sym_off = sym_off; g.nsamp = g.w * g.h * c.planes; sym_off += g.nsamp;
            g.seg_q = ((g.nsamp + e->nseg - 1) / e->nseg + 63) & ~63u;
            const auto hd = e->sp.version == 1 ? ffv1::v1_frame_header_decisions(e->sp) : ffv1::slice_header_decisions(e->sp, sx, sy, e->geom.empty());
            g.hdr_off = uint32_t(hdr.size()); g.hdr_n = uint32_t(hd.size());
            if (g.hdr_n > uint32_t(kStageEntries)) { delete e; return fail(2, "ffv1: %u header decisions do not fit k_resolve's stage", g.hdr_n); }
            for (uint16_t d16 : hd) hdr.push_back(uint16_t(((d16 & 0x100) ? (d16 & 0xFF) : (256 - (d16 & 0xFF))) | (d16 & 0x100)));   // (state, bit) -> t | bit << 8
            const size_t cap = slice_buffer_bytes(d, g.w, g.h, e->sp.version, cfg->slice_buffer_div);
            if (cap >= (size_t(1) << 31)) { delete e; return fail(2, "ffv1: a version 1 frame of %ux%u does not fit the coder's 31-bit byte positions", g.w, g.h); }
            cb += 16;        // slack in front of every slice buffer: k_rangecode's first (empty) second-stage store lands here
            g.cbuf_off_lo = uint32_t(cb); g.cbuf_off_hi = uint32_t(uint64_t(cb) >> 32); g.cbuf_cap = uint32_t(cap);
            cb += cap;
            e->geom.push_back(g);
        }
    e->cbuf_frame_stride = cb;
    e->max_packet = (cb + 15) & ~size_t(15);
    {   // the overlay: every slice's 16 + cap bytes inside its nsamp * 4 bytes of symbols, and room for the per-segment limit to mean something
        const bool fits = overlay_fits(*cfg) && !TIMING_ENV("RCGPU_NO_OVERLAY");
        e->overlay = fits;
        if (fits) {
            for (slice_geom& g : e->geom) {
                const uint64_t off = uint64_t(g.sym_off) * 4 + 16; g.cbuf_off_lo = uint32_t(off); g.cbuf_off_hi = uint32_t(off >> 32);
                g.cbuf_cap = uint32_t(overlay_cap(g.cbuf_cap, g.nsamp));       // (the packet stride callers allocate, max_packet, keeps the caps of buffers of their own)
            }
            e->cbuf_frame_stride = size_t(c.samples_per_frame) * 4;
        }
        c.overlay = fits ? std::max(1u, cfg->slice_buffer_div) : 0u;
    }

    const uint32_t F = cfg->max_batch;
    const size_t nchains = size_t(F) * S, ngroups = (nchains + 63) / 64;
    const uint32_t nseg = e->nseg;
    auto dmalloc = [&](auto** p, size_t bytes) -> hipError_t { return hipMalloc(reinterpret_cast<void**>(p), bytes ? bytes : 16); };
    hipError_t he = hipSuccess;
#define DM(p, b) if (he == hipSuccess) he = dmalloc(&(p), (b))
#define HM(p, b) if (he == hipSuccess) he = hipHostMalloc(reinterpret_cast<void**>(&(p)), (b))
    DM(e->d_const, sizeof(enc_const)); DM(e->d_geom, sizeof(slice_geom) * S); DM(e->d_hdr, hdr.size() * 2 + 16);
    DM(e->d_frame_ptrs, sizeof(void*) * F);
    DM(e->d_sym, size_t(F) * c.samples_per_frame * 4);
    DM(e->d_states, e->lds_states ? 16 : nchains * e->nkeys * 32);
    DM(e->d_ndec, nchains * nseg * 8); DM(e->d_total_n, nchains * 8); DM(e->d_seg_pieces, nchains * nseg * 4 + 8); DM(e->d_group_off, ngroups * nseg * 8);
    DM(e->d_k3_resume, nchains * e->resume_stride); DM(e->d_k4_resume, nchains * sizeof(rc_resume));
    if (!e->overlay) DM(e->d_cbuf, size_t(F) * e->cbuf_frame_stride + 64);
    DM(e->d_out_len, nchains * 4); DM(e->d_tot_len, nchains * 4); DM(e->d_slice_dst, nchains * 8); DM(e->d_err, 16); DM(e->d_events, sizeof(uint2) * kMaxCarryEvents);
    HM(e->h_ndec_pinned, nchains * nseg * 8); HM(e->h_frame_ptrs, sizeof(void*) * F); HM(e->h_total_n, nchains * 8);
    HM(e->h_seg_pieces, nchains * nseg * 4 + 8); HM(e->h_group_off, ngroups * nseg * 8);
#undef DM
#undef HM
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    // Timing build, RCGPU_EXP_PARTITION=k[,shared]: the range coder's kernels on k CUs of every XCD, k_resolve (in run-on mode: the front stream)
    // on the other 32 - k -- or, with ",shared", on all of them.  Bit i of a CU mask is a CU of XCD i % 8 and a mask that leaves an XCD empty is
    // ignored whole (tools/cu_mask_probe.hip).  Same bytes; the shipped library makes plain streams (profiles/r05_partition.jsonl says why).
    if (const char* x = TIMING_ENV("RCGPU_EXP_PARTITION")) {
        const int k = std::max(1, std::min(31, atoi(x)));
        const bool shared = strstr(x, "shared") != nullptr;
        uint32_t mrc[8], mrs[8];
        for (int w = 0; w < 8; w++) { mrc[w] = 0; mrs[w] = 0; }
        for (int i = 0; i < 256; i++) { if (i < 8 * k) mrc[i >> 5] |= 1u << (i & 31); else mrs[i >> 5] |= 1u << (i & 31); }
        if (shared) for (int w = 0; w < 8; w++) mrs[w] = ~0u;
        if (he == hipSuccess) he = hipExtStreamCreateWithCUMask(&e->rc_stream, 8, mrc);
        if (he == hipSuccess) he = hipExtStreamCreateWithCUMask(&e->rr_stream, 8, mrs);
    }
    if (he == hipSuccess && !e->rc_stream) he = hipStreamCreateWithFlags(&e->rc_stream, hipStreamNonBlocking);
    if (he == hipSuccess && !e->rr_stream) he = hipStreamCreateWithFlags(&e->rr_stream, hipStreamNonBlocking);
    e->ev.resize(2 * (7 + 3 * nseg));              // timing events: k_model, tails, footer, scan, gather + up to three kernels per segment
    for (auto& ev : e->ev) if (he == hipSuccess) he = hipEventCreate(&ev);
    e->ev_prev.resize(e->ev.size());
    for (auto& ev : e->ev_prev) if (he == hipSuccess) he = hipEventCreate(&ev);
    for (uint32_t j = 0; j < 2 * nseg; j++) {
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_k3[j], hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_k4[j], hipEventDisableTiming);
        if (he == hipSuccess && j < nseg) he = hipEventCreateWithFlags(&e->ev_rr[j], hipEventDisableTiming);
    }
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    for (auto& ev : e->ev_batch) if (he == hipSuccess) he = hipEventCreate(&ev);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming);
    if (he == hipSuccess) he = hipMemcpy(e->d_const, &e->hc, sizeof(enc_const), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(e->d_geom, e->geom.data(), sizeof(slice_geom) * S, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(e->d_hdr, hdr.data(), hdr.size() * 2, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipFuncSetAttribute(reinterpret_cast<const void*>(e->lds_states ? k_resolve<true> : k_resolve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(e->resolve_lds));
    if (he != hipSuccess) {
        (void)hipGetLastError();                  // (the sticky error goes with this report: see HIP_TRY)
        const int r = fail(100, "ffv1: device setup failed: %s", hipGetErrorString(he));
        rcgpu_ffv1_destroy(e);
        return r;
    }
    *out = e;
    return 0;
}

extern "C" size_t rcgpu_ffv1_config_record(const rcgpu_ffv1* e, uint8_t* out, size_t cap)
{
    if (!e) return 0;
    if (out && cap >= e->record.size()) memcpy(out, e->record.data(), e->record.size());
    return e->record.size();
}

extern "C" size_t rcgpu_ffv1_max_packet_bytes(const rcgpu_ffv1* e) { return e ? e->max_packet : 0; }

extern "C" int rcgpu_ffv1_encode_device(rcgpu_ffv1* e, const void* const* d_frames, uint32_t n, void* d_packets, size_t packet_stride,
                                        uint64_t* d_packet_sizes, void* hip_stream)
{
    clear_error();
    if (!e || !d_frames || !d_packets || !d_packet_sizes) return fail(1, "ffv1: null argument");
    if (!n || n > e->cfg.max_batch) return fail(2, "ffv1: batch of %u frames (max_batch %u)", n, e->cfg.max_batch);
    if (packet_stride < e->max_packet || (packet_stride & 3)) return fail(2, "ffv1: packet_stride must be a multiple of 4 and >= %zu", e->max_packet);
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);       // NULL = the default stream: ordered after the caller's earlier work
    hipStream_t s2 = e->rc_stream;
    static const bool exp_serial = TIMING_ENV("RCGPU_EXP_SERIAL") != nullptr;      // timing runs: every kernel alone on the device, one after the other
    if (exp_serial) s2 = st;
    // run-on mode: this batch goes into the other bank; modelling, resolving and the batch's tail get streams of their own (see rcgpu_ffv1)
    const bool ro = e->run_on && !exp_serial && !e->exp_skip_rc;
    if (ro) {
#define SW(f) std::swap(e->f, e->alt.f)
        SW(d_frame_ptrs); SW(d_sym); SW(d_states); SW(d_ndec); SW(d_total_n); SW(d_seg_pieces); SW(d_group_off); SW(d_k3_resume); SW(d_k4_resume);
        SW(d_cbuf); SW(d_out_len); SW(d_tot_len); SW(d_slice_dst); SW(d_err); SW(d_events); SW(d_ckpt); SW(ckpt_cap); SW(ev_done); SW(used); SW(joined);
#undef SW
    }
    if (!ro) e->input_event = nullptr;                   // (one batch at a time: everything follows the caller's stream anyway)
    hipStream_t ms = ro ? e->model_stream : st;          // k_model and the tables
    hipStream_t fr = ro ? e->front_stream : st;          // k_resolve
    hipStream_t tl = ro ? e->tail_stream : s2;           // footer, scan, gather
    if (ro) {
        // the caller's frames are ready: behind everything on its stream so far -- or behind the event it named (ffv1_set_input_event:
        // the pipeline's stream carries the previous batch's gather, which must not hold this batch's modelling up)
        if (e->input_event) { HIP_TRY(hipStreamWaitEvent(ms, e->input_event, 0)); e->input_event = nullptr; }
        else { HIP_TRY(hipEventRecord(e->ev_in, st)); HIP_TRY(hipStreamWaitEvent(ms, e->ev_in, 0)); }
        if (e->used) HIP_TRY(hipStreamWaitEvent(ms, e->ev_done, 0));                 // the batch before the last, whose bank this is, is finished
    }
    const enc_const& c = e->hc;
    const uint32_t S = c.S, nchains = n * S, ngroups = (nchains + 63) / 64, nseg = e->nseg;
    std::swap(e->ev, e->ev_prev); std::swap(e->ev_kernel, e->ev_kernel_prev); e->ev_used_prev = e->ev_used;      // timing events alternate between two sets
    e->ev_used = 0; e->ev_kernel.clear();
    auto timed = [&](int kernel, hipStream_t stream, auto&& launch) -> hipError_t {
        if (e->ev_used + 2 > e->ev.size()) { launch(); return hipGetLastError(); }
        hipError_t r = hipEventRecord(e->ev[e->ev_used], stream);
        launch();
        if (r == hipSuccess) r = hipEventRecord(e->ev[e->ev_used + 1], stream);
        e->ev_used += 2; e->ev_kernel.push_back(kernel);
        return r;
    };

    for (uint32_t i = 0; i < n; i++) e->h_frame_ptrs[i] = d_frames[i];
    HIP_TRY(rc::copy_by_kernel(e->d_frame_ptrs, e->h_frame_ptrs, sizeof(void*) * n, ms));
    HIP_TRY(hipMemsetAsync(e->d_ndec, 0, size_t(nchains) * nseg * 8, ms));
    HIP_TRY(hipMemsetAsync(e->d_err, 0, 16, ms));
    uint32_t max_tiles = 0;
    for (const slice_geom& g : e->geom) max_tiles = std::max(max_tiles, ((g.w + kTileW - 1) / kTileW) * ((g.h + kTileR - 1) / kTileR));
    HIP_TRY(timed(1, ms, [&] { hipLaunchKernelGGL(k_model, dim3(max_tiles, nchains), dim3(256), size_t(c.planes) * kTileRows * kTileCols * 4, ms,
                                                  e->d_const, e->d_geom, e->d_frame_ptrs, e->d_sym, e->d_ndec); }));
    // states_coded = 0: every context starts at 128 (beside the host round trip below)
    if (!e->lds_states) HIP_TRY(hipMemsetAsync(e->d_states, 0x80, size_t(nchains) * e->nkeys * 32, ms));
    // The exact decision counts size the stream windows: one host round trip per batch.
    HIP_TRY(rc::copy_by_kernel(e->h_ndec_pinned, e->d_ndec, size_t(nchains) * nseg * 8, ms));
    HIP_TRY(hipStreamSynchronize(ms));
    uint64_t total_dec = 0;
    for (uint32_t chain = 0; chain < nchains; chain++) {
        unsigned long long D = e->geom[chain % S].hdr_n, done = 0;
        for (uint32_t j = 0; j < nseg; j++) {
            D += e->h_ndec_pinned[size_t(chain) * nseg + j];
            if (j + 1 == nseg) D += 1;                                                   // end-of-slice bit
            const unsigned long long pieces = j + 1 == nseg ? (D + kPieceEntries - 1) / kPieceEntries : D / kPieceEntries;
            e->h_seg_pieces[size_t(j) * nchains + chain] = uint32_t(pieces - done);
            done = pieces;
        }
        e->h_total_n[chain] = D;
        total_dec += D;
    }
    size_t window_need = 0;
    for (uint32_t j = 0; j < nseg; j++) {
        unsigned long long off = 0;
        for (uint32_t g = 0; g < ngroups; g++) {
            uint32_t mx = 0;
            for (uint32_t l = 0; l < 64 && g * 64 + l < nchains; l++) mx = std::max(mx, e->h_seg_pieces[size_t(j) * nchains + g * 64 + l]);
            e->h_group_off[size_t(j) * ngroups + g] = off;
            off += (unsigned long long)mx * kGroupPieceBytes;
        }
        window_need = std::max<size_t>(window_need, off);
    }
    e->last_decisions = total_dec;
    uint32_t total_spans = 0;
    if (e->span_pieces) {
        e->seg_spans.assign(nseg, 0); e->seg_span_off.assign(nseg, 0);
        for (uint32_t j = 0; j < nseg; j++) {
            uint32_t mx = 0;
            for (uint32_t chain = 0; chain < nchains; chain++) mx = std::max(mx, e->h_seg_pieces[size_t(j) * nchains + chain]);
            e->seg_span_off[j] = total_spans;
            e->seg_spans[j] = std::max(1u, (mx + e->span_pieces - 1) / e->span_pieces);
            if (e->seg_spans[j] > 65535u)      // the spans of a segment are the launch's grid.y
                return fail(2, "ffv1: rc_span %u cuts a segment of %u pieces into %u spans (at most 65535): raise rc_span or segments", e->span_pieces, mx, e->seg_spans[j]);
            total_spans += e->seg_spans[j];
        }
        const size_t need = size_t(total_spans) * nchains * sizeof(rc_ckpt);
        if (need > e->ckpt_cap) {
            // (run-on mode: this bank's checkpoints; its last batch has run -- the model stream waited for it above and the host for the model stream)
            if (e->d_ckpt) HIP_TRY(hipFree(e->d_ckpt));
            e->d_ckpt = nullptr; e->ckpt_cap = 0;
            const size_t want = need + need / 8 + (1u << 20);
            if (hipMalloc(reinterpret_cast<void**>(&e->d_ckpt), want) != hipSuccess) { (void)hipGetLastError(); return fail(101, "ffv1: cannot allocate %zu bytes of range-coder checkpoints for %u frames -- lower max_batch", want, n); }
            e->ckpt_cap = want;
        }
    }
    if (window_need > e->window_cap) {
        if (ro) for (hipStream_t q : { fr, s2, tl }) HIP_TRY(hipStreamSynchronize(q));       // the batch in flight is using the windows that are about to go
        for (auto& w : e->d_window) { if (w) HIP_TRY(hipFree(w)); w = nullptr; }
        e->window_cap = 0;
        const size_t want = window_need + window_need / 8 + (1u << 20);
        for (uint32_t k = 0; k < (nseg > 1 ? e->nwin : 1u); k++) {
            hipError_t he = hipMalloc(reinterpret_cast<void**>(&e->d_window[k]), want);
            if (he != hipSuccess) { (void)hipGetLastError(); return fail(101, "ffv1: cannot allocate %zu bytes for a decision-stream window of %u frames: %s -- lower max_batch", want, n, hipGetErrorString(he)); }
        }
        e->window_cap = want;
    }
    HIP_TRY(rc::copy_by_kernel(e->d_total_n, e->h_total_n, size_t(nchains) * 8, ms));
    HIP_TRY(rc::copy_by_kernel(e->d_seg_pieces, e->h_seg_pieces, (size_t(nchains) * nseg * 4 + 7) & ~size_t(7), ms));
    HIP_TRY(rc::copy_by_kernel(e->d_group_off, e->h_group_off, size_t(ngroups) * nseg * 8, ms));
    if (ro) { HIP_TRY(hipEventRecord(e->ev_model, ms)); HIP_TRY(hipStreamWaitEvent(fr, e->ev_model, 0)); }
    // k_resolve(seg j) on the caller's stream, k_rangecode(seg j) on rc_stream; window j % nwin is reused once k_rangecode(j - nwin) is done.
    // Split coder: k_rc_range(seg j) on rr_stream between the two -- it follows k_resolve(j) and its own previous segment, the spans of
    // segment j follow it and nothing else.
    hipStream_t s3 = exp_serial ? st : (ro && e->chain_stream) ? e->chain_stream : e->rr_stream;
    HIP_TRY(hipEventRecord(e->ev_fork, fr));
    HIP_TRY(hipStreamWaitEvent(s2, e->ev_fork, 0));
    if (e->span_pieces) HIP_TRY(hipStreamWaitEvent(s3, e->ev_fork, 0));
    // Segment G of the encoder's life (gseg + j) hands over through window G % nw, and its two events are G % (2 * nseg) of their rings:
    // the numbering runs through the batches, so that in run-on mode this batch's first segments wait for the previous batch's last ones
    // exactly as a batch's later segments wait for its earlier ones.
    const uint32_t nw = nseg > 1 ? e->nwin : 1u, ring = 2 * nseg;
    for (uint32_t j = 0; j < nseg; j++) {
        const unsigned long long G = e->gseg + j;
        hipEvent_t k3 = e->ev_k3[G % ring], k4 = e->ev_k4[G % ring];
        uint8_t* win = e->d_window[G % nw];
        if (G >= nw) HIP_TRY(hipStreamWaitEvent(fr, e->ev_k4[(G - nw) % ring], 0));
        HIP_TRY(timed(2, fr, [&] {
            if (e->lds_states) hipLaunchKernelGGL(k_resolve<true>, dim3(nchains), dim3(64), e->resolve_lds, fr, e->d_const, e->d_geom, e->d_hdr, e->d_sym, e->d_states,
                                                  e->d_group_off + size_t(j) * ngroups, win, e->nkeys, j, e->d_k3_resume, e->resume_stride, e->resolve_prio);
            else hipLaunchKernelGGL(k_resolve<false>, dim3(nchains), dim3(64), e->resolve_lds, fr, e->d_const, e->d_geom, e->d_hdr, e->d_sym, e->d_states,
                                    e->d_group_off + size_t(j) * ngroups, win, e->nkeys, j, e->d_k3_resume, e->resume_stride, e->resolve_prio); }));
        HIP_TRY(hipEventRecord(k3, fr));
        if (e->exp_skip_rc) { HIP_TRY(hipStreamWaitEvent(s2, k3, 0)); HIP_TRY(hipEventRecord(k4, s2)); continue; }    // timing runs: k_resolve alone
        if (e->span_pieces) {
            rc_ckpt* ck = e->d_ckpt + size_t(e->seg_span_off[j]) * nchains;
            const uint32_t nsp = e->seg_spans[j];
            HIP_TRY(hipStreamWaitEvent(s3, k3, 0));
            HIP_TRY(timed(7, s3, [&] { hipLaunchKernelGGL(k_rc_range, dim3(ngroups), dim3(64), 0, s3, e->d_const, e->d_total_n, e->d_seg_pieces + size_t(j) * nchains,
                                                          j, e->d_k4_resume, e->d_group_off + size_t(j) * ngroups, win, nchains, ck, e->span_pieces, nsp); }));
            HIP_TRY(hipEventRecord(e->ev_rr[j], s3));
            HIP_TRY(hipStreamWaitEvent(s2, e->ev_rr[j], 0));
            static const bool exp_skip_b = TIMING_ENV("RCGPU_EXP_SKIP_B") != nullptr;        // timing runs: no span coder (no valid output)
            if (!exp_skip_b)
            HIP_TRY(timed(3, s2, [&] { hipLaunchKernelGGL(k_rangecode<true>, dim3(ngroups, nsp), dim3(64), 0, s2, e->d_const, e->d_geom, e->d_total_n, e->d_seg_pieces + size_t(j) * nchains,
                                                          j, e->d_k4_resume, e->d_group_off + size_t(j) * ngroups, win, e->cbuf_base(),
                                                          (unsigned long long)e->cbuf_frame_stride, nchains, e->d_out_len, e->d_err, e->d_events, ck, e->span_pieces); }));
        } else {
            HIP_TRY(hipStreamWaitEvent(s2, k3, 0));
            // timing build, RCGPU_EXP_RC_LDS=bytes: dynamic LDS nobody uses, so that a CU takes at most 160 KB / bytes of the coder's workgroups
            // (40960: four, one per SIMD) -- the dispatcher otherwise packs a masked stream's wavefronts onto few SIMDs (profiles/r05_partition.jsonl)
            static const uint32_t exp_rc_lds = TIMING_ENV("RCGPU_EXP_RC_LDS") ? uint32_t(atoi(TIMING_ENV("RCGPU_EXP_RC_LDS"))) : 0u;
            HIP_TRY(timed(3, s2, [&] { hipLaunchKernelGGL(k_rangecode<false>, dim3(ngroups), dim3(64), exp_rc_lds, s2, e->d_const, e->d_geom, e->d_total_n, e->d_seg_pieces + size_t(j) * nchains,
                                                          j, e->d_k4_resume, e->d_group_off + size_t(j) * ngroups, win, e->cbuf_base(),
                                                          (unsigned long long)e->cbuf_frame_stride, nchains, e->d_out_len, e->d_err, e->d_events, static_cast<rc_ckpt*>(nullptr), 0u); }));
        }
        HIP_TRY(hipEventRecord(k4, s2));
    }
    const hipEvent_t k4_last = e->ev_k4[(e->gseg + nseg - 1) % ring];
    e->gseg += nseg;
    if (e->span_pieces && !e->exp_skip_rc)
        HIP_TRY(timed(8, s2, [&] { hipLaunchKernelGGL(k_rc_tails, dim3(ngroups), dim3(64), 0, s2, e->d_const, e->d_geom, e->d_ckpt, total_spans, nchains,
                                                      e->cbuf_base(), (unsigned long long)e->cbuf_frame_stride); }));
    // footer / scan / gather follow the last range-coder segment -- on its stream, or in run-on mode on the tail stream, beside the next
    // batch's first coder segments; then the caller's stream joins: this batch, or in run-on mode the batch before it
    if (tl != s2) {
        if (e->span_pieces && !e->exp_skip_rc) { HIP_TRY(hipEventRecord(e->ev_tails, s2)); HIP_TRY(hipStreamWaitEvent(tl, e->ev_tails, 0)); }    // (behind k_rc_tails, which follows the last span)
        else HIP_TRY(hipStreamWaitEvent(tl, k4_last, 0));
    }
    HIP_TRY(timed(4, tl, [&] { hipLaunchKernelGGL(k_footer, dim3(nchains), dim3(256), 0, tl, e->d_const, e->d_geom, e->cbuf_base(), (unsigned long long)e->cbuf_frame_stride,
                                                  e->d_out_len, e->d_tot_len, e->d_err, e->d_events); }));
    HIP_TRY(timed(5, tl, [&] { hipLaunchKernelGGL(k_scan, dim3(n), dim3(64), 0, tl, e->d_const, e->d_tot_len, e->d_slice_dst, reinterpret_cast<unsigned long long*>(d_packet_sizes)); }));
    if (!e->defer_gather) {
        if (e->gather_wait) { HIP_TRY(hipStreamWaitEvent(tl, e->gather_wait, 0)); e->gather_wait = nullptr; }
        HIP_TRY(timed(6, tl, [&] { hipLaunchKernelGGL(k_gather, dim3(nchains, 8), dim3(256), 0, tl, e->d_const, e->d_geom, e->cbuf_base(), (unsigned long long)e->cbuf_frame_stride,
                                                      e->d_tot_len, e->d_slice_dst, static_cast<uint8_t*>(d_packets), (unsigned long long)packet_stride); }));
    }
#ifdef RCGPU_TIMING_BUILD
    // a timing run that skipped a coder has no valid packets: the batch says so in its error word, whoever reads it
    if (e->exp_skip_rc || TIMING_ENV("RCGPU_EXP_SKIP_B") || TIMING_ENV("RCGPU_EXP_STATES_L2")) HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_err), 8, 1, tl));
#endif
    HIP_TRY(hipEventRecord(e->ev_done, tl));
    HIP_TRY(hipEventRecord(e->ev_batch[e->nbatch % rcgpu_ffv1::kBatchEvents], tl)); e->nbatch++;
    e->used = true; e->joined = false;
    if (!ro) { HIP_TRY(hipStreamWaitEvent(st, e->ev_done, 0)); e->joined = true; }
    if (e->alt.used && !e->alt.joined) { HIP_TRY(hipStreamWaitEvent(st, e->alt.ev_done, 0)); e->alt.joined = true; }
    HIP_TRY(hipGetLastError());
    e->ev_valid = true; e->last_n = n;
    return 0;
}

// Run-on mode: see rcgpu.h.  Switching it on allocates the second bank (symbols, states, tables, coder output: about as much again as
// the encoder holds per batch) and the three streams; switching it off waits for what is in flight.
extern "C" int rcgpu_ffv1_set_run_on(rcgpu_ffv1* e, int on)
{
    clear_error();
    if (!e) return fail(1, "ffv1: null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!on) {
        if (e->run_on) for (hipStream_t q : { e->model_stream, e->front_stream, e->rc_stream, e->tail_stream, e->chain_stream }) if (q) HIP_TRY(hipStreamSynchronize(q));
        e->run_on = false; e->joined = true; e->alt.joined = true;
        return 0;
    }
    if (!e->alt_allocated) {
        const uint32_t F = e->cfg.max_batch, S = e->hc.S, nseg = e->nseg;
        const size_t nchains = size_t(F) * S, ngroups = (nchains + 63) / 64;
        hipError_t he = hipSuccess;
        auto dm = [&](auto** p, size_t bytes) { if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void**>(p), bytes ? bytes : 16); };
        rcgpu_ffv1::bank_t& b = e->alt;
        dm(&b.d_frame_ptrs, sizeof(void*) * F); dm(&b.d_sym, size_t(F) * e->hc.samples_per_frame * 4); dm(&b.d_states, e->lds_states ? 16 : nchains * e->nkeys * 32);
        dm(&b.d_ndec, nchains * nseg * 8); dm(&b.d_total_n, nchains * 8); dm(&b.d_seg_pieces, nchains * nseg * 4 + 8); dm(&b.d_group_off, ngroups * nseg * 8);
        dm(&b.d_k3_resume, nchains * e->resume_stride); dm(&b.d_k4_resume, nchains * sizeof(rc_resume)); if (!e->overlay) dm(&b.d_cbuf, size_t(F) * e->cbuf_frame_stride + 64);
        dm(&b.d_out_len, nchains * 4); dm(&b.d_tot_len, nchains * 4); dm(&b.d_slice_dst, nchains * 8); dm(&b.d_err, 16); dm(&b.d_events, sizeof(uint2) * kMaxCarryEvents);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_model, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_tails, hipEventDisableTiming);
        // Five streams are busy or waiting at any time (the caller's, model, front, coder, tail) and a stream that waits holds up whatever
        // shares its hardware queue: ROCm deals the streams of ONE priority to four hardware queues (GPU_MAX_HW_QUEUES), those of another
        // priority to four others.  Model and tail, the background of the batch in flight, take the low priority's; measured with all five
        // at one priority: k_resolve and k_rangecode shared a queue and ran one after the other, 820 ms per step instead of 480 (and with
        // k_resolve at the HIGH priority its workgroups are dispatched before the coder's every time: 712 ms).
        // (Round 6 measured k_model's stream at the other priorities, profiles/r06_model_priority.jsonl: at the normal one its 29 ms of work take 45 ms,
        // at the low or the high one they trickle through in 375-450 ms beside the batch in flight -- and the step is the same 470-480 ms either
        // way: the round trip that follows k_model hides behind the batch in flight as long as k_model ends before that batch does.)
        int lo = 0, hi = 0;
        if (he == hipSuccess) he = hipDeviceGetStreamPriorityRange(&lo, &hi);
        int model_prio = lo;
        if (const char* x = TIMING_ENV("RCGPU_MODEL_PRIO")) model_prio = atoi(x) < 0 ? hi : atoi(x) > 0 ? lo : 0;      // for measuring: -1 high, 0 normal, 1 low
        if (he == hipSuccess) he = hipStreamCreateWithPriority(&e->model_stream, hipStreamNonBlocking, model_prio);
        if (he == hipSuccess) he = hipStreamCreateWithPriority(&e->tail_stream, hipStreamNonBlocking, lo);
        e->front_stream = e->rr_stream;             // the split coder's stream, idle in this mode: no fifth stream at the normal priority
        if (e->span_pieces) {
            // The split coder in run-on mode (round 6) keeps SIX streams busy: k_rc_range -- the one serial chain left, nine dependent instructions
            // per decision -- gets a stream at the HIGH priority (hardware queues of its own, and its few wavefronts are dispatched first), k_resolve a
            // stream of its own at the normal priority beside the spans' (rc_stream) and the caller's.
            if (he == hipSuccess) he = hipStreamCreateWithPriority(&e->chain_stream, hipStreamNonBlocking, hi);
            if (he == hipSuccess) he = hipStreamCreateWithPriority(&e->front_own, hipStreamNonBlocking, 0);
            e->front_stream = e->front_own;
        }
        if (he != hipSuccess) {
            // nothing of a half-made bank stays behind: the memory it holds is what the first batch's windows need, and a second attempt
            // would overwrite (leak) the pointers
            void* abufs[] = { b.d_frame_ptrs, b.d_sym, b.d_states, b.d_ndec, b.d_total_n, b.d_seg_pieces, b.d_group_off, b.d_k3_resume, b.d_k4_resume, b.d_cbuf,
                              b.d_out_len, b.d_tot_len, b.d_slice_dst, b.d_err, b.d_events };
            for (void* q : abufs) if (q) (void)hipFree(q);
            b.d_frame_ptrs = nullptr; b.d_sym = nullptr; b.d_states = nullptr; b.d_ndec = nullptr; b.d_total_n = nullptr; b.d_seg_pieces = nullptr; b.d_group_off = nullptr;
            b.d_k3_resume = nullptr; b.d_k4_resume = nullptr; b.d_cbuf = nullptr; b.d_out_len = nullptr; b.d_tot_len = nullptr; b.d_slice_dst = nullptr; b.d_err = nullptr; b.d_events = nullptr;
            for (hipEvent_t* q : { &b.ev_done, &e->ev_in, &e->ev_model, &e->ev_tails }) if (*q) { (void)hipEventDestroy(*q); *q = nullptr; }
            for (hipStream_t* q : { &e->model_stream, &e->tail_stream, &e->front_own, &e->chain_stream }) if (*q) { (void)hipStreamDestroy(*q); *q = nullptr; }
            e->front_stream = nullptr;
            e->run_on = false;
            (void)hipGetLastError();
            return fail(101, "ffv1: run-on mode: cannot allocate the second bank for %u frames: %s -- lower max_batch", F, hipGetErrorString(he));
        }
        e->alt_allocated = true;
    }
    e->run_on = true;
    return 0;
}

// Makes `hip_stream` wait for every batch issued so far (run-on mode leaves the last one unjoined).
extern "C" int rcgpu_ffv1_run_on(const rcgpu_ffv1* e) { return e && e->run_on ? 1 : 0; }

extern "C" int rcgpu_ffv1_join(rcgpu_ffv1* e, void* hip_stream)
{
    clear_error();
    if (!e) return fail(1, "ffv1: null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (e->alt.used && !e->alt.joined) { HIP_TRY(hipStreamWaitEvent(st, e->alt.ev_done, 0)); e->alt.joined = true; }
    if (e->used && !e->joined) { HIP_TRY(hipStreamWaitEvent(st, e->ev_done, 0)); e->joined = true; }
    return 0;
}

namespace rc {
void ffv1_set_defer_gather(rcgpu_ffv1* e, bool on) { if (e) e->defer_gather = on; }

// In run-on mode the batch's footer and scan ran on the tail stream: the caller's stream waits for them first, and the bank's next user
// waits for ev_done, which is therefore recorded again behind the gather.
int ffv1_gather(rcgpu_ffv1* e, void* d_packets, size_t packet_stride, void* hip_stream)
{
    if (!e || !d_packets || !e->ev_valid) return fail(1, "ffv1: gather without a batch");
    if (packet_stride < e->max_packet || (packet_stride & 3)) return fail(2, "ffv1: packet_stride must be a multiple of 4 and >= %zu", e->max_packet);
    HIP_TRY(hipSetDevice(e->cfg.device));
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    const uint32_t nchains = e->last_n * e->hc.S;
    if (e->gather_wait) { HIP_TRY(hipStreamWaitEvent(st, e->gather_wait, 0)); e->gather_wait = nullptr; }
    if (e->run_on) { HIP_TRY(hipStreamWaitEvent(st, e->ev_done, 0)); e->joined = true; }
    const bool ev_room = e->ev_used + 2 <= e->ev.size();
    if (ev_room) HIP_TRY(hipEventRecord(e->ev[e->ev_used], st));
    hipLaunchKernelGGL(k_gather, dim3(nchains, 8), dim3(256), 0, st, e->d_const, e->d_geom, e->cbuf_base(), (unsigned long long)e->cbuf_frame_stride,
                       e->d_tot_len, e->d_slice_dst, static_cast<uint8_t*>(d_packets), (unsigned long long)packet_stride);
    if (ev_room) { HIP_TRY(hipEventRecord(e->ev[e->ev_used + 1], st)); e->ev_used += 2; e->ev_kernel.push_back(6); }
    if (e->run_on) HIP_TRY(hipEventRecord(e->ev_done, st));
    HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace rc

// Device time from the end of one batch to the end of the next, for the last batches issued (their events are waited for): what a caller that issues
// batch after batch reads a step's duration from without putting anything of its own between the batches.  ms[0] is the oldest interval.
extern "C" int rcgpu_ffv1_batch_intervals(const rcgpu_ffv1* e, float* ms, int cap)
{
    if (!e || !ms || cap <= 0 || e->nbatch < 2) return 0;
    (void)hipSetDevice(e->cfg.device);
    const unsigned long long have = std::min<unsigned long long>(e->nbatch, rcgpu_ffv1::kBatchEvents);
    const int n = int(std::min<unsigned long long>(have - 1, (unsigned long long)cap));
    if (hipEventSynchronize(e->ev_batch[(e->nbatch - 1) % rcgpu_ffv1::kBatchEvents]) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (int i = 0; i < n; i++) {
        const unsigned long long b = e->nbatch - 1 - (unsigned long long)(n - 1 - i);       // the batch whose end closes interval i
        float t = 0;
        if (hipEventElapsedTime(&t, e->ev_batch[(b - 1) % rcgpu_ffv1::kBatchEvents], e->ev_batch[b % rcgpu_ffv1::kBatchEvents]) != hipSuccess) { (void)hipGetLastError(); return i; }
        ms[i] = t;
    }
    return n;
}

// Sum of the device time of every launch of each kernel in the last encode call (HIP events on the launch stream).
extern "C" int rcgpu_ffv1_last_kernel_times(const rcgpu_ffv1* e, const char** names, float* ms, int cap)
{
    if (!e || !e->ev_valid || !e->ev_used) return 0;
    for (size_t i = 0; i < e->ev_used; i++) if (hipEventSynchronize(e->ev[i]) != hipSuccess) return 0;
    int k = 0;
    for (; k < rcgpu_ffv1::kNumK && k < cap; k++) { names[k] = kKernelNames[k]; ms[k] = 0; }
    for (size_t i = 0; i < e->ev_kernel.size(); i++) {
        float t = 0;
        if (hipEventSynchronize(e->ev[2 * i + 1]) != hipSuccess) continue;
        (void)hipEventElapsedTime(&t, e->ev[2 * i], e->ev[2 * i + 1]);
        if (e->ev_kernel[i] < k) ms[e->ev_kernel[i]] += t;
    }
    return k;
}

namespace rc {
// The same for the call BEFORE the last one (the pipeline starts batch k+1 before batch k has finished).
int ffv1_prev_kernel_times(const rcgpu_ffv1* e, const char** names, float* ms, int cap)
{
    if (!e || !e->ev_used_prev) return 0;
    int k = 0;
    for (; k < rcgpu_ffv1::kNumK && k < cap; k++) { names[k] = kKernelNames[k]; ms[k] = 0; }
    for (size_t i = 0; i < e->ev_kernel_prev.size() && 2 * i + 1 < e->ev_used_prev; i++) {
        float t = 0;
        if (hipEventSynchronize(e->ev_prev[2 * i + 1]) != hipSuccess) continue;
        (void)hipEventElapsedTime(&t, e->ev_prev[2 * i], e->ev_prev[2 * i + 1]);
        if (e->ev_kernel_prev[i] < k) ms[e->ev_kernel_prev[i]] += t;
    }
    return k;
}
// Device-side timeline of the call before the last one, in ms from the start of its k_model: [0] first k_resolve starts, [1] last k_resolve
// ends, [2] sum of the gaps between consecutive k_resolve launches, [3] last k_rangecode ends, [4] k_gather (or k_scan) ends, [5] the NEXT
// call's k_model starts.  For RCGPU_TRACE.
int ffv1_prev_timeline(const rcgpu_ffv1* e, float* t6)
{
    if (!e || !e->ev_used_prev || !e->ev_used) return 0;
    for (int i = 0; i < 6; i++) t6[i] = 0;
    hipEvent_t z = e->ev_prev[0];
    auto at = [&](hipEvent_t ev) { float t = 0; if (hipEventSynchronize(ev) != hipSuccess) return 0.f; (void)hipEventElapsedTime(&t, z, ev); return t; };
    float last_end = 0; bool first = true;
    for (size_t i = 0; i < e->ev_kernel_prev.size() && 2 * i + 1 < e->ev_used_prev; i++) {
        const float s0 = at(e->ev_prev[2 * i]), s1 = at(e->ev_prev[2 * i + 1]);
        const int k = e->ev_kernel_prev[i];
        if (k == 2) { if (first) { t6[0] = s0; first = false; } else t6[2] += std::max(0.f, s0 - last_end); last_end = s1; t6[1] = s1; }
        if (k == 3) t6[3] = s1;
        if (k >= 4) t6[4] = std::max(t6[4], s1);
    }
    t6[5] = at(e->ev[0]);
    return 6;
}
}  // namespace rc

// launches of kernel `index` in the last encode call (k_resolve / k_rangecode run once per segment)
extern "C" int rcgpu_ffv1_last_kernel_launches(const rcgpu_ffv1* e, int index)
{
    if (!e) return 0;
    int n = 0;
    for (int k : e->ev_kernel) n += k == index;
    return n;
}

extern "C" int rcgpu_ffv1_last_stats(const rcgpu_ffv1* e, uint64_t* decisions, uint64_t* packet_bytes)
{
    if (!e) return 1;
    if (decisions) *decisions = e->last_decisions;
    if (packet_bytes) *packet_bytes = e->last_packet_bytes;
    return 0;
}

namespace rc {
const char* ffv1_error_flags_text(uint32_t flags)
{
    if (!flags) return "";
    if (flags & 1u) return "a slice outgrew its byte buffer (content expands beyond 1.5x raw -- or, where the slices' bytes share the symbol buffer, beyond 4 bytes per sample early in a slice: RCGPU_FLAG_OWN_SLICE_BUFFERS, the shim's -rcgpu_own_slice_buffers 1)";
    if (flags & 2u) return "a slice does not fit its footer / the 24-bit slice size field";
    if (flags & 4u) return "more late carries than the event table holds";
    if (flags & 8u) return "timing build: a coder was switched off, the packets are not FFV1";
    return "unknown device error";
}

int ffv1_staging(rcgpu_ffv1* e, enc_staging* out)
{
    if (!e || !out) return fail(1, "ffv1: null argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint32_t F = e->cfg.max_batch;
    if (!e->d_in) {
        e->in_stride = (e->frame_payload + 255) & ~size_t(255);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_in), e->in_stride * F));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_packets), e->max_packet * F));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_psizes), 2 * 8 * F));          // two batches' worth: the pipeline alternates
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_psizes), 8 * F));
    }
    out->d_in = e->d_in; out->in_stride = e->in_stride; out->payload_bytes = e->frame_payload;
    out->d_packets = e->d_packets; out->packet_stride = e->max_packet; out->d_psizes = reinterpret_cast<uint64_t*>(e->d_psizes);
    out->d_err = e->d_err; out->compute_stream = e->own_stream; out->max_batch = F; out->device = e->cfg.device;
    return 0;
}

void ffv1_set_gather_wait(rcgpu_ffv1* e, void* hip_event) { if (e) e->gather_wait = static_cast<hipEvent_t>(hip_event); }
void ffv1_set_input_event(rcgpu_ffv1* e, void* hip_event) { if (e) e->input_event = static_cast<hipEvent_t>(hip_event); }
}  // namespace rc

// Error word of the last batch: bit 0 a slice outgrew its byte buffer, bit 1 a slice does not fit its footer or the 24-bit size
// field (its packet is incomplete), bit 2 more than 4096 late carries.  Synchronises the encoder's streams.
extern "C" int rcgpu_ffv1_last_error_flags(rcgpu_ffv1* e, uint32_t* flags)
{
    clear_error();
    if (!e || !flags) return fail(1, "ffv1: null argument");
    *flags = 0;
    if (!e->ev_valid) return 0;
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!e->h_err) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->h_err), 16));
    // the batch's last kernels run on rc_stream; the caller's stream joined it at the end of the call
    HIP_TRY(hipStreamSynchronize(e->rc_stream));
    if (e->used) HIP_TRY(hipEventSynchronize(e->ev_done));       // (run-on mode: the batch's footer runs on the tail stream)
    HIP_TRY(hipMemcpy(e->h_err, e->d_err, 16, hipMemcpyDeviceToHost));
    *flags = e->h_err[0];
    if (*flags) return fail(102, "ffv1: %s (flags %u)", ffv1_error_flags_text(*flags), *flags);
    return 0;
}

extern "C" int rcgpu_ffv1_encode_host(rcgpu_ffv1* e, const uint8_t* const* frames, uint32_t n, uint8_t* const* out_packets, size_t* out_sizes)
{
    clear_error();
    if (!e || !frames || !out_packets || !out_sizes) return fail(1, "ffv1: null argument");
    if (!n || n > e->cfg.max_batch) return fail(2, "ffv1: batch of %u frames (max_batch %u)", n, e->cfg.max_batch);
    enc_staging sg;
    if (int r = ffv1_staging(e, &sg)) return r;
    hipStream_t st = e->own_stream;
    std::vector<const void*> ptrs(n);
    for (uint32_t i = 0; i < n; i++) {
        // straight from the caller's (pageable, typically memory-mapped) buffer: the runtime stages it in chunks.  Callers with a
        // sequence to encode use rcgpu_ffv1_encode_sequence, which overlaps pinned uploads, encoding and downloads.
        HIP_TRY(hipMemcpyAsync(sg.d_in + i * sg.in_stride, frames[i], e->frame_payload, hipMemcpyHostToDevice, st));
        ptrs[i] = sg.d_in + i * sg.in_stride;
    }
    if (int r = rcgpu_ffv1_encode_device(e, ptrs.data(), n, e->d_packets, e->max_packet, reinterpret_cast<uint64_t*>(e->d_psizes), st)) return r;
    if (e->run_on) if (int r = rcgpu_ffv1_join(e, st)) return r;      // run-on mode leaves the batch unjoined: its coder, footer, scan and gather are still on their streams
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(e->h_psizes, e->d_psizes, 8 * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&err, e->d_err, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (err) return fail(102, "ffv1: %s (flags %u)", ffv1_error_flags_text(err), err);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; i++) {
        out_sizes[i] = size_t(e->h_psizes[i]);
        total += e->h_psizes[i];
        HIP_TRY(hipMemcpyAsync(out_packets[i], e->d_packets + i * e->max_packet, out_sizes[i], hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    e->last_packet_bytes = total;
    return 0;
}

// `-f framemd5` (CLI/Output.cpp:312-332): MD5 of the frames of the LAST batch as FFmpeg's rawvideo bytes (k_rawvideo).  The payloads are
// still where the encoder read them (the caller's device buffers, or the staging area of rcgpu_ffv1_encode_host); the symbol buffer,
// free between batches, holds the rawvideo bytes while they are hashed.
extern "C" int rcgpu_ffv1_framemd5_last(rcgpu_ffv1* e, uint32_t n, uint8_t* out_md5 /* n x 16, host */, uint64_t* frame_bytes)
{
    clear_error();
    if (!e || !out_md5 || !n) return fail(1, "ffv1: null argument");
    if (n > e->last_n) return fail(2, "ffv1: framemd5 of %u frames, the last batch had %u", n, e->last_n);
    const enc_const& c = e->hc;
    if (c.fields == kFieldsExr) return fail(2, "ffv1: framemd5 of EXR input is not supported");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const uint64_t bytes = uint64_t(c.samples_per_frame) * (c.bps == 8 ? 1 : 2);
    const size_t stride = size_t(c.samples_per_frame) * 4;
    hipStream_t st = e->own_stream;
    if (e->run_on) if (int r = rcgpu_ffv1_join(e, st)) return r;      // the rawvideo bytes go into the bank's symbol buffer: not while its k_resolve may still be reading it
    hipLaunchKernelGGL(k_rawvideo, dim3((c.W * c.H + 255) / 256, n), dim3(256), 0, st, e->d_const, e->d_frame_ptrs, reinterpret_cast<uint8_t*>(e->d_sym), stride);
    HIP_TRY(hipGetLastError());
    std::vector<const void*> bufs(n); std::vector<uint64_t> sizes(n, bytes);
    for (uint32_t i = 0; i < n; i++) bufs[i] = reinterpret_cast<const uint8_t*>(e->d_sym) + size_t(i) * stride;
    if (frame_bytes) *frame_bytes = bytes;
    return rcgpu_md5_device(bufs.data(), sizes.data(), n, out_md5, st);
}

// Debug taps for the stage-by-stage parity tests (tests/test_gpu_stages.py): copies an intermediate of the LAST
// batch to the host.  what: 0 planes (int32, first sub-batch only), 1 symbols (u32), 2 per-chain decision counts (u64),
// 3 decision stream of `chain` de-interleaved (u16; needs segments == 1), 4 raw slice bytes of `chain` before the footer.
extern "C" long long rcgpu_ffv1_debug_fetch(rcgpu_ffv1* e, int what, uint32_t chain, void* dst, size_t cap)
{
    if (!e || !e->ev_valid) return -1;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    const enc_const& c = e->hc;
    const uint32_t S = c.S;
    auto d2h = [&](const void* src, size_t bytes) -> long long {
        if (bytes > cap) return -2;
        return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? (long long)bytes : -3;
    };
    switch (what) {
    case 0: {      // int32 planes of the first frames of the last batch: k_unpack on demand (the caller's frame buffers must still be there)
        const uint32_t nf = std::min(e->last_n, kSubBatch);
        int32_t* planes = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&planes), size_t(nf) * c.samples_per_frame * 4) != hipSuccess) return -3;
        hipLaunchKernelGGL(k_unpack, dim3((c.W * c.H + 255) / 256, nf), dim3(256), 0, nullptr, e->d_const, e->d_frame_ptrs, planes, 0u);
        const long long r = hipDeviceSynchronize() == hipSuccess ? d2h(planes, size_t(nf) * c.samples_per_frame * 4) : -3;
        (void)hipFree(planes);
        return r;
    }
    case 1: {
        if (!e->overlay) return d2h(e->d_sym, size_t(e->last_n) * c.samples_per_frame * 4);
        // the slices' bytes have taken the symbols' place: the batch is modelled again into a buffer of its own (the caller's frames must still be there)
        uint32_t* sym = nullptr; unsigned long long* nd = nullptr;
        const uint32_t nchains = e->last_n * S;
        if (hipMalloc(reinterpret_cast<void**>(&sym), size_t(e->last_n) * c.samples_per_frame * 4) != hipSuccess) return -3;
        if (hipMalloc(reinterpret_cast<void**>(&nd), size_t(nchains) * e->nseg * 8) != hipSuccess) { (void)hipFree(sym); return -3; }
        (void)hipMemset(nd, 0, size_t(nchains) * e->nseg * 8);
        uint32_t max_tiles = 0;
        for (const slice_geom& g : e->geom) max_tiles = std::max(max_tiles, ((g.w + kTileW - 1) / kTileW) * ((g.h + kTileR - 1) / kTileR));
        hipLaunchKernelGGL(k_model, dim3(max_tiles, nchains), dim3(256), size_t(c.planes) * kTileRows * kTileCols * 4, nullptr, e->d_const, e->d_geom, e->d_frame_ptrs, sym, nd);
        const long long r = hipDeviceSynchronize() == hipSuccess ? d2h(sym, size_t(e->last_n) * c.samples_per_frame * 4) : -3;
        (void)hipFree(sym); (void)hipFree(nd);
        return r;
    }
    case 2: return d2h(e->d_total_n, size_t(e->last_n) * S * 8);
    case 3: {
        if (chain >= e->last_n * S) return -4;
        if (e->nseg != 1) return -6;
        const uint32_t nchains = e->last_n * S;
        const unsigned long long nd = e->h_total_n[chain];
        const size_t pieces = e->h_seg_pieces[chain];
        if (nd * 2 > cap) return -2;
        std::vector<uint8_t> tmp(pieces * kPieceBytes);
        const uint8_t* base = e->d_window[0] + e->h_group_off[chain >> 6] + (chain & 63) * kPieceBytes;
        (void)nchains;
        if (hipMemcpy2D(tmp.data(), kPieceBytes, base, kGroupPieceBytes, kPieceBytes, pieces, hipMemcpyDeviceToHost) != hipSuccess) return -3;
        uint16_t* o = static_cast<uint16_t*>(dst);
        for (unsigned long long i = 0; i < nd; i++) {        // piece bytes -> state | bit << 8, the oracle's trace form
            const uint8_t* pcs = tmp.data() + (i / kPieceEntries) * kPieceBytes;
            const uint32_t k = uint32_t(i % kPieceEntries), t = pcs[k], bit = (pcs[kPieceEntries + (k >> 3)] >> (k & 7)) & 1;
            o[i] = bit ? uint16_t(t | 0x100) : uint16_t(256 - t);
        }
        return (long long)(nd * 2);
    }
    case 4: {
        if (chain >= e->last_n * S) return -4;
        uint32_t len = 0;
        if (hipMemcpy(&len, e->d_out_len + chain, 4, hipMemcpyDeviceToHost) != hipSuccess) return -3;
        const slice_geom& g = e->geom[chain % S];
        const uint8_t* src = e->cbuf_base() + size_t(chain / S) * e->cbuf_frame_stride + (size_t(g.cbuf_off_hi) << 32 | g.cbuf_off_lo);
        return d2h(src, len);
    }
    case 5: return d2h(e->d_err, 16);      // [0] error flags, [1] carries k_rangecode handed to k_footer (events) in the last call
    default: return -5;
    }
}

#ifdef RCGPU_PROF
extern "C" int rcgpu_debug_prof(unsigned long long* out)      // timing build only: see PROF_T
{
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof z) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z) != hipSuccess;
}
#endif
