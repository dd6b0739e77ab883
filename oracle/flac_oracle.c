/* flac_oracle.c -- CPU oracle for the WAV -> FLAC path.  TEST INFRASTRUCTURE ONLY (see flac_oracle.h).
 *
 * Decoder half: restates the FLAC bitstream exactly as the reference's vendored libFLAC parses it
 *   (Source/Lib/ThirdParty/flac/src/libFLAC/stream_decoder.c: frame header :2159-2466, subframes :2468-2743,
 *    residual :2745-2788, footer :2022-2078; bitreader.c:716-742 for Rice codes; lpc.c:784-812, fixed.c:367-393 for
 *    signal restoration; Source/Lib/CoDec/Wrapper.cpp:247-373 for the WAV byte layout).
 * Encoder half: the deterministic rule the device kernels follow (flac_gpu.hip); every step is integer-exact except
 *   Levinson-Durbin and the coefficient quantiser, which are plain IEEE double +,-,*,/ in a fixed order
 *   (build with -ffp-contract=off on both sides).
 */
#include "flac_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------- CRCs, MD5 */
uint8_t flaco_crc8(const uint8_t* d, size_t n)
{
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= d[i]; for (int k = 0; k < 8; k++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1); }
    return c;
}
uint16_t flaco_crc16(const uint8_t* d, size_t n)
{
    uint16_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= (uint16_t)(d[i] << 8); for (int k = 0; k < 8; k++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1); }
    return c;
}
static inline uint32_t rol32(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
void flaco_md5(const uint8_t* data, size_t size, uint8_t out[16])
{
    static const uint8_t S[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                                   4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
    uint32_t K[64];
    for (int i = 0; i < 64; i++) {          /* floor(2^32 * |sin(i+1)|) without libm: use the published constants' generator */
        static const uint32_t Kc[64] = {
            0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
            0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
            0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
            0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
        K[i] = Kc[i];
    }
    uint32_t h[4] = { 0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476 };
    const uint64_t bits = (uint64_t)size * 8;
    size_t total = ((size + 8) / 64 + 1) * 64;
    for (size_t off = 0; off < total; off += 64) {
        uint8_t blk[64];
        for (int i = 0; i < 64; i++) {
            size_t p = off + (size_t)i;
            blk[i] = p < size ? data[p] : (p == size ? 0x80 : 0);
        }
        if (off + 64 == total) for (int k = 0; k < 8; k++) blk[56 + k] = (uint8_t)(bits >> (8 * k));
        uint32_t m[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) | ((uint32_t)blk[4 * i + 2] << 16) | ((uint32_t)blk[4 * i + 3] << 24);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
        for (int i = 0; i < 64; i++) {
            uint32_t f; int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
            else { f = c ^ (b | ~d); g = (7 * i) & 15; }
            uint32_t t = d; d = c; c = b; b = b + rol32(a + f + K[i] + m[g], S[i]); a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d;
    }
    for (int k = 0; k < 16; k++) out[k] = (uint8_t)(h[k / 4] >> (8 * (k % 4)));
}

/* ---------------------------------------------------------------------------------------------- helpers */
uint32_t flaco_default_block_size(uint32_t rate)
{
    /* largest of the sizes FLAC can signal without an explicit field that fits in 105 ms (FFmpeg's choice) */
    static const uint32_t tab[] = { 192, 256, 512, 576, 1024, 1152, 2048, 2304, 4096, 4608, 8192, 16384 };
    const uint64_t target = (uint64_t)rate * 105 / 1000;
    uint32_t best = 192;
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++) if (tab[i] <= target) best = tab[i];
    return best;
}
static int blocksize_code(uint32_t bs)
{
    switch (bs) {
    case 192: return 1; case 576: return 2; case 1152: return 3; case 2304: return 4; case 4608: return 5;
    case 256: return 8; case 512: return 9; case 1024: return 10; case 2048: return 11; case 4096: return 12;
    case 8192: return 13; case 16384: return 14; case 32768: return 15;
    default: return bs <= 256 ? 6 : 7;
    }
}
static inline int32_t load_sample(const uint8_t* p, uint32_t bps)
{
    if (bps == 8) return (int32_t)p[0] - 128;                       /* WAV 8 bit is offset binary (Wrapper.cpp:270-277) */
    if (bps == 16) return (int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8));
    return (int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8;
}
static inline void store_sample(uint8_t* p, int32_t v, uint32_t bps)
{
    if (bps == 8) { p[0] = (uint8_t)(v + 128); return; }
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8);
    if (bps == 24) p[2] = (uint8_t)(v >> 16);
}

/* MSB-first bit writer */
typedef struct { uint8_t* p; size_t cap; uint64_t bitpos; int overflow; } bitw;
static void bw_put(bitw* w, uint32_t v, int n)          /* n <= 32 */
{
    for (int i = n - 1; i >= 0; i--) {
        size_t byte = (size_t)(w->bitpos >> 3);
        if (byte >= w->cap) { w->overflow = 1; w->bitpos++; continue; }
        if (!(w->bitpos & 7)) w->p[byte] = 0;
        if ((v >> i) & 1) w->p[byte] |= (uint8_t)(0x80 >> (w->bitpos & 7));
        w->bitpos++;
    }
}
static void bw_put_signed(bitw* w, int32_t v, int n) { bw_put(w, (uint32_t)v & (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1)), n); }
static void bw_unary(bitw* w, uint32_t q) { while (q >= 32) { bw_put(w, 0, 32); q -= 32; } bw_put(w, 1, (int)q + 1); }

/* ---------------------------------------------------------------------------------------------- analysis */
#define MAX_ORDER 32
#define QLP_PRECISION 15
#define MAX_PART_ORDER 8

typedef struct {
    int type;            /* 0 constant, 1 verbatim, 2 fixed, 3 lpc */
    int order, shift;
    int32_t qlp[MAX_ORDER];
    int part_order;
    uint8_t rice_k[1 << MAX_PART_ORDER];
    uint64_t bits;       /* exact size of the subframe */
} subframe_plan;

static inline uint32_t zigzag(int32_t r) { return r >= 0 ? (uint32_t)r << 1 : (((uint32_t)(-(r + 1))) << 1) | 1; }

/* fixed-order residual (fixed.c:217-365 semantics): e0 = x, e1 = dx, ... */
static void fixed_residual(const int32_t* x, uint32_t n, int order, int32_t* r)
{
    for (uint32_t i = (uint32_t)order; i < n; i++) {
        int64_t v;
        switch (order) {
        case 0: v = x[i]; break;
        case 1: v = (int64_t)x[i] - x[i - 1]; break;
        case 2: v = (int64_t)x[i] - 2 * (int64_t)x[i - 1] + x[i - 2]; break;
        case 3: v = (int64_t)x[i] - 3 * (int64_t)x[i - 1] + 3 * (int64_t)x[i - 2] - x[i - 3]; break;
        default: v = (int64_t)x[i] - 4 * (int64_t)x[i - 1] + 6 * (int64_t)x[i - 2] - 4 * (int64_t)x[i - 3] + x[i - 4]; break;
        }
        r[i] = (int32_t)v;
    }
}
/* LPC residual (lpc.c:267 semantics, 64-bit accumulation) */
static int lpc_residual(const int32_t* x, uint32_t n, int order, const int32_t* q, int shift, int32_t* r)
{
    for (uint32_t i = (uint32_t)order; i < n; i++) {
        int64_t acc = 0;
        for (int j = 0; j < order; j++) acc += (int64_t)q[j] * x[i - 1 - j];
        int64_t v = (int64_t)x[i] - (acc >> shift);
        if (v > 0x7FFFFFFFll || v < -0x7FFFFFFFll) return 0;      /* residual must fit 32 bit signed (not INT32_MIN) */
        r[i] = (int32_t)v;
    }
    return 1;
}

/* cost model shared with the device: min over k of m*(k+1) + (U >> k) */
static uint64_t rice_cost(uint64_t U, uint64_t m, int kmax, int* kbest)
{
    uint64_t best = ~(uint64_t)0; int kb = 0;
    for (int k = 0; k <= kmax; k++) {
        uint64_t c = m * (uint64_t)(k + 1) + (U >> k);
        if (c < best) { best = c; kb = k; }
    }
    if (kbest) *kbest = kb;
    return best;
}

/* choose partition order and Rice parameters for residual r[order..n); returns estimated bits incl. the 6-bit header */
static uint64_t plan_partitions(const int32_t* r, uint32_t n, int order, int kmax, int param_bits, subframe_plan* sp)
{
    uint64_t* sums = malloc(sizeof(uint64_t) << MAX_PART_ORDER);
    int pmax = 0;
    while (pmax < MAX_PART_ORDER && !((n >> pmax) & 1) && (n >> (pmax + 1)) > (uint32_t)order) pmax++;
    uint64_t best = ~(uint64_t)0;
    for (int po = pmax; po >= 0; po--) {
        const uint32_t parts = 1u << po, plen = n >> po;
        if (po == pmax) {
            for (uint32_t p = 0; p < parts; p++) {
                uint64_t U = 0;
                for (uint32_t i = p ? p * plen : (uint32_t)order; i < (p + 1) * plen; i++) U += zigzag(r[i]);
                sums[p] = U;
            }
        } else {
            for (uint32_t p = 0; p < parts; p++) sums[p] = sums[2 * p] + sums[2 * p + 1];
        }
        uint64_t bits = 6; uint8_t ks[1 << MAX_PART_ORDER];
        for (uint32_t p = 0; p < parts; p++) {
            const uint64_t m = p ? plen : plen - (uint32_t)order;
            int k; bits += (uint64_t)param_bits + rice_cost(sums[p], m, kmax, &k);
            ks[p] = (uint8_t)k;
        }
        if (bits < best) { best = bits; sp->part_order = po; memcpy(sp->rice_k, ks, parts); }
    }
    free(sums);
    return best;
}

static uint64_t exact_residual_bits(const int32_t* r, uint32_t n, int order, int param_bits, const subframe_plan* sp)
{
    const uint32_t parts = 1u << sp->part_order, plen = n >> sp->part_order;
    uint64_t bits = 6;
    for (uint32_t p = 0; p < parts; p++) {
        bits += (uint64_t)param_bits;
        const int k = sp->rice_k[p];
        for (uint32_t i = p ? p * plen : (uint32_t)order; i < (p + 1) * plen; i++) bits += (zigzag(r[i]) >> k) + 1 + (uint64_t)k;
    }
    return bits;
}

/* Welch window in Q15 and integer autocorrelation */
static void autocorrelate(const int32_t* x, uint32_t n, int maxlag, int64_t* ac, int64_t* xs)
{
    const int64_t d = (int64_t)n + 1, d2 = d * d;
    for (uint32_t i = 0; i < n; i++) {
        const int64_t c = 2 * (int64_t)i - ((int64_t)n - 1);
        const int64_t w = ((d2 - c * c) << 15) / d2;
        xs[i] = ((int64_t)x[i] * w) >> 15;
    }
    for (int l = 0; l <= maxlag; l++) {
        int64_t s = 0;
        for (uint32_t i = (uint32_t)l; i < n; i++) s += xs[i] * xs[i - (uint32_t)l];
        ac[l] = s;
    }
}

/* Levinson-Durbin; lp[m-1][0..m-1] = predictor of order m (x^[i] = sum lp[j] * x[i-1-j]); returns usable max order */
static int levinson(const int64_t* ac, int maxorder, double lp[MAX_ORDER][MAX_ORDER])
{
    double a[MAX_ORDER], t[MAX_ORDER];
    double err = (double)ac[0];
    if (!(err > 0.0)) return 0;
    int m;
    for (m = 1; m <= maxorder; m++) {
        double acc = (double)ac[m];
        for (int j = 1; j < m; j++) acc = acc - a[j - 1] * (double)ac[m - j];
        const double k = acc / err;
        for (int j = 1; j < m; j++) t[j - 1] = a[j - 1] - k * a[m - j - 1];
        for (int j = 1; j < m; j++) a[j - 1] = t[j - 1];
        a[m - 1] = k;
        err = err * (1.0 - k * k);
        for (int j = 0; j < m; j++) lp[m - 1][j] = a[j];
        if (!(err > 0.0)) { m++; break; }
    }
    return m - 1;
}

/* quantise predictor coefficients to QLP_PRECISION bits; returns 0 when unusable */
static int quantize(const double* c, int order, int32_t* q, int* shift_out)
{
    double cmax = 0.0;
    for (int j = 0; j < order; j++) { const double a = c[j] < 0 ? -c[j] : c[j]; if (a > cmax) cmax = a; }
    if (!(cmax > 0.0) || cmax > 65536.0) return 0;
    /* e with cmax = f * 2^e, f in [0.5, 1) -- from the IEEE exponent field, no libm */
    uint64_t bits; memcpy(&bits, &cmax, 8);
    const int e = (int)((bits >> 52) & 0x7FF) - 1022;
    int shift = QLP_PRECISION - 1 - e;
    if (shift > 15) shift = 15;
    if (shift < 0) return 0;
    const int32_t qmax = (1 << (QLP_PRECISION - 1)) - 1, qmin = -(1 << (QLP_PRECISION - 1));
    const double scale = (double)(1 << shift);
    double error = 0.0;
    for (int j = 0; j < order; j++) {
        error = error + c[j] * scale;
        /* round half away from zero */
        double rr = error < 0 ? -error : error;
        double fl = (double)(int64_t)(rr + 0.5);
        int64_t qi = (int64_t)(error < 0 ? -fl : fl);
        if (qi > qmax) qi = qmax;
        if (qi < qmin) qi = qmin;
        error = error - (double)qi;
        q[j] = (int32_t)qi;
    }
    *shift_out = shift;
    return 1;
}

/* bps: width of THIS subframe's samples -- the side channel of a stereo pair has one bit more (stream_decoder.c:2036-2062) */
static void plan_subframe(const flaco_params* P, uint32_t bps, const int32_t* x, uint32_t n, subframe_plan* best, int32_t* best_res,
                          int32_t* tmp_res, int64_t* xs)
{
    const int param_bits = bps > 16 ? 5 : 4, kmax = bps > 16 ? 30 : 14;
    memset(best, 0, sizeof *best);
    /* constant */
    int constant = 1;
    for (uint32_t i = 1; i < n; i++) if (x[i] != x[0]) { constant = 0; break; }
    if (constant) { best->type = 0; best->bits = 8 + bps; return; }
    best->type = 1; best->bits = 8 + (uint64_t)n * bps;                /* verbatim */
    subframe_plan cand;
    /* fixed orders 0..4 */
    for (int o = 0; o <= 4 && (uint32_t)o < n; o++) {
        memset(&cand, 0, sizeof cand);
        cand.type = 2; cand.order = o;
        fixed_residual(x, n, o, tmp_res);
        int ok = 1;
        for (uint32_t i = (uint32_t)o; i < n; i++) if (tmp_res[i] == INT32_MIN) ok = 0;
        if (!ok) continue;
        plan_partitions(tmp_res, n, o, kmax, param_bits, &cand);
        cand.bits = 8 + (uint64_t)o * bps + exact_residual_bits(tmp_res, n, o, param_bits, &cand);
        if (cand.bits < best->bits) { *best = cand; memcpy(best_res, tmp_res, n * sizeof(int32_t)); }
    }
    /* LPC orders 1..max */
    int maxorder = (int)P->max_lpc_order;
    if (maxorder > MAX_ORDER) maxorder = MAX_ORDER;
    if ((uint32_t)maxorder >= n) maxorder = (int)n - 1;
    if (maxorder > 0) {
        int64_t ac[MAX_ORDER + 1];
        static __thread double lp[MAX_ORDER][MAX_ORDER];
        autocorrelate(x, n, maxorder, ac, xs);
        const int usable = levinson(ac, maxorder, lp);
        for (int o = 1; o <= usable; o++) {
            memset(&cand, 0, sizeof cand);
            cand.type = 3; cand.order = o;
            if (!quantize(lp[o - 1], o, cand.qlp, &cand.shift)) continue;
            if (!lpc_residual(x, n, o, cand.qlp, cand.shift, tmp_res)) continue;
            plan_partitions(tmp_res, n, o, kmax, param_bits, &cand);
            cand.bits = 8 + (uint64_t)o * bps + 4 + 5 + (uint64_t)o * QLP_PRECISION + exact_residual_bits(tmp_res, n, o, param_bits, &cand);
            if (cand.bits < best->bits) { *best = cand; memcpy(best_res, tmp_res, n * sizeof(int32_t)); }
        }
    }
}

static void write_subframe(bitw* w, int bps, const int32_t* x, uint32_t n, const subframe_plan* sp, const int32_t* res)
{
    const int param_bits = bps > 16 ? 5 : 4;
    switch (sp->type) {
    case 0: bw_put(w, 0x00, 8); bw_put_signed(w, x[0], bps); return;
    case 1: bw_put(w, 0x02, 8); for (uint32_t i = 0; i < n; i++) bw_put_signed(w, x[i], bps); return;
    case 2: bw_put(w, (uint32_t)(0x10 | (sp->order << 1)), 8); break;                 /* 0 001ooo 0 */
    default: bw_put(w, (uint32_t)(0x40 | ((sp->order - 1) << 1)), 8); break;          /* 0 1ooooo 0 */
    }
    for (int i = 0; i < sp->order; i++) bw_put_signed(w, x[i], bps);
    if (sp->type == 3) {
        bw_put(w, QLP_PRECISION - 1, 4);
        bw_put_signed(w, sp->shift, 5);
        for (int j = 0; j < sp->order; j++) bw_put_signed(w, sp->qlp[j], QLP_PRECISION);
    }
    bw_put(w, param_bits == 5 ? 1 : 0, 2);
    bw_put(w, (uint32_t)sp->part_order, 4);
    const uint32_t parts = 1u << sp->part_order, plen = n >> sp->part_order;
    for (uint32_t p = 0; p < parts; p++) {
        const int k = sp->rice_k[p];
        bw_put(w, (uint32_t)k, param_bits);
        for (uint32_t i = p ? p * plen : (uint32_t)sp->order; i < (p + 1) * plen; i++) {
            const uint32_t u = zigzag(res[i]);
            bw_unary(w, u >> k);
            if (k) bw_put(w, u & ((1u << k) - 1), k);
        }
    }
}

static int utf8_put(uint8_t* p, uint64_t v)
{
    if (v < 0x80) { p[0] = (uint8_t)v; return 1; }
    int n = v < 0x800 ? 2 : v < 0x10000 ? 3 : v < 0x200000 ? 4 : v < 0x4000000 ? 5 : v < 0x80000000ull ? 6 : 7;
    static const uint8_t lead[8] = { 0, 0, 0xC0, 0xE0, 0xF0, 0xF8, 0xFC, 0xFE };
    for (int i = n - 1; i > 0; i--) { p[i] = (uint8_t)(0x80 | (v & 0x3F)); v >>= 6; }
    p[0] = (uint8_t)(lead[n] | v);
    return n;
}

long flaco_encode(const flaco_params* P, const uint8_t* pcm, uint64_t pcm_bytes, uint8_t* out, size_t cap, uint32_t* frame_sizes, size_t frame_cap)
{
    const uint32_t ch = P->channels, bps = P->bits_per_sample, bytes_ps = bps / 8;
    if (!ch || ch > 8 || (bps != 8 && bps != 16 && bps != 24)) return -1;
    const uint32_t B = P->block_size ? P->block_size : flaco_default_block_size(P->sample_rate);
    const uint64_t total = pcm_bytes / ((uint64_t)bytes_ps * ch);
    const uint64_t nframes = (total + B - 1) / B;
    if (nframes > frame_cap) return -1;
    /* Two channels are also tried as left/side, side/right and mid/side (channel assignments 8, 9, 10; stream_decoder.c:2299-2321 reads
       them, :2036-2062 undoes them): four candidate signals L, R, M = (L + R) >> 1, S = L - R, the pair with the fewest bits wins, the
       plain pair on a tie. */
    const uint32_t nsig = ch == 2 ? 4 : ch;
    int32_t* x = malloc((size_t)B * nsig * sizeof(int32_t));
    int32_t* res = malloc((size_t)B * nsig * sizeof(int32_t));
    int32_t* tmp = malloc((size_t)B * sizeof(int32_t));
    int64_t* xs = malloc((size_t)B * sizeof(int64_t));
    subframe_plan* plans = malloc(sizeof(subframe_plan) * nsig);
    size_t pos = 0; long ret = (long)nframes;
    for (uint64_t fi = 0; fi < nframes; fi++) {
        const uint32_t n = (uint32_t)((fi + 1) * B <= total ? B : total - fi * B);
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t c = 0; c < ch; c++)
                x[(size_t)c * B + i] = load_sample(pcm + ((fi * B + i) * ch + c) * bytes_ps, bps);
        uint64_t bits = 0;
        uint32_t sig[8], sig_bps[8], assignment = ch - 1;
        for (uint32_t c = 0; c < ch; c++) { sig[c] = c; sig_bps[c] = bps; }
        if (ch == 2) {
            for (uint32_t i = 0; i < n; i++) {
                const int32_t l = x[i], r = x[(size_t)B + i];
                x[(size_t)2 * B + i] = (l + r) >> 1;
                x[(size_t)3 * B + i] = l - r;
            }
        }
        for (uint32_t c = 0; c < nsig; c++) plan_subframe(P, c == 3 && ch == 2 ? bps + 1 : bps, x + (size_t)c * B, n, &plans[c], res + (size_t)c * B, tmp, xs);
        if (ch == 2) {
            const uint64_t lr = plans[0].bits + plans[1].bits, ls = plans[0].bits + plans[3].bits, rs = plans[3].bits + plans[1].bits, ms = plans[2].bits + plans[3].bits;
            uint64_t best = lr; assignment = 1;
            if (ls < best) { best = ls; assignment = 8; }
            if (rs < best) { best = rs; assignment = 9; }
            if (ms < best) { best = ms; assignment = 10; }
            if (assignment == 8) { sig[1] = 3; sig_bps[1] = bps + 1; }
            else if (assignment == 9) { sig[0] = 3; sig_bps[0] = bps + 1; }
            else if (assignment == 10) { sig[0] = 2; sig[1] = 3; sig_bps[1] = bps + 1; }
        }
        for (uint32_t c = 0; c < ch; c++) bits += plans[sig[c]].bits;
        /* frame header (stream_decoder.c:2159-2466), fixed blocking strategy */
        uint8_t hdr[16]; int h = 0;
        hdr[h++] = 0xFF; hdr[h++] = 0xF8;
        const int bsc = blocksize_code(n);
        const int src = P->sample_rate == 44100 ? 9 : P->sample_rate == 48000 ? 10 : P->sample_rate == 96000 ? 11 : 0;
        hdr[h++] = (uint8_t)((bsc << 4) | src);
        hdr[h++] = (uint8_t)((assignment << 4) | ((bps == 8 ? 1 : bps == 16 ? 4 : 6) << 1));
        h += utf8_put(hdr + h, fi);
        if (bsc == 6) hdr[h++] = (uint8_t)(n - 1);
        else if (bsc == 7) { hdr[h++] = (uint8_t)((n - 1) >> 8); hdr[h++] = (uint8_t)(n - 1); }
        hdr[h] = flaco_crc8(hdr, (size_t)h); h++;
        const size_t fsize = (size_t)h + (size_t)((bits + 7) / 8) + 2;
        if (pos + fsize > cap) { ret = -1; break; }
        memcpy(out + pos, hdr, (size_t)h);
        bitw w = { out + pos + h, fsize - (size_t)h - 2, 0, 0 };
        for (uint32_t c = 0; c < ch; c++) write_subframe(&w, (int)sig_bps[c], x + (size_t)sig[c] * B, n, &plans[sig[c]], res + (size_t)sig[c] * B);
        if (w.overflow || w.bitpos != bits) { ret = -2; break; }
        const uint16_t crc = flaco_crc16(out + pos, fsize - 2);
        out[pos + fsize - 2] = (uint8_t)(crc >> 8); out[pos + fsize - 1] = (uint8_t)crc;
        frame_sizes[fi] = (uint32_t)fsize;
        pos += fsize;
    }
    free(x); free(res); free(tmp); free(xs); free(plans);
    return ret;
}

size_t flaco_codec_private(const flaco_params* P, uint64_t total_samples, uint32_t min_frame, uint32_t max_frame, const uint8_t md5[16], uint8_t* out)
{
    const uint32_t B = P->block_size ? P->block_size : flaco_default_block_size(P->sample_rate);
    uint8_t* p = out;
    memcpy(p, "fLaC", 4); p += 4;
    *p++ = 0x80; *p++ = 0; *p++ = 0; *p++ = 34;                      /* last block, STREAMINFO, length 34 */
    *p++ = (uint8_t)(B >> 8); *p++ = (uint8_t)B; *p++ = (uint8_t)(B >> 8); *p++ = (uint8_t)B;
    *p++ = (uint8_t)(min_frame >> 16); *p++ = (uint8_t)(min_frame >> 8); *p++ = (uint8_t)min_frame;
    *p++ = (uint8_t)(max_frame >> 16); *p++ = (uint8_t)(max_frame >> 8); *p++ = (uint8_t)max_frame;
    const uint64_t v = ((uint64_t)P->sample_rate << 44) | ((uint64_t)(P->channels - 1) << 41) | ((uint64_t)(P->bits_per_sample - 1) << 36) | (total_samples & 0xFFFFFFFFFull);
    for (int s = 56; s >= 0; s -= 8) *p++ = (uint8_t)(v >> s);
    memcpy(p, md5, 16); p += 16;
    return (size_t)(p - out);
}

/* ---------------------------------------------------------------------------------------------- decoder */
typedef struct { const uint8_t* p; size_t size; uint64_t bitpos; int err; } bitr;
static uint32_t br_get(bitr* r, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        size_t byte = (size_t)(r->bitpos >> 3);
        if (byte >= r->size) { r->err = 1; return 0; }
        v = (v << 1) | ((r->p[byte] >> (7 - (r->bitpos & 7))) & 1);
        r->bitpos++;
    }
    return v;
}
static int32_t br_get_signed(bitr* r, int n) { uint32_t v = br_get(r, n); return n == 32 ? (int32_t)v : (int32_t)(v << (32 - n)) >> (32 - n); }
static uint32_t br_unary(bitr* r) { uint32_t q = 0; while (!r->err && !br_get(r, 1)) q++; return q; }

long long flaco_decode(const flaco_params* P, const uint8_t* frames, size_t size, uint8_t* pcm, size_t cap)
{
    const uint32_t ch = P->channels, bps = P->bits_per_sample, bytes_ps = bps / 8;
    size_t pos = 0; uint64_t outpos = 0;
    int32_t* x = malloc(sizeof(int32_t) * 65536 * ch);
    long long ret = 0;
    while (pos < size && !ret) {
        bitr r = { frames + pos, size - pos, 0, 0 };
        if (br_get(&r, 15) != 0x7FFC) { ret = -10; break; }             /* sync 11111111 111110 + reserved 0 */
        if (br_get(&r, 1) != 0) { ret = -11; break; }                     /* fixed blocksize stream */
        const uint32_t bsc = br_get(&r, 4), src = br_get(&r, 4), ca = br_get(&r, 4), ssc = br_get(&r, 3);
        if (br_get(&r, 1)) { ret = -12; break; }
        uint32_t first = br_get(&r, 8); int extra = 0;                    /* UTF-8 frame number */
        while (first & 0x80) { extra++; first <<= 1; }
        for (int i = 1; i < extra; i++) br_get(&r, 8);
        uint32_t n;
        if (bsc == 1) n = 192; else if (bsc >= 2 && bsc <= 5) n = 576u << (bsc - 2);
        else if (bsc == 6) n = br_get(&r, 8) + 1; else if (bsc == 7) n = br_get(&r, 16) + 1;
        else if (bsc >= 8) n = 256u << (bsc - 8); else { ret = -13; break; }
        if (src == 12) br_get(&r, 8); else if (src == 13 || src == 14) br_get(&r, 16); else if (src == 15) { ret = -14; break; }
        if (ca != ch - 1 && !(ch == 2 && ca >= 8 && ca <= 10)) { ret = -15; break; }
        const uint32_t want_ssc = bps == 8 ? 1 : bps == 16 ? 4 : 6;
        if (ssc != want_ssc && ssc != 0) { ret = -16; break; }
        const size_t hbytes = (size_t)(r.bitpos >> 3);
        const uint32_t crc8 = br_get(&r, 8);
        if (crc8 != flaco_crc8(frames + pos, hbytes)) { ret = -17; break; }
        for (uint32_t c = 0; c < ch && !ret; c++) {
            int32_t* d = x + (size_t)c * 65536;
            if (br_get(&r, 1)) { ret = -20; break; }
            const uint32_t type = br_get(&r, 6);
            uint32_t wasted = 0;
            if (br_get(&r, 1)) wasted = br_unary(&r) + 1;
            const int side = (ca == 8 && c == 1) || (ca == 9 && c == 0) || (ca == 10 && c == 1);      /* the difference channel: one bit more */
            const int sb = (int)bps + side - (int)wasted;
            if (type == 0) { const int32_t v = br_get_signed(&r, sb); for (uint32_t i = 0; i < n; i++) d[i] = v; }
            else if (type == 1) { for (uint32_t i = 0; i < n; i++) d[i] = br_get_signed(&r, sb); }
            else {
                int order, shift = 0, prec = 0; int32_t q[32];
                const int is_lpc = (type & 0x20) != 0;
                if (is_lpc) order = (int)(type & 0x1F) + 1;
                else if ((type & 0x38) == 0x08) { order = (int)(type & 7); if (order > 4) { ret = -21; break; } }
                else { ret = -22; break; }
                for (int i = 0; i < order; i++) d[i] = br_get_signed(&r, sb);
                if (is_lpc) {
                    prec = (int)br_get(&r, 4) + 1; if (prec == 16) { ret = -23; break; }
                    shift = br_get_signed(&r, 5); if (shift < 0) { ret = -24; break; }
                    for (int j = 0; j < order; j++) q[j] = br_get_signed(&r, prec);
                }
                const uint32_t method = br_get(&r, 2); if (method > 1) { ret = -25; break; }
                const int pbits = method ? 5 : 4; const uint32_t esc = method ? 31 : 15;
                const uint32_t po = br_get(&r, 4), parts = 1u << po, plen = n >> po;
                if ((n >> po) < (uint32_t)order || (po && (n & (parts - 1)))) { ret = -26; break; }
                uint32_t i = (uint32_t)order;
                for (uint32_t p = 0; p < parts && !r.err; p++) {
                    const uint32_t k = br_get(&r, pbits);
                    const uint32_t end = (p + 1) * plen;
                    if (k == esc) { const int raw = (int)br_get(&r, 5); for (; i < end; i++) d[i] = raw ? br_get_signed(&r, raw) : 0; continue; }
                    for (; i < end && !r.err; i++) {
                        const uint32_t qv = br_unary(&r);
                        const uint32_t u = (qv << k) | (k ? br_get(&r, (int)k) : 0);
                        int32_t res = (u & 1) ? -(int32_t)(u >> 1) - 1 : (int32_t)(u >> 1);
                        int64_t pred = 0;
                        if (is_lpc) { int64_t acc = 0; for (int j = 0; j < order; j++) acc += (int64_t)q[j] * d[i - 1 - (uint32_t)j]; pred = acc >> shift; }
                        else switch (order) {
                            case 0: pred = 0; break;
                            case 1: pred = d[i - 1]; break;
                            case 2: pred = 2 * (int64_t)d[i - 1] - d[i - 2]; break;
                            case 3: pred = 3 * (int64_t)d[i - 1] - 3 * (int64_t)d[i - 2] + d[i - 3]; break;
                            default: pred = 4 * (int64_t)d[i - 1] - 6 * (int64_t)d[i - 2] + 4 * (int64_t)d[i - 3] - d[i - 4]; break;
                        }
                        d[i] = (int32_t)(res + pred);
                    }
                }
            }
            if (wasted) for (uint32_t i = 0; i < n; i++) d[i] = (int32_t)((uint32_t)d[i] << wasted);
        }
        if (ret) break;
        if (ca >= 8) {                                                     /* stream_decoder.c:2036-2062 */
            int32_t* a = x; int32_t* b = x + 65536;
            for (uint32_t i = 0; i < n; i++) {
                if (ca == 8) b[i] = a[i] - b[i];                           /* left, side -> right */
                else if (ca == 9) a[i] = a[i] + b[i];                      /* side, right -> left */
                else { const int32_t m2 = (int32_t)(((uint32_t)a[i] << 1) | ((uint32_t)b[i] & 1u)), sd = b[i]; a[i] = (m2 + sd) >> 1; b[i] = (m2 - sd) >> 1; }
            }
        }
        if (r.err) { ret = -30; break; }
        while (r.bitpos & 7) if (br_get(&r, 1)) { ret = -31; break; }
        if (ret) break;
        const size_t fbytes = (size_t)(r.bitpos >> 3);
        if (fbytes + 2 > size - pos) { ret = -32; break; }
        const uint16_t crc = (uint16_t)((frames[pos + fbytes] << 8) | frames[pos + fbytes + 1]);
        if (crc != flaco_crc16(frames + pos, fbytes)) { ret = -33; break; }
        if ((outpos + n) * ch * bytes_ps > cap) { ret = -34; break; }
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t c = 0; c < ch; c++)
                store_sample(pcm + ((outpos + i) * ch + c) * bytes_ps, x[(size_t)c * 65536 + i], bps);
        outpos += n;
        pos += fbytes + 2;
    }
    free(x);
    return ret ? ret : (long long)(outpos * ch * bytes_ps);
}
