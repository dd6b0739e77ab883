// crc_dev.h -- device helpers for the FFV1 CRC-32 (poly 0x04C11DB7, MSB first, init 0, no final xor; ZenCRC32.cpp:1097-1135):
// arithmetic in GF(2)[x]/P used to glue per-thread CRCs together, crc(A||B) = crc(A) * x^(8|B|) + crc(B).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ __forceinline__ uint32_t gf_mulmod(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int i = 31; i >= 0; i--) {
        r = (r << 1) ^ ((r >> 31) ? 0x04C11DB7u : 0u);
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}
__device__ uint32_t gf_xpow8(unsigned long long nbytes)      // x^(8*nbytes) mod P
{
    uint32_t result = 1, base = 0x100;
    while (nbytes) {
        if (nbytes & 1) result = gf_mulmod(result, base);
        base = gf_mulmod(base, base);
        nbytes >>= 1;
    }
    return result;
}

// CRC of `total` bytes at `p` (16-byte aligned) by one 256-thread block; every thread returns its share, the xor over the block
// is the CRC.  16 KB tiles are read fully coalesced -- thread t owns the 64-byte chunk t of every tile; its chunks are 16 KB
// apart, so a Horner recurrence with the constant M = x^(8*16384) accumulates them: acc = acc*M + crc(chunk); multiplication by
// M is four table look-ups.  The sub-tile remainder is done with contiguous per-thread segments.
// T: slicing-by-4 tables, T[k][b] = crc of byte b followed by k zero bytes (filled by crc_tables); TM: scratch [4][256].
__device__ __forceinline__ void crc_tables(uint32_t (*T)[256], int tid)
{
    uint32_t c = uint32_t(tid) << 24;
    for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
    T[0][tid] = c;
    __syncthreads();
    for (int k = 1; k < 4; k++) { const uint32_t p = T[k - 1][tid]; T[k][tid] = (p << 8) ^ T[0][p >> 24]; __syncthreads(); }
}
__device__ __forceinline__ uint32_t block_crc_share(const uint8_t* p, uint32_t total, const uint32_t (*T)[256], uint32_t (*TM)[256], int tid)
{
    constexpr uint32_t kTile = 16384, kChunk = kTile / 256;
    const uint32_t ntiles = total / kTile, rem = total - ntiles * kTile;
    const uint32_t M = gf_xpow8(kTile);
    for (int k = 0; k < 4; k++) TM[k][tid] = gf_mulmod(uint32_t(tid) << (8 * k), M);
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint4* p4 = reinterpret_cast<const uint4*>(p + size_t(t) * kTile + size_t(tid) * kChunk);
        uint32_t cc = 0;
#pragma unroll
        for (int q = 0; q < int(kChunk / 16); q++) {
            const uint4 v = p4[q];
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cc ^= __builtin_bswap32(w[j]);
                cc = T[3][cc >> 24] ^ T[2][(cc >> 16) & 0xFF] ^ T[1][(cc >> 8) & 0xFF] ^ T[0][cc & 0xFF];
            }
        }
        acc = TM[3][acc >> 24] ^ TM[2][(acc >> 16) & 0xFF] ^ TM[1][(acc >> 8) & 0xFF] ^ TM[0][acc & 0xFF] ^ cc;
    }
    if (ntiles) acc = gf_mulmod(acc, gf_xpow8((unsigned long long)kChunk * (255 - tid) + rem));
    const uint8_t* rp = p + size_t(ntiles) * kTile;
    const uint32_t seg = ((rem + 255) / 256 + 3) & ~3u;
    const uint32_t beg = min(rem, uint32_t(tid) * seg), end = min(rem, beg + seg);
    uint32_t c = 0;
    for (uint32_t i = beg; i < end; i++) c = (c << 8) ^ T[0][(c >> 24) ^ rp[i]];
    return gf_mulmod(c, gf_xpow8(rem - end)) ^ acc;
}
