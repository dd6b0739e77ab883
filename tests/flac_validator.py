"""A FLAC frame decoder written from the format specification (RFC 9639 / the FLAC format description), not from this repository's
encoder, its oracle or the vendored libFLAC: stock FFmpeg cannot be run here, so that the streams are decodable by a third party is
shown by decoding them with independent code.  TEST INFRASTRUCTURE.

What it checks on the way: frame sync and reserved bits, block-size / sample-rate / sample-size codes, the UTF-8 coded frame number
counting up from 0, CRC-8 of every frame header and CRC-16 of every frame, subframe types (constant, verbatim, fixed 0-4, LPC 1-32 with
its precision, shift and coefficients), wasted bits, partitioned Rice / Rice2 residuals incl. the escape code, the three stereo
decorrelations, zero padding to the byte boundary.  `parse_stream` returns the PCM as interleaved little-endian bytes (8-bit samples
unsigned, the WAV convention the reference's flac_wrapper restores, Lib/CoDec/Wrapper.cpp:247-373) and, per frame and subframe, the
predictor words -- (order, precision, shift, coefficients) -- which tests/golden/flac_coefficients.json pins.
"""
from __future__ import annotations


class _Bits:
    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p = data, pos * 8

    def u(self, n: int) -> int:
        v = 0
        p = self.p
        d = self.d
        for _ in range(n):
            v = (v << 1) | ((d[p >> 3] >> (7 - (p & 7))) & 1)
            p += 1
        self.p = p
        return v

    def s(self, n: int) -> int:
        v = self.u(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self) -> int:
        n = 0
        while not self.u(1):
            n += 1
        return n


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


_FIXED = [[], [1], [2, -1], [3, -3, 1], [4, -6, 4, -1]]


def _residual(br: _Bits, n: int, order: int, out: list[int], info: dict) -> None:
    method = br.u(2)
    assert method in (0, 1), f"residual coding method {method} is reserved"
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    po = br.u(4)
    info["rice_method"], info["partition_order"] = method, po
    assert n % (1 << po) == 0 and (n >> po) >= order, "partition order does not divide the block / first partition shorter than the predictor"
    params = []
    for part in range(1 << po):
        cnt = (n >> po) - (order if part == 0 else 0)
        k = br.u(pbits)
        params.append(k)
        if k == esc:
            raw = br.u(5)
            for _ in range(cnt):
                out.append(br.s(raw) if raw else 0)
        else:
            for _ in range(cnt):
                q = br.unary()
                u = (q << k) | (br.u(k) if k else 0)
                out.append((u >> 1) ^ -(u & 1))
    info["rice_parameters"] = params


def _subframe(br: _Bits, n: int, bps: int) -> tuple[list[int], dict]:
    assert br.u(1) == 0, "subframe padding bit"
    t = br.u(6)
    wasted = 0
    if br.u(1):
        wasted = br.unary() + 1
        bps -= wasted
    info: dict = {"wasted": wasted}
    if t == 0:
        info["type"] = "constant"
        out = [br.s(bps)] * n
    elif t == 1:
        info["type"] = "verbatim"
        out = [br.s(bps) for _ in range(n)]
    elif 8 <= t <= 12:
        order = t - 8
        info.update(type="fixed", order=order)
        out = [br.s(bps) for _ in range(order)]
        res: list[int] = []
        _residual(br, n, order, res, info)
        c = _FIXED[order]
        for r in res:
            out.append(r + sum(c[j] * out[-1 - j] for j in range(order)))
    elif t >= 32:
        order = t - 31
        out = [br.s(bps) for _ in range(order)]
        prec = br.u(4) + 1
        assert prec != 16, "LPC precision code 1111 is invalid"
        shift = br.s(5)
        assert shift >= 0, "negative LPC shift"
        coefs = [br.s(prec) for _ in range(order)]
        info.update(type="lpc", order=order, precision=prec, shift=shift, coefficients=coefs)
        res = []
        _residual(br, n, order, res, info)
        for r in res:
            out.append(r + (sum(coefs[j] * out[-1 - j] for j in range(order)) >> shift))
    else:
        raise AssertionError(f"reserved subframe type {t}")
    if wasted:
        out = [v << wasted for v in out]
    assert len(out) == n
    return out, info


def parse_frame(data: bytes, pos: int, stream_bps: int, stream_rate: int, stream_channels: int, expect_number: int):
    """One frame at data[pos:]: -> (channel sample lists, end position, per-subframe info)."""
    br = _Bits(data, pos)
    assert br.u(14) == 0x3FFE, f"no frame sync at byte {pos}"
    assert br.u(1) == 0, "reserved bit after the sync code"
    blocking = br.u(1)
    assert blocking == 0, "variable block size streams are not what this encoder writes"
    bs_code, sr_code = br.u(4), br.u(4)
    ch_code, ss_code = br.u(4), br.u(3)
    assert br.u(1) == 0, "reserved bit in the frame header"
    first = br.u(8)                                                     # UTF-8 style frame number
    if first < 0x80:
        number = first
    else:
        extra = 1 if first >> 5 == 0b110 else 2 if first >> 4 == 0b1110 else 3 if first >> 3 == 0b11110 else 4 if first >> 2 == 0b111110 else 5 if first >> 1 == 0b1111110 else None
        assert extra, "bad first byte of the coded frame number"
        number = first & (0x7F >> (extra + 1))
        for _ in range(extra):
            b = br.u(8)
            assert b >> 6 == 0b10, "bad continuation byte of the coded frame number"
            number = (number << 6) | (b & 0x3F)
    assert number == expect_number, f"frame number {number}, expected {expect_number}"
    assert bs_code != 0, "reserved block size code"
    n = 192 if bs_code == 1 else 576 << (bs_code - 2) if bs_code <= 5 else br.u(8) + 1 if bs_code == 6 else br.u(16) + 1 if bs_code == 7 else 256 << (bs_code - 8)
    rates = {0: stream_rate, 1: 88200, 2: 176400, 3: 192000, 4: 8000, 5: 16000, 6: 22050, 7: 24000, 8: 32000, 9: 44100, 10: 48000, 11: 96000}
    rate = rates[sr_code] if sr_code in rates else br.u(8) * 1000 if sr_code == 12 else br.u(16) if sr_code == 13 else br.u(16) * 10 if sr_code == 14 else None
    assert rate == stream_rate, f"sample rate {rate} in a {stream_rate} Hz stream"
    bps = {0: stream_bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}.get(ss_code)
    assert bps == stream_bps, f"sample size code {ss_code} in a {stream_bps}-bit stream"
    assert br.p % 8 == 0
    hdr_end = br.p // 8
    assert crc8(data[pos:hdr_end]) == br.u(8), "frame header CRC-8"
    assert ch_code <= 10, "reserved channel assignment"
    nch = ch_code + 1 if ch_code < 8 else 2
    assert nch == stream_channels
    chans, infos = [], []
    for c in range(nch):
        side = (ch_code == 8 and c == 1) or (ch_code == 9 and c == 0) or (ch_code == 10 and c == 1)
        s, info = _subframe(br, n, bps + (1 if side else 0))
        chans.append(s)
        infos.append(info)
    while br.p % 8:
        assert br.u(1) == 0, "non-zero padding in front of the frame CRC"
    end = br.p // 8
    assert crc16(data[pos:end]) == br.u(16), "frame CRC-16"
    if ch_code == 8:
        chans[1] = [l - s for l, s in zip(chans[0], chans[1])]
    elif ch_code == 9:
        chans[0] = [s + r for s, r in zip(chans[0], chans[1])]
    elif ch_code == 10:
        l, r = [], []
        for m, s in zip(chans[0], chans[1]):
            m = (m << 1) | (s & 1)
            l.append((m + s) >> 1)
            r.append((m - s) >> 1)
        chans = [l, r]
    lim = 1 << (bps - 1)
    assert all(-lim <= v < lim for ch in chans for v in ch), "a decoded sample leaves the sample range"
    return chans, end + 2, {"channel_assignment": ch_code, "block_size": n, "subframes": infos}


def parse_stream(frames: bytes, channels: int, bits: int, rate: int):
    """All frames of a stream of concatenated FLAC frames -> (pcm bytes, [frame info])."""
    pos, k = 0, 0
    pcm = bytearray()
    infos = []
    nb = bits // 8
    while pos < len(frames):
        chans, pos, info = parse_frame(frames, pos, bits, rate, channels, k)
        for i in range(len(chans[0])):
            for c in range(channels):
                v = chans[c][i]
                if bits == 8:
                    v += 128
                pcm += (v & ((1 << bits) - 1)).to_bytes(nb, "little")
        infos.append(info)
        k += 1
    return bytes(pcm), infos


def predictor_words(infos) -> list:
    """What pins a block's predictor choice: per frame [channel assignment, [type, order, precision, shift, coefficients, partition order] per subframe]."""
    out = []
    for f in infos:
        out.append([f["channel_assignment"], [[s["type"], s.get("order", 0), s.get("precision", 0), s.get("shift", 0), s.get("coefficients", []), s.get("partition_order", 0)]
                                              for s in f["subframes"]]])
    return out
