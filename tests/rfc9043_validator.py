"""Structural validator for FFV1 version 3 streams, written from RFC 9043 (not from this repository's encoder or oracle): stock FFmpeg
cannot be run in this environment, so conformance beyond "the reference's decoder rebuilds the files" is checked here field by field.

TEST INFRASTRUCTURE.  Checks, with the RFC section each comes from:
  * configuration record (4.2, 4.3): version == 3, micro_version >= 4 for new streams (4.2.2), coder_type in {0, 1, 2} (4.2.3),
    state_transition_delta yields transitions in 1..255 (4.2.4), colorspace_type in {0, 1} (4.2.5), bits_per_raw_sample (4.2.7),
    chroma subsampling 0 for RGB (4.2.9/10), num_h/v_slices (4.2.12/13), 1 <= quant_table_set_count <= 8 (4.2.14), every quantisation
    table covers exactly 128 entries with run lengths and its levels never decrease (4.9), context count = (product + 1) / 2 (4.9.3),
    states_coded (4.2.15), ec in {0, 1} (4.2.16), intra (4.2.17), CRC-32 parity zero over the record (4.3.2)
  * frame (4.4): slices found from the END of the packet through their footers (4.6, 4.7.5: slice_size is 24 bit, error_status 0),
    they tile the packet exactly; with ec == 1 the CRC over each slice incl. its parity is zero (4.9 "slice_crc_parity")
  * slice header (4.5): keyframe bit in front of the first slice (4.4), slice_x/slice_y inside the grid and every grid position
    present exactly once, slice_width_minus1 / slice_height_minus1 keep the rectangle inside the grid, quant_table_set_index <
    quant_table_set_count per plane group, picture_structure in 0..3, no sar_num without a sar_den
The range decoder below is the one of RFC 9043 section 3.8.1 (binary arithmetic coder, 3.8.1.2 "unsigned/signed integer symbols").

`decode_frame` goes further than structure: it decodes every SAMPLE of a frame with code written from the RFC's text alone -- border
(3.1), sample positions (3.2), median predictor and its 16-bit exception (3.3), quantisation table sets and the context (3.4, 3.5, 4.9),
JPEG 2000 RCT with the offset and the BGR exception for 9..15 bits without alpha (3.7.2), sample differences folded to `bits` (3.8),
signed symbols (3.8.1.2), line-interleaved planes of an RGB slice (3.7.2 / 4.6), sentinel-mode termination (3.8.1.1.1) -- so that three
independent decoders (the reference's, the oracle's restatement of it, and this one) must agree on every golden packet.  There is no
network in this environment: "from the RFC" means from the published text as its author of this file knows it, not from a copy at hand;
what stays unpinned is FFmpeg itself.
"""
from __future__ import annotations

import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))


def default_state_transition() -> list[int]:
    """RFC 9043 4.2.4 / 3.8.1.5 default_state_transition[256]: a bitstream constant; read from the oracle's C table (same 256 numbers)."""
    src = open(os.path.join(_HERE, "..", "oracle", "ffv1_oracle.c")).read()
    m = re.search(r"one_state_default\[256\]\s*=\s*\{([^}]*)\}", src)
    t = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    assert len(t) == 256
    return t


def crc32_mpeg(data: bytes) -> int:
    """4.9.? CRC-32: polynomial 0x04C11DB7, MSB first, initial value 0, no final xor."""
    c = 0
    for b in data:
        c ^= b << 24
        for _ in range(8):
            c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
    return c


_CRC_T = None


def crc32_fast(data: bytes) -> int:
    global _CRC_T
    if _CRC_T is None:
        _CRC_T = []
        for i in range(256):
            c = i << 24
            for _ in range(8):
                c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
            _CRC_T.append(c)
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFFFFFF) ^ _CRC_T[(c >> 24) ^ b]
    return c


class RangeDecoder:
    """RFC 9043 3.8.1: range coder with adaptive 8-bit states."""

    def __init__(self, data: bytes, one_state: list[int]):
        self.d = data
        self.pos = 2
        self.low = (data[0] << 8 | data[1]) if len(data) >= 2 else 0
        self.range = 0xFF00
        if self.low >= 0xFF00:                       # 3.8.1.3 initial values
            self.low = 0xFF00
            self.pos = len(data)
        self.one = one_state
        self.zero = [0] * 256
        for i in range(1, 256):
            self.zero[i] = 256 - one_state[256 - i]

    def _refill(self):
        if self.range < 0x100:
            self.range <<= 8
            self.low <<= 8
            if self.pos < len(self.d):
                self.low += self.d[self.pos]
            self.pos += 1

    def bit(self, st: list[int], i: int) -> int:
        r = (self.range * st[i]) >> 8
        self.range -= r
        if self.low < self.range:
            st[i] = self.zero[st[i]]
            self._refill()
            return 0
        self.low -= self.range
        self.range = r
        st[i] = self.one[st[i]]
        self._refill()
        return 1

    def symbol(self, st: list[int], signed: bool) -> int:
        """3.8.1.2"""
        if self.bit(st, 0):
            return 0
        e = 0
        while self.bit(st, 1 + min(e, 9)):
            e += 1
            assert e <= 31, "range coder: exponent of a symbol above 31"
        a = 1
        for i in range(e - 1, -1, -1):
            a = a * 2 + self.bit(st, 22 + min(i, 9))
        if signed and self.bit(st, 11 + min(e, 10)):
            return -a
        return a


class Record:
    pass


def parse_record(rec: bytes) -> Record:
    assert len(rec) >= 4 + 2, "configuration record too short"
    assert crc32_fast(rec) == 0, "4.3.2: configuration_record_crc_parity does not bring the CRC to zero"
    one = default_state_transition()
    rd = RangeDecoder(rec[:-4], one)
    st = [128] * 32
    r = Record()
    r.version = rd.symbol(st, False)
    assert r.version == 3, f"4.2.1: version {r.version} in a configuration record"
    r.micro_version = rd.symbol(st, False)
    assert r.micro_version >= 4, f"4.2.2: micro_version {r.micro_version} < 4 (experimental range of version 3)"
    r.coder_type = rd.symbol(st, False)
    assert r.coder_type in (0, 1, 2), f"4.2.3: coder_type {r.coder_type}"
    r.one_state = list(one)
    if r.coder_type == 2:
        for i in range(1, 256):
            r.one_state[i] = (one[i] + rd.symbol(st, True)) & 0xFF
            assert 1 <= r.one_state[i] <= 255, f"4.2.4: state transition {i} -> {r.one_state[i]}"
    r.colorspace_type = rd.symbol(st, False)
    assert r.colorspace_type in (0, 1), f"4.2.5: colorspace_type {r.colorspace_type}"
    r.bits_per_raw_sample = rd.symbol(st, False)
    assert 0 <= r.bits_per_raw_sample <= 16, f"4.2.7: bits_per_raw_sample {r.bits_per_raw_sample}"
    if r.bits_per_raw_sample == 0:
        r.bits_per_raw_sample = 8
    r.chroma_planes = rd.bit(st, 0)
    r.log2_h = rd.symbol(st, False)
    r.log2_v = rd.symbol(st, False)
    r.alpha_plane = rd.bit(st, 0)
    if r.colorspace_type == 1:
        assert r.chroma_planes == 1 and r.log2_h == 0 and r.log2_v == 0, "4.2.5: RGB needs chroma_planes 1 and no subsampling"
    r.num_h_slices = rd.symbol(st, False) + 1
    r.num_v_slices = rd.symbol(st, False) + 1
    assert 1 <= r.num_h_slices <= 256 and 1 <= r.num_v_slices <= 256
    r.quant_table_set_count = rd.symbol(st, False)
    assert 1 <= r.quant_table_set_count <= 8, f"4.2.14: quant_table_set_count {r.quant_table_set_count}"
    r.context_count = []
    r.tables = []
    for _ in range(r.quant_table_set_count):
        scale = 1
        tabs = []
        for _t in range(5):
            qs = [128] * 32                                    # 4.9: every table starts from fresh states
            v = 0
            i = 0
            levels = []
            while i < 128:
                n = rd.symbol(qs, False) + 1
                assert i + n <= 128, "4.9.1: quantisation table run crosses index 127"
                levels += [v] * n
                i += n
                v += 1
            assert levels == sorted(levels) and levels[0] == 0, "4.9.1: levels must start at 0 and never decrease"
            tabs.append([x * scale for x in levels])
            scale *= 2 * levels[127] + 1                       # 4.9.3: (2 * len_count - 1) with len_count = number of levels
        ctx = (scale + 1) // 2
        assert 1 <= ctx <= 32768, f"4.9.3: context count {ctx}"
        r.context_count.append(ctx)
        r.tables.append(tabs)
    r.states_coded = [rd.bit(st, 0) for _ in range(r.quant_table_set_count)]
    if any(r.states_coded):
        for k, ctx in enumerate(r.context_count):
            if r.states_coded[k]:
                for _c in range(ctx):
                    s2 = [128] * 32
                    for _j in range(32):
                        rd.symbol(s2, True)
    r.ec = rd.symbol(st, False)
    assert r.ec in (0, 1), f"4.2.16: ec {r.ec}"
    r.intra = rd.symbol(st, False)
    assert r.intra in (0, 1), f"4.2.17: intra {r.intra}"
    return r


def split_slices(r: Record, packet: bytes) -> list[tuple[int, int]]:
    """4.6 / 4.7.5: walk the footers from the end; returns (start, end) of every slice in bitstream order."""
    tail = 8 if r.ec else 3
    out = []
    end = len(packet)
    while end > 0:
        assert end >= tail, "4.7: slice footer does not fit"
        size = int.from_bytes(packet[end - tail:end - tail + 3], "big")
        start = end - tail - size
        assert start >= 0, f"4.7.5: slice_size {size} reaches in front of the packet"
        if r.ec:
            assert packet[end - 5] == 0, f"4.7.6: error_status {packet[end - 5]}"
            assert crc32_fast(packet[start:end]) == 0, f"4.7.7: slice_crc_parity wrong for the slice at {start}"
        out.append((start, end))
        end = start
    out.reverse()
    assert out and out[0][0] == 0
    return out


def validate_frame(r: Record, packet: bytes, width: int, height: int) -> dict:
    slices = split_slices(r, packet)
    assert len(slices) == r.num_h_slices * r.num_v_slices, f"{len(slices)} slices in the packet, grid is {r.num_h_slices}x{r.num_v_slices}"
    seen = {}
    planes_sets = 2 + (1 if r.alpha_plane else 0)              # 4.5.5: 1 + (version <= 3 || chroma_planes) + alpha_plane
    for n, (a, b) in enumerate(slices):
        rd = RangeDecoder(packet[a:b], r.one_state)
        if n == 0:
            st = [128] * 32
            assert rd.bit(st, 0) == 1, "4.4: keyframe bit must be 1 in an intra-only stream"
        st = [128] * 32                                        # 4.5: slice header states
        sx = rd.symbol(st, False)
        sy = rd.symbol(st, False)
        sw = rd.symbol(st, False) + 1
        sh = rd.symbol(st, False) + 1
        assert sx + sw <= r.num_h_slices and sy + sh <= r.num_v_slices, f"4.5.1-4: slice rectangle ({sx},{sy},{sw},{sh}) leaves the grid"
        for yy in range(sy, sy + sh):
            for xx in range(sx, sx + sw):
                assert (xx, yy) not in seen, f"grid position ({xx},{yy}) coded twice"
                seen[(xx, yy)] = n
        for _p in range(planes_sets):
            q = rd.symbol(st, False)
            assert q < r.quant_table_set_count, f"4.5.5: quant_table_set_index {q}"
        ps = rd.symbol(st, False)
        assert 0 <= ps <= 3, f"4.5.6: picture_structure {ps}"
        sar_num = rd.symbol(st, False)
        sar_den = rd.symbol(st, False)
        assert sar_den != 0 or sar_num == 0, "4.5.7/8: sar_num without sar_den"          # 0/0 and 0/1 both mean "unknown"
        # the slice's pixel rectangle must not be empty (4.5.3 slice_pixel_width / height)
        x0, x1 = sx * width // r.num_h_slices, (sx + sw) * width // r.num_h_slices
        y0, y1 = sy * height // r.num_v_slices, (sy + sh) * height // r.num_v_slices
        assert x1 > x0 and y1 > y0, "empty slice"
    assert len(seen) == r.num_h_slices * r.num_v_slices, "some grid positions are not covered by any slice"
    return {"slices": len(slices), "bytes": len(packet)}


def _quant_full(levels128: list[int]) -> list[int]:
    """4.9: a table covers the differences 0..127 as coded; -1..-127 are the negated values, 128 (= -128) mirrors 127."""
    t = list(levels128) + [0] * 128
    for i in range(1, 128):
        t[256 - i] = -t[i]
    t[128] = -t[127]
    return t


def decode_frame(r: Record, packet: bytes, width: int, height: int, only=None):
    """All samples of one version-3 intra frame, decoded from the RFC: returns a list of rows, each a list of pixels, each a tuple of
    components in picture order -- (R, G, B[, A]) for colorspace_type 1, (Y,) for a single plane.  `only`: a set of slice numbers (in
    bitstream order) to decode; the pixels of the others stay None (a 4K slice takes this pure-Python loop a quarter of a minute)."""
    rgb = r.colorspace_type == 1
    assert rgb or not r.chroma_planes, "decode_frame: YCbCr with chroma planes is outside this repository's formats"
    nplanes = (3 + r.alpha_plane) if rgb else 1
    bps = r.bits_per_raw_sample
    bits = bps + 1 if rgb else bps                               # 3.8: RGB differences have one more bit (the RCT widens Cb, Cr)
    mask = (1 << bits) - 1
    cast16 = (not rgb) and bps == 16                             # 3.3 exception: l, t, tl as int16 for 16-bit YCbCr with the range coder
    swap_gb = rgb and not r.alpha_plane and 9 <= bps <= 15       # 3.7.2 exception: B and G trade places
    one, zero = r.one_state, [0] * 256
    for i in range(1, 256):
        zero[i] = (256 - one[256 - i]) & 0xFF                    # 3.8.1.5 zero_state_i = 256 - one_state_(256 - i)
    out = [[None] * width for _ in range(height)]
    for n, (a, b) in enumerate(split_slices(r, packet)):
        if only is not None and n not in only:
            continue
        tail = 8 if r.ec else 3
        rd = RangeDecoder(packet[a:b - tail], one)
        if n == 0:
            assert rd.bit([128] * 32, 0) == 1                    # 4.4 keyframe
        st = [128] * 32
        sx, sy = rd.symbol(st, False), rd.symbol(st, False)
        sw, sh = rd.symbol(st, False) + 1, rd.symbol(st, False) + 1
        nidx = 2 + (1 if r.alpha_plane else 0)
        qidx = [rd.symbol(st, False) for _ in range(nidx)]
        rd.symbol(st, False); rd.symbol(st, False); rd.symbol(st, False)        # picture_structure, sar_num, sar_den
        x0, x1 = sx * width // r.num_h_slices, (sx + sw) * width // r.num_h_slices      # 4.5.x slice_pixel_*
        y0, y1 = sy * height // r.num_v_slices, (sy + sh) * height // r.num_v_slices
        w, h = x1 - x0, y1 - y0
        # per plane group: the five tables of its set, and 32 states of 128 per context (states_coded == 0, keyframe)
        group_of = [0, 1, 1, 2][:nplanes]
        Q = [[_quant_full(t) for t in r.tables[qidx[g]]] for g in range(len(qidx))]
        states = [[[128] * 32 for _ in range(r.context_count[qidx[g]])] for g in range(len(qidx))]
        assert not any(r.states_coded), "decode_frame: initial states in the record are not used by this repository"
        # three rows per plane with the 3.1 border: two columns to the left, one to the right
        rows = [[[0] * (w + 3) for _ in range(3)] for _ in range(nplanes)]
        d, dlen = rd.d, len(rd.d)
        low, rng, pos = rd.low, rd.range, rd.pos
        for y in range(h):
            for p in range(nplanes):
                g = group_of[p]
                q0, q1, q2, q3, q4 = Q[g]
                five = q3[127] != 0 or q4[127] != 0
                stp = states[g]
                cur, prev, pprev = rows[p][y % 3], rows[p][(y + 2) % 3], rows[p][(y + 1) % 3]
                if y == 0:
                    for i in range(w + 3): prev[i] = 0
                if y <= 1:
                    for i in range(w + 3): pprev[i] = 0
                # column index = x + 2.  3.1: the column left of the slice is the leftmost column shifted down by one row (0 on top),
                # the one right of it repeats the rightmost, the second column to the left and everything above are 0
                cur[0] = 0
                cur[1] = prev[2] if y else 0
                prev[w + 2] = prev[w + 1]
                for x in range(w):
                    i = x + 2
                    L2, l = cur[i - 2], cur[i - 1]
                    tl, t, tr, T2 = prev[i - 1], prev[i], prev[i + 1], pprev[i]
                    ctx = q0[(l - tl) & 255] + q1[(tl - t) & 255] + q2[(t - tr) & 255]
                    if five:
                        ctx += q3[(L2 - l) & 255] + q4[(T2 - t) & 255]
                    if cast16:
                        ls = ((l & 0xFFFF) ^ 0x8000) - 0x8000; ts = ((t & 0xFFFF) ^ 0x8000) - 0x8000; tls = ((tl & 0xFFFF) ^ 0x8000) - 0x8000
                        grad = ls + ts - tls
                        pred = sorted((ls, ts, grad))[1]
                    else:
                        grad = l + t - tl
                        pred = l if (t <= l <= grad or grad <= l <= t) else t if (l <= t <= grad or grad <= t <= l) else grad
                    neg = ctx < 0
                    if neg:
                        ctx = -ctx
                    s32 = stp[ctx]
                    # ---- 3.8.1.2 get_symbol(signed), the range decoder of 3.8.1.1 inlined (this loop is the whole cost of the test)
                    def getbit(k):
                        nonlocal low, rng, pos
                        sv = s32[k]
                        rr = (rng * sv) >> 8
                        rng -= rr
                        if low < rng:
                            s32[k] = zero[sv]; bitv = 0
                        else:
                            low -= rng; rng = rr; s32[k] = one[sv]; bitv = 1
                        if rng < 0x100:
                            rng <<= 8; low <<= 8
                            if pos < dlen: low += d[pos]
                            pos += 1
                        return bitv
                    if getbit(0):
                        diff = 0
                    else:
                        e = 0
                        while getbit(1 + (e if e < 9 else 9)):
                            e += 1
                            assert e <= 31
                        aa = 1
                        for k in range(e - 1, -1, -1):
                            aa = aa * 2 + getbit(22 + (k if k < 9 else 9))
                        diff = -aa if getbit(11 + (e if e < 10 else 10)) else aa
                    if neg:
                        diff = -diff
                    cur[i] = (pred + diff) & mask
            # the picture line is complete: 3.7.2 inverse RCT
            for x in range(w):
                if rgb:
                    Y, Cb, Cr = rows[0][y % 3][x + 2], rows[1][y % 3][x + 2] - (1 << bps), rows[2][y % 3][x + 2] - (1 << bps)
                    gg = Y - ((Cb + Cr) >> 2)
                    rr_, bb = Cr + gg, Cb + gg
                    if swap_gb:
                        gg, bb = bb, gg
                    px = (rr_, gg, bb) + ((rows[3][y % 3][x + 2],) if r.alpha_plane else ())
                else:
                    px = (rows[0][y % 3][x + 2],)
                out[y0 + y][x0 + x] = px
        # 3.8.1.1.1 sentinel mode: one more binary symbol with state 129, its value discarded; the bytes must not have run out before
        rd.low, rd.range, rd.pos = low, rng, pos
        rd.bit([129], 0)
        assert rd.pos - 1 <= dlen + 1, "slice data ran out before its last symbol"
    assert only is not None or all(px is not None for row in out for px in row), "some pixels are covered by no slice"
    return out


def validate_stream(record: bytes, packets: list[bytes], width: int, height: int) -> Record:
    r = parse_record(record)
    for p in packets:
        validate_frame(r, p, width, height)
    return r
