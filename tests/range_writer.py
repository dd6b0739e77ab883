"""A range ENCODER for tests: RFC 9043 3.8.1 the other way round (what FFmpeg's rangecoder.c does when it writes), and a writer of FFV1
configuration records from raw field values -- for records no encoder would make (hostile run lengths, out-of-range fields), which the
mutation tests do not reach.  Test infrastructure; the product has its own writer (rawcooked_amd/csrc/ffv1_host.cpp)."""
from __future__ import annotations


class RangeEncoder:
    """RFC 9043 3.8.1 the other way round (what FFmpeg's rangecoder.c writes), for records no encoder would make."""
    def __init__(self):
        from rfc9043_validator import default_state_transition
        self.one = default_state_transition()
        self.zero = [0] * 256
        for i in range(1, 256):
            self.zero[i] = (256 - self.one[256 - i]) & 0xFF
        self.low, self.range, self.out, self.pending, self.pending_ff = 0, 0xFF00, bytearray(), -1, 0

    def transitions(self, one):
        """From here on the stream's own table (rangecoder::AssignStateTransitions, FFV1_RangeCoder.cpp:36-42); the states keep their values."""
        self.one = list(one)
        self.zero = [0] * 256
        for i in range(1, 256):
            self.zero[i] = (256 - self.one[256 - i]) & 0xFF

    def _renorm(self):
        while self.range < 0x100:
            if self.pending < 0:
                self.pending = self.low >> 8
            elif self.low <= 0xFF00:
                self.out.append(self.pending); self.out += b"\xff" * self.pending_ff; self.pending_ff = 0; self.pending = self.low >> 8
            elif self.low >= 0x10000:
                self.out.append(self.pending + 1); self.out += b"\x00" * self.pending_ff; self.pending_ff = 0; self.pending = (self.low >> 8) & 0xFF
            else:
                self.pending_ff += 1
            self.low = (self.low & 0xFF) << 8
            self.range <<= 8

    def bit(self, st, i, b):
        r1 = (self.range * st[i]) >> 8
        if b:
            self.low += self.range - r1; self.range = r1; st[i] = self.one[st[i]]
        else:
            self.range -= r1; st[i] = self.zero[st[i]]
        self._renorm()

    def u(self, st, v):
        v = int(v)
        self.bit(st, 0, v == 0)
        if v == 0:
            return
        e = v.bit_length() - 1
        for k in range(e):
            self.bit(st, 1 + min(k, 9), 1)
        if e < 32:                                  # (a value of 2^32 and more: 32 ones and no terminator -- the exponent the reference condemns)
            self.bit(st, 1 + min(e, 9), 0)
        for k in range(min(e, 31) - 1, -1, -1):
            self.bit(st, 22 + min(k, 9), (v >> k) & 1)

    def s(self, st, v):
        self.bit(st, 0, v == 0)
        if v == 0:
            return
        v = int(v); a = abs(v)
        e = a.bit_length() - 1
        for k in range(e):
            self.bit(st, 1 + min(k, 9), 1)
        self.bit(st, 1 + min(e, 9), 0)
        for k in range(e - 1, -1, -1):
            self.bit(st, 22 + min(k, 9), (a >> k) & 1)
        self.bit(st, 11 + min(e, 10), v < 0)

    def done(self):
        self.range = 0xFF; self.low += 0xFF; self._renorm(); self.range = 0xFF; self._renorm()
        return bytes(self.out)


def sealed(body: bytes) -> bytes:
    """body + configuration_record_crc_parity (RFC 9043 4.3.2): CRC-32/MPEG-2 with initial value 0 over everything is then 0."""
    from rfc9043_validator import crc32_fast
    return body + crc32_fast(body).to_bytes(4, "big")


def record(version=3, micro=4, coder=1, deltas=None, colorspace=1, bits=16, chroma=1, log2h=0, log2v=0, alpha=0, h1=0, v1=0, sets=None, coded=None,
           ec=1, intra=1, tail=b"", upto=None) -> bytes:
    """A configuration record from raw field values, in the order parameters::Parse reads them (RFC 9043 4.2).  `sets`: per table set five
    lists of len_minus1 values (None: one set, every table one run of 128 = a single context); `coded`: per set None or a list of state
    values; `deltas`: 255 state_transition_delta values when coder == 2; `upto`: stop after this many fields (a record cut short)."""
    e, st, n = RangeEncoder(), [128] * 32, [0]

    def more():
        n[0] += 1
        return upto is None or n[0] <= upto
    if more(): e.u(st, version)
    if version >= 3 and more(): e.u(st, micro)
    if more(): e.u(st, coder)
    if coder == 2 and more():
        for d in (deltas or [0] * 255):
            e.s(st, d)
    if more(): e.u(st, colorspace)
    if version and more(): e.u(st, bits)
    if more(): e.bit(st, 0, chroma)
    if more(): e.u(st, log2h); e.u(st, log2v)
    if more(): e.bit(st, 0, alpha)
    sets = [[[127]] * 5] if sets is None else sets
    if version > 1 and more():
        e.u(st, h1); e.u(st, v1); e.u(st, len(sets))
    if more():
        for tables in sets:
            for runs in tables:
                q = [128] * 32
                for r in runs:
                    e.u(q, r)
    if more():
        for i in range(len(sets)):
            c = (coded or [None] * len(sets))[i]
            if version >= 3:
                e.bit(st, 0, 1 if c else 0)
            for v in (c or []):
                e.s(st, v)
    if version >= 3 and more():
        e.u(st, ec)
        if micro: e.u(st, intra)
    return sealed(e.done() + tail)


def one_state_of(deltas):
    """The transition table a record with coder_type 2 and these 255 deltas puts in force (FFV1_Parameters.cpp:41-55)."""
    from rfc9043_validator import default_state_transition
    d = default_state_transition()
    return [d[0]] + [(d[i] + int(deltas[i - 1])) & 0xFF for i in range(1, 256)]


def first_slice(indexes, key=1, xywh=(0, 0, 0, 0), one_state=None, tail=b"\x55" * 24) -> bytes:
    """The first bytes of a version 3 frame: the key frame bit, then the first slice's header (RFC 9043 4.5: slice_x, slice_y,
    slice_width - 1, slice_height - 1, one quant_table_set_index per plane group, picture_structure, sar_num, sar_den) -- under the stream's
    own transition table when it has one (FFV1_Slice.cpp:254-255) -- and `tail` for sample data."""
    e, st = RangeEncoder(), [128] * 32
    k = [128]
    e.bit(k, 0, key)
    if one_state is not None:
        e.transitions(one_state)
    for v in xywh:
        e.u(st, v)
    for v in indexes:
        e.u(st, v)
    e.u(st, 0); e.u(st, 0); e.u(st, 0)
    return e.done() + tail
