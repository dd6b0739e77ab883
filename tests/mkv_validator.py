"""Structural validator for the Matroska files this repository's muxer writes, from the Matroska element specification (RFC 9559) and
EBML (RFC 8794): no stock player or mkvalidator exists in this environment, so what they would check first is checked here.

TEST INFRASTRUCTURE.  validate(path) walks the whole file and asserts:
  * EBML header: DocType "matroska", DocTypeVersion >= DocTypeReadVersion, EBMLMaxIDLength 4, EBMLMaxSizeLength 8
  * exactly one Segment of KNOWN size that ends where its last child ends; anything behind it is a further top-level EBML document
    (RAWcooked's --output-version 2 appends one)
  * every element ID is one the specification defines at that level; sizes nest exactly; integers <= 8 bytes, floats 4 or 8 bytes
  * SeekHead: every Seek points at an element with the ID it names; Info: TimestampScale, MuxingApp, WritingApp, Duration >= the last
    block's end; Tracks: numbers 1..n unique, UIDs unique and non-zero, TrackType 1 with Video{PixelWidth, PixelHeight} / 2 with
    Audio{SamplingFrequency, Channels}, CodecID set, FlagLacing 0
  * Attachments before the first Cluster, each with FileName, FileMimeType, FileUID, FileData; Tags target existing TrackUIDs
  * Clusters: Timestamp first, then SimpleBlocks only; block header = 1-byte track number of an existing track, int16 relative time,
    no lacing; absolute block times per track never decrease
  * Cues: every CuePoint names a Cluster position that is a Cluster, whose Timestamp is <= CueTime, and that holds a block of the
    cue's track at CueTime
"""
from __future__ import annotations

import struct

M, U, S, B, F, D = "master", "uint", "string", "binary", "float", "date"
SCHEMA = {
    None: {0x1A45DFA3: ("EBML", M), 0x18538067: ("Segment", M), 0xEC: ("Void", B)},
    "EBML": {0x4286: ("EBMLVersion", U), 0x42F7: ("EBMLReadVersion", U), 0x42F2: ("EBMLMaxIDLength", U), 0x42F3: ("EBMLMaxSizeLength", U),
             0x4282: ("DocType", S), 0x4287: ("DocTypeVersion", U), 0x4285: ("DocTypeReadVersion", U)},
    "Segment": {0x114D9B74: ("SeekHead", M), 0x1549A966: ("Info", M), 0x1654AE6B: ("Tracks", M), 0x1254C367: ("Tags", M),
                0x1941A469: ("Attachments", M), 0x1F43B675: ("Cluster", M), 0x1C53BB6B: ("Cues", M), 0xEC: ("Void", B)},
    "SeekHead": {0x4DBB: ("Seek", M), 0xEC: ("Void", B)},
    "Seek": {0x53AB: ("SeekID", B), 0x53AC: ("SeekPosition", U)},
    "Info": {0x2AD7B1: ("TimestampScale", U), 0x4D80: ("MuxingApp", S), 0x5741: ("WritingApp", S), 0x73A4: ("SegmentUUID", B), 0x4489: ("Duration", F)},
    "Tracks": {0xAE: ("TrackEntry", M)},
    "TrackEntry": {0xD7: ("TrackNumber", U), 0x73C5: ("TrackUID", U), 0x83: ("TrackType", U), 0x9C: ("FlagLacing", U), 0x22B59C: ("Language", S),
                   0x86: ("CodecID", S), 0x63A2: ("CodecPrivate", B), 0x23E383: ("DefaultDuration", U), 0xE0: ("Video", M), 0xE1: ("Audio", M)},
    "Video": {0x9A: ("FlagInterlaced", U), 0xB0: ("PixelWidth", U), 0xBA: ("PixelHeight", U), 0x54B0: ("DisplayWidth", U), 0x54BA: ("DisplayHeight", U)},
    "Audio": {0xB5: ("SamplingFrequency", F), 0x9F: ("Channels", U), 0x6264: ("BitDepth", U)},
    "Tags": {0x7373: ("Tag", M)},
    "Tag": {0x63C0: ("Targets", M), 0x67C8: ("SimpleTag", M)},
    "Targets": {0x68CA: ("TargetTypeValue", U), 0x63C5: ("TagTrackUID", U)},
    "SimpleTag": {0x45A3: ("TagName", S), 0x447A: ("TagLanguage", S), 0x4487: ("TagString", S)},
    "Attachments": {0x61A7: ("AttachedFile", M)},
    "AttachedFile": {0x466E: ("FileName", S), 0x4660: ("FileMimeType", S), 0x46AE: ("FileUID", U), 0x465C: ("FileData", B)},
    "Cluster": {0xE7: ("Timestamp", U), 0xA3: ("SimpleBlock", B)},
    "Cues": {0xBB: ("CuePoint", M)},
    "CuePoint": {0xB3: ("CueTime", U), 0xB7: ("CueTrackPositions", M)},
    "CueTrackPositions": {0xF7: ("CueTrack", U), 0xF1: ("CueClusterPosition", U)},
}


def _vint(f, pos, keep_marker):
    f.seek(pos)
    first = f.read(1)
    assert first, f"unexpected end of file at {pos}"
    b0 = first[0]
    assert b0, f"EBML: invalid length descriptor 0 at {pos}"
    n = 1
    while not (b0 & (0x80 >> (n - 1))):
        n += 1
    rest = f.read(n - 1)
    assert len(rest) == n - 1
    v = b0 if keep_marker else b0 & (0xFF >> n)
    for c in rest:
        v = v << 8 | c
    unknown = (not keep_marker) and v == (1 << (7 * n)) - 1
    return v, n, unknown


class Node:
    __slots__ = ("name", "type", "id", "pos", "data", "end", "children", "value")

    def child(self, name):
        c = [x for x in self.children if x.name == name]
        assert len(c) == 1, f"{self.name}: expected exactly one {name}, found {len(c)}"
        return c[0]

    def all(self, name):
        return [x for x in self.children if x.name == name]

    def opt(self, name):
        c = self.all(name)
        assert len(c) <= 1
        return c[0] if c else None


def _parse(f, pos, end, level, max_value_bytes=1 << 16):
    out = []
    while pos < end:
        eid, n1, _ = _vint(f, pos, True)
        assert n1 <= 4, f"element ID longer than 4 bytes at {pos}"
        size, n2, unknown = _vint(f, pos + n1, False)
        assert not unknown, f"element {eid:X} at {pos} has an unknown size (not allowed here)"
        schema = SCHEMA[level]
        assert eid in schema, f"element {eid:X} at {pos} is not defined inside {level or 'the top level'}"
        nd = Node()
        nd.name, nd.type = schema[eid]
        nd.id, nd.pos, nd.data, nd.end = eid, pos, pos + n1 + n2, pos + n1 + n2 + size
        assert nd.end <= end, f"{nd.name} at {pos} runs past its parent ({nd.end} > {end})"
        nd.children, nd.value = [], None
        if nd.type == M:
            nd.children = _parse(f, nd.data, nd.end, nd.name)
        else:
            if nd.type in (U, D):
                assert size <= 8, f"{nd.name}: integer of {size} bytes"
            if nd.type == F:
                assert size in (0, 4, 8), f"{nd.name}: float of {size} bytes"
            if size <= max_value_bytes or nd.name == "SimpleBlock":
                f.seek(nd.data)
                raw = f.read(min(size, 16) if nd.name == "SimpleBlock" else size)
                if nd.type == U:
                    nd.value = int.from_bytes(raw, "big")
                elif nd.type == F:
                    nd.value = struct.unpack(">d" if size == 8 else ">f", raw)[0] if size else 0.0
                elif nd.type == S:
                    nd.value = raw.rstrip(b"\0").decode("utf-8")
                else:
                    nd.value = raw
        out.append(nd)
        pos = nd.end
    assert pos == end, f"children of {level} end at {pos}, parent ends at {end}"
    return out


def validate(path: str, on_block=None) -> dict:
    """on_block(track, absolute_time, payload_offset, payload_end): called for every SimpleBlock, with where its payload lies in the file."""
    import os
    fsize = os.path.getsize(path)
    with open(path, "rb") as f:
        # top level: EBML header, Segment, then optionally further EBML documents (RAWcooked v2 reversibility data)
        eid, n1, _ = _vint(f, 0, True)
        assert eid == 0x1A45DFA3, "file does not start with an EBML header"
        top = []
        pos = 0
        while pos < fsize and len(top) < 2:
            top += _parse(f, pos, _next_end(f, pos), None)
            pos = top[-1].end
        head, seg = top[0], top[1]
        assert head.name == "EBML" and seg.name == "Segment"
        assert head.child("DocType").value == "matroska"
        assert head.child("DocTypeVersion").value >= head.child("DocTypeReadVersion").value >= 1
        assert head.child("EBMLMaxIDLength").value == 4 and head.child("EBMLMaxSizeLength").value == 8
        trailing = fsize - seg.end
        if trailing:
            eid, _, _ = _vint(f, seg.end, True)
            assert eid == 0x1A45DFA3, "bytes behind the Segment that are not an EBML document"
        names = [c.name for c in seg.children]
        assert names.count("Info") == 1 and names.count("Tracks") == 1
        first_cluster = names.index("Cluster") if "Cluster" in names else len(names)
        for must_be_early in ("Info", "Tracks", "Attachments", "Tags"):
            if must_be_early in names:
                assert names.index(must_be_early) < first_cluster, f"{must_be_early} behind the first Cluster"
        # SeekHead
        for sh in seg.all("SeekHead"):
            for sk in sh.all("Seek"):
                at = seg.data + sk.child("SeekPosition").value
                f.seek(at)
                want = sk.child("SeekID").value
                assert f.read(len(want)) == want, f"SeekHead entry {want.hex()} points at something else"
        info = seg.child("Info")
        scale = info.child("TimestampScale").value
        assert scale > 0 and info.child("MuxingApp").value and info.child("WritingApp").value
        # Tracks
        tracks = {}
        uids = set()
        for te in seg.child("Tracks").all("TrackEntry"):
            num, uid, typ = te.child("TrackNumber").value, te.child("TrackUID").value, te.child("TrackType").value
            assert num not in tracks and 1 <= num <= 126 and uid and uid not in uids
            uids.add(uid)
            assert te.child("CodecID").value
            lac = te.opt("FlagLacing")
            assert lac is None or lac.value == 0, "blocks are never laced here"
            if typ == 1:
                v = te.child("Video")
                assert v.child("PixelWidth").value > 0 and v.child("PixelHeight").value > 0
            else:
                assert typ == 2
                a = te.child("Audio")
                assert a.child("SamplingFrequency").value > 0 and a.child("Channels").value > 0
            tracks[num] = {"uid": uid, "type": typ, "codec": te.child("CodecID").value, "last": -1, "blocks": 0,
                           "default_duration": (te.opt("DefaultDuration").value if te.opt("DefaultDuration") else 0)}
        assert sorted(tracks) == list(range(1, len(tracks) + 1)), "track numbers must be 1..n"
        for att in (seg.opt("Attachments").all("AttachedFile") if seg.opt("Attachments") else []):
            assert att.child("FileName").value and att.child("FileMimeType").value and att.child("FileUID").value
            att.child("FileData")
        for tag in (seg.opt("Tags").all("Tag") if seg.opt("Tags") else []):
            t = tag.child("Targets").opt("TagTrackUID")
            assert t is None or t.value in uids, "Tag targets a TrackUID that does not exist"
            for st in tag.all("SimpleTag"):
                assert st.child("TagName").value
        # Clusters
        clusters = {}
        end_time = 0
        for c in seg.all("Cluster"):
            assert c.children and c.children[0].name == "Timestamp", "a Cluster must start with its Timestamp"
            ts = c.children[0].value
            held = set()
            for blk in c.children[1:]:
                assert blk.name == "SimpleBlock"
                raw = blk.value
                assert raw[0] & 0x80, "track number of a SimpleBlock must be a 1-byte vint here"
                trk = raw[0] & 0x7F
                assert trk in tracks, f"SimpleBlock of unknown track {trk}"
                rel = struct.unpack(">h", raw[1:3])[0]
                assert (raw[3] & 0x06) == 0, "laced SimpleBlock"
                t_abs = ts + rel
                assert t_abs >= tracks[trk]["last"], f"track {trk}: block time goes backwards ({t_abs} < {tracks[trk]['last']})"
                tracks[trk]["last"] = t_abs
                tracks[trk]["blocks"] += 1
                held.add((trk, t_abs))
                end_time = max(end_time, t_abs)
                assert blk.end - blk.data > 4, "empty SimpleBlock"
                if on_block is not None:
                    on_block(trk, t_abs, blk.data + 4, blk.end)
            clusters[c.pos - seg.data] = (ts, held)
        dur = info.opt("Duration")
        if dur is not None and clusters:
            assert dur.value >= end_time, f"Duration {dur.value} < last block time {end_time}"
        cues = seg.opt("Cues")
        ncues = 0
        if cues is not None:
            for cp in cues.all("CuePoint"):
                t = cp.child("CueTime").value
                for tp in cp.all("CueTrackPositions"):
                    at = tp.child("CueClusterPosition").value
                    assert at in clusters, f"Cue at {t} points at {at}, which is not a Cluster"
                    assert clusters[at][0] <= t and (tp.child("CueTrack").value, t) in clusters[at][1], f"Cue {t}: its Cluster holds no such block"
                    ncues += 1
        return {"tracks": {k: {"type": v["type"], "codec": v["codec"], "blocks": v["blocks"]} for k, v in tracks.items()},
                "clusters": len(clusters), "cues": ncues, "trailing_bytes": trailing, "segment_bytes": seg.end - seg.data}


def _next_end(f, pos):
    eid, n1, _ = _vint(f, pos, True)
    size, n2, unknown = _vint(f, pos + n1, False)
    assert not unknown, "top-level element of unknown size"
    return pos + n1 + n2 + size
