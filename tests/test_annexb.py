"""The byte-stream reader (include/ohevc_annexb.h) on hand-assembled NAL byte strings — no stream, no reference needed — and, where
the reference tree / oracle/_ref is present, against the writer's access-unit table and the reference decoder's own MD5 verdict."""
import hashlib
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from openhevc_amd import annexb as A


def nal(nut, payload=b"", layer=0, tid=0):
    return bytes([(nut << 1) | (layer >> 5), ((layer & 31) << 3) | (tid + 1)]) + payload


SC3, SC4 = b"\x00\x00\x01", b"\x00\x00\x00\x01"
VPS, SPS, PPS, AUD, SEI_P, SEI_S, TRAIL_R, IDR = 32, 33, 34, 35, 39, 40, 1, 19
FIRST, NOT_FIRST = b"\x80\x11\x22", b"\x00\x33\x44"


def test_access_units_split_where_the_parser_splits():
    """hevc_parser.c:40-87: parameter sets / AUD / prefix SEI after a picture open a new unit; so does a first slice segment; later
    slice segments, suffix SEI and enhancement-layer NAL units do not"""
    au0 = SC4 + nal(VPS, b"\x01") + SC4 + nal(SPS, b"\x02") + SC4 + nal(PPS, b"\x03") + SC3 + nal(IDR, FIRST) + SC3 + nal(IDR, NOT_FIRST) + SC3 + nal(SEI_S, b"\x84\x00\x80")
    au1 = SC3 + nal(TRAIL_R, FIRST) + SC3 + nal(TRAIL_R, NOT_FIRST)
    au2 = SC3 + nal(TRAIL_R, FIRST) + SC3 + nal(TRAIL_R, FIRST, layer=1) + SC3 + nal(PPS, b"\x09", layer=1)       # layer 1 never splits
    au3 = SC3 + nal(AUD, b"\x50") + SC3 + nal(SEI_P, b"\x01\x00\x80") + SC3 + nal(TRAIL_R, FIRST)
    au4 = SC3 + nal(48, b"\x00\x80") + SC3 + nal(TRAIL_R, FIRST)                                                    # reserved 48..55 opens one too
    data = au0 + au1 + au2 + au3 + au4
    cuts = [0]
    for a in (au0, au1, au2, au3, au4):
        cuts.append(cuts[-1] + len(a))
    assert A.split(data) == list(zip(cuts[:-1], cuts[1:]))
    # a four-byte start code at a boundary: its leading zero_byte stays with the unit before it (the parser returns i - 5)
    data = au0 + b"\x00" + au1
    assert A.split(data) == [(0, len(au0) + 1), (len(au0) + 1, len(data))]
    assert A.split(b"") == [(0, 0)] and A.split(au1) == [(0, len(au1))]


def test_streaming_scan_agrees_with_the_whole_buffer_scan():
    """the same stream fed in pieces of every size: the state carried across calls (state64, frame_start_found) finds the same cuts,
    reported relative to the piece — negative when the start code began in an earlier piece"""
    import ctypes as C
    aus = [SC3 + nal(IDR, FIRST + bytes(range(7)))] + [SC3 + nal(PPS, b"\x07") + SC3 + nal(TRAIL_R, FIRST + bytes([k] * k)) for k in range(1, 6)]
    data = b"".join(aus)
    want = [a for a, _ in A.split(data)][1:]
    L = A.lib()
    for piece in (1, 2, 3, 5, 7, 64):
        sc = A.OhAuScanner()
        L.oh_au_scanner_init(C.byref(sc))
        got, pos = [], 0
        while pos < len(data):
            chunk = data[pos:pos + piece]
            off = 0
            while off < len(chunk):
                r = L.oh_au_find_frame_end(C.byref(sc), chunk[off:], len(chunk) - off)
                if r == A.END_NOT_FOUND:
                    break
                got.append(pos + off + r)
                off += r + 6                                   # behind the byte that closed the decision
            pos += piece
        assert got == want, piece


def test_nal_scan_and_headers():
    au = b"\x00\x00" + SC4 + nal(SPS, b"\xaa\xbb") + b"\x00\x00" + SC3 + nal(TRAIL_R, FIRST, tid=2) + SC3 + nal(SEI_S, b"\x01", layer=3) + b"\x00"
    units = A.nal_units(au)
    assert [(t, l, tid, f) for _, _, t, l, tid, f in units] == [(SPS, 0, 0, 0), (TRAIL_R, 0, 2, 1), (SEI_S, 3, 0, 0)]
    for off, size, *_ in units:
        assert au[off - 3:off] == SC3
    assert units[0][1] == 4 and units[1][1] == 5 and units[2][1] == 3      # trailing zeros belong to no unit
    with pytest.raises(ValueError):
        A.nal_units(b"\x12\x00\x00\x01" + nal(SPS, b"\x01"))


def test_unescape_drops_emulation_prevention_and_keeps_positions():
    raw = nal(TRAIL_R, b"\x80\x00\x00\x03\x01\xff\x00\x00\x03\x00\x00\x03\x02\x10")
    rbsp, pos, used = A.unescape(raw)
    assert rbsp == nal(TRAIL_R, b"\x80\x00\x00\x01\xff\x00\x00\x00\x00\x02\x10") and used == len(raw)
    assert pos == [4, 8, 10]                                                # index of the second zero in front of each dropped byte
    # stops at the next start code; 00 00 03 at the very end is an escape too
    rbsp, pos, used = A.unescape(raw + SC3 + nal(PPS))
    assert used == len(raw) and rbsp[-1] == 0x10
    rbsp, pos, used = A.unescape(nal(PPS, b"\x11\x00\x00\x03"))
    assert rbsp == nal(PPS, b"\x11\x00\x00") and pos == [4]
    assert A.unescape(nal(PPS, b"\x11\x22"))[0] == nal(PPS, b"\x11\x22")


def escape(b):
    out, zeros = bytearray(), 0
    for v in b:
        if zeros >= 2 and v <= 3:
            out.append(3)
            zeros = 0
        out.append(v)
        zeros = zeros + 1 if v == 0 else 0
    return bytes(out)


def test_picture_hash_sei():
    digests = [hashlib.md5(bytes([c]) * 100).digest() for c in range(3)]
    digests[1] = b"\x00\x00\x00\x01" + digests[1][4:]                       # forces emulation prevention inside the payload
    msg = bytes([132, 49, 0]) + b"".join(digests)
    sei = nal(SEI_S, escape(msg + b"\x80"), tid=0)
    assert b"\x00\x00\x03" in sei
    assert A.picture_hash(sei) == (0, digests)
    # other messages in front of it, a long payload type (0xFF run), hash in a prefix SEI under the type the reference accepts there
    other = bytes([5, 3, 1, 2, 3])
    assert A.picture_hash(nal(SEI_S, escape(other + msg + b"\x80"))) == (0, digests)
    assert A.picture_hash(nal(SEI_P, escape(bytes([255, 1, 49, 0]) + b"".join(digests) + b"\x80"))) == (0, digests)
    assert A.picture_hash(nal(SEI_P, escape(msg + b"\x80"))) is None        # 132 in a PREFIX SEI is not a picture hash
    assert A.picture_hash(nal(SEI_S, escape(other + b"\x80"))) is None
    # CRC and checksum forms
    assert A.picture_hash(nal(SEI_S, bytes([132, 7, 1, 0x12, 0x34, 0x56, 0x78, 0x9a, 0xbc, 0x80]))) == (1, [0x1234, 0x5678, 0x9abc])
    assert A.picture_hash(nal(SEI_S, bytes([132, 13, 2]) + bytes(range(1, 13)) + b"\x80")) == (2, [0x01020304, 0x05060708, 0x090a0b0c])
    with pytest.raises(ValueError):
        A.picture_hash(nal(SEI_S, bytes([132, 49, 0]) + b"\x01" * 10))       # payload longer than the unit
    with pytest.raises(ValueError):
        A.picture_hash(nal(PPS, b"\x01\x02\x03"))


# ---- against the stream writer and the reference decoder ----
import refdec                                                               # noqa: E402
import streamgen                                                            # noqa: E402

need_ref = pytest.mark.skipif(not refdec.have_refdec(), reason="reference tree / oracle/_ref not present")


def test_splitter_on_written_streams():
    """the writer's own access-unit table (every unit opens with a four-byte start code whose zero_byte the parser leaves with the
    previous unit) for several picture structures"""
    for kw in (dict(), dict(n_slices=3), dict(tile_cols=2, tile_rows=2), dict(wpp=1), dict(gop=0), dict(idr_period=3)):
        data, aus = streamgen.write_stream(416, 240, 77, n_pictures=6, gop=kw.pop("gop", 2), **kw)
        got = A.split(data)
        assert [a for a, _ in got][1:] == [a + 1 for a, _ in aus][1:] and got[0][0] == 0 and got[-1][1] == len(data)
        for a, b in got:
            units = A.nal_units(data[a:b])
            assert sum(1 for u in units if u[5]) == 1                        # one first slice segment per access unit


def test_splitter_keeps_the_layers_of_an_access_unit_together():
    """two-layer (SHVC) streams: the enhancement layer's NAL units (nuh_layer_id 1: its parameter sets, its slice segments whose
    first_slice_segment_in_pic_flag is set too) stay in the access unit of their base-layer picture (hevc_parser.c:62, 71 test nuh_layer_id)"""
    for kw in (dict(), dict(n_slices=2), dict(wpp=1), dict(idr_period=2)):
        data, aus = streamgen.write_stream(192, 128, 78, n_pictures=5, gop=2, shvc_el_width=384, shvc_el_height=256, **kw)
        got = A.split(data)
        assert len(got) == 5 and [a for a, _ in got][1:] == [a + 1 for a, _ in aus][1:] and got[-1][1] == len(data)
        for a, b in got:
            units = A.nal_units(data[a:b])
            assert sorted({u[3] for u in units if u[2] < 32}) == [0, 1]      # slice segments of both layers
            assert sum(1 for u in units if u[5] and u[3] == 0) == 1 and sum(1 for u in units if u[5] and u[3] == 1) == 1


@need_ref
def test_hashes_read_from_the_stream_are_what_the_reference_checks():
    """digests written into SEI messages come back through oh_sei_picture_hash; they are the MD5s of the reference's output planes;
    the reference's own verdict on the same stream is "Correct MD5" for every plane"""
    import numpy as np
    data, aus = streamgen.write_stream(416, 240, 5, n_pictures=4, gop=2, bit_depth=10)
    pics = refdec.decode(data)
    want = [refdec.md5_of(p) for p in pics]
    with_sei, _ = streamgen.add_md5(data, aus, want)
    got = []
    for a, b in A.split(with_sei):
        au = with_sei[a:b]
        for off, size, t, *_ in A.nal_units(au):
            if t in (SEI_P, SEI_S):
                h = A.picture_hash(au[off:off + size])
                if h:
                    got.append(h[1])
    assert got == want
    with refdec.captured_stderr() as log:
        refdec.decode(with_sei, check_md5=True)
    assert log.text.count("Correct MD5") == 12 and "Incorrect MD5" not in log.text
