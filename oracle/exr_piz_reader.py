#!/usr/bin/env python3
"""Test infrastructure: an independent reader of scan-line OpenEXR files with PIZ (or no) compression, written from the OpenEXR file-layout
and PIZ descriptions -- NOT from fray_amd/csrc/host_exr.cpp, which it exists to check (the two share the specification and nothing else: this one
is table-driven Python over numpy arrays; neither was derived from the other).  The reference decodes its cubemap faces with the OpenEXR
library (bitmap.cpp:238-264, Imf::RgbaInputFile), which this image lacks, so an independent second implementation is the strongest pin available.

File layout (OpenEXR "File Layout" document):
  magic 0x01312f76, version word; header = attributes {name\\0 type\\0 int32 size, value} up to an empty name; then the line-offset table (one
  uint64 per chunk) and the chunks {int32 y, int32 size, data}.  A PIZ chunk holds 32 scan lines; inside a chunk the uncompressed layout is, per
  scan line, the channels in alphabetical order, each channel's pixels of that line together.
PIZ chunk: uint16 minNonZero, maxNonZero; the bitmap bytes [minNonZero .. maxNonZero] of a 65536-bit "value occurs" map; int32 length; a Huffman
  stream {int32 im, iM, tableLength, nBits, reserved; the packed code-length table; the bits}.  Decoding: canonical Huffman (6-bit lengths, zero
  runs 59..62 short / 63 long, symbol iM = "repeat the previous value n times" with an 8-bit count) -> uint16 values, channel after channel;
  inverse 2-D Haar wavelet per channel (the 14-bit form when the largest value fits in 14 bits, else the 16-bit modular form); the reverse
  look-up table made from the bitmap; then the rows are interleaved back into scan lines.  HALF pixels become floats exactly (IEEE binary16).

    python oracle/exr_piz_reader.py file.exr      # prints size, channels and an FNV-1a-64 of the float32 RGB texels
"""
import struct
import sys

import numpy as np

HUF_ENCBITS, HUF_DECBITS = 16, 14
HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1
HUF_DECSIZE, HUF_DECMASK = 1 << HUF_DECBITS, (1 << HUF_DECBITS) - 1
SHORT_ZEROCODE_RUN, LONG_ZEROCODE_RUN = 59, 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN


class ExrError(ValueError):
    pass


def read_header(b):
    if len(b) < 8 or struct.unpack_from("<I", b, 0)[0] != 20000630:
        raise ExrError("not an OpenEXR file")
    version = struct.unpack_from("<I", b, 4)[0]
    if version & 0xFF != 2 or version & 0x200 or version & 0x800 or version & 0x1000:
        raise ExrError("only single-part scan-line files of version 2 are read here")
    pos, attrs = 8, {}
    while True:
        e = b.index(b"\0", pos)
        name = b[pos:e].decode()
        pos = e + 1
        if not name:
            break
        e = b.index(b"\0", pos)
        typ = b[pos:e].decode()
        pos = e + 1
        size = struct.unpack_from("<i", b, pos)[0]
        pos += 4
        attrs[name] = (typ, b[pos:pos + size])
        pos += size
    return attrs, pos


def parse_channels(v):
    pos, out = 0, []
    while v[pos] != 0:
        e = v.index(b"\0", pos)
        name = v[pos:e].decode()
        pos = e + 1
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", v, pos)
        pos += 16
        out.append((name, ptype, xs, ys))
    return out


# ---- Huffman ------------------------------------------------------------------------------------------------------------------------------------
def unpack_lengths(data, pos, im, iM):
    """The packed table: 6 bits per symbol from im to iM, most significant bit first; zero runs are run-length coded."""
    lengths = np.zeros(HUF_ENCSIZE, np.int64)
    c = lc = 0
    sym = im
    while sym <= iM:
        while lc < 6:
            c = (c << 8) | data[pos]
            pos += 1
            lc += 8
        lc -= 6
        l = (c >> lc) & 63
        if l == LONG_ZEROCODE_RUN:
            while lc < 8:
                c = (c << 8) | data[pos]
                pos += 1
                lc += 8
            lc -= 8
            run = ((c >> lc) & 255) + SHORTEST_LONG_RUN
            if sym + run > iM + 1:
                raise ExrError("code-length table runs past its end")
            sym += run
        elif l >= SHORT_ZEROCODE_RUN:
            run = l - SHORT_ZEROCODE_RUN + 2
            if sym + run > iM + 1:
                raise ExrError("code-length table runs past its end")
            sym += run
        else:
            lengths[sym] = l
            sym += 1
    return lengths, pos


def canonical_codes(lengths):
    """Codes of one length are consecutive, in symbol order; shorter codes have numerically larger prefixes (OpenEXR's canonical table)."""
    n = np.bincount(lengths, minlength=59).astype(np.int64)
    first = [0] * 59
    c = 0
    for l in range(58, 0, -1):
        nc = (c + int(n[l])) >> 1
        first[l] = c
        c = nc
    codes = np.zeros(HUF_ENCSIZE, np.int64)
    nxt = list(first)
    for s in np.nonzero(lengths)[0]:
        l = int(lengths[s])
        codes[s] = nxt[l]
        nxt[l] += 1
    return codes


def huf_uncompress(data, n_out):
    if len(data) < 20:
        raise ExrError("Huffman block too short")
    im, iM, _table_len, nbits, _ = struct.unpack_from("<iiiii", data, 0)
    if not (0 <= im < HUF_ENCSIZE and 0 <= iM < HUF_ENCSIZE and im <= iM):
        raise ExrError("bad Huffman symbol range")
    lengths, pos = unpack_lengths(data, 20, im, iM)
    if nbits > 8 * (len(data) - pos):
        raise ExrError("Huffman bit count exceeds the data")
    codes = canonical_codes(lengths)
    # decoding table indexed by the next 14 bits: short codes resolve directly, longer ones go through a list per 14-bit prefix
    short_len = np.zeros(HUF_DECSIZE, np.int64)
    short_sym = np.zeros(HUF_DECSIZE, np.int64)
    long_lists = {}
    for s in np.nonzero(lengths)[0]:
        l, c = int(lengths[s]), int(codes[s])
        if c >> l:
            raise ExrError("Huffman code does not fit its length")
        if l <= HUF_DECBITS:
            lo = c << (HUF_DECBITS - l)
            short_len[lo:lo + (1 << (HUF_DECBITS - l))] = l
            short_sym[lo:lo + (1 << (HUF_DECBITS - l))] = s
        else:
            long_lists.setdefault(c >> (l - HUF_DECBITS), []).append((l, c, int(s)))
    short_len, short_sym = short_len.tolist(), short_sym.tolist()
    out = np.zeros(n_out, np.uint16)
    o = 0
    rlc = iM
    c = lc = 0
    end = pos + (nbits + 7) // 8
    stream = data[pos:end]
    total_bits = nbits
    consumed = 0
    i = 0
    nbytes = len(stream)

    def emit(sym):
        nonlocal o, c, lc, i, consumed
        if sym == rlc:
            while lc < 8:
                if i >= nbytes:
                    raise ExrError("Huffman stream ends inside a run length")
                c = (c << 8) | stream[i]
                i += 1
                lc += 8
            lc -= 8
            consumed += 8
            run = (c >> lc) & 255
            if o == 0 or o + run > n_out:
                raise ExrError("bad run in the Huffman stream")
            out[o:o + run] = out[o - 1]
            o += run
        else:
            if o >= n_out:
                raise ExrError("Huffman stream holds more values than the block")
            out[o] = sym
            o += 1

    while consumed < total_bits:
        while lc < HUF_DECBITS and i < nbytes:
            c = (c << 8) | stream[i]
            i += 1
            lc += 8
        c &= (1 << lc) - 1 if lc else 0
        if lc >= HUF_DECBITS:
            key = (c >> (lc - HUF_DECBITS)) & HUF_DECMASK
        else:
            key = (c << (HUF_DECBITS - lc)) & HUF_DECMASK
        l = short_len[key]
        if l:
            if l > lc:
                raise ExrError("Huffman stream ends inside a code")
            lc -= l
            consumed += l
            emit(short_sym[key])
        else:
            found = False
            for (ll, cc, ss) in long_lists.get(key, ()):
                while lc < ll and i < nbytes:
                    c = (c << 8) | stream[i]
                    i += 1
                    lc += 8
                if lc >= ll and ((c >> (lc - ll)) & ((1 << ll) - 1)) == cc:
                    lc -= ll
                    consumed += ll
                    emit(ss)
                    found = True
                    break
            if not found:
                raise ExrError("no Huffman code matches the stream")
    if o != n_out:
        raise ExrError("Huffman stream decoded to %d values, the block holds %d" % (o, n_out))
    return out


# ---- wavelet ------------------------------------------------------------------------------------------------------------------------------------
def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    a = ai.astype(np.int16)
    b = (ai - hs).astype(np.int16)
    return a.astype(np.uint16), b.astype(np.uint16)


def _wdec16(l, h):
    m = l.astype(np.int32)
    d = h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def wav2_decode(img, mx):
    """Inverse of OpenEXR's 2-D Haar transform on an (ny, nx) uint16 array, in place: from the coarsest level down, every 2x2 group of a level is
    rebuilt from {average, horizontal, vertical, diagonal} coefficient positions (px, p01, p10, p11)."""
    ny, nx = img.shape
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.zeros(0, int)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.zeros(0, int)
        if len(ys) and len(xs):
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            px, p01, p10, p11 = img[Y, X], img[Y, X + p], img[Y + p, X], img[Y + p, X + p]
            i00, i10 = dec(px, p10)
            i01, i11 = dec(p01, p11)
            a, b = dec(i00, i01)
            img[Y, X], img[Y, X + p] = a, b
            a, b = dec(i10, i11)
            img[Y + p, X], img[Y + p, X + p] = a, b
        if nx & p and len(ys):          # an odd column at this level: 1-D in y
            x = len(xs) * p2
            a, b = dec(img[ys, x], img[ys + p, x])
            img[ys, x], img[ys + p, x] = a, b
        if ny & p:                      # an odd row at this level: 1-D in x
            y = len(ys) * p2
            if len(xs):
                a, b = dec(img[y, xs], img[y, xs + p])
                img[y, xs], img[y, xs + p] = a, b
        p2 = p
        p >>= 1


def reverse_lut(bitmap):
    bits = np.unpackbits(bitmap, bitorder="little").astype(bool)
    bits[0] = True
    present = np.nonzero(bits)[0].astype(np.uint16)
    lut = np.zeros(65536, np.uint16)
    lut[:len(present)] = present
    return lut, len(present) - 1


def piz_block(data, channels, nx, ny):
    """channels: [(words per pixel)] in file order; returns the block's uint16 words in scan-line order."""
    if len(data) < 4:
        raise ExrError("PIZ block too short")
    min_nz, max_nz = struct.unpack_from("<HH", data, 0)
    pos = 4
    bitmap = np.zeros(8192, np.uint8)
    if min_nz <= max_nz:
        if max_nz >= 8192 or pos + max_nz - min_nz + 1 > len(data):
            raise ExrError("bad PIZ bitmap range")
        bitmap[min_nz:max_nz + 1] = np.frombuffer(data, np.uint8, max_nz - min_nz + 1, pos)
        pos += max_nz - min_nz + 1
    lut, mx = reverse_lut(bitmap)
    length = struct.unpack_from("<i", data, pos)[0]
    pos += 4
    if length < 0 or pos + length > len(data):
        raise ExrError("bad PIZ Huffman length")
    total = sum(w for w in channels) * nx * ny
    words = huf_uncompress(bytes(data[pos:pos + length]), total)
    # channel after channel: ny rows of nx pixels of w words
    planes, o = [], 0
    for w in channels:
        plane = words[o:o + nx * ny * w].reshape(ny, nx * w).copy()
        o += nx * ny * w
        for j in range(w):
            sub = plane[:, j::w].copy()
            wav2_decode(sub, mx)
            plane[:, j::w] = sub
        planes.append(lut[plane])
    rows = [np.concatenate([pl[y] for pl in planes]) for y in range(ny)]
    return np.concatenate(rows)


def read_exr(path):
    """-> (width, height, {channel name: float32 [h, w] array}) for HALF / FLOAT channels of a PIZ- or un-compressed scan-line file."""
    b = open(path, "rb").read()
    attrs, pos = read_header(b)
    for k in ("channels", "compression", "dataWindow", "lineOrder"):
        if k not in attrs:
            raise ExrError("header lacks " + k)
    chans = parse_channels(attrs["channels"][1])
    comp = attrs["compression"][1][0]
    if comp not in (0, 4):
        raise ExrError("compression %d is not read here (NONE and PIZ are)" % comp)
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    if w <= 0 or h <= 0 or w > 16384 or h > 16384:
        raise ExrError("bad data window")
    for (_, ptype, xs, ys) in chans:
        if xs != 1 or ys != 1 or ptype not in (1, 2):
            raise ExrError("sub-sampled or UINT channels are not read here")
    words = [1 if ptype == 1 else 2 for (_, ptype, _, _) in chans]
    lines_per_block = 32 if comp == 4 else 1
    nblocks = (h + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from("<%dQ" % nblocks, b, pos)
    out = {name: np.zeros((h, w), np.float32) for (name, _, _, _) in chans}
    row_words = sum(words) * w
    for off in offsets:
        y, size = struct.unpack_from("<ii", b, off)
        ny = min(lines_per_block, y1 - y + 1)
        if y < y0 or ny <= 0 or off + 8 + size > len(b):
            raise ExrError("bad chunk")
        raw = b[off + 8:off + 8 + size]
        if comp == 4 and size < row_words * ny * 2:
            block = piz_block(raw, words, w, ny)
        else:
            block = np.frombuffer(raw, "<u2", row_words * ny)       # stored uncompressed (also what a PIZ writer does when compression does not pay)
        for r in range(ny):
            line = block[r * row_words:(r + 1) * row_words]
            o = 0
            for (name, ptype, _, _), wd in zip(chans, words):
                seg = line[o:o + w * wd]
                o += w * wd
                if ptype == 1:
                    out[name][y - y0 + r] = seg.view(np.float16).astype(np.float32)
                else:
                    out[name][y - y0 + r] = seg.view("<f4")
    return w, h, out


def read_rgb(path):
    """float32 [h, w, 3] RGB as Bitmap::loadEXR leaves it (bitmap.cpp:238-264: Rgba -> Color(r, g, b)); missing colour channels are 0."""
    w, h, ch = read_exr(path)
    img = np.zeros((h, w, 3), np.float32)
    for i, n in enumerate("RGB"):
        if n in ch:
            img[..., i] = ch[n]
    return img


def fnv1a64(a):
    h = 14695981039346656037
    for byte in np.ascontiguousarray(a).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


if __name__ == "__main__":
    for p in sys.argv[1:]:
        im = read_rgb(p)
        print(p, im.shape, "mean %.5f" % float(im.mean()), "min %.5f max %.5f" % (float(im.min()), float(im.max())))
