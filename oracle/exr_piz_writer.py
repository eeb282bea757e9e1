"""Test infrastructure: a small writer of scan-line OpenEXR files with PIZ compression (HALF channels), the inverse of oracle/exr_piz_reader.py --
forward look-up table, forward 2-D Haar wavelet (14-bit or 16-bit modular form), canonical Huffman coding with run-length codes, the packed
code-length table -- written from the same format description.  It exists so that tests can make PIZ files of shapes and value ranges the
shipped cubemap faces do not have (odd sizes, which take the wavelet's 1-D row / column steps; ranges below 2^14, which take the 14-bit
transform; constant areas, which take the run-length code; blocks stored raw because compression did not pay) and require BOTH decoders --
fray_amd/csrc/host_exr.cpp and the independent reader -- to return exactly what was written."""
import heapq
import struct

import numpy as np

HUF_ENCSIZE = (1 << 16) + 1
SHORT_ZEROCODE_RUN, LONG_ZEROCODE_RUN = 59, 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN
LONGEST_LONG_RUN = 255 + SHORTEST_LONG_RUN


def _wenc14(a, b):
    a_s = a.astype(np.int16).astype(np.int32)
    b_s = b.astype(np.int16).astype(np.int32)
    ms = (a_s + b_s) >> 1
    ds = a_s - b_s
    return ms.astype(np.int16).astype(np.uint16), ds.astype(np.int16).astype(np.uint16)


def _wenc16(a, b):
    ao = (a.astype(np.int32) + 0x8000) & 0xFFFF
    bi = b.astype(np.int32)
    m = (ao + bi) >> 1
    d = ao - bi
    m = np.where(d < 0, (m + 0x8000) & 0xFFFF, m)
    d &= 0xFFFF
    return m.astype(np.uint16), d.astype(np.uint16)


def wav2_encode(img, mx):
    ny, nx = img.shape
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        ys = np.arange(0, ny - p2 + 1, p2)
        xs = np.arange(0, nx - p2 + 1, p2)
        Y, X = np.meshgrid(ys, xs, indexing="ij")
        i00, i01 = enc(img[Y, X], img[Y, X + p])
        i10, i11 = enc(img[Y + p, X], img[Y + p, X + p])
        a, b = enc(i00, i10)
        img[Y, X], img[Y + p, X] = a, b
        a, b = enc(i01, i11)
        img[Y, X + p], img[Y + p, X + p] = a, b
        if nx & p:
            x = len(xs) * p2
            a, b = enc(img[ys, x], img[ys + p, x])
            img[ys, x], img[ys + p, x] = a, b
        if ny & p:
            y = len(ys) * p2
            a, b = enc(img[y, xs], img[y, xs + p])
            img[y, xs], img[y, xs + p] = a, b
        p = p2
        p2 <<= 1


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, nbits, value):
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 255)
        self.acc &= (1 << self.n) - 1

    def finish(self):
        if self.n:
            self.out.append((self.acc << (8 - self.n)) & 255)
        return bytes(self.out)


def _code_lengths(freq):
    """Huffman code lengths (<= 58) for the symbols with freq > 0."""
    syms = [int(s) for s in np.nonzero(freq)[0]]
    if len(syms) == 1:
        return {syms[0]: 1}
    heap = [(int(freq[s]), i, (s,)) for i, s in enumerate(syms)]
    heapq.heapify(heap)
    length = dict.fromkeys(syms, 0)
    tie = len(heap)
    while len(heap) > 1:
        fa, _, sa = heapq.heappop(heap)
        fb, _, sb = heapq.heappop(heap)
        for s in sa + sb:
            length[s] += 1
        heapq.heappush(heap, (fa + fb, tie, sa + sb))
        tie += 1
    assert max(length.values()) <= 58
    return length


def huf_compress(words, use_runs=True):
    freq = np.bincount(words, minlength=HUF_ENCSIZE).astype(np.int64)
    im = int(np.nonzero(freq)[0][0])
    iM = int(np.nonzero(freq)[0][-1]) + 1            # the run-length pseudo symbol
    freq[iM] = 1
    lens = _code_lengths(freq)
    lengths = np.zeros(HUF_ENCSIZE, np.int64)
    for s, l in lens.items():
        lengths[s] = l
    # canonical codes, as the reader builds them
    n = np.bincount(lengths, minlength=59)
    first, c = [0] * 59, 0
    for l in range(58, 0, -1):
        nc = (c + int(n[l])) >> 1
        first[l] = c
        c = nc
    codes, nxt = {}, list(first)
    for s in np.nonzero(lengths)[0]:
        l = int(lengths[s])
        codes[int(s)] = (l, nxt[l])
        nxt[l] += 1
    # packed code-length table
    tb = _Bits()
    s = im
    while s <= iM:
        l = int(lengths[s])
        if l == 0:
            run = 1
            while s + run <= iM and run < LONGEST_LONG_RUN and lengths[s + run] == 0:
                run += 1
            if run >= 2:
                if run >= SHORTEST_LONG_RUN:
                    tb.put(6, LONG_ZEROCODE_RUN)
                    tb.put(8, run - SHORTEST_LONG_RUN)
                else:
                    tb.put(6, SHORT_ZEROCODE_RUN + run - 2)
                s += run
                continue
        tb.put(6, l)
        s += 1
    table = tb.finish()
    # the data bits
    db = _Bits()
    nbits = 0
    rl, rc = codes[iM]
    vals = words.tolist()
    i, N = 0, len(vals)
    while i < N:
        v = vals[i]
        run = 1
        while i + run < N and vals[i + run] == v and run < 256:
            run += 1
        l, c = codes[v]
        if use_runs and run > 1 and l + rl + 8 < l * run:
            db.put(l, c); db.put(rl, rc); db.put(8, run - 1)
            nbits += l + rl + 8
        else:
            for _ in range(run):
                db.put(l, c)
            nbits += l * run
        i += run
    data = db.finish()
    return struct.pack("<iiiii", im, iM, len(table), nbits, 0) + table + data


def piz_block(rows, nchan, nx, use_runs=True):
    """rows: uint16 [ny, nchan * nx] in scan-line layout (channel after channel inside a line) -> the compressed block, or None when it does not pay."""
    ny = rows.shape[0]
    planes = [rows[:, c * nx:(c + 1) * nx].copy() for c in range(nchan)]
    used = np.zeros(65536, bool)
    for pl in planes:
        used[pl.ravel()] = True
    used[0] = False                                  # zero is always present, never stored
    nz = np.nonzero(np.packbits(used, bitorder="little"))[0]
    bitmap = np.packbits(used, bitorder="little")
    present = np.nonzero(np.concatenate([[True], used[1:]]))[0]
    fwd = np.zeros(65536, np.uint16)
    fwd[present] = np.arange(len(present), dtype=np.uint16)
    mx = len(present) - 1
    coded = []
    for pl in planes:
        t = fwd[pl]
        wav2_encode(t, mx)
        coded.append(t.ravel())
    huf = huf_compress(np.concatenate(coded).astype(np.int64), use_runs)
    if len(nz):
        lo, hi = int(nz[0]), int(nz[-1])
        head = struct.pack("<HH", lo, hi) + bitmap[lo:hi + 1].tobytes()
    else:
        head = struct.pack("<HH", 8191, 0)
    out = head + struct.pack("<i", len(huf)) + huf
    return out if len(out) < rows.size * 2 else None


def write_exr(path, channels, use_runs=True):
    """channels: {name: float array [h, w]} written as HALF; PIZ, 32-line blocks, increasing y."""
    names = sorted(channels)
    h, w = channels[names[0]].shape
    half = {n: np.ascontiguousarray(channels[n], np.float32).astype(np.float16).view(np.uint16) for n in names}
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1, 0, 1, 1) for n in names) + b"\0"

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = struct.pack("<II", 20000630, 2)
    header += attr("channels", "chlist", chlist) + attr("compression", "compression", b"\x04") + attr("dataWindow", "box2i", box)
    header += attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\x00") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    nblocks = (h + 31) // 32
    chunks = []
    for bi in range(nblocks):
        y = bi * 32
        ny = min(32, h - y)
        rows = np.concatenate([half[n][y:y + ny] for n in names], axis=1)
        blk = piz_block(rows, len(names), w, use_runs)
        data = blk if blk is not None else rows.astype("<u2").tobytes()
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    off = len(header) + 8 * nblocks
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(header + table + b"".join(chunks))
    return {n: half[n].view(np.float16).astype(np.float32) for n in names}
