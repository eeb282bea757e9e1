// OpenEXR reader for the cubemap faces of a CubemapEnvironment (reference: Bitmap::loadEXR,
// src/bitmap.cpp:238-264, which goes through the OpenEXR library's RgbaInputFile).  OpenEXR is a
// third-party dependency the reference does not vendor; this is a from-scratch decoder of the
// published file format, restricted to what the shipped faces use (and failing cleanly otherwise):
// single-part scan-line files, HALF / FLOAT channels named R, G, B (A ignored), no subsampling,
// compression NONE or PIZ (wavelet + Huffman, 32-line blocks).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "host_scene.h"

namespace frayhost {

namespace {

struct ExrError : std::runtime_error {
    explicit ExrError(const char* m) : std::runtime_error(m) {}
};

struct Reader {
    const unsigned char* p;
    const unsigned char* e;
    void need(size_t n) const { if ((size_t)(e - p) < n) throw ExrError("truncated file"); }
    uint8_t u8() { need(1); return *p++; }
    uint16_t u16() { need(2); uint16_t v = (uint16_t)(p[0] | (p[1] << 8)); p += 2; return v; }
    int32_t i32() { need(4); uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); p += 4; return (int32_t)v; }
    uint64_t u64() { uint64_t lo = (uint32_t)i32(); uint64_t hi = (uint32_t)i32(); return lo | (hi << 32); }
    std::string str() { std::string s; for (;;) { char c = (char)u8(); if (!c) break; s += c; if (s.size() > 255) throw ExrError("bad string"); } return s; }
};

float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h >> 15) << 31, ex = (h >> 10) & 31, man = h & 1023;
    uint32_t bits;
    if (ex == 0) {
        if (man == 0) bits = sign;
        else {   // subnormal half -> normal float
            int sh = 0;
            while (!(man & 1024)) { man <<= 1; sh++; }
            man &= 1023;
            bits = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (man << 13);
        }
    } else if (ex == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((ex + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

// ---- PIZ: Huffman coding of 16-bit symbols (format as published in OpenEXR's ImfHuf) -------------
const int HUF_ENCBITS = 16, HUF_DECBITS = 14;
const int HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1, HUF_DECSIZE = 1 << HUF_DECBITS, HUF_DECMASK = HUF_DECSIZE - 1;

struct Dec { int len = 0; int lit = 0; std::vector<int> longs; };

struct BitIn {
    const unsigned char* p;
    const unsigned char* e;
    uint64_t c = 0;
    int lc = 0;
    void byte() { if (p >= e) throw ExrError("huffman: out of data"); c = (c << 8) | *p++; lc += 8; }
    uint64_t bits(int n) { while (lc < n) byte(); lc -= n; return (c >> lc) & ((1ull << n) - 1); }
};

void canonical_codes(std::vector<uint64_t>& h)
{
    uint64_t n[59] = {0};
    for (int i = 0; i < HUF_ENCSIZE; i++) { if (h[i] > 58) throw ExrError("huffman: bad code length"); n[h[i]]++; }
    uint64_t c = 0;
    for (int i = 58; i > 0; --i) { uint64_t nc = (c + n[i]) >> 1; n[i] = c; c = nc; }
    for (int i = 0; i < HUF_ENCSIZE; i++) { int l = (int)h[i]; if (l > 0) h[i] = (uint64_t)l | (n[l]++ << 6); }
}

void huf_uncompress(const unsigned char* data, int nData, std::vector<uint16_t>& out)
{
    if (nData == 0) { if (!out.empty()) throw ExrError("huffman: no data"); return; }
    if (nData < 20) throw ExrError("huffman: short header");
    auto rd = [&](int o) { return (int)((uint32_t)data[o] | ((uint32_t)data[o + 1] << 8) | ((uint32_t)data[o + 2] << 16) | ((uint32_t)data[o + 3] << 24)); };
    int im = rd(0), iM = rd(4), nBits = rd(12);
    if (im < 0 || im >= HUF_ENCSIZE || iM < 0 || iM >= HUF_ENCSIZE) throw ExrError("huffman: bad table size");
    // ---- packed code-length table: 6-bit lengths with run-length codes for zeros
    std::vector<uint64_t> hcode(HUF_ENCSIZE, 0);
    BitIn tb{data + 20, data + nData};
    for (int s = im; s <= iM; s++) {
        uint64_t l = hcode[s] = tb.bits(6);
        int zerun = 0;
        if (l == 63) zerun = (int)tb.bits(8) + 6;
        else if (l >= 59) zerun = (int)l - 59 + 2;
        if (zerun) {
            if (s + zerun > iM + 1) throw ExrError("huffman: table too long");
            while (zerun--) hcode[s++] = 0;
            s--;
        }
    }
    const unsigned char* in = tb.p;
    if (nBits < 0 || (int64_t)nBits > 8 * (int64_t)(data + nData - in)) throw ExrError("huffman: bad bit count");
    canonical_codes(hcode);
    // ---- decoding table: 14-bit primary index, overflow lists for longer codes
    std::vector<Dec> dec(HUF_DECSIZE);
    for (int s = im; s <= iM; s++) {
        uint64_t c = hcode[s] >> 6;
        int l = (int)(hcode[s] & 63);
        if (c >> l) throw ExrError("huffman: bad table entry");
        if (l > HUF_DECBITS) {
            Dec& d = dec[c >> (l - HUF_DECBITS)];
            if (d.len) throw ExrError("huffman: bad table entry");
            d.longs.push_back(s);
        } else if (l) {
            size_t base = (size_t)(c << (HUF_DECBITS - l));
            for (size_t i = 0; i < ((size_t)1 << (HUF_DECBITS - l)); i++) {
                Dec& d = dec[base + i];
                if (d.len || !d.longs.empty()) throw ExrError("huffman: bad table entry");
                d.len = l; d.lit = s;
            }
        }
    }
    // ---- decode; symbol iM is the run-length escape: repeat the previous output <next byte> times
    BitIn b{in, in + (nBits + 7) / 8};
    size_t o = 0;
    const size_t no = out.size();
    auto emit = [&](int sym) {
        if (sym == iM) {
            if (b.lc < 8) b.byte();
            b.lc -= 8;
            unsigned cs = (unsigned)((b.c >> b.lc) & 255);
            if (o + cs > no) throw ExrError("huffman: too much data");
            if (o == 0) throw ExrError("huffman: run without a value");
            uint16_t s = out[o - 1];
            while (cs-- > 0) out[o++] = s;
        } else {
            if (o >= no) throw ExrError("huffman: too much data");
            out[o++] = (uint16_t)sym;
        }
    };
    while (b.p < b.e) {
        b.byte();
        while (b.lc >= HUF_DECBITS) {
            const Dec& d = dec[(b.c >> (b.lc - HUF_DECBITS)) & HUF_DECMASK];
            if (d.len) {
                b.lc -= d.len;
                emit(d.lit);
            } else {
                if (d.longs.empty()) throw ExrError("huffman: invalid code");
                size_t j = 0;
                for (; j < d.longs.size(); j++) {
                    int l = (int)(hcode[d.longs[j]] & 63);
                    while (b.lc < l && b.p < b.e) b.byte();
                    if (b.lc >= l && (hcode[d.longs[j]] >> 6) == ((b.c >> (b.lc - l)) & ((1ull << l) - 1))) {
                        b.lc -= l;
                        emit(d.longs[j]);
                        break;
                    }
                }
                if (j == d.longs.size()) throw ExrError("huffman: invalid code");
            }
        }
    }
    int i = (8 - nBits) & 7;
    b.c >>= i;
    b.lc -= i;
    while (b.lc > 0) {
        const Dec& d = dec[(b.c << (HUF_DECBITS - b.lc)) & HUF_DECMASK];
        if (!d.len) throw ExrError("huffman: invalid code");
        b.lc -= d.len;
        emit(d.lit);
    }
    if (o != no) throw ExrError("huffman: not enough data");
}

// ---- PIZ: inverse 2-D Haar-like wavelet on 16-bit data (format as published in OpenEXR's ImfWav) -
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
{
    int16_t ls = (int16_t)l, hs = (int16_t)h;
    int hi = hs;
    int ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
{
    int m = l, d = h;
    int bb = (m - (d >> 1)) & 0xffff;
    int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}
void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx)
{
    const bool w14 = mx < (1 << 14);
    int n = nx > ny ? ny : nx;
    int p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* ey = in + oy * (ny - p2);
        const int oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                uint16_t* p10 = px + oy1;
                uint16_t* p11 = p10 + ox1;
                if (w14) {
                    wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11);
                    wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11);
                } else {
                    wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11);
                    wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11);
                }
            }
            if (nx & p) {
                uint16_t* p10 = px + oy1;
                if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t* px = py;
            uint16_t* ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

struct Channel { std::string name; int type; int size; };   // size in 16-bit words per sample

// One PIZ block -> the block's scan-line-interleaved raw layout (line, channel, x), as 16-bit words.
void piz_uncompress(const unsigned char* data, int nData, const std::vector<Channel>& ch, int nx, int ny, std::vector<uint16_t>& raw)
{
    Reader r{data, data + nData};
    uint16_t minNonZero = r.u16(), maxNonZero = r.u16();
    const int BITMAP_SIZE = 8192;
    if (maxNonZero >= BITMAP_SIZE) throw ExrError("piz: bad bitmap range");
    std::vector<unsigned char> bitmap(BITMAP_SIZE, 0);
    if (minNonZero <= maxNonZero) {
        r.need((size_t)(maxNonZero - minNonZero + 1));
        memcpy(&bitmap[minNonZero], r.p, (size_t)(maxNonZero - minNonZero + 1));
        r.p += maxNonZero - minNonZero + 1;
    }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; i++)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t maxValue = (uint16_t)(k - 1);
    int length = r.i32();
    if (length < 0 || (size_t)length > (size_t)(r.e - r.p)) throw ExrError("piz: bad huffman length");
    size_t total = 0;
    for (auto& c : ch) total += (size_t)nx * ny * c.size;
    std::vector<uint16_t> tmp(total);
    huf_uncompress(r.p, length, tmp);
    size_t off = 0;
    std::vector<size_t> start;
    for (auto& c : ch) {
        start.push_back(off);
        for (int j = 0; j < c.size; j++) wav2_decode(&tmp[off + j], nx, c.size, ny, nx * c.size, maxValue);
        off += (size_t)nx * ny * c.size;
    }
    for (auto& v : tmp) v = lut[v];
    raw.resize(total);
    size_t o = 0;
    std::vector<size_t> cur = start;
    for (int y = 0; y < ny; y++)
        for (size_t c = 0; c < ch.size(); c++) {
            size_t n = (size_t)nx * ch[c].size;
            memcpy(&raw[o], &tmp[cur[c]], n * 2);
            o += n;
            cur[c] += n;
        }
}

void decode(const std::vector<unsigned char>& file, Image& img)
{
    Reader r{file.data(), file.data() + file.size()};
    if (r.i32() != 20000630) throw ExrError("not an OpenEXR file");
    int version = r.i32();
    if ((version & 0xff) != 2 || (version & 0x1a00)) throw ExrError("unsupported EXR flavour (tiled / multi-part / deep)");
    std::vector<Channel> ch;
    int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1, lineOrder = 0;
    for (;;) {
        std::string name = r.str();
        if (name.empty()) break;
        std::string type = r.str();
        int size = r.i32();
        if (size < 0) throw ExrError("bad attribute size");
        r.need((size_t)size);
        Reader a{r.p, r.p + size};
        r.p += size;
        if (name == "channels") {
            for (;;) {
                std::string cn = a.str();
                if (cn.empty()) break;
                int pt = a.i32();
                a.u8(); a.u8(); a.u8(); a.u8();
                int xs = a.i32(), ys = a.i32();
                if (xs != 1 || ys != 1) throw ExrError("subsampled channels are not supported");
                if (pt != 1 && pt != 2) throw ExrError("only HALF and FLOAT channels are supported");
                ch.push_back(Channel{cn, pt, pt == 1 ? 1 : 2});
            }
        } else if (name == "compression") compression = a.u8();
        else if (name == "dataWindow") { x0 = a.i32(); y0 = a.i32(); x1 = a.i32(); y1 = a.i32(); }
        else if (name == "lineOrder") lineOrder = a.u8();
    }
    if (ch.empty() || x1 < x0 || y1 < y0) throw ExrError("missing channels / dataWindow");
    if (compression != 0 && compression != 4) throw ExrError("only NONE and PIZ compression are supported");
    if (lineOrder > 1) throw ExrError("unsupported line order");
    // in 64 bits: x1 - x0 + 1 of two arbitrary int32 corners does not fit an int
    const int64_t W64 = (int64_t)x1 - (int64_t)x0 + 1, H64 = (int64_t)y1 - (int64_t)y0 + 1;
    if (W64 <= 0 || H64 <= 0 || W64 > 8192 || H64 > 8192) throw ExrError("image too large");
    const int W = (int)W64, H = (int)H64;
    const int lines = compression == 4 ? 32 : 1;
    const int nChunks = (H + lines - 1) / lines;
    std::vector<uint64_t> offsets(nChunks);
    for (auto& o : offsets) o = r.u64();
    size_t wordsPerLine = 0;
    for (auto& c : ch) wordsPerLine += (size_t)W * c.size;
    img.w = W; img.h = H;
    img.rgb.assign((size_t)W * H * 3, 0.0f);
    for (int k = 0; k < nChunks; k++) {
        if (offsets[k] > file.size() || file.size() - offsets[k] < 8) throw ExrError("bad chunk offset");   // no `offset + 8`: that wraps for offsets near 2^64
        Reader c{file.data() + offsets[k], file.data() + file.size()};
        int y = c.i32(), size = c.i32();
        if (size < 0) throw ExrError("bad chunk size");
        c.need((size_t)size);
        const int64_t yb64 = (int64_t)y - (int64_t)y0;
        if (yb64 < 0 || yb64 >= H || yb64 % lines) throw ExrError("bad chunk row");
        const int yb = (int)yb64;
        int ny = std::min(lines, H - yb);
        std::vector<uint16_t> raw;
        if (compression == 4 && (size_t)size < wordsPerLine * ny * 2) {
            piz_uncompress(c.p, size, ch, W, ny, raw);
        } else {   // stored uncompressed (little-endian words)
            if ((size_t)size != wordsPerLine * ny * 2) throw ExrError("bad raw chunk size");
            raw.resize(wordsPerLine * ny);
            for (size_t i = 0; i < raw.size(); i++) raw[i] = (uint16_t)(c.p[2 * i] | (c.p[2 * i + 1] << 8));
        }
        size_t o = 0;
        for (int line = 0; line < ny; line++)
            for (auto& cc : ch) {
                int comp = cc.name == "R" ? 0 : cc.name == "G" ? 1 : cc.name == "B" ? 2 : -1;
                for (int x = 0; x < W; x++) {
                    float v;
                    if (cc.size == 1) v = half_to_float(raw[o]);
                    else { uint32_t bits = (uint32_t)raw[o] | ((uint32_t)raw[o + 1] << 16); memcpy(&v, &bits, 4); }
                    o += cc.size;
                    if (comp >= 0) img.rgb[((size_t)(yb + line) * W + x) * 3 + comp] = v;
                }
            }
    }
}

}  // namespace

bool load_exr(const char* path, Image& img, std::string& err)
{
    FILE* f = fopen(path, "rb");
    if (!f) { err = "cannot open file"; return false; }
    std::vector<unsigned char> file;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    try {
        decode(file, img);
        return true;
    } catch (const std::exception& e) {
        err = std::string("EXR: ") + e.what();
        img = Image();
        return false;
    }
}

}  // namespace frayhost
