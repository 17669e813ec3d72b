// JPEG bitmap textures (ITU T.81): baseline, extended-sequential and progressive Huffman coding, 8-bit samples, 1, 3 or 4
// components, restart intervals, interleaved and single-component scans, chroma subsampled 1x or 2x in either direction (other
// integer ratios by sample repetition).  Own decoder -- the reference reads textures with its vendored stb_image
// (R/CRTTextureBitmap.cpp:10), which is third party and is not used here -- but a JPEG has no single right answer: the pixels
// depend on the decoder's fixed-point inverse DCT, its chroma filter and its colour matrix.  So this file takes stb_image's
// NUMERICAL conventions (studied in R/stb_image/stb_image.h:2439-2540, 3478-3560, 3676-3700, 3878-4050) and restates them:
//   * inverse DCT: the Loeffler / Ligtenberg / Moschytz factorisation with 12-bit constants; columns keep 2 extra bits (round
//     at bit 10), rows round at bit 17 after adding the +128 level shift; 32-bit wrap-around arithmetic;
//   * a coefficient is the product of the decoded value and its quantiser, truncated to 16 bits;
//   * chroma up-sampling: 3:1 triangle filter horizontally and / or vertically (2x), rounding constants 2 of 4 and 8 of 16;
//   * YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits first, the Cb term of green truncated to its
//     upper 16 bits; Adobe CMYK / YCCK through the (x * y + 128) * 257 >> 16 product; RGB-tagged files pass through;
//   * output: 3 bytes per texel for colour files, 1 for grey files (what stbi_load(..., 0) returns).
// Known answers from the reference's own CRTTextureBitmap over seeded files: tests/golden/bitmap_known_answers.json.
#include "image_decode.h"

#include <cstdint>
#include <cstring>
#include <stdexcept>

namespace crt {
namespace {

[[noreturn]] void bad(const std::string& what, const char* why) { throw std::runtime_error(what + ": JPEG: " + why); }

// position in the block (row-major) of the k-th coefficient in coding order; 15 spare entries catch runs past the end
const unsigned char kNatural[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };

struct CodeTable {
    bool defined = false;
    unsigned char symbol[256];
    int count[17];           // codes of each length
    int firstCode[18];       // canonical code of the first symbol of a length
    int firstIndex[17];      // its index in symbol[]
    uint16_t quick[512];     // by the next 9 bits: (length << 8) | symbol, 0 when the code is longer

    void build(const std::string& what)
    {
        int code = 0, index = 0;
        std::memset(quick, 0, sizeof(quick));
        for (int len = 1; len <= 16; len++) {
            firstCode[len] = code;
            firstIndex[len] = index;
            if (count[len] && code + count[len] - 1 >= (1 << len)) bad(what, "code lengths of a Huffman table do not fit");
            if (len <= 9)
                for (int i = 0; i < count[len]; i++) {
                    const int c = (code + i) << (9 - len);
                    for (int f = 0; f < (1 << (9 - len)); f++) quick[c + f] = static_cast<uint16_t>((len << 8) | symbol[index + i]);
                }
            code = (code + count[len]) << 1;
            index += count[len];
        }
        firstCode[17] = code;
        defined = true;
    }
};

// the entropy-coded segment as a bit stream: 0xFF 0x00 is a data byte 0xFF, any other 0xFF xx ends the data (zeros follow)
struct Bits {
    const unsigned char* p;
    const unsigned char* end;
    uint32_t acc = 0;
    int have = 0;
    int marker = 0; // the marker that ended the data, 0 while none

    void fill()
    {
        while (have <= 24) {
            unsigned b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    unsigned c = p < end ? *p++ : 0xD9u;
                    while (c == 0xFF) c = p < end ? *p++ : 0xD9u; // fill bytes
                    if (c != 0) {
                        marker = static_cast<int>(c);
                        b = 0;
                    }
                }
            }
            acc |= b << (24 - have);
            have += 8;
        }
    }
    int peek9()
    {
        if (have < 16) fill();
        return static_cast<int>(acc >> 23);
    }
    void drop(int n) { acc <<= n; have -= n; }
    int get(int n) // n = 0..16 bits, most significant first
    {
        if (n == 0) return 0;
        if (have < n) fill();
        const int v = static_cast<int>(acc >> (32 - n));
        drop(n);
        return v;
    }
    void reset() { acc = 0; have = 0; marker = 0; }
};

int decodeSymbol(Bits& b, const CodeTable& t, const std::string& what)
{
    const int q = t.quick[b.peek9()];
    if (q) {
        b.drop(q >> 8);
        return q & 255;
    }
    int code = b.get(9);
    for (int len = 10; len <= 16; len++) {
        code = (code << 1) | b.get(1);
        if (t.count[len] && code >= t.firstCode[len] && code < t.firstCode[len] + t.count[len]) return t.symbol[t.firstIndex[len] + code - t.firstCode[len]];
    }
    bad(what, "bit pattern that is no Huffman code");
}

// the `n` bits that follow a magnitude category: values of the lower half of the range are negative
int receiveExtend(Bits& b, int n)
{
    if (n == 0) return 0;
    const int v = b.get(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                // tables of the current scan
    int x = 0, y = 0;                  // samples
    int blocksW = 0, blocksH = 0;      // padded to whole MCUs
    int pred = 0;
    std::vector<int16_t> coef;         // blocksH x blocksW x 64, dequantised for sequential files, raw for progressive ones
    std::vector<unsigned char> plane;  // (blocksH * 8) x (blocksW * 8)
};

inline int32_t wrapMul(int32_t a, int32_t b) { return static_cast<int32_t>(static_cast<uint32_t>(a) * static_cast<uint32_t>(b)); }
inline int32_t wrapAdd(int32_t a, int32_t b) { return static_cast<int32_t>(static_cast<uint32_t>(a) + static_cast<uint32_t>(b)); }
inline int32_t wrapSub(int32_t a, int32_t b) { return static_cast<int32_t>(static_cast<uint32_t>(a) - static_cast<uint32_t>(b)); }
inline int fix12(float x) { return static_cast<int>(x * 4096 + 0.5); }
inline unsigned char clamp8(int v) { return static_cast<unsigned char>(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// one 8-point inverse DCT of the LLM factorisation on values scaled by 4096: even part into e0..e3, odd part into o0..o3;
// output k = e[k] + o[3-k], output 7-k = e[k] - o[3-k]
struct Llm {
    int32_t e0, e1, e2, e3, o0, o1, o2, o3;
    Llm(int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t s4, int32_t s5, int32_t s6, int32_t s7)
    {
        const int32_t z = wrapMul(wrapAdd(s2, s6), fix12(0.5411961f));
        const int32_t lo = wrapAdd(z, wrapMul(s6, fix12(-1.847759065f)));
        const int32_t hi = wrapAdd(z, wrapMul(s2, fix12(0.765366865f)));
        const int32_t sum = wrapMul(wrapAdd(s0, s4), 4096), dif = wrapMul(wrapSub(s0, s4), 4096);
        e0 = wrapAdd(sum, hi);
        e3 = wrapSub(sum, hi);
        e1 = wrapAdd(dif, lo);
        e2 = wrapSub(dif, lo);
        const int32_t a = wrapAdd(s7, s3), b = wrapAdd(s5, s1), c = wrapAdd(s7, s1), d = wrapAdd(s5, s3);
        const int32_t all = wrapMul(wrapAdd(a, b), fix12(1.175875602f));
        const int32_t pc = wrapAdd(all, wrapMul(c, fix12(-0.899976223f)));
        const int32_t pd = wrapAdd(all, wrapMul(d, fix12(-2.562915447f)));
        const int32_t pa = wrapMul(a, fix12(-1.961570560f));
        const int32_t pb = wrapMul(b, fix12(-0.390180644f));
        o3 = wrapAdd(wrapMul(s1, fix12(1.501321110f)), wrapAdd(pc, pb));
        o2 = wrapAdd(wrapMul(s3, fix12(3.072711026f)), wrapAdd(pd, pa));
        o1 = wrapAdd(wrapMul(s5, fix12(2.053119869f)), wrapAdd(pd, pb));
        o0 = wrapAdd(wrapMul(s7, fix12(0.298631336f)), wrapAdd(pc, pa));
    }
};

void inverseDct(const int16_t* c, unsigned char* out, int stride)
{
    int32_t mid[64];
    for (int x = 0; x < 8; x++) { // columns: 12 bits of constants come off, 2 bits stay
        const Llm t(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
        const int32_t e[4] = { wrapAdd(t.e0, 512), wrapAdd(t.e1, 512), wrapAdd(t.e2, 512), wrapAdd(t.e3, 512) };
        const int32_t o[4] = { t.o3, t.o2, t.o1, t.o0 };
        for (int k = 0; k < 4; k++) {
            mid[8 * k + x] = wrapAdd(e[k], o[k]) >> 10;
            mid[8 * (7 - k) + x] = wrapSub(e[k], o[k]) >> 10;
        }
    }
    const int32_t bias = 65536 + (128 << 17); // half of 2^17 to round, and the level shift
    for (int y = 0; y < 8; y++) {
        const int32_t* m = mid + 8 * y;
        const Llm t(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        const int32_t e[4] = { wrapAdd(t.e0, bias), wrapAdd(t.e1, bias), wrapAdd(t.e2, bias), wrapAdd(t.e3, bias) };
        const int32_t o[4] = { t.o3, t.o2, t.o1, t.o0 };
        unsigned char* row = out + static_cast<size_t>(y) * stride;
        for (int k = 0; k < 4; k++) {
            row[k] = clamp8(wrapAdd(e[k], o[k]) >> 17);
            row[7 - k] = clamp8(wrapSub(e[k], o[k]) >> 17);
        }
    }
}

struct Decoder {
    const std::vector<unsigned char>& f;
    const std::string& what;
    size_t at = 2;
    int width = 0, height = 0, nComp = 0, hMax = 1, vMax = 1, mcusX = 0, mcusY = 0;
    bool progressive = false, haveFrame = false, jfif = false;
    int adobeTransform = -1, rgbTagged = 0, restartInterval = 0;
    uint16_t quant[4][64] = {};
    CodeTable dc[4], ac[4];
    Component comp[4];
    // current scan
    int scanN = 0, order[4] = { 0, 0, 0, 0 }, ss = 0, se = 63, ah = 0, al = 0, eobRun = 0;

    Decoder(const std::vector<unsigned char>& file, const std::string& name) : f(file), what(name) {}

    unsigned u8()
    {
        if (at >= f.size()) bad(what, "file ends inside a marker segment");
        return f[at++];
    }
    unsigned u16()
    {
        const unsigned hi = u8();
        return (hi << 8) | u8();
    }

    void tables(unsigned marker)
    {
        if (marker == 0xDD) {
            if (u16() != 4) bad(what, "restart-interval segment of the wrong length");
            restartInterval = static_cast<int>(u16());
            return;
        }
        int len = static_cast<int>(u16()) - 2;
        if (len < 0) bad(what, "marker segment shorter than its own length field");
        if (marker == 0xDB) {
            while (len > 0) {
                const unsigned q = u8();
                const unsigned wide = q >> 4, t = q & 15;
                if (wide > 1) bad(what, "quantisation table of unknown precision");
                if (t > 3) bad(what, "quantisation table number above 3");
                for (int k = 0; k < 64; k++) quant[t][kNatural[k]] = static_cast<uint16_t>(wide ? u16() : u8());
                len -= wide ? 129 : 65;
            }
            if (len != 0) bad(what, "quantisation segment of the wrong length");
        } else if (marker == 0xC4) {
            while (len > 0) {
                const unsigned q = u8();
                if ((q >> 4) > 1 || (q & 15) > 3) bad(what, "Huffman table of unknown class or number");
                CodeTable& t = (q >> 4) ? ac[q & 15] : dc[q & 15];
                int n = 0;
                t.count[0] = 0;
                for (int l = 1; l <= 16; l++) n += (t.count[l] = static_cast<int>(u8()));
                if (n > 256) bad(what, "Huffman table with more than 256 symbols");
                for (int i = 0; i < n; i++) t.symbol[i] = static_cast<unsigned char>(u8());
                t.build(what);
                len -= 17 + n;
            }
            if (len != 0) bad(what, "Huffman segment of the wrong length");
        } else if ((marker >= 0xE0 && marker <= 0xEF) || marker == 0xFE) {
            if (marker == 0xE0 && len >= 5) {
                static const unsigned char tag[5] = { 'J', 'F', 'I', 'F', 0 };
                bool same = true;
                for (int i = 0; i < 5; i++) same &= u8() == tag[i];
                jfif |= same;
                len -= 5;
            } else if (marker == 0xEE && len >= 12) {
                static const unsigned char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool same = true;
                for (int i = 0; i < 6; i++) same &= u8() == tag[i];
                len -= 6;
                if (same) {
                    u8(); u16(); u16();
                    adobeTransform = static_cast<int>(u8());
                    len -= 6;
                }
            }
            if (at + static_cast<size_t>(len) > f.size()) bad(what, "file ends inside an application segment");
            at += static_cast<size_t>(len);
        } else {
            bad(what, "marker this decoder does not know (arithmetic coding, lossless and hierarchical files are not supported)");
        }
    }

    void frame(unsigned marker)
    {
        if (haveFrame) bad(what, "second frame header");
        progressive = marker == 0xC2;
        const unsigned len = u16();
        if (len < 11) bad(what, "frame header too short");
        if (u8() != 8) bad(what, "only 8-bit samples are supported");
        height = static_cast<int>(u16());
        width = static_cast<int>(u16());
        if (height == 0) bad(what, "frame without a height (DNL) is not supported");
        if (width == 0) bad(what, "zero width");
        nComp = static_cast<int>(u8());
        if (nComp != 1 && nComp != 3 && nComp != 4) bad(what, "component count is not 1, 3 or 4");
        if (len != 8u + 3u * static_cast<unsigned>(nComp)) bad(what, "frame header of the wrong length");
        if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > (1ull << 28)) bad(what, "more than 2^28 texels");
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            c.id = static_cast<int>(u8());
            if (nComp == 3 && c.id == "RGB"[i]) rgbTagged++;
            const unsigned q = u8();
            c.h = static_cast<int>(q >> 4);
            c.v = static_cast<int>(q & 15);
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) bad(what, "sampling factor outside 1..4");
            c.tq = static_cast<int>(u8());
            if (c.tq > 3) bad(what, "quantisation table number above 3");
            hMax = c.h > hMax ? c.h : hMax;
            vMax = c.v > vMax ? c.v : vMax;
        }
        for (int i = 0; i < nComp; i++)
            if (hMax % comp[i].h != 0 || vMax % comp[i].v != 0) bad(what, "sampling factors that are not whole ratios");
        mcusX = (width + 8 * hMax - 1) / (8 * hMax);
        mcusY = (height + 8 * vMax - 1) / (8 * vMax);
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            c.x = (width * c.h + hMax - 1) / hMax;
            c.y = (height * c.v + vMax - 1) / vMax;
            c.blocksW = mcusX * c.h;
            c.blocksH = mcusY * c.v;
            c.coef.assign(static_cast<size_t>(c.blocksW) * c.blocksH * 64, 0);
        }
        haveFrame = true;
    }

    // ---- one block of each kind of scan
    void sequentialBlock(Bits& b, Component& c, int16_t* d)
    {
        const CodeTable& hd = dc[c.td];
        const CodeTable& ha = ac[c.ta];
        const uint16_t* q = quant[c.tq];
        const int t = decodeSymbol(b, hd, what);
        if (t > 15) bad(what, "magnitude category above 15");
        const int diff = receiveExtend(b, t);
        const int64_t sum = static_cast<int64_t>(c.pred) + diff;
        if (sum < INT32_MIN || sum > INT32_MAX) bad(what, "DC prediction leaves the 32-bit range");
        c.pred = static_cast<int>(sum);
        const int64_t first = sum * q[0];
        if (first < -32768 || first > 32767) bad(what, "DC coefficient leaves the 16-bit range");
        std::memset(d, 0, 64 * sizeof(int16_t));
        d[0] = static_cast<int16_t>(first);
        int k = 1;
        do {
            const int rs = decodeSymbol(b, ha, what);
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                const int z = kNatural[k++];
                d[z] = static_cast<int16_t>(static_cast<uint32_t>(receiveExtend(b, s)) * q[z]);
            }
        } while (k < 64);
    }

    void progressiveDc(Bits& b, Component& c, int16_t* d)
    {
        if (ah == 0) {
            const int t = decodeSymbol(b, dc[c.td], what);
            if (t > 15) bad(what, "magnitude category above 15");
            const int diff = receiveExtend(b, t);
            const int64_t sum = static_cast<int64_t>(c.pred) + diff;
            if (sum < INT32_MIN || sum > INT32_MAX) bad(what, "DC prediction leaves the 32-bit range");
            c.pred = static_cast<int>(sum);
            const int64_t v = sum * (1 << al);
            if (v < -32768 || v > 32767) bad(what, "DC coefficient leaves the 16-bit range");
            std::memset(d, 0, 64 * sizeof(int16_t));
            d[0] = static_cast<int16_t>(v);
        } else if (b.get(1)) {
            d[0] = static_cast<int16_t>(d[0] + static_cast<int16_t>(1 << al));
        }
    }

    void refineOne(Bits& b, int16_t& v, int16_t bit)
    {
        if (b.get(1) && (v & bit) == 0) v = static_cast<int16_t>(v > 0 ? v + bit : v - bit);
    }

    void progressiveAc(Bits& b, Component& c, int16_t* d)
    {
        const CodeTable& ha = ac[c.ta];
        if (ah == 0) { // first pass over this band
            if (eobRun) {
                eobRun--;
                return;
            }
            int k = ss;
            do {
                const int rs = decodeSymbol(b, ha, what);
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eobRun = (1 << r) - 1;
                        if (r) eobRun += b.get(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int z = kNatural[k++];
                    d[z] = static_cast<int16_t>(static_cast<uint32_t>(receiveExtend(b, s)) * static_cast<uint32_t>(1 << al));
                }
            } while (k <= se);
            return;
        }
        // refinement: one more bit of the coefficients that are already non-zero, and new coefficients of magnitude 1
        const int16_t bit = static_cast<int16_t>(1 << al);
        if (eobRun) {
            eobRun--;
            for (int k = ss; k <= se; k++) {
                int16_t& v = d[kNatural[k]];
                if (v != 0) refineOne(b, v, bit);
            }
            return;
        }
        int k = ss;
        do {
            const int rs = decodeSymbol(b, ha, what);
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eobRun = (1 << r) - 1;
                    if (r) eobRun += b.get(r);
                    r = 64; // to the end of the band
                }
            } else {
                if (s != 1) bad(what, "refinement scan with a new coefficient that is not 1 or -1");
                s = b.get(1) ? bit : -bit;
            }
            while (k <= se) {
                int16_t& v = d[kNatural[k++]];
                if (v != 0) {
                    refineOne(b, v, bit);
                } else {
                    if (r == 0) {
                        v = static_cast<int16_t>(s);
                        break;
                    }
                    r--;
                }
            }
        } while (k <= se);
    }

    void block(Bits& b, Component& c, int bx, int by)
    {
        int16_t* d = c.coef.data() + (static_cast<size_t>(by) * c.blocksW + bx) * 64;
        if (!progressive) sequentialBlock(b, c, d);
        else if (ss == 0) progressiveDc(b, c, d);
        else progressiveAc(b, c, d);
    }

    // true: the interval ended at a restart marker and decoding goes on; false: the scan's data ends here
    bool restart(Bits& b)
    {
        if (b.have < 24) b.fill();
        if (b.marker < 0xD0 || b.marker > 0xD7) return false;
        b.reset();
        for (int i = 0; i < nComp; i++) comp[i].pred = 0;
        eobRun = 0;
        return true;
    }

    void scan()
    {
        if (!haveFrame) bad(what, "scan before the frame header");
        const unsigned len = u16();
        scanN = static_cast<int>(u8());
        if (scanN < 1 || scanN > 4 || scanN > nComp) bad(what, "scan with a component count outside the frame's");
        if (len != 6u + 2u * static_cast<unsigned>(scanN)) bad(what, "scan header of the wrong length");
        for (int i = 0; i < scanN; i++) {
            const int id = static_cast<int>(u8());
            const unsigned q = u8();
            int which = 0;
            while (which < nComp && comp[which].id != id) which++;
            if (which == nComp) bad(what, "scan names a component the frame does not have");
            comp[which].td = static_cast<int>(q >> 4);
            comp[which].ta = static_cast<int>(q & 15);
            if (comp[which].td > 3 || comp[which].ta > 3) bad(what, "Huffman table number above 3");
            order[i] = which;
        }
        ss = static_cast<int>(u8());
        se = static_cast<int>(u8());
        const unsigned a = u8();
        ah = static_cast<int>(a >> 4);
        al = static_cast<int>(a & 15);
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) bad(what, "progressive scan with impossible band or bit positions");
            if (ss != 0 && scanN != 1) bad(what, "interleaved AC scan");
        } else {
            if (ss != 0 || ah != 0 || al != 0) bad(what, "sequential scan with progressive parameters");
            se = 63;
        }
        for (int i = 0; i < scanN; i++) {
            const Component& c = comp[order[i]];
            const bool needDc = !progressive || ss == 0, needAc = !progressive || ss != 0;
            if (needDc && ah == 0 && !dc[c.td].defined) bad(what, "scan uses a DC Huffman table that was never defined");
            if (needAc && !ac[c.ta].defined) bad(what, "scan uses an AC Huffman table that was never defined");
        }
        Bits b;
        b.p = f.data() + at;
        b.end = f.data() + f.size();
        for (int i = 0; i < nComp; i++) comp[i].pred = 0;
        eobRun = 0;
        int todo = restartInterval ? restartInterval : 0x7FFFFFFF;
        bool more = true;
        if (scanN == 1) { // the component's own blocks, row by row
            Component& c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int by = 0; by < h && more; by++)
                for (int bx = 0; bx < w && more; bx++) {
                    block(b, c, bx, by);
                    if (--todo <= 0) {
                        more = restart(b);
                        todo = restartInterval ? restartInterval : 0x7FFFFFFF;
                    }
                }
        } else { // interleaved: every component's blocks of one MCU together
            for (int my = 0; my < mcusY && more; my++)
                for (int mx = 0; mx < mcusX && more; mx++) {
                    for (int i = 0; i < scanN; i++) {
                        Component& c = comp[order[i]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) block(b, c, mx * c.h + x, my * c.v + y);
                    }
                    if (--todo <= 0) {
                        more = restart(b);
                        todo = restartInterval ? restartInterval : 0x7FFFFFFF;
                    }
                }
        }
        // where the next marker is: the one the bit reader ran into, else the first 0xFF xx (xx != 0) behind what it consumed
        if (b.marker) {
            at = static_cast<size_t>(b.p - f.data());
            pending = b.marker;
        } else {
            at = static_cast<size_t>(b.p - f.data());
            pending = 0;
        }
    }
    int pending = 0; // marker already consumed from the byte stream by the entropy decoder

    unsigned nextMarker()
    {
        if (pending) {
            const unsigned m = static_cast<unsigned>(pending);
            pending = 0;
            return m;
        }
        for (;;) { // anything up to the next 0xFF xx with xx not 0 and not 0xFF is skipped
            if (at >= f.size()) bad(what, "file ends without an end-of-image marker");
            if (f[at++] != 0xFF) continue;
            while (at < f.size() && f[at] == 0xFF) at++;
            if (at >= f.size()) bad(what, "file ends without an end-of-image marker");
            const unsigned m = f[at++];
            if (m != 0) return m;
        }
    }

    void reconstruct()
    {
        for (int i = 0; i < nComp; i++) {
            Component& c = comp[i];
            const int stride = c.blocksW * 8;
            c.plane.assign(static_cast<size_t>(stride) * c.blocksH * 8, 0);
            const uint16_t* q = quant[c.tq];
            for (int by = 0; by < c.blocksH; by++)
                for (int bx = 0; bx < c.blocksW; bx++) {
                    int16_t* d = c.coef.data() + (static_cast<size_t>(by) * c.blocksW + bx) * 64;
                    if (progressive)
                        for (int k = 0; k < 64; k++) d[k] = static_cast<int16_t>(static_cast<uint32_t>(static_cast<int32_t>(d[k])) * q[k]);
                    inverseDct(d, c.plane.data() + static_cast<size_t>(by) * 8 * stride + bx * 8, stride);
                }
        }
    }
};

// ---- chroma rows to full resolution
inline unsigned char quarter(int v) { return static_cast<unsigned char>(v >> 2); }
inline unsigned char sixteenth(int v) { return static_cast<unsigned char>(v >> 4); }

const unsigned char* upsampleRow(int hs, int vs, unsigned char* out, const unsigned char* nearRow, const unsigned char* farRow, int w)
{
    if (hs == 1 && vs == 1) return nearRow;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = quarter(3 * nearRow[i] + farRow[i] + 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) {
            out[0] = out[1] = nearRow[0];
            return out;
        }
        out[0] = nearRow[0];
        out[1] = quarter(3 * nearRow[0] + nearRow[1] + 2);
        for (int i = 1; i < w - 1; i++) {
            out[2 * i] = quarter(3 * nearRow[i] + 2 + nearRow[i - 1]);
            out[2 * i + 1] = quarter(3 * nearRow[i] + 2 + nearRow[i + 1]);
        }
        out[2 * (w - 1)] = quarter(3 * nearRow[w - 2] + nearRow[w - 1] + 2); // (sic: the reference decoder weights the last-but-one sample here)
        out[2 * (w - 1) + 1] = nearRow[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        int prev = 3 * nearRow[0] + farRow[0]; // the vertical filter first (16 x the value), then the horizontal one
        if (w == 1) {
            out[0] = out[1] = quarter(prev + 2);
            return out;
        }
        out[0] = quarter(prev + 2);
        for (int i = 1; i < w; i++) {
            const int cur = 3 * nearRow[i] + farRow[i];
            out[2 * i - 1] = sixteenth(3 * prev + cur + 8);
            out[2 * i] = sixteenth(3 * cur + prev + 8);
            prev = cur;
        }
        out[2 * w - 1] = quarter(prev + 2);
        return out;
    }
    for (int i = 0; i < w; i++) // any other ratio: the sample repeated
        for (int j = 0; j < hs; j++) out[i * hs + j] = nearRow[i];
    return out;
}

inline unsigned char mul255(unsigned a, unsigned b)
{
    const unsigned t = a * b + 128;
    return static_cast<unsigned char>((t + (t >> 8)) >> 8);
}

inline int fix20(float x) { return static_cast<int>(x * 4096.0f + 0.5f) << 8; }

void yccToRgb(unsigned char* out, int y, int cb, int cr)
{
    const int32_t base = (y << 20) + (1 << 19);
    const int32_t r = wrapAdd(base, wrapMul(cr - 128, fix20(1.40200f)));
    const uint32_t gcb = static_cast<uint32_t>(wrapMul(cb - 128, -fix20(0.34414f))) & 0xFFFF0000u;
    const int32_t g = wrapAdd(wrapAdd(base, wrapMul(cr - 128, -fix20(0.71414f))), static_cast<int32_t>(gcb));
    const int32_t b = wrapAdd(base, wrapMul(cb - 128, fix20(1.77200f)));
    out[0] = clamp8(r >> 20);
    out[1] = clamp8(g >> 20);
    out[2] = clamp8(b >> 20);
}

} // namespace

DecodedImage decodeJpeg(const std::vector<unsigned char>& file, const std::string& what)
{
    Decoder d(file, what);
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) bad(what, "no start-of-image marker");
    bool done = false;
    while (!done) {
        const unsigned m = d.nextMarker();
        if (m == 0xD9) done = true;
        else if (m == 0xC0 || m == 0xC1 || m == 0xC2) d.frame(m);
        else if (m == 0xDA) d.scan();
        else if (m >= 0xD0 && m <= 0xD7) continue; // a restart marker outside its interval: nothing to do
        else if (m == 0xDC) bad(what, "frame height given after the first scan (DNL) is not supported");
        else d.tables(m);
    }
    if (!d.haveFrame) bad(what, "no frame");
    d.reconstruct();

    DecodedImage img;
    img.width = d.width;
    img.height = d.height;
    const int n = d.nComp >= 3 ? 3 : 1;
    img.channels = n;
    img.pixels.assign(static_cast<size_t>(d.width) * d.height * n, 0);
    const bool passThrough = d.nComp == 3 && (d.rgbTagged == 3 || (d.adobeTransform == 0 && !d.jfif));

    struct Walk { int hs, vs, step, row, wLow; const unsigned char* line0; const unsigned char* line1; std::vector<unsigned char> buf; } walk[4];
    for (int k = 0; k < d.nComp; k++) {
        Walk& r = walk[k];
        r.hs = d.hMax / d.comp[k].h;
        r.vs = d.vMax / d.comp[k].v;
        r.step = r.vs >> 1;
        r.row = 0;
        r.wLow = (d.width + r.hs - 1) / r.hs;
        r.line0 = r.line1 = d.comp[k].plane.data();
        r.buf.assign(static_cast<size_t>(d.width) + 8, 0);
    }
    const unsigned char* rows[4] = { nullptr, nullptr, nullptr, nullptr };
    for (int y = 0; y < d.height; y++) {
        for (int k = 0; k < d.nComp; k++) {
            Walk& r = walk[k];
            const bool lower = r.step >= (r.vs >> 1); // this output row lies in the lower half of its source row: the row below is the far one
            rows[k] = upsampleRow(r.hs, r.vs, r.buf.data(), lower ? r.line1 : r.line0, lower ? r.line0 : r.line1, r.wLow);
            if (++r.step >= r.vs) {
                r.step = 0;
                r.line0 = r.line1;
                if (++r.row < d.comp[k].y) r.line1 += static_cast<size_t>(d.comp[k].blocksW) * 8;
            }
        }
        unsigned char* out = img.pixels.data() + static_cast<size_t>(y) * d.width * n;
        for (int x = 0; x < d.width; x++, out += n) {
            if (d.nComp == 1) {
                out[0] = rows[0][x];
            } else if (d.nComp == 3) {
                if (passThrough) {
                    out[0] = rows[0][x]; out[1] = rows[1][x]; out[2] = rows[2][x];
                } else {
                    yccToRgb(out, rows[0][x], rows[1][x], rows[2][x]);
                }
            } else if (d.adobeTransform == 0) { // CMYK as Adobe writes it (inverted): colour x black
                const unsigned k = rows[3][x];
                out[0] = mul255(rows[0][x], k); out[1] = mul255(rows[1][x], k); out[2] = mul255(rows[2][x], k);
            } else if (d.adobeTransform == 2) { // YCCK
                yccToRgb(out, rows[0][x], rows[1][x], rows[2][x]);
                const unsigned k = rows[3][x];
                out[0] = mul255(255u - out[0], k); out[1] = mul255(255u - out[1], k); out[2] = mul255(255u - out[2], k);
            } else { // three colour components and a fourth that is ignored
                yccToRgb(out, rows[0][x], rows[1][x], rows[2][x]);
            }
        }
    }
    return img;
}

} // namespace crt
