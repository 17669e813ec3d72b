// Bitmap texture decoders: PNG (with its own inflate), BMP, TGA, binary PPM / PGM.  See image_decode.h.
#include "image_decode.h"

#include <cstdint>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace crt {
namespace {

[[noreturn]] void bad(const std::string& what, const char* why) { throw std::runtime_error("texture '" + what + "': " + why); }

constexpr long long kMaxTexels = 1ll << 28;

// ---------------------------------------------------------------------------------------------- inflate (RFC 1951)
struct BitReader {
    const unsigned char* p;
    size_t size, pos = 0;
    uint32_t hold = 0;
    int bits = 0;
    const std::string& what;
    uint32_t take(int n) // n <= 16
    {
        while (bits < n) {
            if (pos >= size) bad(what, "compressed data ends early");
            hold |= static_cast<uint32_t>(p[pos++]) << bits;
            bits += 8;
        }
        const uint32_t v = hold & ((1u << n) - 1u);
        hold >>= n;
        bits -= n;
        return v;
    }
    void alignToByte()
    {
        hold = 0;
        bits = 0;
    }
};

// canonical Huffman code given as code lengths: count[len] codes of each length, symbols ordered by (length, value)
struct Huffman {
    uint16_t count[16];
    uint16_t symbol[288];
    bool build(const uint8_t* lengths, int n)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1; // over-subscribed or incomplete sets are rejected (a single code of length 1 is allowed: distance trees)
        int used = 0;
        for (int len = 1; len < 16; len++) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;
            used += count[len];
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
        return left == 0 || used <= 1;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= static_cast<int>(br.take(1));
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

void inflateBlock(BitReader& br, const Huffman& lit, const Huffman& dist, std::vector<unsigned char>& out, size_t limit)
{
    for (;;) {
        const int sym = lit.decode(br);
        if (sym < 0) bad(br.what, "bad Huffman code");
        if (sym < 256) {
            if (out.size() >= limit) bad(br.what, "decompressed data larger than the image");
            out.push_back(static_cast<unsigned char>(sym));
        } else if (sym == 256) {
            return;
        } else {
            if (sym > 285) bad(br.what, "bad length symbol");
            const size_t len = kLenBase[sym - 257] + br.take(kLenExtra[sym - 257]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds > 29) bad(br.what, "bad distance symbol");
            const size_t d = kDistBase[ds] + br.take(kDistExtra[ds]);
            if (d > out.size()) bad(br.what, "distance reaches before the start of the data");
            if (out.size() + len > limit) bad(br.what, "decompressed data larger than the image");
            size_t from = out.size() - d;
            for (size_t i = 0; i < len; i++) out.push_back(out[from++]);
        }
    }
}

} // namespace

std::vector<unsigned char> zlibInflate(const unsigned char* data, size_t size, size_t expectedSize, const std::string& what)
{
    if (size < 2 || (data[0] & 0x0F) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20)) bad(what, "bad zlib header");
    BitReader br{ data + 2, size - 2, 0, 0, 0, what };
    std::vector<unsigned char> out;
    out.reserve(expectedSize);
    Huffman lit, dist;
    for (;;) {
        const uint32_t last = br.take(1), type = br.take(2);
        if (type == 0) {
            br.alignToByte();
            if (br.pos + 4 > br.size) bad(what, "compressed data ends early");
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen) bad(what, "bad stored block");
            if (br.pos + len > br.size) bad(what, "compressed data ends early");
            if (out.size() + len > expectedSize) bad(what, "decompressed data larger than the image");
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1) {
            uint8_t l[288], d[30];
            for (int i = 0; i < 144; i++) l[i] = 8;
            for (int i = 144; i < 256; i++) l[i] = 9;
            for (int i = 256; i < 280; i++) l[i] = 7;
            for (int i = 280; i < 288; i++) l[i] = 8;
            for (int i = 0; i < 30; i++) d[i] = 5;
            lit.build(l, 288);
            dist.build(d, 30);
            inflateBlock(br, lit, dist, out, expectedSize);
        } else if (type == 2) {
            const int nlen = static_cast<int>(br.take(5)) + 257, ndist = static_cast<int>(br.take(5)) + 1, ncode = static_cast<int>(br.take(4)) + 4;
            if (nlen > 286 || ndist > 30) bad(what, "bad dynamic block header");
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t lengths[320];
            std::memset(lengths, 0, sizeof(lengths));
            for (int i = 0; i < ncode; i++) lengths[order[i]] = static_cast<uint8_t>(br.take(3));
            Huffman cl;
            if (!cl.build(lengths, 19)) bad(what, "bad code-length code");
            uint8_t all[320];
            int i = 0;
            while (i < nlen + ndist) {
                const int sym = cl.decode(br);
                if (sym < 0) bad(what, "bad Huffman code");
                if (sym < 16) {
                    all[i++] = static_cast<uint8_t>(sym);
                } else {
                    uint8_t prev = 0;
                    int rep;
                    if (sym == 16) {
                        if (i == 0) bad(what, "repeat without a previous length");
                        prev = all[i - 1];
                        rep = 3 + static_cast<int>(br.take(2));
                    } else if (sym == 17) {
                        rep = 3 + static_cast<int>(br.take(3));
                    } else {
                        rep = 11 + static_cast<int>(br.take(7));
                    }
                    if (i + rep > nlen + ndist) bad(what, "too many code lengths");
                    while (rep--) all[i++] = prev;
                }
            }
            if (all[256] == 0) bad(what, "no end-of-block code");
            if (!lit.build(all, nlen)) bad(what, "bad literal/length code");
            if (!dist.build(all + nlen, ndist)) bad(what, "bad distance code");
            inflateBlock(br, lit, dist, out, expectedSize);
        } else {
            bad(what, "bad block type");
        }
        if (last) break;
    }
    return out;
}

namespace {

uint32_t be32(const unsigned char* p) { return (static_cast<uint32_t>(p[0]) << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
uint32_t le32(const unsigned char* p) { return (static_cast<uint32_t>(p[3]) << 24) | (p[2] << 16) | (p[1] << 8) | p[0]; }
uint32_t le16(const unsigned char* p) { return static_cast<uint32_t>(p[1] << 8) | p[0]; }

// ---------------------------------------------------------------------------------------------- PNG
int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the scanline filters of one (sub-)image of w x h pixels whose lines are `stride` bytes long, each preceded by its filter byte
void unfilter(unsigned char* data, size_t w, size_t h, size_t stride, size_t bpp, const std::string& what)
{
    (void)w;
    for (size_t y = 0; y < h; y++) {
        unsigned char* line = data + y * (stride + 1);
        const int filter = line[0];
        unsigned char* cur = line + 1;
        const unsigned char* up = y ? cur - (stride + 1) : nullptr;
        if (filter > 4) bad(what, "bad scanline filter");
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = cur[i];
            if (filter == 1) v += a;
            else if (filter == 2) v += b;
            else if (filter == 3) v += (a + b) >> 1;
            else if (filter == 4) v += paeth(a, b, c);
            cur[i] = static_cast<unsigned char>(v);
        }
    }
}

DecodedImage decodePng(const std::vector<unsigned char>& f, const std::string& what)
{
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    bool haveHeader = false, haveTrns = false;
    unsigned char palette[256 * 4];
    int paletteLen = 0;
    for (int i = 0; i < 256; i++) palette[4 * i + 3] = 255;
    unsigned char trnsKey[3] = { 0, 0, 0 };
    uint32_t trns16[3] = { 0, 0, 0 };
    std::vector<unsigned char> idat;
    for (;;) {
        if (pos + 8 > f.size()) bad(what, "PNG ends before IEND");
        const uint32_t len = be32(&f[pos]);
        const unsigned char* type = &f[pos + 4];
        if (len > f.size() || pos + 12 + len > f.size()) bad(what, "PNG chunk runs past the end of the file");
        const unsigned char* d = &f[pos + 8];
        if (!haveHeader && std::memcmp(type, "IHDR", 4) != 0) bad(what, "PNG does not start with IHDR");
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13 || haveHeader) bad(what, "bad IHDR");
            w = be32(d); h = be32(d + 4); depth = d[8]; color = d[9]; interlace = d[12];
            if (w == 0 || h == 0 || static_cast<long long>(w) * h > kMaxTexels) bad(what, "bad PNG size");
            if (d[10] != 0 || d[11] != 0 || interlace > 1) bad(what, "unknown PNG compression / filter / interlace method");
            const bool ok = (color == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                            ((color == 2 || color == 4 || color == 6) && (depth == 8 || depth == 16)) ||
                            (color == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8));
            if (!ok) bad(what, "bad PNG colour type / bit depth");
            haveHeader = true;
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) bad(what, "bad PLTE");
            paletteLen = static_cast<int>(len / 3);
            for (int i = 0; i < paletteLen; i++) std::memcpy(&palette[4 * i], d + 3 * i, 3);
        } else if (std::memcmp(type, "tRNS", 4) == 0) {
            if (!idat.empty()) bad(what, "tRNS after IDAT");
            if (color == 3) {
                if (len > static_cast<uint32_t>(paletteLen)) bad(what, "bad tRNS");
                for (uint32_t i = 0; i < len; i++) palette[4 * i + 3] = d[i];
                haveTrns = true;
            } else if (color == 0 || color == 2) {
                const uint32_t n = color == 0 ? 1u : 3u;
                if (len != 2 * n) bad(what, "bad tRNS");
                for (uint32_t i = 0; i < n; i++) {
                    trns16[i] = static_cast<uint32_t>(d[2 * i] << 8) | d[2 * i + 1];
                    // samples are compared after their reduction to 8 bits, as stb_image does
                    static const int scale[9] = { 0, 255, 85, 0, 17, 0, 0, 0, 1 };
                    trnsKey[i] = depth == 16 ? static_cast<unsigned char>(trns16[i] >> 8) : static_cast<unsigned char>((trns16[i] & 255u) * scale[depth]);
                }
                haveTrns = true;
            } else {
                bad(what, "tRNS in a PNG with an alpha channel");
            }
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            if (color == 3 && paletteLen == 0) bad(what, "palette PNG without PLTE");
            idat.insert(idat.end(), d, d + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        } else if (!(type[0] & 32)) {
            bad(what, "unknown critical PNG chunk");
        }
        pos += 12 + len;
    }
    if (idat.empty()) bad(what, "PNG without image data");
    const int fileCh = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4;
    const size_t bitsPerPixel = static_cast<size_t>(fileCh) * depth;
    const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;
    // the seven Adam7 passes, or the one pass of a plain image
    static const int x0[7] = { 0, 4, 0, 2, 0, 1, 0 }, y0[7] = { 0, 0, 4, 0, 2, 0, 1 }, dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
    const int passes = interlace ? 7 : 1;
    size_t rawSize = 0;
    size_t pw[7], ph[7];
    for (int p = 0; p < passes; p++) {
        pw[p] = interlace ? (w - x0[p] + dx[p] - 1) / dx[p] : w;
        ph[p] = interlace ? (h - y0[p] + dy[p] - 1) / dy[p] : h;
        if (interlace && (static_cast<uint32_t>(x0[p]) >= w || static_cast<uint32_t>(y0[p]) >= h)) pw[p] = ph[p] = 0;
        if (pw[p] && ph[p]) rawSize += ph[p] * (1 + (pw[p] * bitsPerPixel + 7) / 8);
    }
    std::vector<unsigned char> raw = zlibInflate(idat.data(), idat.size(), rawSize, what);
    if (raw.size() != rawSize) bad(what, "PNG image data has the wrong size");
    // samples as 8-bit values, file channel count
    std::vector<unsigned char> samples(static_cast<size_t>(w) * h * fileCh);
    size_t at = 0;
    for (int p = 0; p < passes; p++) {
        if (!pw[p] || !ph[p]) continue;
        const size_t stride = (pw[p] * bitsPerPixel + 7) / 8;
        unfilter(&raw[at], pw[p], ph[p], stride, bpp, what);
        for (size_t y = 0; y < ph[p]; y++) {
            const unsigned char* line = &raw[at + y * (stride + 1) + 1];
            const size_t oy = interlace ? y0[p] + y * dy[p] : y;
            for (size_t x = 0; x < pw[p]; x++) {
                const size_t ox = interlace ? x0[p] + x * dx[p] : x;
                unsigned char* dst = &samples[(oy * w + ox) * fileCh];
                if (depth == 8) {
                    std::memcpy(dst, line + x * fileCh, static_cast<size_t>(fileCh));
                } else if (depth == 16) {
                    for (int c = 0; c < fileCh; c++) dst[c] = line[(x * fileCh + c) * 2]; // the high byte
                } else { // 1, 2, 4 bits, one channel
                    const size_t bit = x * depth;
                    const unsigned v = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    static const int scale[5] = { 0, 255, 85, 0, 17 };
                    dst[0] = static_cast<unsigned char>(color == 3 ? v : v * scale[depth]);
                }
            }
        }
        at += ph[p] * (stride + 1);
    }
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    const size_t n = static_cast<size_t>(w) * h;
    if (color == 3) {
        img.channels = haveTrns ? 4 : 3;
        img.pixels.resize(n * img.channels);
        for (size_t i = 0; i < n; i++) {
            const int idx = samples[i];
            if (idx >= paletteLen) bad(what, "palette index out of range");
            std::memcpy(&img.pixels[i * img.channels], &palette[4 * idx], static_cast<size_t>(img.channels));
        }
    } else if (haveTrns) { // a colour key becomes an alpha channel
        img.channels = fileCh + 1;
        img.pixels.resize(n * img.channels);
        for (size_t i = 0; i < n; i++) {
            bool key = true;
            for (int c = 0; c < fileCh; c++) {
                img.pixels[i * img.channels + c] = samples[i * fileCh + c];
                key = key && samples[i * fileCh + c] == trnsKey[c];
            }
            img.pixels[i * img.channels + fileCh] = key ? 0 : 255;
        }
    } else {
        img.channels = fileCh;
        img.pixels = std::move(samples);
    }
    return img;
}

// ---------------------------------------------------------------------------------------------- BMP
DecodedImage decodeBmp(const std::vector<unsigned char>& f, const std::string& what)
{
    if (f.size() < 26) bad(what, "BMP file too short");
    const uint32_t dataOffset = le32(&f[10]), hsz = le32(&f[14]);
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) bad(what, "unknown BMP header");
    if (14 + static_cast<size_t>(hsz) > f.size()) bad(what, "BMP file too short");
    long long w, h;
    uint32_t bpp, compression = 0;
    if (hsz == 12) {
        w = static_cast<long long>(le16(&f[18]));
        h = static_cast<long long>(le16(&f[20]));
        bpp = le16(&f[24]);
    } else {
        w = static_cast<int32_t>(le32(&f[18]));
        h = static_cast<int32_t>(le32(&f[22]));
        bpp = le16(&f[28]);
        compression = le32(&f[30]);
    }
    const bool flip = h > 0; // rows bottom-up unless the height is negative
    if (h < 0) h = -h;
    if (w <= 0 || h <= 0 || w * h > kMaxTexels) bad(what, "bad BMP size");
    if (compression == 1 || compression == 2) bad(what, "run-length coded BMP files are not supported (the reference's decoder rejects them too)");
    if (compression >= 4) bad(what, "BMP files that wrap a JPEG or PNG are not supported (the reference's decoder rejects them too)");
    if (compression == 3 && bpp != 16 && bpp != 32) bad(what, "BMP channel masks need 16 or 32 bits per pixel");
    // channel masks, as the reference's decoder settles them (R/stb_image/stb_image.h:5484-5597): 16 bits default to 5-5-5, 32 bits
    // to 8-8-8-8 BGRA; BI_BITFIELDS files carry their own -- behind a 40-byte header (and, as that decoder has it, behind a 56-byte
    // one too, whose own mask fields it skips), or in the V4 / V5 header, where they only count in BI_BITFIELDS mode
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
    bool alphaMayBeUnused = false; // 32-bit default masks: an alpha channel that is zero everywhere means "no alpha"
    auto defaults = [&]() {
        if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        else if (bpp == 32) { mr = 0x00FF0000u; mg = 0x0000FF00u; mb = 0x000000FFu; ma = 0xFF000000u; alphaMayBeUnused = true; }
        else { mr = mg = mb = ma = 0; }
    };
    if (hsz == 40 || hsz == 56) {
        if (bpp == 16 || bpp == 32) {
            if (compression == 0) defaults();
            else {
                const size_t maskAt = 14 + static_cast<size_t>(hsz);
                if (maskAt + 12 > f.size()) bad(what, "BMP file too short");
                mr = le32(&f[maskAt]); mg = le32(&f[maskAt + 4]); mb = le32(&f[maskAt + 8]);
                if (mr == mg && mg == mb) bad(what, "BMP channel masks that are all the same");
            }
        }
    } else if (hsz == 108 || hsz == 124) {
        mr = le32(&f[14 + 40]); mg = le32(&f[14 + 44]); mb = le32(&f[14 + 48]); ma = le32(&f[14 + 52]);
        if (compression != 3) defaults();
    }
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    const size_t W = static_cast<size_t>(w), H = static_cast<size_t>(h);
    if (bpp == 8 || bpp == 4 || bpp == 1) {
        const size_t entry = hsz == 12 ? 3 : 4, palAt = 14 + static_cast<size_t>(hsz);
        size_t colours = hsz == 12 ? 0 : le32(&f[46]);
        if (colours == 0 || colours > (1u << bpp)) colours = 1u << bpp;
        if (palAt + colours * entry > f.size()) bad(what, "BMP palette runs past the end of the file");
        const size_t stride = ((W * bpp + 31) / 32) * 4;
        if (dataOffset > f.size() || stride * H > f.size() - dataOffset) bad(what, "BMP pixel data runs past the end of the file");
        img.channels = 3;
        img.pixels.resize(W * H * 3);
        for (size_t y = 0; y < H; y++) {
            const unsigned char* line = &f[dataOffset + (flip ? H - 1 - y : y) * stride];
            for (size_t x = 0; x < W; x++) {
                const size_t bit = x * bpp;
                const unsigned idx = bpp == 8 ? line[x] : (line[bit >> 3] >> (8 - bpp - (bit & 7))) & ((1u << bpp) - 1u);
                if (idx >= colours) bad(what, "BMP palette index out of range");
                const unsigned char* c = &f[palAt + idx * entry];
                unsigned char* dst = &img.pixels[(y * W + x) * 3];
                dst[0] = c[2]; dst[1] = c[1]; dst[2] = c[0];
            }
        }
        return img;
    }
    if (bpp != 16 && bpp != 24 && bpp != 32) bad(what, "BMP files of this bit depth are not supported");
    const size_t bytes = bpp / 8, stride = ((W * bpp + 31) / 32) * 4;
    if (dataOffset > f.size() || stride * H > f.size() - dataOffset) bad(what, "BMP pixel data runs past the end of the file");
    img.channels = (bpp == 24 && ma == 0xFF000000u) ? 3 : (ma ? 4 : 3);
    img.pixels.resize(W * H * img.channels);
    const bool bgr = bpp == 24, bgra = bpp == 32 && mb == 0xFFu && mg == 0xFF00u && mr == 0x00FF0000u && ma == 0xFF000000u;
    if (bgr || bgra) {
        bool anyAlpha = false;
        for (size_t y = 0; y < H; y++) {
            const unsigned char* line = &f[dataOffset + (flip ? H - 1 - y : y) * stride];
            for (size_t x = 0; x < W; x++) {
                const unsigned char* s = line + x * bytes;
                unsigned char* dst = &img.pixels[(y * W + x) * img.channels];
                dst[0] = s[2]; dst[1] = s[1]; dst[2] = s[0];
                if (img.channels == 4) {
                    dst[3] = bgra ? s[3] : 255;
                    anyAlpha = anyAlpha || (bgra && s[3] != 0);
                }
            }
        }
        if (bgra && alphaMayBeUnused && !anyAlpha) // (stb_image does the same)
            for (size_t i = 0; i < W * H; i++) img.pixels[i * 4 + 3] = 255;
        return img;
    }
    // any other masks: a channel's bits are moved to the top of a byte and the byte is filled by repeating them
    if (!mr || !mg || !mb) bad(what, "BMP without channel masks");
    auto highBit = [](uint32_t z) { int n = -1; while (z) { n++; z >>= 1; } return n; };
    auto bitCount = [](uint32_t z) { int n = 0; while (z) { n += static_cast<int>(z & 1u); z >>= 1; } return n; };
    const uint32_t mask[4] = { mr, mg, mb, ma };
    int shift[4], count[4];
    for (int k = 0; k < 4; k++) {
        shift[k] = highBit(mask[k]) - 7;
        count[k] = bitCount(mask[k]);
        if (count[k] > 8) bad(what, "BMP channel mask wider than 8 bits");
    }
    auto widen = [](uint32_t v, int shiftBy, int bits) -> unsigned char {
        static const unsigned mul[9] = { 0, 0xFF, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01 };
        static const unsigned down[9] = { 0, 0, 0, 1, 0, 2, 4, 6, 0 };
        v = shiftBy < 0 ? v << -shiftBy : v >> shiftBy;
        v = (v & 0xFFu) >> (8 - bits); // (a mask whose bits are not contiguous leaves stray bits above: only the byte counts)
        return static_cast<unsigned char>((v * mul[bits]) >> down[bits]);
    };
    for (size_t y = 0; y < H; y++) {
        const unsigned char* line = &f[dataOffset + (flip ? H - 1 - y : y) * stride];
        for (size_t x = 0; x < W; x++) {
            const uint32_t v = bpp == 16 ? le16(line + 2 * x) : le32(line + 4 * x);
            unsigned char* dst = &img.pixels[(y * W + x) * img.channels];
            for (int k = 0; k < 3; k++) dst[k] = widen(v & mask[k], shift[k], count[k]);
            if (img.channels == 4) dst[3] = widen(v & ma, shift[3], count[3]);
        }
    }
    return img;
}

// ---------------------------------------------------------------------------------------------- TGA
DecodedImage decodeTga(const std::vector<unsigned char>& f, const std::string& what)
{
    // Truevision TGA as the reference's decoder reads it (R/stb_image/stb_image.h, stbi__tga_load): true colour 24 / 32 bits, 15 / 16
    // bits (5-5-5, scaled c * 255 / 31, no alpha), grey 8 bits, grey + alpha 16 bits, colour-mapped with 8- or 16-bit indices into a
    // palette of any of those entry sizes; raw or run-length coded; bottom-up unless descriptor bit 5 is set (bit 4, right-to-left,
    // is ignored there and here); an index outside the palette reads entry 0; the palette's "first entry" field is skipped as BYTES.
    if (f.size() < 18) bad(what, "TGA file too short");
    const unsigned idLen = f[0], mapped = f[1], palStart = le16(&f[3]), palLen = le16(&f[5]), palBits = f[7];
    const size_t w = le16(&f[12]), h = le16(&f[14]);
    const unsigned bpp = f[16], descriptor = f[17];
    unsigned type = f[2];
    const bool rle = type >= 8;
    if (rle) type -= 8;
    if (mapped > 1) bad(what, "TGA colour-map type that is not 0 or 1");
    if (mapped ? type != 1 : (type != 2 && type != 3)) bad(what, "TGA image type that is not colour-mapped, true colour or grey");
    if (mapped && bpp != 8 && bpp != 16) bad(what, "colour-mapped TGA whose indices are not 8 or 16 bits");
    if (w == 0 || h == 0) bad(what, "bad TGA size");
    auto layout = [&](unsigned bits, bool grey, bool& packed555) -> unsigned { // bytes per texel of the result
        packed555 = false;
        if (bits == 8) return 1;
        if (bits == 16 && grey) return 2;
        if (bits == 15 || bits == 16) { packed555 = true; return 3; }
        if (bits == 24 || bits == 32) return bits / 8;
        bad(what, "TGA files of this bit depth are not supported");
    };
    bool packed = false;
    const unsigned comp = mapped ? layout(palBits, false, packed) : layout(bpp, type == 3, packed);
    const size_t n = w * h;
    size_t pos = 18 + static_cast<size_t>(idLen);
    auto need = [&](size_t bytes) { if (pos > f.size() || bytes > f.size() - pos) bad(what, "TGA data runs past the end of the file"); };
    auto rgb555 = [&](unsigned char* out) {
        need(2);
        const unsigned px = le16(&f[pos]);
        pos += 2;
        out[0] = static_cast<unsigned char>((((px >> 10) & 31u) * 255u) / 31u);
        out[1] = static_cast<unsigned char>((((px >> 5) & 31u) * 255u) / 31u);
        out[2] = static_cast<unsigned char>(((px & 31u) * 255u) / 31u);
    };
    std::vector<unsigned char> palette;
    if (mapped) {
        if (palLen == 0) bad(what, "colour-mapped TGA without a palette");
        need(palStart);
        pos += palStart;
        palette.resize(static_cast<size_t>(palLen) * comp);
        if (packed) {
            for (unsigned i = 0; i < palLen; i++) rgb555(&palette[static_cast<size_t>(i) * comp]);
        } else {
            need(palette.size());
            std::memcpy(palette.data(), &f[pos], palette.size());
            pos += palette.size();
        }
    }
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    img.channels = static_cast<int>(comp);
    img.pixels.resize(n * comp);
    unsigned char texel[4] = { 0, 0, 0, 0 };
    unsigned left = 0;       // texels left in the current packet
    bool repeating = false;  // ... which repeats `texel`
    for (size_t i = 0; i < n; i++) {
        bool readOne = true;
        if (rle) {
            if (left == 0) {
                need(1);
                const unsigned head = f[pos++];
                left = 1 + (head & 127u);
                repeating = (head >> 7) != 0;
            } else if (repeating) {
                readOne = false;
            }
        }
        if (readOne) {
            if (mapped) {
                need(bpp / 8);
                size_t idx = bpp == 8 ? f[pos] : le16(&f[pos]);
                pos += bpp / 8;
                if (idx >= palLen) idx = 0;
                std::memcpy(texel, &palette[idx * comp], comp);
            } else if (packed) {
                rgb555(texel);
            } else {
                need(comp);
                std::memcpy(texel, &f[pos], comp);
                pos += comp;
            }
        }
        std::memcpy(&img.pixels[i * comp], texel, comp);
        if (rle) left--;
    }
    if (!((descriptor >> 5) & 1u)) // stored bottom-up
        for (size_t y = 0; y * 2 < h; y++)
            for (size_t x = 0; x < w * comp; x++) std::swap(img.pixels[y * w * comp + x], img.pixels[(h - 1 - y) * w * comp + x]);
    if (comp >= 3 && !packed) // stored blue first
        for (size_t i = 0; i < n; i++) std::swap(img.pixels[i * comp], img.pixels[i * comp + 2]);
    return img;
}
DecodedImage decodePnm(const std::vector<unsigned char>& f, const std::string& what)
{
    size_t pos = 2;
    auto nextInt = [&]() -> long long {
        for (;;) { // whitespace and # comments
            if (pos >= f.size()) bad(what, "bad PNM header");
            const int c = f[pos];
            if (c == '#') {
                while (pos < f.size() && f[pos] != '\n') pos++;
            } else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') {
                pos++;
            } else {
                break;
            }
        }
        long long v = 0;
        int digits = 0;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9' && digits < 10) {
            v = v * 10 + (f[pos++] - '0');
            digits++;
        }
        if (!digits) bad(what, "bad PNM header");
        return v;
    };
    const long long w = nextInt(), h = nextInt(), maxv = nextInt();
    pos++; // the single whitespace byte after maxval
    if (w <= 0 || h <= 0 || maxv > 65535 || w * h > kMaxTexels) bad(what, "bad PNM header");
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    img.channels = f[1] == '6' ? 3 : 1;
    // samples are taken as they are, whatever the maximum says; above 255 they are two bytes each, of which the reference's decoder
    // keeps the SECOND (it reads the big-endian pairs as host-order words on a little-endian machine and keeps the word's high byte)
    const size_t wide = maxv > 255 ? 2 : 1, samples = static_cast<size_t>(w) * h * img.channels, bytes = samples * wide;
    if (pos > f.size() || bytes > f.size() - pos) bad(what, "truncated");
    img.pixels.resize(samples);
    for (size_t i = 0; i < samples; i++) img.pixels[i] = f[pos + i * wide + (wide - 1)];
    return img;
}

// GIF 87a / 89a: the FIRST image of the file on a transparent-black canvas, four bytes per texel (what the reference's stb_image
// hands to R/CRTTextureBitmap.cpp:10 for a GIF; conventions studied in R/stb_image/stb_image.h:6702-7060).  Indices whose palette
// entry is the graphic-control extension's transparent colour leave the canvas untouched; where the first image does not cover
// the canvas and the header names a background index above 0, the uncovered texels take that palette entry -- with red and blue
// exchanged, as that decoder stores it (it copies its B, G, R, A palette entry without the swap it applies to drawn texels).
// A file that ends inside the image data yields the rows decoded so far (its reader returns zeros past the end, and a zero is
// the data's own terminator); a damaged code stream is an error.
DecodedImage decodeGif(const std::vector<unsigned char>& f, const std::string& what)
{
    size_t at = 6;
    auto u8 = [&]() -> unsigned { return at < f.size() ? f[at++] : (at++, 0u); };
    auto u16 = [&]() -> unsigned { const unsigned lo = u8(); return lo | (u8() << 8); };
    if (f.size() < 13 || (f[4] != '7' && f[4] != '9') || f[5] != 'a') bad(what, "not a GIF 87a / 89a file");
    DecodedImage img;
    img.width = static_cast<int>(u16());
    img.height = static_cast<int>(u16());
    const unsigned flags = u8(), background = u8();
    u8(); // aspect ratio
    if (img.width == 0 || img.height == 0) bad(what, "GIF canvas without an extent");
    if (static_cast<uint64_t>(img.width) * static_cast<uint64_t>(img.height) > (1ull << 28)) bad(what, "more than 2^28 texels");
    struct Entry { unsigned char r, g, b, a; };
    Entry global[256] = {}, local[256] = {};
    auto readTable = [&](Entry* t, unsigned n, int transparent) {
        for (unsigned i = 0; i < n; i++) {
            t[i].r = static_cast<unsigned char>(u8());
            t[i].g = static_cast<unsigned char>(u8());
            t[i].b = static_cast<unsigned char>(u8());
            t[i].a = static_cast<int>(i) == transparent ? 0 : 255;
        }
    };
    if (flags & 0x80) readTable(global, 2u << (flags & 7), -1);
    img.channels = 4;
    const size_t texels = static_cast<size_t>(img.width) * img.height;
    img.pixels.assign(texels * 4, 0);
    std::vector<unsigned char> drawn(texels, 0);
    unsigned control = 0;
    int transparent = -1;
    for (;;) {
        const unsigned tag = u8();
        if (tag == 0x21) { // extension: only the graphic control block matters (transparent colour)
            const unsigned label = u8();
            unsigned len;
            if (label == 0xF9) {
                len = u8();
                if (len == 4) {
                    control = u8();
                    u16(); // delay
                    if (transparent >= 0) global[transparent].a = 255;
                    if (control & 1) {
                        transparent = static_cast<int>(u8());
                        global[transparent].a = 0;
                    } else {
                        u8();
                        transparent = -1;
                    }
                } else {
                    at += len;
                    continue;
                }
            }
            while ((len = u8()) != 0) at += len;
            if (at > f.size() + 4096) bad(what, "GIF extension runs past the end of the file");
            continue;
        }
        if (tag != 0x2C) bad(what, tag == 0x3B ? "GIF without an image" : "unknown block in a GIF");
        const unsigned x0 = u16(), y0 = u16(), w = u16(), h = u16();
        if (x0 + w > static_cast<unsigned>(img.width) || y0 + h > static_cast<unsigned>(img.height)) bad(what, "GIF image outside its canvas");
        const unsigned lflags = u8();
        const Entry* table = global;
        if (lflags & 0x80) {
            readTable(local, 2u << (lflags & 7), (control & 1) ? transparent : -1);
            table = local;
        } else if (!(flags & 0x80)) {
            bad(what, "GIF image without a colour table");
        }
        // rows in the order they are stored: every row once, or four interlace passes (every 8th from 0, every 8th from 4, every
        // 4th from 2, every 2nd from 1)
        unsigned row = 0, step = (lflags & 0x40) ? 8u : 1u, pass = (lflags & 0x40) ? 3u : 0u, col = 0;
        bool rowsLeft = w != 0 && h != 0;
        auto put = [&](unsigned index) {
            if (!rowsLeft) return;
            const size_t t = static_cast<size_t>(y0 + row) * img.width + x0 + col;
            drawn[t] = 1;
            const Entry& e = table[index];
            if (e.a > 128) {
                unsigned char* p = img.pixels.data() + 4 * t;
                p[0] = e.r; p[1] = e.g; p[2] = e.b; p[3] = e.a;
            }
            if (++col >= w) {
                col = 0;
                row += step;
                while (row >= h && pass > 0) {
                    step = 1u << pass;
                    row = step >> 1;
                    pass--;
                }
                if (row >= h) rowsLeft = false;
            }
        };
        // LZW, variable code size, least significant bit first, data in sub-blocks of up to 255 bytes
        const unsigned minSize = u8();
        if (minSize > 12) bad(what, "GIF code size above 12");
        const int clear = 1 << minSize;
        struct Code { int16_t prefix; unsigned char first, suffix; };
        std::vector<Code> codes(8192);
        for (int i = 0; i < clear; i++) codes[static_cast<size_t>(i)] = { -1, static_cast<unsigned char>(i), static_cast<unsigned char>(i) };
        int size = static_cast<int>(minSize) + 1, mask = (1 << size) - 1, avail = clear + 2, old = -1;
        bool sawClear = false;
        uint32_t bits = 0;
        int have = 0;
        unsigned left = 0;
        std::vector<unsigned char> run(8192);
        for (;;) {
            if (have < size) {
                if (left == 0) {
                    left = u8();
                    if (left == 0) break; // terminator (or the end of a truncated file)
                }
                left--;
                bits |= static_cast<uint32_t>(u8()) << have;
                have += 8;
                continue;
            }
            const int code = static_cast<int>(bits & static_cast<uint32_t>(mask));
            bits >>= size;
            have -= size;
            if (code == clear) {
                size = static_cast<int>(minSize) + 1;
                mask = (1 << size) - 1;
                avail = clear + 2;
                old = -1;
                sawClear = true;
            } else if (code == clear + 1) {
                break; // end of information: whatever follows in the file is not needed for the first image
            } else if (code <= avail) {
                if (!sawClear) bad(what, "GIF image data does not start with a clear code");
                if (old >= 0) {
                    if (avail >= 8192) bad(what, "GIF image data defines more than 8192 codes");
                    Code& c = codes[static_cast<size_t>(avail++)];
                    c.prefix = static_cast<int16_t>(old);
                    c.first = codes[static_cast<size_t>(old)].first;
                    c.suffix = codes[static_cast<size_t>(code)].first; // (for the code being defined right now that is c.first itself)
                } else if (code == avail) {
                    bad(what, "GIF image data uses a code before defining it");
                }
                size_t n = 0;
                for (int k = code; k >= 0; k = codes[static_cast<size_t>(k)].prefix) run[n++] = codes[static_cast<size_t>(k)].suffix;
                while (n) put(run[--n]);
                if ((avail & mask) == 0 && avail <= 0x0FFF) {
                    size++;
                    mask = (1 << size) - 1;
                }
                old = code;
            } else {
                bad(what, "GIF image data uses a code before defining it");
            }
        }
        if (background > 0)
            for (size_t t = 0; t < texels; t++)
                if (!drawn[t]) {
                    unsigned char* p = img.pixels.data() + 4 * t;
                    p[0] = global[background].b; p[1] = global[background].g; p[2] = global[background].r; p[3] = 255;
                }
        return img;
    }
}

// Radiance .hdr (RGBE, "#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w"), flat or run-length coded scanlines, turned
// into three 8-bit channels the way the reference's stb_image does for stbi_load (R/stb_image/stb_image.h:1886-1912, 7241-7400):
// value = mantissa * 2^(exponent - 136); byte = trunc(pow(value, 1 / 2.2f) * 255 + 0.5), clamped.  (pow in double precision from
// the C library: the known answers come from the same library here; another platform's pow may round a texel differently.)
DecodedImage decodeHdr(const std::vector<unsigned char>& f, const std::string& what)
{
    size_t at = 0;
    auto eof = [&]() { return at >= f.size(); };
    auto u8 = [&]() -> unsigned { return at < f.size() ? f[at++] : 0u; };
    auto line = [&]() { // up to the next newline (at most 1023 characters are kept, the rest of the line is dropped)
        std::string t;
        unsigned c = u8();
        while (!eof() && c != '\n') {
            t.push_back(static_cast<char>(c));
            if (t.size() == 1023) {
                while (!eof() && u8() != '\n') {}
                break;
            }
            c = u8();
        }
        return t;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") bad(what, "not a Radiance HDR file");
    bool rgbe = false;
    for (;;) {
        const std::string t = line();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") rgbe = true;
    }
    if (!rgbe) bad(what, "HDR file that is not 32-bit_rle_rgbe");
    const std::string dims = line();
    char* rest = nullptr;
    if (dims.compare(0, 3, "-Y ") != 0) bad(what, "HDR file with an unsupported scanline order");
    const long h = std::strtol(dims.c_str() + 3, &rest, 10);
    while (*rest == ' ') rest++;
    if (std::strncmp(rest, "+X ", 3) != 0) bad(what, "HDR file with an unsupported scanline order");
    const long w = std::strtol(rest + 3, nullptr, 10);
    if (w <= 0 || h <= 0 || static_cast<uint64_t>(w) * static_cast<uint64_t>(h) > (1ull << 28)) bad(what, "HDR size missing or above 2^28 texels");
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    img.channels = 3;
    std::vector<float> lin(static_cast<size_t>(w) * h * 3, 0.0f);
    auto convert = [&](float* out, const unsigned char* q) {
        if (q[3] != 0) {
            const float scale = static_cast<float>(std::ldexp(1.0f, static_cast<int>(q[3]) - (128 + 8)));
            out[0] = q[0] * scale; out[1] = q[1] * scale; out[2] = q[2] * scale;
        } else {
            out[0] = out[1] = out[2] = 0.0f;
        }
    };
    auto flatFrom = [&](size_t firstTexel) { // four bytes per texel, row after row
        for (size_t t = firstTexel; t < static_cast<size_t>(w) * h; t++) {
            unsigned char q[4];
            for (int k = 0; k < 4; k++) q[k] = static_cast<unsigned char>(u8());
            convert(lin.data() + 3 * t, q);
        }
    };
    if (w < 8 || w >= 32768) {
        flatFrom(0);
    } else {
        std::vector<unsigned char> scan(static_cast<size_t>(w) * 4);
        for (long j = 0; j < h; j++) {
            const unsigned c1 = u8(), c2 = u8(), hi = u8();
            if (c1 != 2 || c2 != 2 || (hi & 0x80)) {
                // not a run-length coded scanline: these four bytes are a texel, and that decoder goes on reading flat texels from
                // the frame's SECOND texel whatever row it was in (so only a file that is flat from its first row decodes sensibly)
                const unsigned char q[4] = { static_cast<unsigned char>(c1), static_cast<unsigned char>(c2), static_cast<unsigned char>(hi), static_cast<unsigned char>(u8()) };
                convert(lin.data(), q);
                flatFrom(1);
                break;
            }
            if (static_cast<long>((hi << 8) | u8()) != w) bad(what, "HDR scanline of the wrong length");
            for (int k = 0; k < 4; k++) {
                long i = 0;
                while (i < w) {
                    unsigned count = u8();
                    const bool run = count > 128;
                    if (run) count -= 128;
                    if (count == 0 || static_cast<long>(count) > w - i) bad(what, "damaged run-length data in an HDR file");
                    const unsigned value = run ? u8() : 0u;
                    for (unsigned z = 0; z < count; z++) scan[static_cast<size_t>(i++) * 4 + k] = static_cast<unsigned char>(run ? value : u8());
                }
            }
            for (long i = 0; i < w; i++) convert(lin.data() + 3 * (static_cast<size_t>(j) * w + i), scan.data() + 4 * i);
        }
    }
    img.pixels.resize(lin.size());
    const float gammaInv = 1.0f / 2.2f, scaleInv = 1.0f;
    for (size_t i = 0; i < lin.size(); i++) {
        float z = static_cast<float>(std::pow(lin[i] * scaleInv, gammaInv)) * 255 + 0.5f;
        if (z < 0) z = 0;
        if (z > 255) z = 255;
        img.pixels[i] = static_cast<unsigned char>(static_cast<int>(z));
    }
    return img;
}

// Photoshop .psd: the merged image of an 8- or 16-bit RGB file, raw or PackBits-coded planes, as four 8-bit channels (16-bit
// samples reduced to their high byte, missing planes 0 / alpha 255); with an alpha plane the colours are un-matted from white
// as the reference's stb_image does (R/stb_image/stb_image.h:6211-6390): c' = trunc(c / a + 255 (1 - 1 / a)) for 0 < alpha < 255.
DecodedImage decodePsd(const std::vector<unsigned char>& f, const std::string& what)
{
    size_t at = 4;
    auto u8 = [&]() -> unsigned { return at < f.size() ? f[at++] : (at++, 0u); };
    auto u16 = [&]() -> unsigned { const unsigned hi = u8(); return (hi << 8) | u8(); };
    auto u32 = [&]() -> uint32_t { const uint32_t hi = u16(); return (hi << 16) | u16(); };
    auto skip = [&](uint64_t n) {
        if (n > f.size() || at + n > f.size() + 16) bad(what, "PSD section runs past the end of the file");
        at += static_cast<size_t>(n);
    };
    if (u16() != 1) bad(what, "PSD version that is not 1");
    skip(6);
    const unsigned channels = u16();
    if (channels > 16) bad(what, "PSD with more than 16 channels");
    const uint32_t h = u32(), w = u32();
    const unsigned depth = u16();
    if (depth != 8 && depth != 16) bad(what, "PSD bit depth that is not 8 or 16");
    if (u16() != 3) bad(what, "PSD that is not in RGB mode");
    skip(u32()); // colour mode data
    skip(u32()); // image resources
    skip(u32()); // layers and masks
    const unsigned compression = u16();
    if (compression > 1) bad(what, "PSD with an unknown compression");
    if (w == 0 || h == 0 || static_cast<uint64_t>(w) * h > (1ull << 28)) bad(what, "PSD size missing or above 2^28 texels");
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    img.channels = 4;
    const size_t texels = static_cast<size_t>(w) * h;
    img.pixels.assign(texels * 4, 0);
    if (compression) skip(static_cast<uint64_t>(h) * channels * 2); // the byte counts of the coded rows: not needed, the planes are decoded as one stream
    for (unsigned c = 0; c < 4; c++) {
        unsigned char* p = img.pixels.data() + c;
        if (c >= channels) {
            for (size_t i = 0; i < texels; i++) p[4 * i] = c == 3 ? 255 : 0;
        } else if (compression) { // PackBits over the whole plane
            size_t done = 0;
            while (done < texels) {
                if (at >= f.size()) bad(what, "PSD plane ends early");
                unsigned len = u8();
                if (len == 128) continue;
                if (len < 128) {
                    len++;
                    if (len > texels - done) bad(what, "damaged run-length data in a PSD file");
                    for (unsigned k = 0; k < len; k++) p[4 * done++] = static_cast<unsigned char>(u8());
                } else {
                    len = 257 - len;
                    if (len > texels - done) bad(what, "damaged run-length data in a PSD file");
                    const unsigned v = u8();
                    for (unsigned k = 0; k < len; k++) p[4 * done++] = static_cast<unsigned char>(v);
                }
            }
        } else {
            if (at >= f.size() && texels) bad(what, "PSD plane ends early");
            for (size_t i = 0; i < texels; i++) p[4 * i] = static_cast<unsigned char>(depth == 16 ? (u16() >> 8) : u8());
        }
    }
    if (channels >= 4)
        for (size_t i = 0; i < texels; i++) {
            unsigned char* px = img.pixels.data() + 4 * i;
            if (px[3] != 0 && px[3] != 255) {
                const float a = px[3] / 255.0f, ra = 1.0f / a, inv = 255.0f * (1 - ra);
                for (int k = 0; k < 3; k++) px[k] = static_cast<unsigned char>(static_cast<int>(px[k] * ra + inv)); // (colours matted on white stay in 0..255)
            }
        }
    return img;
}

// Softimage .pic: 8-bit channels in up to ten packets per scanline (each naming its channels with a mask: 0x80 red .. 0x10 alpha),
// raw, pure run-length or mixed run-length coded; three channels, or four when some packet carries alpha; channels no packet
// carries are 255 (conventions of the reference's stb_image, R/stb_image/stb_image.h:6440-6590).
DecodedImage decodePic(const std::vector<unsigned char>& f, const std::string& what)
{
    size_t at = 92;
    auto need = [&]() { if (at >= f.size()) bad(what, "PIC file ends early"); };
    auto u8 = [&]() -> unsigned { return at < f.size() ? f[at++] : (at++, 0u); };
    auto u16 = [&]() -> unsigned { const unsigned hi = u8(); return (hi << 8) | u8(); };
    if (f.size() < 104 || std::memcmp(f.data() + 88, "PICT", 4) != 0) bad(what, "not a Softimage PIC file");
    const unsigned w = u16(), h = u16();
    if (w == 0 || h == 0 || static_cast<uint64_t>(w) * h > (1ull << 28)) bad(what, "PIC size missing or above 2^28 texels");
    at += 8; // ratio, fields, pad
    struct Packet { unsigned type, mask; } packets[10];
    int nPackets = 0;
    unsigned all = 0, chained;
    do {
        if (nPackets == 10) bad(what, "PIC file with more than ten packets per scanline");
        chained = u8();
        const unsigned bits = u8();
        packets[nPackets].type = u8();
        packets[nPackets].mask = u8();
        all |= packets[nPackets].mask;
        nPackets++;
        need();
        if (bits != 8) bad(what, "PIC packet that is not 8 bits per channel");
    } while (chained);
    std::vector<unsigned char> rgba(static_cast<size_t>(w) * h * 4, 0xFF);
    auto read = [&](unsigned mask, unsigned char* dest) {
        for (int i = 0; i < 4; i++)
            if (mask & (0x80u >> i)) {
                need();
                dest[i] = static_cast<unsigned char>(u8());
            }
    };
    auto copy = [&](unsigned mask, unsigned char* dest, const unsigned char* src) {
        for (int i = 0; i < 4; i++)
            if (mask & (0x80u >> i)) dest[i] = src[i];
    };
    for (unsigned y = 0; y < h; y++)
        for (int k = 0; k < nPackets; k++) {
            const Packet& pk = packets[k];
            unsigned char* dest = rgba.data() + static_cast<size_t>(y) * w * 4;
            if (pk.type == 0) {
                for (unsigned x = 0; x < w; x++, dest += 4) read(pk.mask, dest);
            } else if (pk.type == 1) { // runs only; a run longer than the rest of the line is cut
                unsigned left = w;
                while (left > 0) {
                    unsigned count = u8();
                    need();
                    if (count > left) count = left;
                    unsigned char v[4] = { 0, 0, 0, 0 };
                    read(pk.mask, v);
                    for (unsigned i = 0; i < count; i++, dest += 4) copy(pk.mask, dest, v);
                    left -= count;
                    if (count == 0 && at >= f.size()) bad(what, "PIC file ends early");
                }
            } else if (pk.type == 2) { // runs (128: 16-bit length follows, above: length - 127) and literal stretches (below 128: length + 1)
                unsigned left = w;
                while (left > 0) {
                    unsigned count = u8();
                    need();
                    if (count >= 128) {
                        count = count == 128 ? u16() : count - 127;
                        if (count > left) bad(what, "PIC run longer than its scanline");
                        unsigned char v[4] = { 0, 0, 0, 0 };
                        read(pk.mask, v);
                        for (unsigned i = 0; i < count; i++, dest += 4) copy(pk.mask, dest, v);
                        if (count == 0 && at >= f.size()) bad(what, "PIC file ends early");
                    } else {
                        count++;
                        if (count > left) bad(what, "PIC run longer than its scanline");
                        for (unsigned i = 0; i < count; i++, dest += 4) read(pk.mask, dest);
                    }
                    left -= count;
                }
            } else {
                bad(what, "PIC packet with an unknown compression type");
            }
        }
    DecodedImage img;
    img.width = static_cast<int>(w);
    img.height = static_cast<int>(h);
    img.channels = (all & 0x10) ? 4 : 3;
    if (img.channels == 4) {
        img.pixels.swap(rgba);
    } else {
        img.pixels.resize(static_cast<size_t>(w) * h * 3);
        for (size_t t = 0; t < static_cast<size_t>(w) * h; t++) std::memcpy(img.pixels.data() + 3 * t, rgba.data() + 4 * t, 3);
    }
    return img;
}

bool endsWith(const std::string& s, const char* suffix)
{
    const size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) {
        const char a = s[s.size() - n + i], b = suffix[i];
        if ((a >= 'A' && a <= 'Z' ? a + 32 : a) != b) return false;
    }
    return true;
}

} // namespace

DecodedImage decodeImage(const std::vector<unsigned char>& f, const std::string& what)
{
    static const unsigned char pngMagic[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (f.size() >= 8 && std::memcmp(f.data(), pngMagic, 8) == 0) return decodePng(f, what);
    if (f.size() >= 3 && f[0] == 0xFF && f[1] == 0xD8 && f[2] == 0xFF) return decodeJpeg(f, what);
    if (f.size() >= 6 && std::memcmp(f.data(), "GIF8", 4) == 0) return decodeGif(f, what);
    if (f.size() >= 4 && std::memcmp(f.data(), "8BPS", 4) == 0) return decodePsd(f, what);
    if (f.size() >= 4 && std::memcmp(f.data(), "\x53\x80\xF6\x34", 4) == 0) return decodePic(f, what);
    if ((f.size() >= 11 && std::memcmp(f.data(), "#?RADIANCE\n", 11) == 0) || (f.size() >= 7 && std::memcmp(f.data(), "#?RGBE\n", 7) == 0)) return decodeHdr(f, what);
    if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') return decodeBmp(f, what);
    if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) return decodePnm(f, what);
    if (endsWith(what, ".tga")) return decodeTga(f, what); // TGA has no magic number: by its name
    bad(what, "not a PNG, JPEG, GIF, BMP, TGA, PSD, Radiance HDR, Softimage PIC or binary PPM / PGM file");
}

} // namespace crt
