// Bitmap texture files.  The reference loads whatever its vendored stb_image decodes (R/CRTTextureBitmap.cpp:10,
// stbi_load(path, &w, &h, &channels, 0)); stb_image is third party and is not used here.  These decoders cover every format it
// reads -- PNG (all colour types and bit depths, Adam7 interlace, tRNS), JPEG (baseline and progressive, jpeg_decode.cpp), GIF
// (first image), Photoshop PSD (merged RGB image), Radiance HDR (through that decoder's tone curve), Softimage PIC, BMP (palettes,
// 16 / 24 / 32 bit with channel masks, uncompressed), TGA (true colour, grey, colour-mapped; raw and run-length coded), binary
// PPM / PGM -- and yield what stbi_load yields for them, that decoder's oddities included where a texel depends on them: rows top
// to bottom, `channels` bytes per texel in the file's own channel count (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA; 16-bit PNG / PSD
// samples reduced to their high byte).  Known answers: tests/golden/bitmap_known_answers.json, produced by the reference's own
// CRTTextureBitmap over the same 73 files (oracle/make_golden.py).
#pragma once

#include <string>
#include <vector>

namespace crt {

struct DecodedImage {
    int width = 0, height = 0, channels = 0;
    std::vector<unsigned char> pixels; // height x width x channels
};

// `what` names the file in error messages.  Throws std::runtime_error on anything that is not a well-formed file of a supported
// kind (never reads outside `file`, never allocates more than the header's width x height x 4 after checking it against 2^28 texels).
DecodedImage decodeImage(const std::vector<unsigned char>& file, const std::string& what);

// jpeg_decode.cpp: baseline / progressive Huffman JPEG with the reference decoder's numerical conventions (see the file's header)
DecodedImage decodeJpeg(const std::vector<unsigned char>& file, const std::string& what);

// RFC 1950 / 1951 (zlib stream around deflate), exposed for tests; throws std::runtime_error
std::vector<unsigned char> zlibInflate(const unsigned char* data, size_t size, size_t expectedSize, const std::string& what);

} // namespace crt
