// png.cpp -- see png.hpp. Own chunk walk, APNG frame assembly and scanline unfiltering; zlib only inflates.
#include "png.hpp"

#include <zlib.h>

#include <cstring>
#include <string>

#include "vox.hpp"  // ParseError

namespace dust::png {
namespace {

[[noreturn]] void bad(const std::string& what, bool unsupported = false) {
  dust::vox::ParseError e;
  e.what = "png: " + what;
  e.unsupported = unsupported;
  throw e;
}
uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

std::vector<uint8_t> inflate_all(const std::vector<uint8_t>& z, size_t expect) {
  std::vector<uint8_t> out(expect);
  uLongf n = uLongf(expect);
  const int rc = uncompress(out.data(), &n, z.data(), uLong(z.size()));
  if (rc != Z_OK || n != expect) bad("corrupt image data stream");
  return out;
}

// PNG filter types 0-4 (None, Sub, Up, Average, Paeth) undone in place, one scanline at a time
void unfilter(std::vector<uint8_t>& raw, uint32_t height, size_t stride, uint32_t bpp, uint8_t* dst) {
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t type = raw[size_t(y) * (stride + 1)];
    const uint8_t* src = raw.data() + size_t(y) * (stride + 1) + 1;
    uint8_t* cur = dst + size_t(y) * stride;
    const uint8_t* up = y ? dst + size_t(y - 1) * stride : zero.data();
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
      int pred = 0;
      switch (type) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: bad("unknown scanline filter");
      }
      cur[i] = uint8_t(src[i] + pred);
    }
  }
}

}  // namespace

ImageArray load(const uint8_t* bytes, size_t n) {
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (n < 8 || std::memcmp(bytes, kSig, 8) != 0) bad("not a PNG file");
  ImageArray img;
  uint32_t bit_depth = 0, color_type = 0, interlace = 0, num_frames = 1;
  bool have_ihdr = false, animated = false;
  // frame 0 is the IDAT image; every fcTL after it opens another frame made of fdAT chunks
  std::vector<std::vector<uint8_t>> streams(1);
  bool seen_idat = false, frame_covers = true;
  size_t pos = 8;
  for (;;) {
    if (pos + 12 > n) bad("truncated chunk");
    const uint32_t len = be32(bytes + pos);
    const uint8_t* type = bytes + pos + 4;
    const uint8_t* data = bytes + pos + 8;
    if (size_t(len) > n - pos - 12) bad("truncated chunk");
    if (uint32_t(crc32(crc32(0L, Z_NULL, 0), type, uInt(len) + 4u)) != be32(data + len)) bad("chunk checksum mismatch");
    pos += size_t(len) + 12;
    auto is = [&](const char* t) { return std::memcmp(type, t, 4) == 0; };
    if (is("IHDR")) {
      if (len != 13) bad("bad IHDR");
      img.width = be32(data); img.height = be32(data + 4);
      bit_depth = data[8]; color_type = data[9]; interlace = data[12];
      if (img.width == 0 || img.height == 0 || img.width > 16384 || img.height > 16384) bad("bad image size");
      have_ihdr = true;
    } else if (is("acTL")) {
      if (len != 8) bad("bad acTL");
      num_frames = be32(data);
      animated = true;
    } else if (is("fcTL")) {
      if (len != 26) bad("bad fcTL");
      if (!have_ihdr) bad("fcTL before IHDR");
      if (be32(data + 4) != img.width || be32(data + 8) != img.height || be32(data + 12) != 0 || be32(data + 16) != 0) frame_covers = false;
      if (seen_idat) streams.emplace_back();
    } else if (is("IDAT")) {
      if (!have_ihdr) bad("IDAT before IHDR");
      seen_idat = true;
      streams[0].insert(streams[0].end(), data, data + len);
    } else if (is("fdAT")) {
      if (len < 4 || !seen_idat || streams.size() < 2) bad("fdAT out of place");
      streams.back().insert(streams.back().end(), data + 4, data + len);
    } else if (is("IEND")) {
      break;
    }
  }
  if (!have_ihdr || !seen_idat) bad("no image data");
  if (color_type == 3) bad("indexed colour", true);                   // png.rs:107,122: UnsupportedPngColorTypeError
  if (bit_depth != 8 && bit_depth != 16) bad("bit depth below 8", true);  // png.rs:176-183
  if (interlace != 0) bad("interlaced image", true);
  uint32_t src_channels = 0;
  switch (color_type) {
    case 0: src_channels = 1; img.channels = 1; break;
    case 4: src_channels = 2; img.channels = 2; break;
    case 2: src_channels = 3; img.channels = 4; break;  // Rgb is stored as Rgba, the fourth byte zero (png.rs:150-162)
    case 6: src_channels = 4; img.channels = 4; break;
    default: bad("unknown colour type");
  }
  if (!animated) num_frames = 1;
  if (num_frames == 0 || num_frames > streams.size()) bad("fewer frames than the animation control announces");
  if (animated && !frame_covers) bad("animation frame smaller than the image", true);
  img.layers = num_frames;
  img.bytes_per_channel = bit_depth / 8;
  const uint32_t bpp = src_channels * img.bytes_per_channel;
  const size_t stride = size_t(img.width) * bpp;
  const size_t dst_px = size_t(img.channels) * img.bytes_per_channel;
  // A file is untrusted input: IHDR and acTL alone must not decide how much memory is taken. Deflate expands at most ~1032:1,
  // so a frame whose compressed stream is too short to inflate to (stride + 1) * height bytes is rejected before anything of
  // that size is allocated.
  for (uint32_t f = 0; f < num_frames; ++f)
    if ((stride + 1) * img.height > streams[f].size() * 1032 + 1024) bad("image data too short for the announced size");
  img.texels.assign(size_t(num_frames) * img.height * img.width * dst_px, 0);
  std::vector<uint8_t> frame(stride * img.height);
  for (uint32_t f = 0; f < num_frames; ++f) {
    std::vector<uint8_t> raw = inflate_all(streams[f], (stride + 1) * img.height);
    unfilter(raw, img.height, stride, bpp, frame.data());
    uint8_t* dst = img.texels.data() + size_t(f) * img.height * img.width * dst_px;
    if (dst_px == bpp) {
      std::memcpy(dst, frame.data(), frame.size());
    } else {  // Rgb -> Rgba, padding zero
      const size_t px = size_t(img.width) * img.height;
      for (size_t i = 0; i < px; ++i) std::memcpy(dst + i * dst_px, frame.data() + i * bpp, bpp);
    }
  }
  return img;
}

}  // namespace dust::png
