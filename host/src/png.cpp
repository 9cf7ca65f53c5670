// png.cpp -- `Canvas::save_to_png` (src/graphics.rs:77-81): the RGBA8 image of `to_image` as a
// non-interlaced 8-bit RGBA PNG.  The reference writes through the `image`/`png` crates; the
// compressed bytes differ between encoders, the decoded pixels are the same.
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "../include/rtk_host.hpp"

namespace rtkhost {
namespace {

void put_be32(std::vector<uint8_t>& out, uint32_t v) {
  out.push_back((uint8_t)(v >> 24));
  out.push_back((uint8_t)(v >> 16));
  out.push_back((uint8_t)(v >> 8));
  out.push_back((uint8_t)v);
}

void chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
  put_be32(out, (uint32_t)n);
  const size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  if (n) out.insert(out.end(), data, data + n);
  put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}

}  // namespace

void Canvas::save_to_png(const std::string& path) const {
  if (width_ == 0 || height_ == 0) throw Error(RTK_ERR_INVALID, "save_to_png: empty canvas");
  // scanlines with filter type 0 in front of each
  const size_t stride = (size_t)width_ * 4;
  std::vector<uint8_t> raw((stride + 1) * height_);
  for (uint32_t y = 0; y < height_; ++y) {
    raw[(stride + 1) * y] = 0;
    std::memcpy(&raw[(stride + 1) * y + 1], &rgba8_[stride * y], stride);
  }
  uLongf bound = compressBound((uLong)raw.size());
  std::vector<uint8_t> packed(bound);
  if (compress2(packed.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK)
    throw Error(RTK_ERR_INVALID, "save_to_png: deflate failed");

  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, width_);
  put_be32(ihdr, height_);
  ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});  // 8 bits per channel, RGBA, deflate, adaptive filtering, no interlace
  chunk(out, "IHDR", ihdr.data(), ihdr.size());
  chunk(out, "IDAT", packed.data(), bound);
  chunk(out, "IEND", nullptr, 0);

  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw Error(RTK_ERR_INVALID, "save_to_png: cannot open " + path);
  const size_t written = std::fwrite(out.data(), 1, out.size(), f);
  if (std::fclose(f) != 0 || written != out.size()) throw Error(RTK_ERR_INVALID, "save_to_png: short write to " + path);
}

}  // namespace rtkhost
