#include "image_io.hpp"

#include <zlib.h>

#include <cctype>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace image_io {
namespace {

std::uint32_t crc32(const std::uint8_t* p, std::size_t n, std::uint32_t crc = 0) {
  static std::uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (std::uint32_t i = 0; i < 256; ++i) {
      std::uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  crc = ~crc;
  for (std::size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}

void put32be(std::vector<std::uint8_t>& v, std::uint32_t x) {
  v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

void chunk(std::ofstream& f, const char tag[4], const std::vector<std::uint8_t>& data) {
  std::vector<std::uint8_t> buf;
  put32be(buf, (std::uint32_t)data.size());
  f.write((const char*)buf.data(), 4);
  std::vector<std::uint8_t> body(tag, tag + 4);
  body.insert(body.end(), data.begin(), data.end());
  f.write((const char*)body.data(), body.size());
  buf.clear();
  put32be(buf, crc32(body.data(), body.size()));
  f.write((const char*)buf.data(), 4);
}

template <class T>
void put(std::vector<std::uint8_t>& v, T x) {
  const std::uint8_t* p = reinterpret_cast<const std::uint8_t*>(&x);
  v.insert(v.end(), p, p + sizeof(T));
}
void putStr(std::vector<std::uint8_t>& v, const char* s) { v.insert(v.end(), s, s + std::strlen(s) + 1); }

void attr(std::vector<std::uint8_t>& v, const char* name, const char* type, const std::vector<std::uint8_t>& val) {
  putStr(v, name);
  putStr(v, type);
  put<std::int32_t>(v, (std::int32_t)val.size());
  v.insert(v.end(), val.begin(), val.end());
}

}  // namespace

namespace {
std::string lowerExtension(const std::string& fileName) {
  const auto dot = fileName.find_last_of('.');
  std::string ext = dot == std::string::npos ? std::string() : fileName.substr(dot);
  for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
  return ext;
}
std::ofstream openForWriting(const std::string& fileName) {
  std::ofstream f(fileName, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open '" + fileName + "' for writing.");
  return f;
}
void put16le(std::vector<std::uint8_t>& v, std::uint32_t x) { v.push_back(x & 0xff); v.push_back((x >> 8) & 0xff); }
void put32le(std::vector<std::uint8_t>& v, std::uint32_t x) { put16le(v, x & 0xffff); put16le(v, x >> 16); }

// 24-bit BMP, bottom-up rows padded to four bytes; pixels are B,G,R in the file as in memory
void writeBmp(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  auto f = openForWriting(fileName);
  const std::size_t stride = (3 * width + 3) & ~std::size_t(3);
  std::vector<std::uint8_t> h;
  h.push_back('B'); h.push_back('M');
  put32le(h, (std::uint32_t)(54 + stride * height)); put32le(h, 0); put32le(h, 54);
  put32le(h, 40); put32le(h, (std::uint32_t)width); put32le(h, (std::uint32_t)height); put16le(h, 1); put16le(h, 24);
  put32le(h, 0); put32le(h, (std::uint32_t)(stride * height)); put32le(h, 2835); put32le(h, 2835); put32le(h, 0); put32le(h, 0);
  f.write((const char*)h.data(), (std::streamsize)h.size());
  std::vector<std::uint8_t> row(stride, 0);
  for (std::size_t r = height; r-- > 0;) {
    std::copy(bgr8 + r * width * 3, bgr8 + (r + 1) * width * 3, row.begin());
    f.write((const char*)row.data(), (std::streamsize)stride);
  }
}

std::vector<std::uint8_t> toRgb(const std::uint8_t* bgr8, std::size_t pixels) {
  std::vector<std::uint8_t> rgb(pixels * 3);
  for (std::size_t i = 0; i < pixels; ++i) { rgb[3 * i] = bgr8[3 * i + 2]; rgb[3 * i + 1] = bgr8[3 * i + 1]; rgb[3 * i + 2] = bgr8[3 * i]; }
  return rgb;
}

void writePpm(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  auto f = openForWriting(fileName);
  f << "P6\n" << width << " " << height << "\n255\n";
  const auto rgb = toRgb(bgr8, width * height);
  f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
}

// Baseline TIFF, little-endian, one strip of uncompressed 8-bit RGB
void writeTiff(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  auto f = openForWriting(fileName);
  const auto rgb = toRgb(bgr8, width * height);
  std::vector<std::uint8_t> h;
  h.push_back('I'); h.push_back('I'); put16le(h, 42); put32le(h, (std::uint32_t)(8 + rgb.size()));   // IFD behind the pixels
  f.write((const char*)h.data(), 8);
  f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
  std::vector<std::uint8_t> d;
  const std::uint32_t ifd = (std::uint32_t)(8 + rgb.size()), nEntries = 10, bitsAt = ifd + 2 + nEntries * 12 + 4;
  auto entry = [&](std::uint32_t tag, std::uint32_t type, std::uint32_t count, std::uint32_t value) {
    put16le(d, tag); put16le(d, type); put32le(d, count);
    if (type == 3 && count == 1) { put16le(d, value); put16le(d, 0); } else put32le(d, value);
  };
  put16le(d, nEntries);
  entry(256, 4, 1, (std::uint32_t)width);        // ImageWidth
  entry(257, 4, 1, (std::uint32_t)height);       // ImageLength
  entry(258, 3, 3, bitsAt);                      // BitsPerSample -> 8, 8, 8
  entry(259, 3, 1, 1);                           // Compression: none
  entry(262, 3, 1, 2);                           // PhotometricInterpretation: RGB
  entry(273, 4, 1, 8);                           // StripOffsets
  entry(277, 3, 1, 3);                           // SamplesPerPixel
  entry(278, 4, 1, (std::uint32_t)height);       // RowsPerStrip
  entry(279, 4, 1, (std::uint32_t)rgb.size());   // StripByteCounts
  entry(284, 3, 1, 1);                           // PlanarConfiguration: chunky
  put32le(d, 0);                                 // no further IFD
  put16le(d, 8); put16le(d, 8); put16le(d, 8);
  f.write((const char*)d.data(), (std::streamsize)d.size());
}
}  // namespace

bool ldrWriterFor(const std::string& fileName) {
  const std::string e = lowerExtension(fileName);
  return e == ".png" || e == ".bmp" || e == ".ppm" || e == ".pnm" || e == ".tif" || e == ".tiff";
}

void writeLdr(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  const std::string e = lowerExtension(fileName);
  if (e == ".png") writePng(fileName, bgr8, width, height);
  else if (e == ".bmp") writeBmp(fileName, bgr8, width, height);
  else if (e == ".ppm" || e == ".pnm") writePpm(fileName, bgr8, width, height);
  else if (e == ".tif" || e == ".tiff") writeTiff(fileName, bgr8, width, height);
  else throw std::runtime_error("could not find a writer for the specified extension of '" + fileName + "' (built in: .png .bmp .ppm .pnm .tif .tiff)");
}

void writePng(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  std::ofstream f(fileName, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open '" + fileName + "' for writing.");
  const std::uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  f.write((const char*)sig, 8);
  std::vector<std::uint8_t> ihdr;
  put32be(ihdr, (std::uint32_t)width);
  put32be(ihdr, (std::uint32_t)height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit RGB
  chunk(f, "IHDR", ihdr);
  // raw scanlines (filter 0), B,G,R -> R,G,B
  const std::size_t stride = 1 + 3 * width;
  std::vector<std::uint8_t> raw(stride * height);
  for (std::size_t r = 0; r < height; ++r) {
    std::uint8_t* row = &raw[r * stride];
    row[0] = 0;
    for (std::size_t c = 0; c < width; ++c) {
      const std::uint8_t* px = bgr8 + (r * width + c) * 3;
      row[1 + 3 * c + 0] = px[2];
      row[1 + 3 * c + 1] = px[1];
      row[1 + 3 * c + 2] = px[0];
    }
  }
  // zlib stream (deflate level 3; cv::imwrite's PNG default is a fast level too)
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<std::uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 3) != Z_OK) throw std::runtime_error("PNG: deflate failed for '" + fileName + "'.");
  z.resize(zlen);
  chunk(f, "IDAT", z);
  chunk(f, "IEND", {});
}

void writeExr(const std::string& fileName, const float* bgr, std::size_t width, std::size_t height) {
  std::vector<std::uint8_t> h;
  put<std::uint32_t>(h, 20000630u);  // magic
  put<std::uint32_t>(h, 2u);         // version 2, scanline, no flags
  std::vector<std::uint8_t> ch;
  for (const char* name : {"B", "G", "R"}) {
    putStr(ch, name);
    put<std::int32_t>(ch, 2);  // FLOAT
    ch.push_back(0); ch.push_back(0); ch.push_back(0); ch.push_back(0);  // pLinear + reserved
    put<std::int32_t>(ch, 1); put<std::int32_t>(ch, 1);                  // sampling
  }
  ch.push_back(0);
  attr(h, "channels", "chlist", ch);
  attr(h, "compression", "compression", {0});
  std::vector<std::uint8_t> box;
  put<std::int32_t>(box, 0); put<std::int32_t>(box, 0);
  put<std::int32_t>(box, (std::int32_t)width - 1); put<std::int32_t>(box, (std::int32_t)height - 1);
  attr(h, "dataWindow", "box2i", box);
  attr(h, "displayWindow", "box2i", box);
  attr(h, "lineOrder", "lineOrder", {0});
  std::vector<std::uint8_t> one; put<float>(one, 1.f);
  attr(h, "pixelAspectRatio", "float", one);
  std::vector<std::uint8_t> v2; put<float>(v2, 0.f); put<float>(v2, 0.f);
  attr(h, "screenWindowCenter", "v2f", v2);
  attr(h, "screenWindowWidth", "float", one);
  h.push_back(0);
  const std::size_t rowBytes = 3 * width * sizeof(float);
  const std::uint64_t tableStart = h.size();
  const std::uint64_t dataStart = tableStart + 8 * height;
  for (std::size_t y = 0; y < height; ++y) put<std::uint64_t>(h, dataStart + y * (8 + rowBytes));
  std::ofstream f(fileName, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open '" + fileName + "' for writing.");
  f.write((const char*)h.data(), h.size());
  std::vector<float> row(3 * width);
  for (std::size_t y = 0; y < height; ++y) {
    std::int32_t yy = (std::int32_t)y, sz = (std::int32_t)rowBytes;
    f.write((const char*)&yy, 4);
    f.write((const char*)&sz, 4);
    for (int c = 0; c < 3; ++c)  // channels in alphabetical order B, G, R == the film's B,G,R order
      for (std::size_t x = 0; x < width; ++x) row[c * width + x] = bgr[(y * width + x) * 3 + c];
    f.write((const char*)row.data(), rowBytes);
  }
}

bool readExr(const std::string& fileName, std::vector<float>& bgr, std::size_t& width, std::size_t& height) {
  std::ifstream f(fileName, std::ios::binary);
  if (!f) return false;
  std::vector<std::uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (d.size() < 8 || *reinterpret_cast<std::uint32_t*>(d.data()) != 20000630u) return false;
  std::size_t p = 8;
  width = height = 0;
  while (p < d.size() && d[p] != 0) {
    std::string name((const char*)&d[p]); p += name.size() + 1;
    std::string type((const char*)&d[p]); p += type.size() + 1;
    std::int32_t sz; std::memcpy(&sz, &d[p], 4); p += 4;
    if (name == "dataWindow") {
      std::int32_t b[4]; std::memcpy(b, &d[p], 16);
      width = b[2] - b[0] + 1; height = b[3] - b[1] + 1;
    }
    p += sz;
  }
  p += 1;
  if (!width || !height) return false;
  bgr.assign(width * height * 3, 0.f);
  const std::size_t rowBytes = 3 * width * 4;
  for (std::size_t y = 0; y < height; ++y) {
    std::uint64_t off; std::memcpy(&off, &d[p + 8 * y], 8);
    if (off + 8 + rowBytes > d.size()) return false;
    const float* row = reinterpret_cast<const float*>(&d[off + 8]);
    for (int c = 0; c < 3; ++c)
      for (std::size_t x = 0; x < width; ++x) bgr[(y * width + x) * 3 + c] = row[c * width + x];
  }
  return true;
}

}  // namespace image_io
