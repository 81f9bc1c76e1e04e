#include "image_io.hpp"

#include <zlib.h>

#include <cctype>
#include <cmath>

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
  // the IFD goes behind the pixels, on a WORD boundary as TIFF 6.0 requires: one pad byte where width x height is odd
  const std::uint32_t pad = (std::uint32_t)(rgb.size() & 1u), ifd = (std::uint32_t)(8 + rgb.size()) + pad;
  h.push_back('I'); h.push_back('I'); put16le(h, 42); put32le(h, ifd);
  f.write((const char*)h.data(), 8);
  f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
  if (pad) f.put('\0');
  std::vector<std::uint8_t> d;
  const std::uint32_t nEntries = 10, bitsAt = ifd + 2 + nEntries * 12 + 4;
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

// Baseline JPEG (JFIF, 8-bit, YCbCr 4:4:4, the Annex K tables scaled to quality 95 -- cv::imwrite's default quality --, the
// Annex K Huffman tables).  A small encoder so that --outfile x.jpg means what it means to cv::imwrite; not byte-compatible with
// libjpeg (no chroma subsampling, float DCT).
struct JpegBits {
  std::vector<std::uint8_t>& out;
  std::uint32_t acc = 0;
  int n = 0;
  void put(std::uint32_t code, int len) {
    acc = (acc << len) | (code & ((1u << len) - 1u));
    n += len;
    while (n >= 8) {
      const std::uint8_t b = (std::uint8_t)(acc >> (n - 8));
      out.push_back(b);
      if (b == 0xff) out.push_back(0);
      n -= 8;
    }
  }
  void flush() { if (n) put(0x7f, 8 - n); }
};

struct JpegHuff { std::uint16_t code[256]; std::uint8_t len[256]; };
JpegHuff jpegHuff(const std::uint8_t counts[16], const std::uint8_t* symbols) {
  JpegHuff h{};
  std::uint16_t code = 0;
  int k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < counts[l - 1]; ++i) { h.code[symbols[k]] = code++; h.len[symbols[k]] = (std::uint8_t)l; ++k; }
    code <<= 1;
  }
  return h;
}

void writeJpeg(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  static const std::uint8_t zig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  static const std::uint8_t qY[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                      18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
  static const std::uint8_t qC[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
  static const std::uint8_t dcLc[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, dcCc[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
  static const std::uint8_t dcSym[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
  static const std::uint8_t acLc[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
  static const std::uint8_t acLs[162] = {
      0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
      0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
      0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
      0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
      0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
      0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
  static const std::uint8_t acCc[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
  static const std::uint8_t acCs[162] = {
      0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
      0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
      0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
      0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
      0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
      0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
  if (width == 0 || height == 0 || width > 65535 || height > 65535) throw std::runtime_error("JPEG: image size out of range for '" + fileName + "'.");
  auto f = openForWriting(fileName);
  const int quality = 95, scale = 200 - 2 * quality;                    // libjpeg's quality scaling above 50
  std::uint8_t q[2][64];
  for (int i = 0; i < 64; ++i) {
    q[0][i] = (std::uint8_t)std::min(255, std::max(1, (qY[i] * scale + 50) / 100));
    q[1][i] = (std::uint8_t)std::min(255, std::max(1, (qC[i] * scale + 50) / 100));
  }
  std::vector<std::uint8_t> o;
  auto marker = [&](std::uint8_t m, std::size_t len) { o.push_back(0xff); o.push_back(m); o.push_back((std::uint8_t)((len + 2) >> 8)); o.push_back((std::uint8_t)(len + 2)); };
  o.push_back(0xff); o.push_back(0xd8);                                  // SOI
  marker(0xe0, 14);                                                      // APP0 JFIF 1.01, no thumbnail
  { const std::uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0}; o.insert(o.end(), jfif, jfif + 14); }
  for (int t = 0; t < 2; ++t) { marker(0xdb, 65); o.push_back((std::uint8_t)t); for (int i = 0; i < 64; ++i) o.push_back(q[t][zig[i]]); }   // DQT
  marker(0xc0, 15);                                                      // SOF0
  o.push_back(8); o.push_back((std::uint8_t)(height >> 8)); o.push_back((std::uint8_t)height); o.push_back((std::uint8_t)(width >> 8)); o.push_back((std::uint8_t)width);
  o.push_back(3);
  for (int c = 0; c < 3; ++c) { o.push_back((std::uint8_t)(c + 1)); o.push_back(0x11); o.push_back(c ? 1 : 0); }
  auto dht = [&](std::uint8_t id, const std::uint8_t* counts, const std::uint8_t* syms, std::size_t n) {
    marker(0xc4, 17 + n); o.push_back(id); o.insert(o.end(), counts, counts + 16); o.insert(o.end(), syms, syms + n);
  };
  dht(0x00, dcLc, dcSym, 12); dht(0x10, acLc, acLs, 162); dht(0x01, dcCc, dcSym, 12); dht(0x11, acCc, acCs, 162);
  marker(0xda, 10);                                                      // SOS
  o.push_back(3);
  for (int c = 0; c < 3; ++c) { o.push_back((std::uint8_t)(c + 1)); o.push_back(c ? 0x11 : 0x00); }
  o.push_back(0); o.push_back(63); o.push_back(0);
  const JpegHuff hdc[2] = {jpegHuff(dcLc, dcSym), jpegHuff(dcCc, dcSym)}, hac[2] = {jpegHuff(acLc, acLs), jpegHuff(acCc, acCs)};
  float cosT[8][8];
  for (int u = 0; u < 8; ++u) for (int x = 0; x < 8; ++x) cosT[u][x] = (u ? 0.5f : 0.35355339f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.f);
  JpegBits bits{o};
  int prevDc[3] = {0, 0, 0};
  for (std::size_t by = 0; by < height; by += 8) {
    for (std::size_t bx = 0; bx < width; bx += 8) {
      float blk[3][64];
      for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {           // edge blocks repeat the last row / column
        const std::size_t r = std::min(by + y, height - 1), c = std::min(bx + x, width - 1);
        const std::uint8_t* px = bgr8 + (r * width + c) * 3;
        const float B = px[0], G = px[1], R = px[2];
        blk[0][y * 8 + x] = 0.299f * R + 0.587f * G + 0.114f * B - 128.f;
        blk[1][y * 8 + x] = -0.168736f * R - 0.331264f * G + 0.5f * B;
        blk[2][y * 8 + x] = 0.5f * R - 0.418688f * G - 0.081312f * B;
      }
      for (int c = 0; c < 3; ++c) {
        float tmp[64], dct[64];
        for (int y = 0; y < 8; ++y) for (int u = 0; u < 8; ++u) { float a = 0; for (int x = 0; x < 8; ++x) a += blk[c][y * 8 + x] * cosT[u][x]; tmp[y * 8 + u] = a; }
        for (int v = 0; v < 8; ++v) for (int u = 0; u < 8; ++u) { float a = 0; for (int y = 0; y < 8; ++y) a += tmp[y * 8 + u] * cosT[v][y]; dct[v * 8 + u] = a; }
        int zz[64];
        const int t = c ? 1 : 0;
        for (int i = 0; i < 64; ++i) zz[i] = (int)std::lround(dct[zig[i]] / (float)q[t][zig[i]]);
        auto magnitude = [](int v, int& nbits, std::uint32_t& code) {
          int a = v < 0 ? -v : v;
          nbits = 0;
          while (a) { ++nbits; a >>= 1; }
          code = (std::uint32_t)(v < 0 ? v + (1 << nbits) - 1 : v);
        };
        int nb; std::uint32_t code;
        magnitude(zz[0] - prevDc[c], nb, code);
        prevDc[c] = zz[0];
        bits.put(hdc[t].code[nb], hdc[t].len[nb]);
        if (nb) bits.put(code, nb);
        int run = 0;
        for (int i = 1; i < 64; ++i) {
          if (zz[i] == 0) { ++run; continue; }
          while (run > 15) { bits.put(hac[t].code[0xf0], hac[t].len[0xf0]); run -= 16; }
          magnitude(zz[i], nb, code);
          const int sym = (run << 4) | nb;
          bits.put(hac[t].code[sym], hac[t].len[sym]);
          bits.put(code, nb);
          run = 0;
        }
        if (run) bits.put(hac[t].code[0x00], hac[t].len[0x00]);           // EOB
      }
    }
  }
  bits.flush();
  o.push_back(0xff); o.push_back(0xd9);                                   // EOI
  f.write((const char*)o.data(), (std::streamsize)o.size());
}
}  // namespace

bool ldrWriterFor(const std::string& fileName) {
  const std::string e = lowerExtension(fileName);
  return e == ".png" || e == ".bmp" || e == ".ppm" || e == ".pnm" || e == ".tif" || e == ".tiff" || e == ".jpg" || e == ".jpeg" || e == ".jpe";
}

void writeLdr(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height) {
  const std::string e = lowerExtension(fileName);
  if (e == ".png") writePng(fileName, bgr8, width, height);
  else if (e == ".bmp") writeBmp(fileName, bgr8, width, height);
  else if (e == ".ppm" || e == ".pnm") writePpm(fileName, bgr8, width, height);
  else if (e == ".tif" || e == ".tiff") writeTiff(fileName, bgr8, width, height);
  else if (e == ".jpg" || e == ".jpeg" || e == ".jpe") writeJpeg(fileName, bgr8, width, height);
  else throw std::runtime_error("could not find a writer for the specified extension of '" + fileName + "' (built in: .png .jpg .bmp .ppm .pnm .tif .tiff)");
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
