#include "host_threads.hpp"
#include "AccumulatedImage.hpp"

#include <algorithm>
#include <cmath>

#include "image_io.hpp"

void saveHdrImage(const Image3<float>& hdrImage, const std::string& fileName) {
  auto baseName = fileName.substr(0, fileName.find_last_of('.'));
  image_io::writeExr(baseName + ".exr", hdrImage.data.data(), hdrImage.cols, hdrImage.rows);
}

AccumulatedImage::AccumulatedImage(std::size_t w, std::size_t h) { hdrImage.create(h, w); }
AccumulatedImage::~AccumulatedImage() {}

const Image3<std::uint8_t>& AccumulatedImage::updateLdrImage(std::size_t step, float exposure, float gamma) {
  image.create(hdrImage.rows, hdrImage.cols);
  const float scale = 1.f / step;
  const float exposureScale = std::pow(2.f, exposure);
  const float invGamma = 1.f / gamma;
#pragma omp parallel for schedule(static) num_threads(hostLoopThreads())
  for (std::size_t r = 0; r < hdrImage.rows; ++r) {
    const float* in = hdrImage.ptr(r);
    std::uint8_t* out = image.ptr(r);
    for (std::size_t i = 0; i < 3 * hdrImage.cols; ++i) {
      // pow(x * 2^exposure, 1/gamma) * 255 with cv::Mat::convertTo's rounding and saturation
      float v = std::pow(in[i] * scale * exposureScale, invGamma) * 255.0f;
      out[i] = (std::uint8_t)std::min(255.0f, std::max(0.0f, std::nearbyint(v)));
    }
  }
  return image;
}

void AccumulatedImage::saveImages(const std::string& fileName, std::size_t step, float exposure, float gamma) {
  const auto& ldr = updateLdrImage(step, exposure, gamma);
  image_io::writeLdr(fileName, ldr.data.data(), ldr.cols, ldr.rows);   // cv::imwrite: the codec follows the extension
  // The accumulated image is divided by the number of steps so the integrand is divided by the total
  // sample count (each step added one per-pixel mean).
  Image3<float> scaled = hdrImage;
  const float s = 1.f / step;
  for (auto& v : scaled.data) v *= s;
  saveHdrImage(scaled, fileName);
}

void AccumulatedImage::accumulate(const std::vector<TraceRecord>& traces) {
#pragma omp parallel for schedule(static) num_threads(hostLoopThreads())
  for (std::size_t i = 0; i < traces.size(); ++i) {
    const auto& t = traces[i];
    const std::size_t c = t.u, r = t.v;
    if (c >= hdrImage.cols || r >= hdrImage.rows) continue;  // worklist padding
    const float scale = 1.f / t.sampleCount;
    float* px = hdrImage.ptr(r) + 3 * c;
    px[0] += t.b * scale;
    px[1] += t.g * scale;
    px[2] += t.r * scale;
  }
}

void AccumulatedImage::reset() { std::fill(hdrImage.data.begin(), hdrImage.data.end(), 0.f); }
