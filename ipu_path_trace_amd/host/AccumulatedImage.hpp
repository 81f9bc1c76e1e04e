// HDR film: reference src/AccumulatedImage.hpp:13-33.  cv::Mat CV_32FC3 becomes a plain float buffer
// (height x width x 3, B,G,R order as OpenCV stores it); OpenCV is not available on this platform.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "TraceRecord.hpp"

/// Minimal row-major 3-channel image (the role cv::Mat plays in the reference).
template <class T>
struct Image3 {
  std::size_t rows = 0, cols = 0;
  std::vector<T> data;
  void create(std::size_t r, std::size_t c) { rows = r; cols = c; data.assign(r * c * 3, T(0)); }
  T* ptr(std::size_t r) { return data.data() + r * cols * 3; }
  const T* ptr(std::size_t r) const { return data.data() + r * cols * 3; }
};

void saveHdrImage(const Image3<float>& hdrImage, const std::string& fileName);

struct AccumulatedImage {
  AccumulatedImage(std::size_t w, std::size_t h);
  virtual ~AccumulatedImage();

  /// Tone map the HDR image and return a reference to the result (AccumulatedImage.cpp:23-46).
  const Image3<std::uint8_t>& updateLdrImage(std::size_t step, float exposure, float gamma);
  /// Write <fileName> (PNG) and <basename>.exr (AccumulatedImage.cpp:48-56).
  void saveImages(const std::string& fileName, std::size_t step, float exposure, float gamma);
  /// Accumulate the trace results converting from RGB to BGR in the process (AccumulatedImage.cpp:59-74).
  void accumulate(const std::vector<TraceRecord>& traces);
  void reset();
  /// Return a copy of the raw HDR image.
  Image3<float> getHdrImage() const { return hdrImage; }

private:
  Image3<float> hdrImage;
  Image3<std::uint8_t> image;
};
