// Dependency-free image writers standing in for cv::imwrite (reference src/AccumulatedImage.cpp:16-19,49,55):
// 8-bit BGR -> PNG (stored deflate blocks) and float BGR -> OpenEXR (uncompressed scanlines, FLOAT channels).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace image_io {

/// bgr8: height x width x 3 bytes in B,G,R order (OpenCV convention); written as an RGB PNG.
void writePng(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height);
/// bgr: height x width x 3 floats in B,G,R order; written as a 3-channel (B,G,R) float EXR.
void writeExr(const std::string& fileName, const float* bgr, std::size_t width, std::size_t height);
/// Reader for the EXR subset writeExr produces (used by tests).
bool readExr(const std::string& fileName, std::vector<float>& bgr, std::size_t& width, std::size_t& height);

}  // namespace image_io
