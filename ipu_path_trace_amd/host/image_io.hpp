// Dependency-free image writers standing in for cv::imwrite (reference src/AccumulatedImage.cpp:16-19,49,55):
// 8-bit BGR -> PNG (zlib deflate), BMP, PPM, TIFF by extension, and float BGR -> OpenEXR (uncompressed scanlines, FLOAT channels).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace image_io {

/// bgr8: height x width x 3 bytes in B,G,R order (OpenCV convention); written as an RGB PNG.
void writePng(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height);
/// The low-dynamic-range image as cv::imwrite(fileName, image) would save it (reference src/AccumulatedImage.cpp:49): the codec
/// is chosen by the file name's extension, case-insensitive -- .png, .jpg / .jpeg / .jpe (baseline, quality 95, 4:4:4), .bmp,
/// .ppm / .pnm (binary P6), .tif / .tiff (uncompressed RGB).  Any other extension throws, as cv::imwrite does for one it has no
/// writer for.
void writeLdr(const std::string& fileName, const std::uint8_t* bgr8, std::size_t width, std::size_t height);
/// True if writeLdr has a writer for this file name (checked once at start-up, before anything is rendered).
bool ldrWriterFor(const std::string& fileName);
/// bgr: height x width x 3 floats in B,G,R order; written as a 3-channel (B,G,R) float EXR.
void writeExr(const std::string& fileName, const float* bgr, std::size_t width, std::size_t height);
/// Reader for the EXR subset writeExr produces (used by tests).
bool readExr(const std::string& fileName, std::vector<float>& bgr, std::size_t& width, std::size_t& height);

}  // namespace image_io
