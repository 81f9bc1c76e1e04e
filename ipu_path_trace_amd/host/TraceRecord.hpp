// Work item / accumulator exchanged between host and device.
//
// Field order, types and the 20-byte size are fixed by the reference (src/codelets/TraceRecord.hpp:7-19): the record
// is the wire format of the `trace_buffer` stream, and pt_trace_record in include/ptmi.h is declared to match it.
#pragma once
#include <cstddef>
#include <cstdint>

struct TraceRecord {
  std::uint16_t u = 0;            // pixel column
  std::uint16_t v = 0;            // pixel row
  float r = 0.f, g = 0.f, b = 0.f;  // radiance summed over `sampleCount` samples
  std::uint16_t sampleCount = 0;
  std::uint16_t pathLength = 0;   // summed contribution-stack sizes ("rays")

  TraceRecord() = default;
  /// A fresh work item for pixel (pixelU, pixelV): coordinates set, accumulators zero.
  TraceRecord(std::uint16_t pixelU, std::uint16_t pixelV) : u(pixelU), v(pixelV) {}

  void clearAccumulators() { r = g = b = 0.f; sampleCount = 0; pathLength = 0; }
};

static_assert(sizeof(TraceRecord) == 20 && alignof(TraceRecord) == 4, "TraceRecord is the 20-byte wire format");
static_assert(offsetof(TraceRecord, r) == 4 && offsetof(TraceRecord, sampleCount) == 16 &&
              offsetof(TraceRecord, pathLength) == 18, "TraceRecord field offsets");
