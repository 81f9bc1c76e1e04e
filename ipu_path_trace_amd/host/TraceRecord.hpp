// Work item / accumulator shared by host and device: reference src/codelets/TraceRecord.hpp:7-19.
// Same 20-byte layout as pt_trace_record in include/ptmi.h (the wire format of the boundary).
#pragma once
#include <cstdint>

struct TraceRecord {
  std::uint16_t u, v;  // Image pixel coord.
  float r, g, b;       // Accumulated RGB contribution.
  std::uint16_t sampleCount;
  std::uint16_t pathLength;

  TraceRecord(std::uint16_t pixelU, std::uint16_t pixelV)
      : u(pixelU), v(pixelV), r(0.f), g(0.f), b(0.f), sampleCount(0), pathLength(0) {}
  TraceRecord() : TraceRecord(0, 0) {}
};
static_assert(sizeof(TraceRecord) == 20, "TraceRecord is the 20-byte wire format");
