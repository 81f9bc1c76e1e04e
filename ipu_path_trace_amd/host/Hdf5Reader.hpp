// Dependency-free reader for the subset of HDF5 that Keras/h5py write for a saved model (libhdf5 is not available on
// this platform; the reference links libhdf5_cpp: src/keras/Hdf5Model.hpp:3, CMakeLists).
//
// Supported (HDF5 File Format Specification, version 0/1 superblock era -- what h5py's default libver produces):
//   superblock v0/v1 (optionally behind a user block), version-1 object headers with continuation blocks,
//   old-style groups (symbol-table message -> v1 B-tree -> SNOD nodes -> local heap), dataspace v1/v2,
//   IEEE float datatypes, contiguous, compact and CHUNKED data layouts (v1-v3; chunks indexed by the version-1 chunk
//   B-tree, edge chunks clipped, missing chunks read as zeros), the filters h5py applies for `compression="gzip"`,
//   `shuffle=True` and `fletcher32=True` (deflate through zlib, byte shuffle, checksum stripped -- a per-chunk filter
//   mask is honoured), attributes v1-v3 holding fixed-length or variable-length (global heap) strings.
// Anything else (new-style groups, v2 object headers, version-4 layouts with their v2 B-tree / extensible-array chunk
// indices, other filters) raises std::runtime_error naming exactly what was found -- e.g. "filter 32015 (zstd)".
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace h5 {

struct Dataset {
  std::vector<std::size_t> shape;
  std::size_t elementSize = 0;   // bytes per element (2 = float16, 4 = float32)
  bool isFloat = false;
  std::vector<std::uint8_t> bytes;
};

class File {
public:
  explicit File(const std::string& fileName);

  /// String attribute of the root group (fixed-length or variable-length); throws if absent.
  std::string readStringAttribute(const std::string& name) const { return readStringAttribute("/", name); }
  std::string readStringAttribute(const std::string& objectPath, const std::string& name) const;
  bool hasAttribute(const std::string& objectPath, const std::string& name) const;
  /// Names of the links of a group, in B-tree (name-sorted) order.
  std::vector<std::string> listGroup(const std::string& groupPath) const;
  bool exists(const std::string& path) const;
  Dataset openDataSet(const std::string& path) const;

private:
  struct Message { std::uint16_t type; std::uint8_t flags; std::size_t offset, size; };
  struct Attribute { std::vector<std::uint8_t> datatype, dataspace, data; };

  std::uint64_t u(std::size_t off, int bytes) const;
  std::uint64_t addr(std::size_t off) const { return u(off, sizeOffsets); }
  void need(std::size_t off, std::size_t n, const char* what) const;
  std::vector<Message> objectMessages(std::uint64_t headerAddress) const;
  std::map<std::string, std::uint64_t> groupLinks(std::uint64_t headerAddress) const;
  // expectLevel: -1 at the root (any level up to 16), below it the parent's level - 1; visited: nodes seen so far, bounded by the file size
  void walkBtree(std::uint64_t node, std::uint64_t heapData, std::map<std::string, std::uint64_t>& out, int expectLevel, std::size_t& visited) const;
  void checkBtreeNode(int level, int expectLevel, std::size_t& visited, const char* what) const;
  std::uint64_t resolve(const std::string& path) const;
  std::map<std::string, Attribute> attributes(std::uint64_t headerAddress) const;
  std::string globalHeapObject(std::uint64_t collection, std::uint32_t index) const;
  struct Filter { std::uint16_t id; std::vector<std::uint32_t> values; };
  struct ChunkedLayout { std::uint64_t btree = 0; std::vector<std::size_t> dims; };   // dims: chunk shape, then the element size
  void readChunks(const std::string& path, const ChunkedLayout& layout, const std::vector<Filter>& filters, Dataset& out) const;
  void walkChunkBtree(std::uint64_t node, std::size_t nDims, int expectLevel, std::size_t& visited,
                      std::vector<std::pair<std::vector<std::uint64_t>, std::pair<std::uint64_t, std::pair<std::uint32_t, std::uint32_t>>>>& chunks) const;

  std::vector<std::uint8_t> d;
  std::uint64_t base = 0;
  int sizeOffsets = 8, sizeLengths = 8;
  std::uint64_t rootHeader = 0;
};

}  // namespace h5
