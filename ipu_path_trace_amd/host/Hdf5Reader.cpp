#include "Hdf5Reader.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace h5 {
namespace {
const unsigned char kSignature[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
constexpr std::uint64_t kUndefined = ~0ull;
[[noreturn]] void fail(const std::string& m) { throw std::runtime_error("HDF5 reader: " + m); }
std::size_t pad8(std::size_t n) { return (n + 7) & ~std::size_t(7); }
const char* filterName(unsigned id) {
  switch (id) {
    case 1: return "deflate (gzip)";
    case 2: return "shuffle";
    case 3: return "fletcher32";
    case 4: return "szip";
    case 5: return "nbit";
    case 6: return "scaleoffset";
    case 307: return "bzip2";
    case 32000: return "lzf";
    case 32001: return "blosc";
    case 32004: return "lz4";
    case 32008: return "bitshuffle";
    case 32013: return "zfp";
    case 32015: return "zstd";
    case 32026: return "blosc2";
    default: return "unregistered or third-party";
  }
}
}  // namespace

void File::need(std::size_t off, std::size_t n, const char* what) const {
  if (off > d.size() || n > d.size() - off) fail(std::string("truncated file while reading ") + what);
}

std::uint64_t File::u(std::size_t off, int bytes) const {
  need(off, bytes, "an integer");
  std::uint64_t v = 0;
  for (int i = bytes - 1; i >= 0; --i) v = (v << 8) | d[off + i];
  if (bytes < 8 && v == ((1ull << (8 * bytes)) - 1) && bytes == sizeOffsets) return kUndefined;
  return v;
}

File::File(const std::string& fileName) {
  std::ifstream f(fileName, std::ios::binary);
  if (!f) fail("could not open '" + fileName + "'");
  d.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  // the superblock sits at 0 or, behind a user block, at 512, 1024, 2048, ...
  std::size_t sb = std::string::npos;
  for (std::size_t off = 0; off + 8 <= d.size(); off = off ? off * 2 : 512) {
    if (std::memcmp(&d[off], kSignature, 8) == 0) { sb = off; break; }
  }
  if (sb == std::string::npos) fail("'" + fileName + "' has no HDF5 signature");
  need(sb, 24, "the superblock");
  const int version = d[sb + 8];
  if (version > 1) fail("superblock version " + std::to_string(version) + " (written with a newer libver) is not supported");
  sizeOffsets = d[sb + 13];
  sizeLengths = d[sb + 14];
  if ((sizeOffsets != 4 && sizeOffsets != 8) || (sizeLengths != 4 && sizeLengths != 8)) fail("unsupported offset/length size");
  std::size_t p = sb + 24 + (version == 1 ? 4 : 0);
  base = addr(p);
  if (base == kUndefined) base = 0;
  if (base != sb && base != 0) base = sb;   // addresses are relative to the superblock when a user block is present
  if (base == 0 && sb != 0) base = sb;
  p += 4 * sizeOffsets;                      // base, free-space, end-of-file, driver-info addresses
  // root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
  rootHeader = addr(p + sizeOffsets);
  if (rootHeader == kUndefined) fail("root group has no object header");
}

std::vector<File::Message> File::objectMessages(std::uint64_t headerAddress) const {
  std::size_t p = (std::size_t)(base + headerAddress);
  need(p, 16, "an object header");
  if (d[p] != 1) {
    if (std::memcmp(&d[p], "OHDR", 4) == 0) fail("version-2 object headers (libver='latest') are not supported");
    fail("unexpected object header version " + std::to_string(d[p]));
  }
  const std::size_t total = u(p + 2, 2);
  const std::size_t firstSize = u(p + 8, 4);
  std::vector<Message> out;
  std::vector<std::pair<std::size_t, std::size_t>> blocks{{p + 16, firstSize}};
  for (std::size_t b = 0; b < blocks.size() && out.size() < total; ++b) {
    std::size_t q = blocks[b].first, end = q + blocks[b].second;
    need(q, blocks[b].second, "object header messages");
    while (q + 8 <= end && out.size() < total) {
      Message m{(std::uint16_t)u(q, 2), d[q + 4], q + 8, (std::size_t)u(q + 2, 2)};
      need(m.offset, m.size, "a header message");
      if (m.type == 0x0010) blocks.push_back({(std::size_t)(base + addr(m.offset)), (std::size_t)u(m.offset + sizeOffsets, sizeLengths)});
      out.push_back(m);
      q = m.offset + m.size;
    }
  }
  return out;
}

// A node header is 8 bytes + two sibling addresses: no well-formed file holds more nodes than size / that.  Together with
// "a child sits exactly one level below its parent" this bounds the walk of a crafted file (a cycle of internal nodes would
// otherwise cost fan-out ^ depth calls).
void File::checkBtreeNode(int level, int expectLevel, std::size_t& visited, const char* what) const {
  if (level > 16) fail(std::string(what) + " is too deep");
  if (expectLevel >= 0 && level != expectLevel) fail(std::string(what) + ": a child node is not one level below its parent");
  if (++visited > d.size() / (8 + 2 * (std::size_t)sizeOffsets) + 1) fail(std::string(what) + " has more nodes than the file can hold");
}

void File::walkBtree(std::uint64_t node, std::uint64_t heapData, std::map<std::string, std::uint64_t>& out, int expectLevel, std::size_t& visited) const {
  std::size_t p = (std::size_t)(base + node);
  need(p, 24, "a B-tree node");
  if (std::memcmp(&d[p], "TREE", 4) != 0) fail("bad B-tree node signature");
  if (d[p + 4] != 0) fail("expected a group B-tree");
  const int level = d[p + 5];
  checkBtreeNode(level, expectLevel, visited, "group B-tree");
  const std::size_t used = u(p + 6, 2);
  std::size_t q = p + 8 + 2 * sizeOffsets;   // past the sibling pointers
  for (std::size_t i = 0; i < used; ++i) {
    q += sizeLengths;                       // key i
    const std::uint64_t child = addr(q);
    q += sizeOffsets;
    if (level > 0) { walkBtree(child, heapData, out, level - 1, visited); continue; }
    std::size_t s = (std::size_t)(base + child);
    need(s, 8, "a symbol table node");
    if (std::memcmp(&d[s], "SNOD", 4) != 0) fail("bad symbol table node signature");
    const std::size_t n = u(s + 6, 2);
    const std::size_t entry = 2 * sizeOffsets + 24;
    for (std::size_t k = 0; k < n; ++k) {
      std::size_t e = s + 8 + k * entry;
      need(e, entry, "a symbol table entry");
      const std::uint64_t nameOff = addr(e);
      std::size_t np = (std::size_t)(base + heapData + nameOff);
      need(np, 1, "a link name");
      std::string name(reinterpret_cast<const char*>(&d[np]), strnlen(reinterpret_cast<const char*>(&d[np]), d.size() - np));
      out[name] = addr(e + sizeOffsets);
    }
  }
}

std::map<std::string, std::uint64_t> File::groupLinks(std::uint64_t headerAddress) const {
  std::map<std::string, std::uint64_t> out;
  bool found = false;
  for (const Message& m : objectMessages(headerAddress)) {
    if (m.type == 0x0002 || m.type == 0x0006) fail("new-style groups (link messages) are not supported");
    if (m.type != 0x0011) continue;
    found = true;
    const std::uint64_t btree = addr(m.offset), heap = addr(m.offset + sizeOffsets);
    std::size_t hp = (std::size_t)(base + heap);
    need(hp, 8 + 2 * sizeLengths + sizeOffsets, "a local heap");
    if (std::memcmp(&d[hp], "HEAP", 4) != 0) fail("bad local heap signature");
    const std::uint64_t heapData = addr(hp + 8 + 2 * sizeLengths);
    std::size_t visited = 0;
    walkBtree(btree, heapData, out, -1, visited);
  }
  if (!found) fail("object is not a group");
  return out;
}

std::uint64_t File::resolve(const std::string& path) const {
  std::uint64_t cur = rootHeader;
  std::stringstream ss(path);
  std::string part;
  while (std::getline(ss, part, '/')) {
    if (part.empty()) continue;
    auto links = groupLinks(cur);
    auto it = links.find(part);
    if (it == links.end()) fail("no object '" + part + "' on path '" + path + "'");
    cur = it->second;
  }
  return cur;
}

bool File::exists(const std::string& path) const {
  try { (void)resolve(path); return true; } catch (const std::runtime_error&) { return false; }
}

std::vector<std::string> File::listGroup(const std::string& groupPath) const {
  std::vector<std::string> names;
  for (auto& kv : groupLinks(resolve(groupPath))) names.push_back(kv.first);
  return names;
}

std::map<std::string, File::Attribute> File::attributes(std::uint64_t headerAddress) const {
  std::map<std::string, Attribute> out;
  for (const Message& m : objectMessages(headerAddress)) {
    if (m.type != 0x000C) continue;
    const std::size_t p = m.offset;
    if (m.size < 8) fail("attribute message is too short");
    need(p, 8, "an attribute message");
    const int version = d[p];
    if (version < 1 || version > 3) fail("unsupported attribute message version");
    const std::size_t nameSize = u(p + 2, 2), dtSize = u(p + 4, 2), dsSize = u(p + 6, 2);
    std::size_t q = p + 8 + (version == 3 ? 1 : 0);
    auto step = [&](std::size_t n) { std::size_t at = q; q += (version == 1) ? pad8(n) : n; return at; };
    const std::size_t nameAt = step(nameSize), dtAt = step(dtSize), dsAt = step(dsSize);
    need(nameAt, nameSize, "an attribute name");
    std::string name(reinterpret_cast<const char*>(&d[nameAt]), strnlen(reinterpret_cast<const char*>(&d[nameAt]), nameSize));
    Attribute a;
    if (q > m.offset + m.size) fail("attribute message overruns its header message");   // sizes come from the file: check first
    need(dtAt, dtSize, "an attribute datatype");
    need(dsAt, dsSize, "an attribute dataspace");
    a.datatype.assign(d.begin() + dtAt, d.begin() + dtAt + dtSize);
    a.dataspace.assign(d.begin() + dsAt, d.begin() + dsAt + dsSize);
    a.data.assign(d.begin() + q, d.begin() + m.offset + m.size);
    out[name] = std::move(a);
  }
  return out;
}

bool File::hasAttribute(const std::string& objectPath, const std::string& name) const {
  return attributes(resolve(objectPath)).count(name) != 0;
}

std::string File::globalHeapObject(std::uint64_t collection, std::uint32_t index) const {
  std::size_t p = (std::size_t)(base + collection);
  need(p, 8 + sizeLengths, "a global heap collection");
  if (std::memcmp(&d[p], "GCOL", 4) != 0) fail("bad global heap signature");
  const std::size_t size = u(p + 8, sizeLengths);
  std::size_t q = p + 8 + sizeLengths, end = p + size;
  while (q + 8 + sizeLengths <= end) {
    const std::uint32_t idx = (std::uint32_t)u(q, 2);
    const std::size_t osize = u(q + 8, sizeLengths);
    if (idx == 0) break;   // free space
    if (idx == index) {
      need(q + 8 + sizeLengths, osize, "a global heap object");
      return std::string(reinterpret_cast<const char*>(&d[q + 8 + sizeLengths]), osize);
    }
    q += 8 + sizeLengths + pad8(osize);
  }
  fail("global heap object not found");
}

std::string File::readStringAttribute(const std::string& objectPath, const std::string& name) const {
  auto attrs = attributes(resolve(objectPath));
  auto it = attrs.find(name);
  if (it == attrs.end()) fail("no attribute '" + name + "' on '" + objectPath + "'");
  const Attribute& a = it->second;
  if (a.datatype.size() < 8) fail("malformed attribute datatype");
  const int cls = a.datatype[0] & 0x0f;
  std::uint32_t tsize;
  std::memcpy(&tsize, &a.datatype[4], 4);
  if (cls == 3) {   // fixed-length string, null terminated or padded
    if (a.data.size() < tsize) fail("attribute data shorter than its string type");
    return std::string(reinterpret_cast<const char*>(a.data.data()), strnlen(reinterpret_cast<const char*>(a.data.data()), tsize));
  }
  if (cls == 9 && (a.datatype[1] & 0x0f) == 1) {   // variable-length string: {length, collection address, index}
    if (a.data.size() < 4u + sizeOffsets + 4u) fail("malformed variable-length string attribute");
    std::uint32_t len, idx;
    std::memcpy(&len, &a.data[0], 4);
    std::uint64_t coll = 0;
    for (int i = sizeOffsets - 1; i >= 0; --i) coll = (coll << 8) | a.data[4 + i];
    std::memcpy(&idx, &a.data[4 + sizeOffsets], 4);
    std::string s = globalHeapObject(coll, idx);
    if (s.size() > len) s.resize(len);
    return s;
  }
  fail("attribute '" + name + "' is not a string");
}

Dataset File::openDataSet(const std::string& path) const {
  Dataset out;
  bool haveSpace = false, haveType = false, haveLayout = false, isChunked = false;
  ChunkedLayout chunked;
  std::vector<Filter> filters;
  for (const Message& m : objectMessages(resolve(path))) {
    const std::size_t p = m.offset;
    if (m.type == 0x0001) {   // dataspace
      need(p, 8, "a dataspace message");
      const int version = d[p], rank = d[p + 1];
      std::size_t q;
      if (version == 1) q = p + 8;
      else if (version == 2) q = p + 4;
      else fail("unsupported dataspace version");
      for (int i = 0; i < rank; ++i) out.shape.push_back((std::size_t)u(q + i * sizeLengths, sizeLengths));
      haveSpace = true;
    } else if (m.type == 0x0003) {   // datatype
      need(p, 8, "a datatype message");
      const int cls = d[p] & 0x0f;
      out.isFloat = (cls == 1);
      out.elementSize = (std::size_t)u(p + 4, 4);
      if (cls == 1 && (d[p + 1] & 1)) fail("big-endian floating point data is not supported");
      haveType = true;
    } else if (m.type == 0x0008) {   // data layout
      need(p, 8, "a data layout message");
      const int version = d[p];
      if (version == 3) {
        const int cls = d[p + 1];
        if (cls == 1) {
          const std::uint64_t a = addr(p + 2);
          const std::size_t n = u(p + 2 + sizeOffsets, sizeLengths);
          if (a != kUndefined) { need((std::size_t)(base + a), n, "dataset data"); out.bytes.assign(d.begin() + base + a, d.begin() + base + a + n); }
        } else if (cls == 0) {
          const std::size_t n = u(p + 2, 2);
          need(p + 4, n, "compact dataset data");
          out.bytes.assign(d.begin() + p + 4, d.begin() + p + 4 + n);
        } else if (cls == 2) {   // chunked: dimensionality, chunk B-tree address, chunk dims (the last one is the element size)
          const int nd = d[p + 2];
          if (nd < 2 || nd > 9) fail("chunked dataset '" + path + "' has an unsupported dimensionality");
          chunked.btree = addr(p + 3);
          for (int i = 0; i < nd; ++i) chunked.dims.push_back((std::size_t)u(p + 3 + sizeOffsets + 4 * i, 4));
          isChunked = true;
        } else fail("data layout class " + std::to_string(cls) + " of '" + path + "' is not supported (virtual dataset?)");
      } else if (version == 1 || version == 2) {
        const int rank = d[p + 1], cls = d[p + 2];
        std::size_t q = p + 8;
        if (cls == 1) {
          const std::uint64_t a = addr(q);
          q += sizeOffsets;
          std::size_t n = 1;
          for (int i = 0; i < rank; ++i) {   // last dimension is the element size
            const std::size_t dim = (std::size_t)u(q + 4 * i, 4);
            if (dim && n > d.size() / dim) fail("dataset '" + path + "' is larger than the file");
            n *= dim;
          }
          if (a != kUndefined) { need((std::size_t)(base + a), n, "dataset data"); out.bytes.assign(d.begin() + base + a, d.begin() + base + a + n); }
        } else if (cls == 0) {
          q += 4 * rank;
          const std::size_t n = u(q, 4);
          need(q + 4, n, "compact dataset data");
          out.bytes.assign(d.begin() + q + 4, d.begin() + q + 4 + n);
        } else if (cls == 2) {
          if (rank < 2 || rank > 9) fail("chunked dataset '" + path + "' has an unsupported dimensionality");
          chunked.btree = addr(q);
          for (int i = 0; i < rank; ++i) chunked.dims.push_back((std::size_t)u(q + sizeOffsets + 4 * i, 4));
          isChunked = true;
        } else fail("data layout class " + std::to_string(cls) + " of '" + path + "' is not supported");
      } else if (version == 4) {
        fail("version-4 data layout of '" + path + "' (written with libver='latest': v2 B-tree / extensible-array / fixed-array chunk "
             "index) is not supported -- rewrite the file with the default libver");
      } else fail("unsupported data layout version " + std::to_string(version));
      haveLayout = true;
    } else if (m.type == 0x000B) {   // filter pipeline (v1: padded names and values; v2: neither, no name below id 256)
      need(p, 2, "a filter pipeline message");
      const int version = d[p], count = d[p + 1];
      if (version != 1 && version != 2) fail("unsupported filter pipeline version " + std::to_string(version));
      std::size_t q = p + (version == 1 ? 8 : 2);
      for (int i = 0; i < count; ++i) {
        need(q, 8, "a filter description");
        Filter f;
        f.id = (std::uint16_t)u(q, 2);
        std::size_t nameLen = 0;
        if (version == 1 || f.id >= 256) { nameLen = u(q + 2, 2); q += 2; }
        q += 2;                                   // id (+ name length) consumed
        const std::size_t nvals = u(q + 2, 2);    // flags at q, number of client values at q + 2
        q += 4;
        q += (version == 1) ? pad8(nameLen) : nameLen;
        need(q, 4 * nvals, "filter client data");
        for (std::size_t k = 0; k < nvals; ++k) f.values.push_back((std::uint32_t)u(q + 4 * k, 4));
        q += 4 * nvals;
        if (version == 1 && (nvals & 1)) q += 4;
        filters.push_back(std::move(f));
      }
    }
  }
  if (!haveSpace || !haveType || !haveLayout) fail("'" + path + "' is not a dataset");
  std::size_t n = out.elementSize;
  for (auto s : out.shape) {
    if (s && n > (std::size_t(1) << 40) / s) fail("dataset '" + path + "' has an implausible shape");
    n *= s;
  }
  for (const Filter& f : filters)
    if (f.id < 1 || f.id > 3)
      fail("dataset '" + path + "' uses filter " + std::to_string(f.id) + " (" + filterName(f.id) + "), which this reader cannot undo: "
           "supported are deflate (1), shuffle (2) and fletcher32 (3)");
  if (isChunked) {
    if (n > 64 * d.size() + (std::size_t(1) << 20)) fail("dataset '" + path + "' has a shape implausibly larger than the file");
    readChunks(path, chunked, filters, out);
  } else {
    if (!filters.empty()) fail("dataset '" + path + "' has a filter pipeline but no chunked layout");
    if (out.bytes.size() < n) fail("dataset '" + path + "' holds fewer bytes than its shape needs");
  }
  out.bytes.resize(n);
  return out;
}

void File::walkChunkBtree(std::uint64_t node, std::size_t nDims, int expectLevel, std::size_t& visited,
                          std::vector<std::pair<std::vector<std::uint64_t>, std::pair<std::uint64_t, std::pair<std::uint32_t, std::uint32_t>>>>& chunks) const {
  if (node == kUndefined) return;                      // no chunk was ever written: the dataset reads as zeros
  std::size_t p = (std::size_t)(base + node);
  need(p, 8 + 2 * sizeOffsets, "a chunk B-tree node");
  if (std::memcmp(&d[p], "TREE", 4) != 0) fail("bad chunk B-tree node signature");
  if (d[p + 4] != 1) fail("expected a raw-data-chunk B-tree (node type 1), found type " + std::to_string(d[p + 4]));
  const int level = d[p + 5];
  checkBtreeNode(level, expectLevel, visited, "chunk B-tree");
  const std::size_t used = u(p + 6, 2);
  const std::size_t keySize = 8 + 8 * nDims;
  std::size_t q = p + 8 + 2 * sizeOffsets;
  for (std::size_t i = 0; i < used; ++i) {
    need(q, keySize + sizeOffsets, "a chunk B-tree entry");
    const std::uint32_t bytes = (std::uint32_t)u(q, 4), mask = (std::uint32_t)u(q + 4, 4);
    std::vector<std::uint64_t> offset(nDims);
    for (std::size_t k = 0; k < nDims; ++k) offset[k] = u(q + 8 + 8 * k, 8);
    const std::uint64_t child = addr(q + keySize);
    q += keySize + sizeOffsets;
    if (level > 0) walkChunkBtree(child, nDims, level - 1, visited, chunks);
    else chunks.push_back({std::move(offset), {child, {bytes, mask}}});
    if (chunks.size() > (1u << 22)) fail("too many chunks");
  }
}

void File::readChunks(const std::string& path, const ChunkedLayout& layout, const std::vector<Filter>& filters, Dataset& out) const {
  const std::size_t rank = out.shape.size();
  if (layout.dims.size() != rank + 1) fail("chunk rank of '" + path + "' does not match its dataspace");
  const std::size_t es = layout.dims[rank];
  if (es != out.elementSize) fail("chunk element size of '" + path + "' does not match its datatype");
  std::size_t total = es, chunkBytes = es;
  for (std::size_t k = 0; k < rank; ++k) {
    if (layout.dims[k] == 0) fail("zero-sized chunk dimension in '" + path + "'");
    total *= out.shape[k];
    if (chunkBytes > (std::size_t(1) << 32) / layout.dims[k]) fail("implausible chunk size in '" + path + "'");
    chunkBytes *= layout.dims[k];
  }
  if (chunkBytes > 64 * d.size() + (std::size_t(1) << 20)) fail("chunk shape of '" + path + "' is implausibly larger than the file");
  out.bytes.assign(total, 0);                          // chunks never written read as the fill value (zero)
  if (total == 0) return;
  std::vector<std::pair<std::vector<std::uint64_t>, std::pair<std::uint64_t, std::pair<std::uint32_t, std::uint32_t>>>> chunks;
  std::size_t visited = 0;
  walkChunkBtree(layout.btree, rank + 1, -1, visited, chunks);
  std::vector<std::uint8_t> raw, tmp;
  for (const auto& c : chunks) {
    const auto& offset = c.first;
    const std::uint64_t at = c.second.first;
    const std::uint32_t stored = c.second.second.first, mask = c.second.second.second;
    need((std::size_t)(base + at), stored, "a dataset chunk");
    raw.assign(d.begin() + base + at, d.begin() + base + at + stored);
    // undo the pipeline, last filter first (a set bit i of the chunk's mask: filter i was skipped for this chunk)
    for (std::size_t fi = filters.size(); fi-- > 0;) {
      if (mask & (1u << fi)) continue;
      const Filter& f = filters[fi];
      if (f.id == 3) {                                 // fletcher32: four checksum bytes behind the data
        if (raw.size() < 4) fail("chunk of '" + path + "' is too short for its checksum");
        raw.resize(raw.size() - 4);
      } else if (f.id == 1) {                          // deflate
        // a fletcher32 filter applied BEFORE deflate (h5repack -f FLET -f GZIP, older h5py) leaves its four checksum bytes
        // inside the deflated stream: the inflated chunk is that much longer than the chunk shape
        std::size_t extra = 0;
        for (std::size_t fj = 0; fj < fi; ++fj) if (filters[fj].id == 3 && !(mask & (1u << fj))) extra += 4;
        tmp.assign(chunkBytes + extra, 0);
        uLongf got = (uLongf)tmp.size();
        const int rc = ::uncompress(tmp.data(), &got, raw.data(), (uLong)raw.size());
        if (rc != Z_OK) fail("could not inflate a chunk of '" + path + "' (zlib error " + std::to_string(rc) + ")");
        tmp.resize(got);
        raw.swap(tmp);
      } else if (f.id == 2) {                          // byte shuffle: [all byte 0s][all byte 1s]... -> elements
        const std::size_t width = f.values.empty() ? es : f.values[0];
        if (width > 1 && raw.size() >= width) {
          const std::size_t count = raw.size() / width;
          tmp.assign(raw.size(), 0);
          for (std::size_t b = 0; b < width; ++b)
            for (std::size_t e = 0; e < count; ++e) tmp[e * width + b] = raw[b * count + e];
          std::copy(raw.begin() + count * width, raw.end(), tmp.begin() + count * width);
          raw.swap(tmp);
        }
      }
    }
    if (raw.size() < chunkBytes) fail("a chunk of '" + path + "' holds fewer bytes than the chunk shape needs");
    // copy the part of the chunk that lies inside the dataset, one innermost row at a time
    for (std::size_t k = 0; k < rank; ++k)
      if (offset[k] >= out.shape[k] || offset[k] % layout.dims[k]) fail("chunk offset outside the dataset '" + path + "'");
    const std::size_t inner = rank ? std::min<std::size_t>(layout.dims[rank - 1], out.shape[rank - 1] - offset[rank - 1]) : 1;
    std::vector<std::size_t> idx(rank, 0);             // position inside the chunk (all but the innermost dimension)
    for (;;) {
      std::size_t src = 0, dst = 0;
      bool inside = true;
      for (std::size_t k = 0; k + 1 < rank; ++k) {
        src = src * layout.dims[k] + idx[k];
        const std::size_t g = (std::size_t)offset[k] + idx[k];
        if (g >= out.shape[k]) inside = false;
        dst = dst * out.shape[k] + g;
      }
      if (inside) {
        if (rank) { src = src * layout.dims[rank - 1]; dst = dst * out.shape[rank - 1] + (std::size_t)offset[rank - 1]; }
        std::memcpy(&out.bytes[dst * es], &raw[src * es], inner * es);
      }
      std::size_t k = rank > 1 ? rank - 1 : 0;         // odometer over the outer dimensions
      bool done = true;
      while (k-- > 0) {
        if (++idx[k] < layout.dims[k]) { done = false; break; }
        idx[k] = 0;
      }
      if (done) break;
    }
  }
}

}  // namespace h5
