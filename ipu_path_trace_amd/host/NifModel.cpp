#include "NifModel.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <stdexcept>

#include "Hdf5Model.hpp"
#include "json.hpp"
#include "logging.hpp"

namespace {

std::string slurp(const std::string& file) {
  std::ifstream f(file, std::ios::binary);
  if (!f) return std::string();
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

std::uint16_t floatToHalf(float f) {  // round-to-nearest-even
  std::uint32_t x; std::memcpy(&x, &f, 4);
  std::uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (std::uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));
  if (ax >= 0x477ff000u) return (std::uint16_t)(sign | 0x7c00u);
  if (ax < 0x33000001u) return (std::uint16_t)sign;
  int e = (int)(ax >> 23) - 127;
  std::uint32_t m = (ax & 0x7fffffu) | 0x800000u, shift = e < -14 ? 13 + (-14 - e) : 13, hexp = e < -14 ? 0 : e + 15;
  std::uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) q += 1u;
  return (std::uint16_t)(sign | (hexp == 0 ? q : ((hexp - 1u) << 10) + q));
}

}  // namespace

NifMetaData::NifMetaData(const std::string& file) {
  const std::string text = slurp(file);
  if (text.empty()) throw std::runtime_error("Empty property tree after parsing file: '" + file + "'");
  try {
    const json::Value pt = json::parse(text);
    embeddingDimension = (std::size_t)pt.at("embedding_dimension").num;
    name = pt.at("name").str;
    for (auto& v : pt.at("original_image_shape").arr) imageShape.push_back((std::size_t)v.num);
    const auto& enc = pt.at("encode_params");
    eps = (float)enc.at("eps").num;
    logToneMap = enc.at("log_tone_map").b;
    max = (float)enc.at("max").num;
    for (auto& v : enc.at("mean").arr) mean.push_back((float)v.num);
    if (mean.size() != 3) throw std::runtime_error("encode_params.mean must have 3 entries");
    if (logToneMap) for (auto& m : mean) m -= eps;  // fold the inverse eps into the mean (NifMetaData.cpp:48-53)
    bool nextSize = false, nextCount = false;
    for (auto& v : pt.at("train_command").arr) {  // NifMetaData.cpp:56-65
      if (nextSize) { hiddenSize = (std::size_t)std::atoi(v.str.c_str()); nextSize = false; }
      if (nextCount) { layerCount = (std::size_t)std::atoi(v.str.c_str()); nextCount = false; }
      if (v.str == "--layer-size") nextSize = true;
      if (v.str == "--layer-count") nextCount = true;
    }
  } catch (const std::exception& e) {
    std::stringstream ss;
    ss << "Error reading property: " << e.what() << " from file: '" << file << "'";
    throw std::runtime_error(ss.str());
  }
}

NifModel::Data::Data(const std::string& metaFile) : metaData(metaFile) {
  pt_log::info_("Loading model metadata from file: '{}'", metaFile);
  pt_log::debug_("NIF embedding dimension: {}", metaData.embeddingDimension);
  pt_log::debug_("NIF hidden dimension: {}", metaData.hiddenSize);
}

NifModel::Data::Data(const std::string& weightFile, const std::string& metaFile) : Data(metaFile) { setupModel(weightFile); }

// converted.ptnif: "PTNIF1\0\0", u32 n_layers, u32 embedding_dim, then per layer
// u32 rows, cols, dtype (0 = float16, 1 = float32), relu, has_bias followed by the kernel bytes [rows][cols] and bias bytes.
void NifModel::Data::setupModel(const std::string& weightFile) {
  if (weightFile.size() > 5 && (weightFile.rfind(".hdf5") == weightFile.size() - 5 || weightFile.rfind(".h5") == weightFile.size() - 3)) {
    // the reference's own asset format: NifModel.cpp:51-85 on top of Hdf5Model
    Hdf5Model h5model(weightFile);
    std::size_t i = 0;
    for (const auto& l : h5model.get()) {
      layers.emplace_back(l.kernelData.shape, l.kernelData.isHalf() ? "float16" : "float32", l.activation, l.name);
      auto& newLayer = layers.back();
      newLayer.kernel.data = l.kernelData.storage;
      if (l.useBias) newLayer.bias.data = l.biasData.storage;
      if (newLayer.activationFunction == "linear") newLayer.activationFunction = "none";   // NifModel.cpp:73-76
      pt_log::debug_("Layer {}: weight tensors: {} ({} x {})", i, newLayer.kernel.getName(), l.kernelData.shape[0], l.kernelData.shape[1]);
      i += 1;
    }
    return;
  }
  std::ifstream f(weightFile, std::ios::binary);
  if (!f) throw std::runtime_error("Could not open NIF weight file '" + weightFile + "'");
  char magic[8];
  f.read(magic, 8);
  if (std::memcmp(magic, "PTNIF1\0\0", 8) != 0) throw std::runtime_error("'" + weightFile + "' is not a PTNIF1 file");
  f.seekg(0, std::ios::end);
  const std::uint64_t fileSize = (std::uint64_t)f.tellg();
  f.seekg(8, std::ios::beg);
  std::uint32_t n = 0, emb = 0;
  f.read((char*)&n, 4);
  f.read((char*)&emb, 4);
  if (!f || n == 0 || n > 16) throw std::runtime_error("'" + weightFile + "': layer count must be in 1..16");
  if (emb != metaData.embeddingDimension) throw std::runtime_error("embedding dimension of weights and metadata differ");
  for (std::uint32_t i = 0; i < n; ++i) {
    std::uint32_t hdr[5];
    f.read((char*)hdr, sizeof(hdr));
    if (!f) throw std::runtime_error("Truncated NIF weight file '" + weightFile + "'");
    if (hdr[2] > 1) throw std::runtime_error("Only float16 and float32 weights are supported.");
    const std::size_t esz = hdr[2] == 0 ? 2 : 4;
    // sizes come from the file: bound them by what the file can hold before allocating
    if (hdr[0] == 0 || hdr[1] == 0 || (std::uint64_t)hdr[0] * hdr[1] * esz > fileSize)
      throw std::runtime_error("'" + weightFile + "': layer " + std::to_string(i) + " has an impossible shape");
    layers.emplace_back(std::vector<std::size_t>{hdr[0], hdr[1]}, hdr[2] == 0 ? "float16" : "float32", hdr[3] ? "relu" : "none",
                        "dense_" + std::to_string(i));
    auto& l = layers.back();
    l.kernel.data.resize((std::size_t)hdr[0] * hdr[1] * esz);
    f.read((char*)l.kernel.data.data(), l.kernel.data.size());
    if (hdr[4]) {
      l.bias.data.resize((std::size_t)hdr[1] * esz);
      f.read((char*)l.bias.data.data(), l.bias.data.size());
    }
    if (!f) throw std::runtime_error("Truncated NIF weight file '" + weightFile + "'");
    pt_log::debug_("Layer {}: weight tensors: {} ({} x {})", i, l.kernel.getName(), hdr[0], hdr[1]);
  }
}

std::shared_ptr<NifModel::Data> NifModel::Data::synthetic(const std::string& metaFile, std::uint32_t seed) {
  std::shared_ptr<Data> d(new Data(metaFile));
  const auto& m = d->metaData;
  if (!m.hiddenSize || !m.layerCount) throw std::runtime_error("metadata has no --layer-size/--layer-count");
  const std::size_t inDim = 4 * m.embeddingDimension, skip = m.layerCount / 2;
  std::mt19937 rng(seed);
  std::size_t fanIn = inDim;
  for (std::size_t l = 0; l <= m.layerCount; ++l) {
    const bool head = l == m.layerCount;
    if (!head && l == skip) fanIn += inDim;  // the concat NifModel.cpp:305-308 detects
    const std::size_t out = head ? 3 : m.hiddenSize;
    d->layers.emplace_back(std::vector<std::size_t>{fanIn, out}, "float16", head ? "none" : "relu", "dense_" + std::to_string(l));
    auto& L = d->layers.back();
    std::normal_distribution<float> nw(0.f, (head ? 0.25f : 1.f) * std::sqrt(2.f / fanIn)), nb(0.f, 0.05f);
    L.kernel.data.resize(fanIn * out * 2);
    L.bias.data.resize(out * 2);
    auto* k = reinterpret_cast<std::uint16_t*>(L.kernel.data.data());
    auto* b = reinterpret_cast<std::uint16_t*>(L.bias.data.data());
    for (std::size_t i = 0; i < fanIn * out; ++i) k[i] = floatToHalf(nw(rng));
    for (std::size_t i = 0; i < out; ++i) b[i] = floatToHalf(nb(rng));
    fanIn = out;
  }
  return d;
}

NifModel::NifModel(std::shared_ptr<Data>& sharedData, const std::string& modelName) : data(sharedData), name(modelName) {}
NifModel::~NifModel() {}

std::size_t NifModel::flopsPerSample() const {
  std::size_t flops = 0;
  for (const auto& l : data->getLayers()) {
    flops += 2 * l.kernel.shape[0] * l.kernel.shape[1];
    if (l.hasBias()) flops += l.bias.shape[0];
  }
  return flops;
}

void NifModel::analyseModel(std::size_t sampleCount) const {
  std::size_t parametersBytes = 0;
  for (const auto& l : data->getLayers()) parametersBytes += l.kernel.data.size() + l.bias.data.size();
  pt_log::info_("NIF {} layers: {}", name, data->getLayers().size());
  pt_log::info_("NIF {} Hidden size: {}", name, data->getLayers().front().kernel.shape[1]);
  pt_log::info_("NIF {} batch size: {}", name, sampleCount);
  pt_log::info_("NIF {} model FLOPS: {}", name, flopsPerSample() * sampleCount);
  pt_log::info_("NIF {} parameter size: {} KiB", name, parametersBytes / 1024.f);
}

void NifModel::upload(pt_handle device) const {
  std::vector<pt_layer> ls;
  std::size_t converted = 0;
  for (const auto& l : data->getLayers()) {
    pt_layer p{};
    p.rows = (std::uint32_t)l.kernel.shape[0];
    p.cols = (std::uint32_t)l.kernel.shape[1];
    p.kernel = l.kernel.data.data();
    p.bias = l.hasBias() ? l.bias.data.data() : nullptr;
    p.relu = l.activationFunction == "relu";
    // NifModel.cpp:58-60 accepts float16 and float32 layers and gives every matmul its kernel's type (:314).  An all-float32
    // model runs in float as the reference's would; a model that MIXES the types is an extension of this port (its casts
    // between layers are the port's own: the reference ships no such model and has no cast there)
    if (l.kernel.type == "float16") p.dtype = PT_DTYPE_F16;
    else if (l.kernel.type == "float32") { p.dtype = PT_DTYPE_F32; converted += 1; }
    else throw std::runtime_error("Unsupported NIF weight type '" + l.kernel.type + "' (expected float16 or float32).");
    ls.push_back(p);
  }
  if (converted == ls.size()) pt_log::info_("NIF {}: all {} layers are float32: the model runs on the float path (fp32 matrix rate)", name, ls.size());
  else if (converted) pt_log::warn_("NIF {}: {} of {} layers are float32 and the rest float16: a MIXED model is an extension of this port (each layer in its own type, activations cast between them by rules of its own -- the reference has no such model and no fixture for it)", name, converted, ls.size());
  const auto& m = data->getMetaData();
  if (pt_upload_nif(device, ls.data(), (std::uint32_t)ls.size(), (std::uint32_t)m.embeddingDimension, m.max, m.mean.data(),
                    m.logToneMap ? 1 : 0))
    throw std::runtime_error(std::string("init_nif_weights failed: ") + pt_last_error(device));
}
