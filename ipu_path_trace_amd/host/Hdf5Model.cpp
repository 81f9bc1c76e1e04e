#include "Hdf5Model.hpp"

#include <stdexcept>

#include "json.hpp"
#include "logging.hpp"

std::vector<Hdf5Model::JsonLayer> parseJsonModel(const std::string& s) {
  const json::Value pt = json::parse(s);
  if (pt.at("class_name").str != "Functional") throw std::runtime_error("Expdected a Keras 'Functional' Model");
  std::vector<Hdf5Model::JsonLayer> layers;
  for (const json::Value& node : pt.at("config").at("layers").arr) {
    const std::string cn = node.at("class_name").str;
    if (cn == "Dense") {
      const json::Value& cfg = node.at("config");
      Hdf5Model::JsonLayer l;
      l.name = cfg.at("name").str;
      l.activation = cfg.at("activation").str;
      const json::Value& dt = cfg.at("dtype");   // a string, or a policy object {"class_name": "Policy", "config": {"name": ...}}
      l.dtype = dt.type == json::Value::String ? dt.str : dt.at("config").at("name").str;
      l.units = (std::size_t)cfg.at("units").num;
      l.useBias = cfg.at("use_bias").b;
      pt_log::debug_("Layer: {} {} (act: {} bias: {})", cn, l.name, l.activation, l.useBias);
      layers.push_back(l);
    } else if (cn == "InputLayer" || cn == "Concatenate") {
      pt_log::warn_("Ignoring layer (classname: {})", cn);   // implemented by the NIF kernel itself
    } else {
      throw std::runtime_error("Layer class: '" + cn + "' not supported by Hdf5Model loader.");
    }
  }
  return layers;
}

Hdf5Model::Data::Data(const h5::Dataset& dset) : shape(dset.shape), storage(dset.bytes) {
  if (!dset.isFloat || (dset.elementSize != 2 && dset.elementSize != 4))
    throw std::runtime_error("Only float32 and float16 weights are supported.");   // Hdf5Model.cpp:112-117
  isHalfFloat = dset.elementSize == 2;
  numElements = storage.size() / dset.elementSize;
}

Hdf5Model::Hdf5Model(const std::string& file) : hdf(file) {
  pt_log::info_("Reading weights saved from '{}', keras_version {}, backend {}", file, readStringAttribute("keras_version"),
                readStringAttribute("backend"));
  sequential = parseJsonModel(readStringAttribute("model_config"));
  for (auto& l : sequential) {   // dataset paths as Hdf5Model.cpp:71-82
    const std::string base = "/model_weights/" + l.name + "/" + l.name;
    l.kernelData = Data(hdf.openDataSet(base + "/kernel:0"));
    if (l.useBias) l.biasData = Data(hdf.openDataSet(base + "/bias:0"));
  }
  pt_log::info_("Finished reading model description");
}
