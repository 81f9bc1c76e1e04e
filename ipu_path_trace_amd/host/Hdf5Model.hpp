// Keras H5 model -> list of Dense layers: the reference's Hdf5Model (src/keras/Hdf5Model.hpp:12-49) on top of the
// dependency-free reader.  Same public surface (Data, JsonLayer, get(), readStringAttribute).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "Hdf5Reader.hpp"

struct Hdf5Model {
  struct Data {
    Data() = default;
    explicit Data(const h5::Dataset& dset);
    std::size_t rank() const { return shape.size(); }
    std::size_t elements() const { return numElements; }
    bool isHalf() const { return isHalfFloat; }
    std::vector<std::size_t> shape;
    std::vector<std::uint8_t> storage;

  private:
    std::size_t numElements = 0;
    bool isHalfFloat = false;
  };

  struct JsonLayer {
    std::string name, activation, dtype;
    std::size_t units = 0;
    bool useBias = false;
    Data kernelData, biasData;
  };

  explicit Hdf5Model(const std::string& file);
  virtual ~Hdf5Model() = default;

  std::string readStringAttribute(const std::string& attrName) const { return hdf.readStringAttribute(attrName); }
  const std::vector<JsonLayer>& get() const { return sequential; }

private:
  h5::File hdf;
  std::vector<JsonLayer> sequential;
};

/// Dense layers of a Keras "Functional" model config, in layer order (Hdf5Model.cpp:8-55): Dense layers are kept,
/// InputLayer / Concatenate are skipped, anything else is an error.
std::vector<Hdf5Model::JsonLayer> parseJsonModel(const std::string& modelConfig);
