// Host side of the NIF environment light: metadata, layer containers and the weight hand-off to the device.
// Mirrors the reference's src/neural_networks/{NifMetaData,DenseLayer,NifModel}.hpp; the Poplar graph
// building half of NifModel is what libptmi.so's MFMA kernel replaces.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "ptmi.h"

using TensorShape = std::vector<std::size_t>;

/// NifMetaData.hpp:11-24 / NifMetaData.cpp:11-71.
struct NifMetaData {
  NifMetaData(const std::string& file);
  virtual ~NifMetaData() {}

  std::string name;
  std::size_t embeddingDimension = 0;
  std::size_t hiddenSize = 0;
  std::size_t layerCount = 0;
  TensorShape imageShape;

  // Encoder params:
  std::vector<float> mean;  // -eps already folded in when log tone-mapped (NifMetaData.cpp:48-53)
  float eps = 0.f;
  float max = 0.f;
  bool logToneMap = false;
};

/// DenseLayer.hpp:5-31 with poplar::Type replaced by a string ("float16" / "float32").
struct HostTensor {
  HostTensor(const std::vector<std::size_t>& shape, const std::string& dtype, const std::string& name)
      : shape(shape), nameSuffix(name), type(dtype) {}
  const std::string& getName() const { return nameSuffix; }
  std::vector<std::size_t> shape;
  std::string nameSuffix;
  std::string type;
  std::vector<std::uint8_t> data;
};

struct DenseLayer {
  DenseLayer(const std::vector<std::size_t>& shape, const std::string& dtype, const std::string& activation,
             const std::string& layerName)
      : kernel(shape, dtype, layerName + "/kernel"), bias({shape.back()}, dtype, layerName + "/bias"),
        activationFunction(activation) {}
  bool hasBias() const { return !bias.data.empty(); }
  HostTensor kernel;
  HostTensor bias;
  std::string activationFunction;
};

struct NifModel {
  /// Container for all NIF data (shared between devices): NifModel.hpp:21-37.
  struct Data {
    /// `weightFile`: <assets>/converted.hdf5 as the reference reads it (Hdf5Model.cpp:62-87; here through the
    /// dependency-free reader of Hdf5Reader.hpp), or the flat side-car <assets>/converted.ptnif
    /// (ipu_path_trace_amd/nif_assets.py: write_ptnif documents the format).
    Data(const std::string& weightFile, const std::string& metaFile);
    /// Seeded stand-in weights of the architecture named by the metadata (the trained weights are not
    /// shipped with the reference checkout).
    static std::shared_ptr<Data> synthetic(const std::string& metaFile, std::uint32_t seed);

    const NifMetaData& getMetaData() const { return metaData; }
    NifMetaData& getMetaData() { return metaData; }
    const std::vector<DenseLayer>& getLayers() const { return layers; }
    std::vector<DenseLayer>& getLayers() { return layers; }

  private:
    explicit Data(const std::string& metaFile);
    void setupModel(const std::string& weightFile);
    NifMetaData metaData;
    std::vector<DenseLayer> layers;
  };

  NifModel(std::shared_ptr<Data>& sharedData, const std::string& modelName);
  virtual ~NifModel();

  /// Log layers / hidden size / FLOPs / parameter size (NifModel.cpp:122-144).
  void analyseModel(std::size_t sampleCount) const;
  std::size_t flopsPerSample() const;

  /// The role of connectStreams() + program "init_nif_weights" (NifModel.cpp:375-401,
  /// PathTracerApp.cpp:612-613): hand every kernel, bias, max and mean to the device.
  void upload(pt_handle device) const;

private:
  std::shared_ptr<Data> data;
  const std::string name;
};
