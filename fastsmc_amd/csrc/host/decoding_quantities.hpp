// decoding_quantities.hpp -- the precomputed HMM tables ("decoding quantities").
// Mirrors the data the reference's DecodingQuantities holds (DecodingQuantities.hpp:47-87) and its
// gzipped-text parser (DecodingQuantities.cpp:60-345), with the per-distance vectors stored as dense
// row-major tables [key][state] (the shape the device wants) instead of unordered_map<float, vector>.
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace fsmc_host
{

// Per-genetic-distance vectors: rows ordered as in the file, looked up by exact float key
// (the reference uses unordered_map<float,...>::at, HMM.cpp:795-797).
struct KeyedTable {
  std::vector<float> keys;
  std::vector<float> values; // [keys.size()][states]
  std::unordered_map<uint32_t, int> index; // float bit pattern (+0 normalised) -> row

  void add(float key, const std::vector<float>& row);
  int find(float key) const; // -1 if absent
  const float* row(int r, int states) const { return values.data() + static_cast<size_t>(r) * states; }
  size_t size() const { return keys.size(); }
};

class DecodingQuantities
{
public:
  DecodingQuantities() = default;
  // Throws std::runtime_error if the file is missing or does not start with "TransitionType"
  // (DecodingQuantities.cpp:39-58).
  explicit DecodingQuantities(const std::string& fileName);

  static void validateDecodingQuantitiesFile(const std::string& fileName);

  unsigned int states = 0;
  int CSFSSamples = 0;
  std::vector<float> initialStateProb;
  std::vector<float> expectedTimes;
  std::vector<float> discretization;
  std::vector<float> timeVector;
  std::vector<float> columnRatios;
  std::vector<std::vector<float>> classicEmissionTable;
  std::vector<std::vector<float>> compressedEmissionTable;
  KeyedTable Dvectors, Bvectors, Uvectors, rowRatioVectors;
  std::unordered_map<int, std::vector<float>> homozygousEmissionMap;
  std::vector<std::vector<std::vector<float>>> CSFSmap, foldedCSFSmap, ascertainedCSFSmap, foldedAscertainedCSFSmap;

private:
  void parse(const std::string& fileName);
};

} // namespace fsmc_host
