// data.hpp -- haplotype data for the decode path.
// Reads the reference's input surface (Data.cpp): Oxford .hap[s][.gz] + .samples + .map[.gz], folds
// to minor alleles, counts alleles per site, and keeps genotypes as a packed bit matrix
// [haplotype][site/64] (bit s%64 of word s/64) -- the layout uploaded to HBM (fsmc_haps_upload) --
// instead of per-individual vector<bool>.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "decoding_params.hpp"
#include "util.hpp"

namespace fsmc_host
{

// Individual.hpp:25-40: the two haplotypes of one diploid sample as bit vectors.  The decode path keeps genotypes
// packed (Data::bits); this is the reference's view of them for API users (Data::individuals()).
class Individual
{
public:
  std::vector<bool> genotype1;
  std::vector<bool> genotype2;

  explicit Individual(int numOfSites = 0) : genotype1(static_cast<size_t>(numOfSites)), genotype2(static_cast<size_t>(numOfSites)) {}
  void setGenotype(int_least8_t hap, int pos, bool val) // Individual.cpp:25-32: hap 1 -> genotype1, anything else -> genotype2
  {
    if (hap == 1) {
      genotype1[static_cast<size_t>(pos)] = val;
    } else {
      genotype2[static_cast<size_t>(pos)] = val;
    }
  }
};

class Data
{
public:
  Data() = default;
  explicit Data(const DecodingParams& params); // Data.cpp:36-95

  // Build from arrays instead of files (synthetic inputs): alleles[hap*sites + site] in {0,1},
  // haplotypes 2i and 2i+1 belong to individual i; positions in bp and centimorgans.  Follows the
  // FastSMC-mode conventions (gen = float(cM / 100.f), Data.cpp:549-565).
  static Data fromArrays(const uint8_t* alleles, size_t nHaps, size_t nSites, const int64_t* bp, const double* cm,
                         bool foldToMinor, bool useKnownSeed, int chrNumber = 1);

  static int countHapLines(const std::string& inFileRoot);   // Data.cpp:264-287
  static int countSamplesLines(const std::string& inFileRoot); // Data.cpp:289-318

  // Data::calculateUndistinguishedCounts (Data.cpp:567-599); throws where the reference exits.
  std::vector<std::vector<int>> calculateUndistinguishedCounts(int numCsfsSamples) const;

  bool genotype(size_t hapRow, size_t site) const
  {
    return (bits[hapRow * wordsPerHap + (site >> 6)] >> (site & 63)) & 1ull;
  }
  std::vector<bool> genotypeVector(size_t hapRow) const;
  // the reference's Data::individuals (Data.hpp:36), unpacked on demand from the bit matrix
  std::vector<Individual> individuals() const;
  size_t numIndividuals() const { return FamIDList.size(); }
  size_t numHapRows() const { return 2 * FamIDList.size(); }
  // number of local haplotype row r in the whole file: 2 * (sample line of its individual) + r % 2
  unsigned globalHapId(size_t hapRow) const { return 2u * globalIndIndex[hapRow / 2] + static_cast<unsigned>(hapRow % 2); }
  std::vector<unsigned> globalIndIndex; // sample-file line of every loaded individual

  std::vector<std::string> FamIDList, IIDList, famAndIndNameList;
  unsigned long sampleSize = 0;        // individuals in the file
  unsigned long haploidSampleSize = 0; // 2 * sampleSize
  int sites = 0;
  bool decodingUsesCSFS = false;
  bool foldToMinorAlleles = false;
  int chrNumber = 0;
  std::vector<float> geneticPositions; // Morgans
  std::vector<int> physicalPositions;
  std::vector<float> recRateAtMarker;
  std::vector<bool> siteWasFlippedDuringFolding;
  std::vector<int> totalSamplesCount;
  std::vector<int> derivedAlleleCounts;
  std::vector<std::string> SNP_IDs;

  // job windows over individuals (Data.cpp:62-80)
  bool mJobbing = false;
  int windowSize = 0, w_i = 0, w_j = 0;
  bool is_j_above_diag = false;

  // packed genotypes of the individuals this job loaded: row 2*ind + (hap-1)
  std::vector<uint64_t> bits;
  // the generator the emission preparation draws its seeds from (the reference: the process's rand(), seeded in the
  // constructor, Data.cpp:62-70; here the object's own -- same numbers, not perturbed by other threads' rand() calls)
  mutable GlibcRand rng;
  size_t wordsPerHap = 0;

private:
  void setupJobWindows(int jobID, int jobs);
  bool readSample(unsigned linesProcessed, int jobID, int jobs) const; // Data.cpp:251-262
  void readSamplesList(const std::string& inFileRoot, int jobID, int jobs);
  void readHapsAsmc(const std::string& inFileRoot);
  void readHapsFastSMC(const std::string& inFileRoot, int jobID, int jobs,
                       const std::vector<std::pair<unsigned long, double>>& geneticMap);
  void readMapAsmc(const std::string& inFileRoot);
  static std::vector<std::pair<unsigned long, double>> readMapFastSMC(const std::string& inFileRoot);
  void addMarker(unsigned long physicalPos, double geneticPos, unsigned pos);
  void addMarkerFromMap(unsigned long bp, const std::vector<std::pair<unsigned long, double>>& map, unsigned& cur,
                        unsigned pos);
  void allocateBits();
  void setBit(size_t hapRow, size_t site) { bits[hapRow * wordsPerHap + (site >> 6)] |= 1ull << (site & 63); }
};

} // namespace fsmc_host
