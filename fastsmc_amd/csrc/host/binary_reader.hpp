// binary_reader.hpp -- reader for the binary IBD output (.bibd.gz) written by HMM::writeIbd in BIN_OUT mode.
// Wire format (reference writer HMM.cpp:383-401 header, 1146-1176 records; reference reader
// BinaryDataReader.hpp:87-174): header = 3 bools (has length, has posterior-mean age, has MAP age), int chr,
// uint n, then n x {uint len, famId bytes, uint len, iid bytes}; record = uint ind1, u8 hap1, uint ind2, u8 hap2,
// int bpStart, int bpEnd, [float length_cM], float score, [float postMean], [float MAP].
#pragma once

#include <zlib.h>

#include <cstdint>
#include <iomanip>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fsmc_host
{

struct IbdPairDataLine {
  std::string ind1FamId = "0_00", ind1Id = "0_00";
  int ind1Hap = -1;
  std::string ind2FamId = "0_00", ind2Id = "0_00";
  int ind2Hap = -1;
  int chromosome = -1;
  int ibdStart = -1, ibdEnd = -1;
  float lengthInCentimorgans = -1.f, ibdScore = -1.f, postEst = -1.f, mapEst = -1.f;

  std::string toString() const
  {
    std::stringstream line;
    line << std::setprecision(std::numeric_limits<float>::digits10 + 1);
    line << ind1FamId << '\t' << ind1Id << '\t' << ind1Hap << '\t' << ind2FamId << '\t' << ind2Id << '\t' << ind2Hap
         << '\t' << chromosome << '\t' << ibdStart << '\t' << ibdEnd;
    if (lengthInCentimorgans != -1.f) line << '\t' << lengthInCentimorgans;
    line << '\t' << ibdScore;
    if (postEst != -1.f) line << '\t' << postEst;
    if (mapEst != -1.f) line << '\t' << mapEst;
    return line.str();
  }
};

class BinaryDataReader
{
public:
  explicit BinaryDataReader(const std::string& binaryFile)
  {
    mFile = gzopen(binaryFile.c_str(), "rb");
    if (!mFile) {
      throw std::runtime_error("cannot open " + binaryFile);
    }
    readHeader();
    peek();
  }
  ~BinaryDataReader()
  {
    if (mFile) gzclose(mFile);
  }
  BinaryDataReader(const BinaryDataReader&) = delete;
  BinaryDataReader& operator=(const BinaryDataReader&) = delete;

  bool moreLinesInFile() const { return mMore; }

  IbdPairDataLine getNextLine()
  {
    if (!mMore) {
      throw std::runtime_error("no more records in the binary IBD file");
    }
    IbdPairDataLine line;
    const unsigned ind1 = mNextInd1;
    unsigned ind2 = 0;
    std::uint_least8_t hap1 = 0, hap2 = 0;
    get(&hap1, sizeof(hap1));
    get(&ind2, sizeof(ind2));
    get(&hap2, sizeof(hap2));
    get(&line.ibdStart, sizeof(int));
    get(&line.ibdEnd, sizeof(int));
    if (mHasLength) get(&line.lengthInCentimorgans, sizeof(float));
    get(&line.ibdScore, sizeof(float));
    if (mHasPostMean) get(&line.postEst, sizeof(float));
    if (mHasMap) get(&line.mapEst, sizeof(float));
    line.ind1Hap = static_cast<int>(hap1);
    line.ind2Hap = static_cast<int>(hap2);
    line.chromosome = mChr;
    line.ind1FamId = mFamIds.at(ind1);
    line.ind1Id = mIIds.at(ind1);
    line.ind2FamId = mFamIds.at(ind2);
    line.ind2Id = mIIds.at(ind2);
    peek();
    return line;
  }

private:
  void get(void* dst, unsigned n)
  {
    if (gzread(mFile, dst, n) < static_cast<int>(n)) {
      throw std::runtime_error("truncated binary IBD file");
    }
  }
  void readHeader()
  {
    get(&mHasLength, sizeof(bool));
    get(&mHasPostMean, sizeof(bool));
    get(&mHasMap, sizeof(bool));
    get(&mChr, sizeof(int));
    unsigned n = 0;
    get(&n, sizeof(unsigned));
    mFamIds.reserve(n);
    mIIds.reserve(n);
    for (unsigned i = 0; i < n; ++i) {
      for (std::vector<std::string>* dst : {&mFamIds, &mIIds}) {
        unsigned len = 0;
        get(&len, sizeof(unsigned));
        std::string s(len, '\0');
        if (len) get(&s[0], len);
        dst->push_back(std::move(s));
      }
    }
  }
  void peek() { mMore = gzread(mFile, &mNextInd1, sizeof(unsigned)) == static_cast<int>(sizeof(unsigned)); }

  gzFile mFile = nullptr;
  bool mHasLength = false, mHasPostMean = false, mHasMap = false;
  int mChr = -1;
  std::vector<std::string> mFamIds, mIIds;
  unsigned mNextInd1 = 0;
  bool mMore = true;
};

} // namespace fsmc_host
