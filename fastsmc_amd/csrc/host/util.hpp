// util.hpp -- small host utilities: gz-transparent line reader, tokeniser, number parsing with the
// reference's semantics (StringUtils.cpp:36-44: parse as long double, then narrow).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fsmc_host
{

inline bool fileExists(const std::string& path)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (f) {
    std::fclose(f);
    return true;
  }
  return false;
}

// Reads plain or gzipped text line by line (zlib's gz* functions read both transparently).
class LineReader
{
public:
  explicit LineReader(const std::string& path) : mPath(path)
  {
    mFile = gzopen(path.c_str(), "rb");
    if (!mFile) {
      throw std::runtime_error("ERROR: could not open " + path);
    }
    gzbuffer(mFile, 1u << 20);
  }
  ~LineReader()
  {
    if (mFile) {
      gzclose(mFile);
    }
  }
  LineReader(const LineReader&) = delete;
  LineReader& operator=(const LineReader&) = delete;

  // returns false at end of file; strips the trailing '\n' (a '\r' is kept, like std::getline)
  bool getline(std::string& line)
  {
    line.clear();
    bool any = false;
    for (;;) {
      if (!gzgets(mFile, mBuf, sizeof(mBuf))) {
        return any;
      }
      any = true;
      const size_t n = std::strlen(mBuf);
      if (n && mBuf[n - 1] == '\n') {
        line.append(mBuf, n - 1);
        return true;
      }
      line.append(mBuf, n);
      if (n + 1 < sizeof(mBuf)) {
        return true; // last line without newline
      }
    }
  }

private:
  std::string mPath;
  gzFile mFile = nullptr;
  char mBuf[1 << 16];
};

inline std::vector<std::string> splitWhitespace(const std::string& line)
{
  std::vector<std::string> out;
  std::istringstream iss(line);
  std::string tok;
  while (iss >> tok) {
    out.push_back(tok);
  }
  return out;
}

// StringUtils::stof / stod of the reference (StringUtils.cpp:36-44)
inline float refStof(const std::string& s)
{
  return static_cast<float>(std::stold(s));
}
inline double refStod(const std::string& s)
{
  return static_cast<double>(std::stold(s));
}

inline std::string toLower(std::string s)
{
  for (char& c : s) {
    if (c >= 'A' && c <= 'Z') {
      c = static_cast<char>(c - 'A' + 'a');
    }
  }
  return s;
}

inline uint32_t floatBits(float f)
{
  uint32_t u;
  std::memcpy(&u, &f, sizeof(u));
  return u;
}


// glibc's rand() / srand() (random_r TYPE_3: the additive feedback generator r[i] = r[i-31] + r[i-3] over 32-bit words,
// seeded by the Lehmer sequence 16807 * x mod 2^31-1, the first 310 outputs discarded), as an object of its own.  The
// reference seeds the PROCESS-GLOBAL generator in Data's constructor (Data.cpp:62-70) and draws the seeds of the
// emission preparation from it in HMM's (Data.cpp:144-160): any other thread of the process that calls rand() in between
// -- the HIP runtime does while it initialises -- shifts the sequence.  A Data object carries its own generator instead:
// the same numbers as glibc's for the same seed (tests/test_host_prep.py compares with the oracle, which calls the real
// one), whatever else runs in the process.
class GlibcRand
{
public:
  void seed(unsigned int s)
  {
    if (s == 0) {
      s = 1;
    }
    int32_t word = static_cast<int32_t>(s);
    r_[0] = static_cast<uint32_t>(word);
    for (int i = 1; i < 31; ++i) {
      const long hi = word / 127773;
      const long lo = word % 127773;
      long next = 16807 * lo - 2836 * hi;
      if (next < 0) {
        next += 2147483647;
      }
      word = static_cast<int32_t>(next);
      r_[i] = static_cast<uint32_t>(word);
    }
    for (int i = 31; i < 34; ++i) {
      r_[i] = r_[i - 31];
    }
    n_ = 34;
    for (int k = 0; k < 310; ++k) {
      step();
    }
  }
  int next() { return static_cast<int>(step() >> 1); }

private:
  uint32_t step()
  {
    const uint32_t v = r_[(n_ - 31) % 34] + r_[(n_ - 3) % 34];
    r_[n_ % 34] = v;
    ++n_;
    return v;
  }
  uint32_t r_[34] = {};
  unsigned long long n_ = 34;
};

} // namespace fsmc_host
