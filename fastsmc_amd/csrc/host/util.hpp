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

} // namespace fsmc_host
