// data.cpp -- input surface of the decode path (reference: Data.cpp).  Errors that make the
// reference print and exit(1) are thrown as std::runtime_error here (a library must not exit).
#include "data.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <memory>
#include <iostream>
#include <mutex>
#include <random>
#include <stdexcept>
#include <thread>

#include "util.hpp"

namespace fsmc_host
{

namespace
{

std::string findWithSuffix(const std::string& root, const std::vector<std::string>& suffixes, const char* what)
{
  for (const std::string& s : suffixes) {
    if (fileExists(root + s)) {
      return root + s;
    }
  }
  std::string msg = std::string("ERROR. Could not find ") + what + " file in";
  for (const std::string& s : suffixes) {
    msg += " " + root + s;
  }
  throw std::runtime_error(msg);
}

std::string hapsPath(const std::string& root)
{
  return findWithSuffix(root, {".hap.gz", ".hap", ".haps.gz", ".haps"}, "hap");
}
std::string samplesPath(const std::string& root)
{
  return findWithSuffix(root, {".samples", ".sample"}, "sample");
}
std::string mapPath(const std::string& root)
{
  return findWithSuffix(root, {".map.gz", ".map"}, "map");
}

bool isSamplesHeader(const std::vector<std::string>& t)
{
  return t.size() >= 3 && ((t[0] == "ID_1" && t[1] == "ID_2" && t[2] == "missing") ||
                           (t[0] == "0" && t[1] == "0" && t[2] == "0"));
}

// Splits "chr snpID bp alleleA alleleB<rest>" like `stream >> a >> b >> c >> d >> e; getline(rest)`.
bool splitHapsLine(const std::string& line, std::string (&field)[5], size_t& restBegin)
{
  size_t p = 0;
  const size_t n = line.size();
  for (int f = 0; f < 5; ++f) {
    while (p < n && std::isspace(static_cast<unsigned char>(line[p]))) {
      ++p;
    }
    if (p >= n) {
      return false;
    }
    const size_t b = p;
    while (p < n && !std::isspace(static_cast<unsigned char>(line[p]))) {
      ++p;
    }
    field[f].assign(line, b, p - b);
  }
  restBegin = p;
  return true;
}

// Host threads of the start-up work (parsing, bit transposes, the emission preparation's shuffles): the machine's, at
// most 16 (a GPU box gives a rank that many), FSMC_HOST_THREADS overrides.
unsigned hostThreads()
{
  if (const char* e = std::getenv("FSMC_HOST_THREADS")) {
    const int v = std::atoi(e);
    if (v >= 1) {
      return static_cast<unsigned>(std::min(v, 256));
    }
  }
  const unsigned hw = std::thread::hardware_concurrency();
  return std::max(1u, std::min(hw ? hw : 1u, 16u));
}

// body(i) for i in [0, n), dynamically scheduled over the host threads; the first exception is rethrown
template <typename F> void parallelFor(size_t n, F&& body)
{
  const unsigned T = static_cast<unsigned>(std::min<size_t>(hostThreads(), n));
  if (T <= 1) {
    for (size_t i = 0; i < n; ++i) {
      body(i);
    }
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr err;
  std::mutex errMutex;
  auto work = [&]() {
    try {
      for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
        body(i);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(errMutex);
      if (!err) {
        err = std::current_exception();
      }
      next.store(n);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < T; ++t) {
    pool.emplace_back(work);
  }
  work();
  for (std::thread& t : pool) {
    t.join();
  }
  if (err) {
    std::rethrow_exception(err);
  }
}

// In-place transpose of a 64 x 64 bit matrix held as 64 words (row i = word i, column j = bit j).
void transpose64(uint64_t (&a)[64])
{
  uint64_t m = 0x00000000FFFFFFFFull;
  for (int j = 32; j != 0; j >>= 1, m ^= (m << j)) {
    for (int k = 0; k < 64; k = (k + j + 1) & ~j) {
      const uint64_t t = ((a[k] >> j) ^ a[k + j]) & m;
      a[k] ^= t << j;
      a[k + j] ^= t;
    }
  }
}

// The allele characters of a haps line, packed: `a` points at the separator in front of the first allele, the allele of
// haplotype i is a[2*i + 1] ('0' or '1', Data.cpp:465-471).  Writes ceil(n / 64) words (bit i = haplotype i carries '1'),
// returns false when a character is neither.
bool packAlleles(const char* a, size_t n, uint64_t* out)
{
  const size_t words = (n + 63) / 64;
  unsigned char bad = 0;
  size_t i = 0;
  for (size_t w = 0; w < words; ++w) {
    uint64_t acc = 0;
    const size_t end = std::min(n, (w + 1) * 64);
    int bit = 0;
    // four haplotypes per 8-byte load: their characters sit in bytes 1, 3, 5, 7; '0' = 0x30, '1' = 0x31
    for (; i + 4 <= end; i += 4, bit += 4) {
      uint64_t x;
      std::memcpy(&x, a + 2 * i, 8);
      const uint64_t c = (x >> 8) & 0x00FF00FF00FF00FFull;
      bad |= static_cast<unsigned char>((((c & 0x00FE00FE00FE00FEull) ^ 0x0030003000300030ull) != 0) ? 1 : 0);
      const uint64_t y = c & 0x0001000100010001ull;
      // (bits 0, 16, 32, 48 of y land on bits 48, 49, 50, 51 of the product; the other partial products fall on distinct
      //  lower bits or beyond bit 63: no carries)
      acc |= (((y * 0x0001000200040008ull) >> 48) & 0xFull) << bit;
    }
    for (; i < end; ++i, ++bit) {
      const unsigned char ch = static_cast<unsigned char>(a[2 * i + 1]);
      bad |= static_cast<unsigned char>((ch & 0xFE) != 0x30);
      acc |= static_cast<uint64_t>(ch & 1u) << bit;
    }
    out[w] = acc;
  }
  return bad == 0;
}

// dst bits [dstPos, dstPos + n) = src bits [srcPos, srcPos + n) (dst bits there are zero on entry)
void copyBits(uint64_t* dst, size_t dstPos, const uint64_t* src, size_t srcPos, size_t n)
{
  while (n) {
    const size_t so = srcPos & 63, dof = dstPos & 63;
    const size_t take = std::min<size_t>(n, std::min<size_t>(64 - so, 64 - dof));
    uint64_t v = src[srcPos >> 6] >> so;
    if (take < 64) {
      v &= (1ull << take) - 1;
    }
    dst[dstPos >> 6] |= v << dof;
    srcPos += take;
    dstPos += take;
    n -= take;
  }
}

} // namespace

int Data::countHapLines(const std::string& inFileRoot)
{
  LineReader br(hapsPath(inFileRoot));
  std::string line;
  int n = 0;
  while (br.getline(line)) {
    n++;
  }
  return n;
}

int Data::countSamplesLines(const std::string& inFileRoot)
{
  LineReader br(samplesPath(inFileRoot));
  std::string line;
  int n = 0;
  while (br.getline(line)) {
    const auto t = splitWhitespace(line);
    if (t.empty() || isSamplesHeader(t)) {
      continue;
    }
    n++;
  }
  return n;
}

void Data::setupJobWindows(int jobID, int jobs)
{
  // Data.cpp:46-80: the job grid is over windows of individuals; window size is the side of a square
  mJobbing = (jobID != -1) && (jobs != -1);
  if (!mJobbing) {
    return;
  }
  const double n = static_cast<double>(sampleSize);
  windowSize = static_cast<int>(std::ceil(std::sqrt((2. * std::pow(n, 2) - n) * 2. / jobs)));
  if (windowSize % 2 != 0) {
    windowSize++;
  }
  w_i = 1;
  int cptJob = 1;
  int cptTotJob = 1;
  while (cptTotJob < jobID) {
    w_i++;
    cptJob += 2;
    cptTotJob += cptJob;
  }
  w_j = static_cast<int>(std::ceil(static_cast<float>(cptJob - (cptTotJob - jobID)) / 2));
  is_j_above_diag = (cptJob - (cptTotJob - jobID)) % 2 == 1;
}

bool Data::readSample(unsigned d, int jobID, int jobs) const
{
  if (!mJobbing) {
    return true;
  }
  return (d >= static_cast<unsigned>((w_i - 1) * windowSize) / 2 && d < static_cast<unsigned>(w_i * windowSize) / 2) ||
         (d >= static_cast<unsigned>((w_j - 1) * windowSize) / 2 && d < static_cast<unsigned>(w_j * windowSize) / 2) ||
         (jobs == jobID && d >= static_cast<unsigned>((w_j - 1) * windowSize) / 2);
}

void Data::readSamplesList(const std::string& inFileRoot, int jobID, int jobs)
{
  LineReader br(samplesPath(inFileRoot));
  std::string line;
  unsigned linesProcessed = 0;
  while (br.getline(line)) {
    const auto t = splitWhitespace(line);
    if (t.empty() || isSamplesHeader(t)) {
      continue;
    }
    if (t.size() < 2) {
      throw std::runtime_error("ERROR: malformed samples line: " + line);
    }
    if (readSample(linesProcessed, jobID, jobs)) {
      FamIDList.push_back(t[0]);
      IIDList.push_back(t[1]);
      famAndIndNameList.push_back(t[0] + "\t" + t[1]);
      globalIndIndex.push_back(linesProcessed);
    }
    linesProcessed++;
  }
}

void Data::allocateBits()
{
  wordsPerHap = (static_cast<size_t>(sites) + 63) / 64;
  bits.assign(numHapRows() * wordsPerHap, 0ull);
}

Data::Data(const DecodingParams& params)
{
  const std::string& root = params.inFileRoot;
  foldToMinorAlleles = params.foldData;
  decodingUsesCSFS = params.usingCSFS;

  // FastSMC mode reads the haps file ONCE (the reference counts its lines first, Data.cpp:264-287, then reads it, then
  // reads it again for the hashing words): the number of sites is known when the single pass ends
  sites = params.FastSMC ? 0 : countHapLines(root);
  sampleSize = static_cast<unsigned long>(countSamplesLines(root));
  haploidSampleSize = sampleSize * 2ul;
  siteWasFlippedDuringFolding.assign(static_cast<size_t>(sites), false);

  // (Data.cpp:62-70 seeds the process-global generator; here the object's own, util.hpp)
  if (params.useKnownSeed) {
    rng.seed(1234u);
  } else {
    std::random_device rd;
    rng.seed(rd());
  }
  setupJobWindows(params.jobInd, params.jobs);
  readSamplesList(root, params.jobInd, params.jobs);
  if (params.FastSMC) {
    const auto geneticMap = readMapFastSMC(root);
    readHapsFastSMC(root, params.jobInd, params.jobs, geneticMap); // (sets `sites`, allocates and fills the bit matrix)
  } else {
    allocateBits();
    readHapsAsmc(root);
    readMapAsmc(root);
  }
}

std::vector<std::pair<unsigned long, double>> Data::readMapFastSMC(const std::string& inFileRoot)
{
  // Data.cpp:98-141: columns "bp <ignored> cM"; a row whose first field is not an integer is a header
  LineReader br(mapPath(inFileRoot));
  std::vector<std::pair<unsigned long, double>> geneticMap;
  std::string line;
  std::string field[3];
  while (br.getline(line)) {
    std::stringstream ss(line);
    ss >> field[0] >> field[1] >> field[2]; // fields keep their previous value on short lines, like the reference
    if (field[0].empty()) {
      continue;
    }
    try {
      (void)std::stoi(field[0]);
    } catch (const std::invalid_argument&) {
      continue;
    }
    geneticMap.emplace_back(std::stol(field[0]), std::stod(field[2]));
  }
  if (geneticMap.empty()) {
    throw std::runtime_error("ERROR: genetic map " + inFileRoot + ".map[.gz] has no usable rows");
  }
  return geneticMap;
}

void Data::addMarker(unsigned long physicalPos, double geneticPos, unsigned pos)
{
  // Data.cpp:549-565
  geneticPositions.push_back(static_cast<float>(geneticPos / 100.f));
  physicalPositions.push_back(static_cast<int>(physicalPos));
  if (pos > 0) {
    const double genDistFromPrevious = geneticPositions[pos] - geneticPositions[pos - 1];
    const unsigned long physDistFromPrevious =
        static_cast<unsigned long>(physicalPositions[pos] - physicalPositions[pos - 1]);
    const float recRate = static_cast<float>(genDistFromPrevious / physDistFromPrevious);
    if (pos == 1) {
      recRateAtMarker.push_back(recRate);
    }
    recRateAtMarker.push_back(recRate);
  }
}

void Data::addMarkerFromMap(unsigned long bp, const std::vector<std::pair<unsigned long, double>>& map,
                            unsigned& cur, unsigned pos)
{
  // Data::readGeneticMap (Data.cpp:523-547): exact hit, before-first, or linear interpolation
  while (bp > map[cur].first && cur < map.size() - 1) {
    cur++;
  }
  double cm;
  if (bp >= map[cur].first || cur == 0) {
    cm = map[cur].second;
  } else {
    cm = map[cur - 1].second + (bp - map[cur - 1].first) * (map[cur].second - map[cur - 1].second) /
                                   (map[cur].first - map[cur - 1].first);
  }
  addMarker(bp, cm, pos);
}

void Data::readHapsFastSMC(const std::string& inFileRoot, int jobID, int jobs,
                           const std::vector<std::pair<unsigned long, double>>& geneticMap)
{
  // Data.cpp:397-521, as ONE pass over the file and on every host core: a reader thread inflates the file and cuts it
  // into blocks of whole lines; worker threads parse the blocks -- fields, allele count, the loaded haplotypes' alleles
  // packed 64 to a word, one row of words per SITE --; the blocks are then joined in file order (position checks,
  // genetic-map interpolation: the sequential part, Data.cpp:523-565) and the site-major bit rows transposed into the
  // haplotype-major matrix the device wants.  At the 10 000-haplotype x 100 000-site shape the text is 2 GB: the
  // reference's two to three sequential passes over it (Data.cpp:264-287, 397-521, FastSMC.cpp:144-227) are what a
  // multi-GPU run would wait for.
  const size_t nHapsFile = 2 * sampleSize;
  const size_t wordsFile = (nHapsFile + 63) / 64;
  // the loaded individuals as ranges of haplotype columns (job windows select up to three ranges, Data.cpp:251-262)
  std::vector<std::pair<size_t, size_t>> ranges; // [first hap, last hap) of the file
  for (unsigned d = 0; d < sampleSize; d++) {
    if (readSample(d, jobID, jobs)) {
      if (!ranges.empty() && ranges.back().second == 2 * static_cast<size_t>(d)) {
        ranges.back().second += 2;
      } else {
        ranges.emplace_back(2 * static_cast<size_t>(d), 2 * static_cast<size_t>(d) + 2);
      }
    }
  }
  const size_t nLoaded = numHapRows();
  const size_t wordsLoaded = (nLoaded + 63) / 64;
  const int totalSamples = static_cast<int>(2 * sampleSize);

  struct Block {
    std::vector<char> text; // whole lines, the last one newline-terminated or the file's last
    size_t nLines = 0;      // lines of the block (every line counts as a site, like countHapLines)
    size_t index = 0;       // position of the block in the file
    // filled by the worker:
    size_t nRows = 0; // lines that are haps rows (five fields)
    std::vector<unsigned long> bp;
    std::vector<int> da;
    std::vector<char> flipped;
    std::vector<uint64_t> rows; // [nRows][wordsLoaded]
    std::string firstChrField;
  };
  std::deque<std::unique_ptr<Block>> todo;
  std::vector<std::unique_ptr<Block>> done; // by block index
  std::mutex mu;
  std::condition_variable cvWork, cvRoom;
  bool eof = false;
  std::exception_ptr err;
  const unsigned T = hostThreads();
  const size_t maxQueued = 2 * static_cast<size_t>(T) + 2;
  size_t blocksMade = 0;

  auto parseBlock = [&](Block& blk) {
    blk.bp.reserve(blk.nLines);
    blk.da.reserve(blk.nLines);
    blk.flipped.reserve(blk.nLines);
    blk.rows.assign(blk.nLines * wordsLoaded, 0ull);
    std::vector<uint64_t> all(wordsFile);
    const char* p = blk.text.data();
    const char* const end = p + blk.text.size();
    while (p < end) {
      const char* nl = static_cast<const char*>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
      const char* le = nl ? nl : end; // line = [p, le)
      // the five leading fields (Data.cpp:413-428)
      const char* q = p;
      const char* fb[5];
      const char* fe[5];
      bool ok = true;
      for (int f = 0; f < 5; ++f) {
        while (q < le && std::isspace(static_cast<unsigned char>(*q))) {
          ++q;
        }
        if (q >= le) {
          ok = false;
          break;
        }
        fb[f] = q;
        while (q < le && !std::isspace(static_cast<unsigned char>(*q))) {
          ++q;
        }
        fe[f] = q;
      }
      if (ok) {
        const size_t restLen = static_cast<size_t>(le - q);
        if (!(restLen == 4 * sampleSize || restLen == 4 * sampleSize + 1)) {
          throw std::runtime_error("ERROR: haps line has wrong length. Length is " + std::to_string(restLen) +
                                   ", but should be 4 * " + std::to_string(sampleSize));
        }
        if (blk.nRows == 0) {
          blk.firstChrField.assign(fb[0], fe[0]);
        }
        blk.bp.push_back(std::stoul(std::string(fb[2], fe[2])));
        if (!packAlleles(q, nHapsFile, all.data())) {
          throw std::runtime_error("ERROR: hap is not '0' or '1'");
        }
        int DAcount = 0;
        for (size_t w = 0; w < wordsFile; ++w) {
          DAcount += __builtin_popcountll(all[w]);
        }
        const bool minorAlleleValue = foldToMinorAlleles ? (DAcount <= totalSamples - DAcount) : true;
        blk.flipped.push_back(minorAlleleValue ? 0 : 1);
        blk.da.push_back(foldToMinorAlleles ? std::min(DAcount, totalSamples - DAcount) : DAcount);
        if (!minorAlleleValue) { // the stored bit says "carries the minor allele"
          for (size_t w = 0; w < wordsFile; ++w) {
            all[w] = ~all[w];
          }
        }
        uint64_t* row = &blk.rows[blk.nRows * wordsLoaded];
        size_t dst = 0;
        for (const auto& r : ranges) {
          copyBits(row, dst, all.data(), r.first, r.second - r.first);
          dst += r.second - r.first;
        }
        blk.nRows++;
      }
      p = nl ? nl + 1 : end;
    }
    blk.rows.resize(blk.nRows * wordsLoaded);
    std::vector<char>().swap(blk.text);
  };

  auto worker = [&]() {
    for (;;) {
      std::unique_ptr<Block> blk;
      size_t idx = 0;
      {
        std::unique_lock<std::mutex> lock(mu);
        cvWork.wait(lock, [&] { return !todo.empty() || eof || err; });
        if (err || (todo.empty() && eof)) {
          return;
        }
        blk = std::move(todo.front());
        todo.pop_front();
        idx = blk->index;
        cvRoom.notify_one();
      }
      try {
        parseBlock(*blk);
      } catch (...) {
        std::lock_guard<std::mutex> lock(mu);
        if (!err) {
          err = std::current_exception();
        }
        cvWork.notify_all();
        cvRoom.notify_all();
        return;
      }
      std::lock_guard<std::mutex> lock(mu);
      if (done.size() <= idx) {
        done.resize(idx + 1);
      }
      done[idx] = std::move(blk);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < T; ++t) {
    pool.emplace_back(worker);
  }
  // the reader: inflate, cut at line ends
  try {
    gzFile f = gzopen(hapsPath(inFileRoot).c_str(), "rb");
    if (!f) {
      throw std::runtime_error("ERROR: could not open " + hapsPath(inFileRoot));
    }
    gzbuffer(f, 1u << 20);
    size_t kBlockBytes = 8u << 20;
    if (const char* e = std::getenv("FSMC_HOST_BLOCK_BYTES")) { // tests: many blocks from a small file
      const long v = std::atol(e);
      if (v >= 64) {
        kBlockBytes = static_cast<size_t>(v);
      }
    }
    std::vector<char> carry;
    bool more = true;
    while (more) {
      auto blk = std::make_unique<Block>();
      blk->text.resize(carry.size() + kBlockBytes);
      std::memcpy(blk->text.data(), carry.data(), carry.size());
      size_t have = carry.size();
      carry.clear();
      const int got = gzread(f, blk->text.data() + have, static_cast<unsigned>(kBlockBytes));
      if (got < 0) {
        gzclose(f);
        throw std::runtime_error("ERROR: could not read " + hapsPath(inFileRoot));
      }
      have += static_cast<size_t>(got);
      more = got > 0 && !gzeof(f);
      if (got == 0) {
        more = false;
      }
      blk->text.resize(have);
      if (more) { // keep the unfinished last line for the next block
        size_t cut = have;
        while (cut > 0 && blk->text[cut - 1] != '\n') {
          --cut;
        }
        carry.assign(blk->text.begin() + static_cast<long>(cut), blk->text.end());
        blk->text.resize(cut);
      }
      if (blk->text.empty()) {
        continue;
      }
      size_t lines = 0;
      for (const char* c = blk->text.data(), *e = c + blk->text.size(); c < e;) {
        const char* nl = static_cast<const char*>(std::memchr(c, '\n', static_cast<size_t>(e - c)));
        lines++;
        c = nl ? nl + 1 : e;
      }
      blk->nLines = lines;
      blk->index = blocksMade++;
      std::unique_lock<std::mutex> lock(mu);
      cvRoom.wait(lock, [&] { return todo.size() < maxQueued || err; });
      if (err) {
        break;
      }
      todo.push_back(std::move(blk));
      cvWork.notify_one();
    }
    gzclose(f);
  } catch (...) {
    std::lock_guard<std::mutex> lock(mu);
    if (!err) {
      err = std::current_exception();
    }
  }
  {
    std::lock_guard<std::mutex> lock(mu);
    eof = true;
  }
  cvWork.notify_all();
  for (std::thread& t : pool) {
    t.join();
  }
  if (err) {
    std::rethrow_exception(err);
  }

  // ---- join the blocks in file order (the sequential part: Data.cpp:431-452, 523-565)
  size_t nLinesTotal = 0, nRowsTotal = 0;
  for (const auto& blk : done) {
    nLinesTotal += blk->nLines;
    nRowsTotal += blk->nRows;
  }
  sites = static_cast<int>(nLinesTotal);
  if (nRowsTotal != nLinesTotal) {
    throw std::runtime_error("ERROR: read " + std::to_string(nRowsTotal) + " haps rows, expected " + std::to_string(sites));
  }
  siteWasFlippedDuringFolding.assign(static_cast<size_t>(sites), false);
  totalSamplesCount.assign(static_cast<size_t>(sites), totalSamples);
  derivedAlleleCounts.assign(static_cast<size_t>(sites), 0);
  geneticPositions.reserve(static_cast<size_t>(sites));
  physicalPositions.reserve(static_cast<size_t>(sites));
  recRateAtMarker.reserve(static_cast<size_t>(sites));
  unsigned long largestBp = 0;
  unsigned pos = 0, curG = 0;
  std::vector<size_t> firstRow(done.size() + 1, 0);
  for (size_t bI = 0; bI < done.size(); ++bI) {
    const Block& blk = *done[bI];
    firstRow[bI + 1] = firstRow[bI] + blk.nRows;
    for (size_t r = 0; r < blk.nRows; ++r, ++pos) {
      const unsigned long bp = blk.bp[r];
      if (bp > largestBp) {
        largestBp = bp;
      } else {
        throw std::runtime_error("ERROR: rows in haps data file must be ordered by increasing physical position, but "
                                 "two consecutive values were " + std::to_string(largestBp) + " and " +
                                 std::to_string(bp));
      }
      if (pos == 0) {
        const std::string chr = blk.firstChrField.substr(0, blk.firstChrField.find(':'));
        try {
          chrNumber = std::stoi(chr);
        } catch (const std::exception&) {
          chrNumber = 0;
        }
        if (chrNumber <= 0 || chrNumber > 1260) {
          chrNumber = 0;
        }
      }
      addMarkerFromMap(bp, geneticMap, curG, pos);
      siteWasFlippedDuringFolding[pos] = blk.flipped[r] != 0;
      derivedAlleleCounts[pos] = blk.da[r];
    }
  }
  // ---- site-major rows -> the haplotype-major matrix: 64 sites x 64 haplotypes at a time, all cores
  allocateBits();
  const size_t siteBlocks = (static_cast<size_t>(sites) + 63) / 64;
  auto rowOfSite = [&](size_t site) -> const uint64_t* {
    const size_t bI = static_cast<size_t>(std::upper_bound(firstRow.begin(), firstRow.end(), site) - firstRow.begin()) - 1;
    return &done[bI]->rows[(site - firstRow[bI]) * wordsLoaded];
  };
  parallelFor(siteBlocks, [&](size_t sb) {
    const size_t s0 = sb * 64;
    const size_t nS = std::min<size_t>(64, static_cast<size_t>(sites) - s0);
    const uint64_t* src[64];
    for (size_t i = 0; i < nS; ++i) {
      src[i] = rowOfSite(s0 + i);
    }
    for (size_t j = 0; j < wordsLoaded; ++j) {
      uint64_t a[64];
      for (size_t i = 0; i < 64; ++i) {
        a[i] = i < nS ? src[i][j] : 0ull;
      }
      transpose64(a);
      const size_t nH = std::min<size_t>(64, nLoaded - 64 * j);
      for (size_t k = 0; k < nH; ++k) {
        bits[(64 * j + k) * wordsPerHap + sb] = a[k];
      }
    }
  });
}

void Data::readHapsAsmc(const std::string& inFileRoot)
{
  // Data.cpp:320-395: reads the first 2*numIndividuals() haplotype columns and counts alleles over them
  LineReader br(hapsPath(inFileRoot));
  totalSamplesCount.assign(static_cast<size_t>(sites), 0);
  derivedAlleleCounts.assign(static_cast<size_t>(sites), 0);
  std::string line;
  std::string field[5];
  unsigned pos = 0;
  const unsigned nHap = static_cast<unsigned>(numHapRows());
  const int totalSamples = static_cast<int>(nHap);
  while (br.getline(line)) {
    size_t rest = 0;
    if (!splitHapsLine(line, field, rest)) {
      continue;
    }
    const size_t restLen = line.size() - rest;
    if (!(restLen == 4 * sampleSize || restLen == 4 * sampleSize + 1)) {
      throw std::runtime_error("ERROR: haps line has wrong length. Length is " + std::to_string(restLen) +
                               ", should be 4*" + std::to_string(numIndividuals()));
    }
    if (pos >= static_cast<unsigned>(sites)) {
      break;
    }
    const char* a = line.data() + rest;
    int DAcount = 0;
    for (unsigned i = 0; i < nHap; i++) {
      const char c = a[2 * i + 1];
      if (c == '1') {
        DAcount++;
      } else if (c != '0') {
        throw std::runtime_error("ERROR: hap is not '0' or '1'");
      }
    }
    totalSamplesCount[pos] = totalSamples;
    const bool minorAlleleValue = foldToMinorAlleles ? (DAcount <= totalSamples - DAcount) : true;
    siteWasFlippedDuringFolding[pos] = !minorAlleleValue;
    for (unsigned i = 0; i < nHap; i++) {
      if ((a[2 * i + 1] == '1') == minorAlleleValue) {
        setBit(i, pos);
      }
    }
    derivedAlleleCounts[pos] = foldToMinorAlleles ? std::min(DAcount, totalSamples - DAcount) : DAcount;
    pos++;
  }
  if (pos != static_cast<unsigned>(sites)) {
    throw std::runtime_error("ERROR: read " + std::to_string(pos) + " haps rows, expected " + std::to_string(sites));
  }
}

void Data::readMapAsmc(const std::string& inFileRoot)
{
  // Data.cpp:162-210: plink map "chr id cM bp"; gen = stof(cM) / 100.f in float
  LineReader br(mapPath(inFileRoot));
  SNP_IDs.assign(static_cast<size_t>(sites), "");
  geneticPositions.assign(static_cast<size_t>(sites), 0.f);
  recRateAtMarker.assign(static_cast<size_t>(sites), 0.f);
  physicalPositions.assign(static_cast<size_t>(sites), 0);
  std::string line;
  int pos = 0;
  while (br.getline(line)) {
    const auto t = splitWhitespace(line);
    if (t.size() < 4) {
      if (t.empty()) {
        continue;
      }
      throw std::runtime_error("ERROR: malformed map line: " + line);
    }
    if (pos >= sites) {
      pos++;
      break;
    }
    SNP_IDs[pos] = t[1];
    geneticPositions[pos] = refStof(t[2]) / 100.f;
    physicalPositions[pos] = std::stoi(t[3]);
    if (pos > 0) {
      const float genDist = geneticPositions[pos] - geneticPositions[pos - 1];
      const int physDist = physicalPositions[pos] - physicalPositions[pos - 1];
      recRateAtMarker[pos] = genDist / physDist;
    }
    pos++;
  }
  if (pos != sites) {
    throw std::runtime_error("ERROR. Read " + std::to_string(pos) + " from map file, expected " +
                             std::to_string(sites));
  }
}

Data Data::fromArrays(const uint8_t* alleles, size_t nHaps, size_t nSites, const int64_t* bp, const double* cm,
                      bool foldToMinor, bool useKnownSeed, int chrNumber)
{
  if (nHaps == 0 || nHaps % 2 != 0 || nSites == 0) {
    throw std::runtime_error("fromArrays: need an even, positive number of haplotypes and at least one site");
  }
  Data d;
  d.foldToMinorAlleles = foldToMinor;
  d.decodingUsesCSFS = true;
  d.sites = static_cast<int>(nSites);
  d.sampleSize = nHaps / 2;
  d.haploidSampleSize = nHaps;
  d.chrNumber = chrNumber;
  d.siteWasFlippedDuringFolding.assign(nSites, false);
  if (useKnownSeed) {
    d.rng.seed(1234u);
  } else {
    std::random_device rd;
    d.rng.seed(rd());
  }
  d.setupJobWindows(1, 1);
  for (size_t i = 0; i < nHaps / 2; ++i) {
    const std::string id = "1_" + std::to_string(i + 1);
    d.FamIDList.push_back(id);
    d.IIDList.push_back(id);
    d.famAndIndNameList.push_back(id + "\t" + id);
    d.globalIndIndex.push_back(static_cast<unsigned>(i));
  }
  d.allocateBits();
  d.totalSamplesCount.assign(nSites, static_cast<int>(nHaps));
  d.derivedAlleleCounts.assign(nSites, 0);
  // The input is haplotype-major: both passes walk it along its rows, on every host thread (a walk down the columns --
  // one cache line per allele -- took 16 s for 10 000 haplotypes x 100 000 sites; this takes under one).
  // pass 1: derived-allele count of every site, a block of 4096 sites per work item
  constexpr size_t kSiteBlock = 4096;
  const size_t nBlocks = (nSites + kSiteBlock - 1) / kSiteBlock;
  std::vector<int> count(nSites, 0);
  parallelFor(nBlocks, [&](size_t b) {
    const size_t s0 = b * kSiteBlock, s1 = std::min(nSites, s0 + kSiteBlock);
    int* c = count.data();
    for (size_t h = 0; h < nHaps; ++h) {
      const uint8_t* row = alleles + h * nSites;
      for (size_t s = s0; s < s1; ++s) {
        c[s] += row[s] ? 1 : 0;
      }
    }
  });
  const int total = static_cast<int>(nHaps);
  std::vector<uint8_t> minorValue(nSites, 1); // the allele value stored as 1 (the minor allele when folding)
  for (size_t s = 0; s < nSites; ++s) {
    d.addMarker(static_cast<unsigned long>(bp[s]), cm[s], static_cast<unsigned>(s));
    const int DAcount = count[s];
    const bool minorAlleleValue = foldToMinor ? (DAcount <= total - DAcount) : true;
    d.siteWasFlippedDuringFolding[s] = !minorAlleleValue;
    minorValue[s] = minorAlleleValue ? 1 : 0;
    d.derivedAlleleCounts[s] = foldToMinor ? std::min(DAcount, total - DAcount) : DAcount;
  }
  // pass 2: a haplotype's row of words, 64 sites to a word
  parallelFor(nHaps, [&](size_t h) {
    const uint8_t* row = alleles + h * nSites;
    uint64_t* out = d.bits.data() + h * d.wordsPerHap;
    for (size_t w = 0; w * 64 < nSites; ++w) {
      const size_t s0 = w * 64, n = std::min<size_t>(64, nSites - s0);
      uint64_t word = 0;
      for (size_t i = 0; i < n; ++i) {
        word |= static_cast<uint64_t>((row[s0 + i] != 0) == (minorValue[s0 + i] != 0)) << i;
      }
      out[w] = word;
    }
  });
  return d;
}

std::vector<bool> Data::genotypeVector(size_t hapRow) const
{
  std::vector<bool> v(static_cast<size_t>(sites));
  for (size_t s = 0; s < v.size(); ++s) {
    v[s] = genotype(hapRow, s);
  }
  return v;
}

std::vector<Individual> Data::individuals() const
{
  std::vector<Individual> out(numIndividuals(), Individual(sites));
  for (size_t i = 0; i < out.size(); ++i) {
    out[i].genotype1 = genotypeVector(2 * i);
    out[i].genotype2 = genotypeVector(2 * i + 1);
  }
  return out;
}

std::vector<std::vector<int>> Data::calculateUndistinguishedCounts(const int numCsfsSamples) const
{
  // Data.cpp:567-599.  A draw shuffles a vector of totalSamples - 2 entries with a generator seeded from std::rand()
  // (Data.cpp:144-160; here the object's own glibc-compatible generator: util.hpp, GlibcRand): 3 x sites shuffles of cohort size -- 3 G element moves at the 10 000-haplotype x 100 000-site
  // shape.  Only the SEEDS depend on each other (the std::rand() sequence, in site and `distinguished` order, drawn only
  // where the reference draws): they are taken sequentially, the shuffles then run on every host core.
  const size_t n = derivedAlleleCounts.size();
  std::vector<std::vector<int>> undistinguished(n, std::vector<int>(3));
  struct Draw {
    uint32_t site;
    int distinguished;
    int seed;
  };
  std::vector<Draw> draws;
  draws.reserve(3 * n);
  for (size_t i = 0; i < n; ++i) {
    const int derivedAlleles = derivedAlleleCounts[i];
    const int totalSamples = totalSamplesCount[i];
    if (decodingUsesCSFS && numCsfsSamples > totalSamples) {
      throw std::runtime_error("ERROR. SNP with numerical ID " + std::to_string(i) + " has " +
                               std::to_string(totalSamples) + " non-missing individuals, but the CSFS requires " +
                               std::to_string(numCsfsSamples));
    }
    if (foldToMinorAlleles && derivedAlleles > totalSamples - derivedAlleles) {
      throw std::runtime_error("Minor alleles has frequency > 50%. Data is supposed to be folded.");
    }
    for (int distinguished = 0; distinguished < 3; distinguished++) {
      const int successes = derivedAlleles - distinguished;
      if (successes < 0 || successes > totalSamples - 2) {
        undistinguished[i][static_cast<size_t>(distinguished)] = -1; // (no draw: Data.cpp:146-148)
      } else {
        draws.push_back(Draw{static_cast<uint32_t>(i), distinguished, rng.next()});
      }
    }
  }
  constexpr size_t kGrain = 64; // draws per task
  parallelFor((draws.size() + kGrain - 1) / kGrain, [&](size_t task) {
    std::vector<unsigned short> samplingVector;
    const size_t lo = task * kGrain, hi = std::min(draws.size(), lo + kGrain);
    for (size_t d = lo; d < hi; ++d) {
      const Draw& dr = draws[d];
      const int populationSize = totalSamplesCount[dr.site] - 2;
      const int successes = derivedAlleleCounts[dr.site] - dr.distinguished;
      const int sampleSizeCsfs = numCsfsSamples - 2;
      samplingVector.assign(static_cast<size_t>(populationSize), 0);
      for (int i = 0; i < successes; i++) {
        samplingVector[static_cast<size_t>(i)] = 1;
      }
      std::shuffle(samplingVector.begin(), samplingVector.end(), std::mt19937(static_cast<unsigned>(dr.seed)));
      int ret = 0;
      for (int i = 0; i < sampleSizeCsfs; i++) {
        ret += samplingVector[static_cast<size_t>(i)];
      }
      undistinguished[dr.site][static_cast<size_t>(dr.distinguished)] = ret;
    }
  });
  for (size_t i = 0; i < n; ++i) {
    for (int distinguished = 0; distinguished < 3; distinguished++) {
      int& sample = undistinguished[i][static_cast<size_t>(distinguished)];
      if (foldToMinorAlleles && (sample + distinguished > numCsfsSamples / 2)) {
        sample = numCsfsSamples - 2 - sample;
      }
    }
  }
  return undistinguished;
}

} // namespace fsmc_host
