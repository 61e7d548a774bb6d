// undist_counts.cpp -- oracle restatement of the RNG-dependent part of emission preparation.
// TEST INFRASTRUCTURE ONLY (see hmm_oracle.h).
//
// The reference draws "undistinguished" allele counts with glibc rand() seeding a libstdc++
// mt19937 that drives std::shuffle (Data.cpp:144-160), after std::srand(1234) when
// useKnownSeed is set (Data.cpp:55-60).  The resulting emissions therefore depend on the
// exact libc/libstdc++ algorithms, so this piece is C++ and calls the same std functions
// in the same order (Data.cpp:567-599).
#include <algorithm>
#include <cstdlib>
#include <random>
#include <vector>

namespace
{
// Data.cpp:144-160
int sample_hypergeometric(int populationSize, int numberOfSuccesses, int sampleSize)
{
  if (numberOfSuccesses < 0 || numberOfSuccesses > populationSize) {
    return -1; // no RNG draw on this path
  }
  std::vector<unsigned short> urn(static_cast<size_t>(populationSize), 0);
  std::fill(urn.begin(), urn.begin() + numberOfSuccesses, static_cast<unsigned short>(1));
  std::shuffle(urn.begin(), urn.end(), std::mt19937(std::rand()));
  int drawn = 0;
  for (int i = 0; i < sampleSize; ++i) {
    drawn += urn[static_cast<size_t>(i)];
  }
  return drawn;
}
} // namespace

extern "C" {

// Data.cpp:55-60 (known seed) followed by Data.cpp:567-599.
// derived/total: per-site derived (minor if folded) allele count and total haploid samples.
// out: [nSites][3].  Returns 0, or -1 if a site violates the reference's checks
// (CSFS needs more samples than available; folded data with minor count > 50 %).
int fo_undistinguished_counts(const int* derived, const int* total, long nSites, int numCsfsSamples, int fold,
                              int decodingUsesCSFS, int seedKnown, int* out)
{
  if (seedKnown) {
    std::srand(1234u);
  }
  for (long i = 0; i < nSites; ++i) {
    const int derivedAlleles = derived[i];
    const int totalSamples = total[i];
    if (decodingUsesCSFS && numCsfsSamples > totalSamples) {
      return -1;
    }
    if (fold && derivedAlleles > totalSamples - derivedAlleles) {
      return -1;
    }
    for (int distinguished = 0; distinguished < 3; ++distinguished) {
      int sample = sample_hypergeometric(totalSamples - 2, derivedAlleles - distinguished, numCsfsSamples - 2);
      if (fold && (sample + distinguished > numCsfsSamples / 2)) {
        sample = numCsfsSamples - 2 - sample;
      }
      out[i * 3 + distinguished] = sample;
    }
  }
  return 0;
}
}
