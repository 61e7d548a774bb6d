/*
 * hmm_oracle.h -- CPU restatement of FastSMC's pairwise HMM decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/ASMC_SRC/SRC).  The arithmetic is the reference's NO_SSE
 * variant: separately rounded IEEE fp32 multiply/add, exact 1.0f/x division,
 * sequential k-ascending sums (see DESIGN.md "Which reference build").
 *
 * PARITY STATUS: the helpers (rounding, scaling, window padding, bit subsets)
 * are pinned by the reference's own known-answer tests
 * (TESTS/test_hmm_utils.cpp, restated in tests/test_oracle_known_answers.py).
 * The forward/backward/IBD chain is "parity unpinned" against reference golden
 * outputs: every golden regression in the reference needs a
 * *.decodingQuantities.gz blob that is missing from the checkout, and the
 * reference itself cannot be built here without stand-ins for Eigen/Boost.
 */
#ifndef FSMC_HMM_ORACLE_H
#define FSMC_HMM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Constant inputs of the path (HMM.cpp:65-127, DecodingQuantities.hpp:51-68). */
typedef struct {
  int32_t K;             /* states */
  int32_t S;             /* sites (sequenceLength) */
  const float* pi;       /* initialStateProb [K] */
  const float* colRatios;/* columnRatios [K], zero padded */
  const float* expTimes; /* expectedTimes [K] */
  int32_t nRows;         /* rows of the transition tables */
  const float* Dt;       /* [nRows][K] */
  const float* Bt;       /* [nRows][K] */
  const float* Ut;       /* [nRows][K] */
  const float* RRt;      /* [nRows][K] */
  const int32_t* stepRow;/* [S]: table row for the step (p-1 -> p), p >= 1 */
  const float* e1;       /* emission1AtSite      [S][K] */
  const float* e0m1;     /* emission0minus1AtSite[S][K] */
  const float* e2m0;     /* emission2minus0AtSite[S][K] */
  /* sequence mode (decodingSequence, HMM.cpp:760-770, 915-925); all NULL / 0 in array mode.  Rows are indexed by
   * the later site q of the gap (q-1, q):
   *   gapRowF[q]  key roundMorgans(recDist_q - rate[q])   siteRowF[q] key rate[q]      (forward, into site q)
   *   gapRowB[q]  key roundMorgans(recDist_q - rate[q-1]) siteRowB[q] key rate[q-1]    (backward, out of site q)
   * with recDist_q = roundMorgans(gen[q]-gen[q-1]), rate[p] = roundMorgans(recRateAtMarker[p]);
   *   hom[q] = homozygousEmissionMap[roundPhysical(phys[q]-phys[q-1]-1)]  [S][K] */
  int32_t sequence;
  const int32_t* gapRowF;
  const int32_t* siteRowF;
  const int32_t* gapRowB;
  const int32_t* siteRowB;
  const float* hom;
} fo_model;

/* One IBD record as handed to writePairIBD (HMM.cpp:1110-1177). */
typedef struct {
  uint32_t pair;   /* caller-defined ordinal of the pair */
  int32_t start;   /* site index of first site */
  int32_t end;     /* site index of last site (inclusive) */
  float prob;      /* cumulative posterior over the segment ("posteriorIBD") */
  float postMean;  /* getPosteriorMean(per-state sums), 0 if not requested */
  float map;       /* getMAP(per-state sums), 0 if not requested */
} fo_ibd_record;

/* HmmUtils.cpp:65-79 */
float fo_round_morgans(float value, int precision, float min);
/* HmmUtils.cpp:81-94 */
int fo_round_physical(int value, int precision);
/* HmmUtils.cpp:102-130 (NO_SSE branch) */
void fo_calculate_scaling_batch(const float* vec, float* scalings, float* sums, int batchSize, int numStates);
/* HmmUtils.cpp:132-151 */
void fo_apply_scaling_batch(float* vec, const float* scalings, int batchSize, int numStates);
/* HmmUtils.cpp:153-164 */
unsigned fo_get_from_position(const float* gen, unsigned n, unsigned from, float cmDist);
/* HmmUtils.cpp:166-177 */
unsigned fo_get_to_position(const float* gen, unsigned n, unsigned to, float cmDist);
/* HmmUtils.cpp:31-63; vectors are one byte per site; returns length written */
unsigned long fo_subset_xor(const uint8_t* v1, const uint8_t* v2, unsigned long n, unsigned long from,
                            unsigned long to, uint8_t* out);
unsigned long fo_subset_and(const uint8_t* v1, const uint8_t* v2, unsigned long n, unsigned long from,
                            unsigned long to, uint8_t* out);

/*
 * HMM::decodeBatch (HMM.cpp:639-722) = forwardBatch (725-784) + backwardBatch (882-940) +
 * combine/normalise (669-692, NO_SSE branch); array mode, or sequence mode when m->sequence.
 * Sequence mode keeps the reference's buffer semantics: `previousAlpha = nextAlpha` (HMM.cpp:767) and
 * `lastComputedBeta = previousBeta` (922) are Eigen::Map assignments, i.e. they COPY the half-step result over
 * the stored vector of the neighbouring site, so the posterior of site p is built from the un-scaled alpha after
 * the homozygous half-step towards p+1 (p < to-1) and the beta after the half-step towards p-1 (p > from).
 * obsBits / homMinorBits: [B][to-from] bytes (what makeBits produced for the window).
 * alpha, beta: caller buffers of S*K*B floats, layout (pos*K + k)*B + v.
 * On return alpha holds the posterior for pos in [from,to); beta the scaled betas.
 * alphaFwd (optional, may be NULL): copy of the scaled forward alphas before combine.
 */
void fo_decode_batch(const fo_model* m, const uint8_t* obsBits, const uint8_t* homMinorBits, int B, unsigned from,
                     unsigned to, float* alpha, float* beta, float* alphaFwd);

/*
 * The reference's SCALAR path for one pair -- HMM::decode (HMM.cpp:1469-1495) = forward (1533-1609, getNextAlpha
 * 1611-1633) + backward (1636-1690, getPreviousBeta 1692-1721) + element-wise product and column normalisation
 * (HmmUtils.hpp:132-244) -- "for debugging and pedagogical reasons" in the reference, and its own second
 * implementation of the same mathematics: plain loops over the states, one pair, a division per posterior entry.
 * A second statement of the chain inside the oracle: tests assert fo_decode_batch == fo_decode_scalar to 1e-5
 * (the two differ in rounding only: reciprocal-multiply against division, the first site's emission from the prepared
 * rows, see the source).  obsBits / homMinorBits: [S] bytes, absolute sites.  posterior: [K][S], caller-zeroed;
 * columns outside [from, to) stay 0.
 */
void fo_decode_scalar(const fo_model* m, const uint8_t* obsBits, const uint8_t* homMinorBits, unsigned from,
                      unsigned to, float* posterior);

/*
 * HMM::augmentSumOverPairs (HMM.cpp:1044-1085).  obs vectors here are indexed by
 * absolute site (whole-sequence bits, as in non-hashing mode): [paddedB][S].
 * sum* are [S][K] row-major accumulators (any may be NULL when its flag is 0).
 */
void fo_augment_sum_over_pairs(const fo_model* m, const float* post, int actualB, int paddedB,
                               const uint8_t* obsBits, const uint8_t* homMinorBits, int doSums, int doMajorMinor,
                               float* sum, float* sum00, float* sum01, float* sum11);

/*
 * HMM::writePerPairOutput (HMM.cpp:1360-1458), the numeric part.
 * meanPost [actualB][S]; MAP [actualB][S]; perPairPost [actualB][K][S] (post * expected time);
 * sumOfPost [K][S] accumulated.  Any output may be NULL.
 */
void fo_per_pair_output(const fo_model* m, const float* post, int actualB, int paddedB, const float* expCoalTimes,
                        float* meanPost, int32_t* MAP, float* perPairPost, float* sumOfPost);

/* HMM::getPosteriorMean (HMM.cpp:1087-1097) / getMAP (1099-1107) on a per-state vector of length n */
float fo_posterior_mean(const fo_model* m, const float* vec, int n);
float fo_map(const fo_model* m, const float* vec, int n);

/*
 * HMM::writePerPairOutputFastSMC (HMM.cpp:1179-1357) for pair v of a decoded batch.
 * Scans pos in [fromV, toV), appends records (record.pair = pairOrdinal) to out (capacity cap);
 * returns the number of records this pair produced (even if > space left; nothing is
 * written past cap).
 */
int fo_ibd_scan_pair(const fo_model* m, const float* post, int paddedB, int v, unsigned fromV, unsigned toV,
                     unsigned stateThreshold, unsigned ageThreshold, float probabilityThreshold, int wantMean,
                     int wantMAP, uint32_t pairOrdinal, fo_ibd_record* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
