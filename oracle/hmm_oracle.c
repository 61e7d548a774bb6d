/*
 * hmm_oracle.c -- CPU restatement (plain C99) of FastSMC's pairwise HMM decode path.
 * TEST INFRASTRUCTURE ONLY -- see hmm_oracle.h for the rules and the parity status.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no -ffast-math, no FMA).
 * Layout of every batch buffer is the reference's: index (pos*K + k)*B + v, the pair
 * index v fastest (HMM.cpp:697, 741).  Citations are to /root/reference/ASMC_SRC/SRC.
 */
#include "hmm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- helpers */

/* HmmUtils.cpp:65-79: fp32 throughout, libm log10f/powf/roundf */
float fo_round_morgans(float value, int precision, float min)
{
  if (value <= min) {
    return min;
  }
  const float correction = 10.f - (float)precision;
  float L10 = floorf(log10f(value)) + correction;
  if (L10 < 0.f) {
    L10 = 0.f;
  }
  const float factor = powf(10.f, 10.f - L10);
  return roundf(value * factor) / factor;
}

/* HmmUtils.cpp:81-94: double log10/pow/round, integer result */
int fo_round_physical(int value, int precision)
{
  if (value <= 1) {
    return 1;
  }
  int L10 = (int)floor(log10((double)value)) - precision;
  if (L10 < 0) {
    L10 = 0;
  }
  int factor = (int)pow(10.0, (double)L10);
  return (int)round(value / (double)factor) * factor;
}

/* HmmUtils.cpp:102-130, NO_SSE branch: sums over states ascending from 0.f, exact division */
void fo_calculate_scaling_batch(const float* vec, float* scalings, float* sums, int batchSize, int numStates)
{
  for (int v = 0; v < batchSize; ++v) {
    sums[v] = 0.f;
  }
  for (int k = 0; k < numStates; ++k) {
    const float* row = vec + (size_t)k * batchSize;
    for (int v = 0; v < batchSize; ++v) {
      sums[v] += row[v];
    }
  }
  for (int v = 0; v < batchSize; ++v) {
    scalings[v] = 1.0f / sums[v];
  }
}

/* HmmUtils.cpp:132-151 */
void fo_apply_scaling_batch(float* vec, const float* scalings, int batchSize, int numStates)
{
  for (int k = 0; k < numStates; ++k) {
    float* row = vec + (size_t)k * batchSize;
    for (int v = 0; v < batchSize; ++v) {
      row[v] *= scalings[v];
    }
  }
}

/* HmmUtils.cpp:153-164 */
unsigned fo_get_from_position(const float* gen, unsigned n, unsigned from, float cmDist)
{
  (void)n;
  float cum = 0.f;
  while (cum < cmDist && from > 0u) {
    from--;
    cum += (gen[from + 1u] - gen[from]) * 100.f;
  }
  return from;
}

/* HmmUtils.cpp:166-177 */
unsigned fo_get_to_position(const float* gen, unsigned n, unsigned to, float cmDist)
{
  float cum = 0.f;
  while (cum < cmDist && to + 1u < n) {
    to++;
    cum += (gen[to] - gen[to - 1u]) * 100.f;
  }
  return (to + 1u < n) ? to + 1u : n;
}

/* HmmUtils.cpp:31-46 */
unsigned long fo_subset_xor(const uint8_t* v1, const uint8_t* v2, unsigned long n, unsigned long from,
                            unsigned long to, uint8_t* out)
{
  const unsigned long minTo = n < to ? n : to;
  for (unsigned long i = from; i < minTo; ++i) {
    out[i - from] = (uint8_t)((v1[i] ^ v2[i]) & 1u);
  }
  return minTo - from;
}

/* HmmUtils.cpp:48-63 */
unsigned long fo_subset_and(const uint8_t* v1, const uint8_t* v2, unsigned long n, unsigned long from,
                            unsigned long to, uint8_t* out)
{
  const unsigned long minTo = n < to ? n : to;
  for (unsigned long i = from; i < minTo; ++i) {
    out[i - from] = (uint8_t)(v1[i] & v2[i] & 1u);
  }
  return minTo - from;
}

/* ----------------------------------------------------------- decodeBatch */

/* Expand one site's observation bits to the reference's 0/1 floats (HMM.cpp:647-652). */
static void expand_obs(const uint8_t* obsBits, const uint8_t* homMinorBits, int B, unsigned len, unsigned rel,
                       float* isZero, float* isTwo)
{
  for (int v = 0; v < B; ++v) {
    isZero[v] = obsBits[(size_t)v * len + rel] ? 0.0f : 1.0f;
    isTwo[v] = homMinorBits[(size_t)v * len + rel] ? 1.0f : 0.0f;
  }
}

/* HMM::getNextAlphaBatched, NO_SSE branch (HMM.cpp:787-830) */
static void next_alpha(const fo_model* m, int row, int B, const float* prev, float* next, float* alphaC, float* AU,
                       const float* isZero, const float* isTwo, const float* e1, const float* e0m1,
                       const float* e2m0)
{
  const int K = m->K;
  const float* Bv = m->Bt + (size_t)row * K;
  const float* Uv = m->Ut + (size_t)row * K;
  const float* Dv = m->Dt + (size_t)row * K;
  const float* cR = m->colRatios;

  /* alphaC[k] = sum_{i>=k} prev[i], accumulated from the top down (799-814) */
  memcpy(alphaC + (size_t)(K - 1) * B, prev + (size_t)(K - 1) * B, (size_t)B * sizeof(float));
  for (int k = K - 2; k >= 0; --k) {
    for (int v = 0; v < B; ++v) {
      alphaC[(size_t)k * B + v] = alphaC[(size_t)(k + 1) * B + v] + prev[(size_t)k * B + v];
    }
  }
  for (int v = 0; v < B; ++v) {
    AU[v] = 0.f;
  }
  for (int k = 0; k < K; ++k) {
    for (int v = 0; v < B; ++v) {
      if (k) {
        AU[v] = Uv[k - 1] * prev[(size_t)(k - 1) * B + v] + cR[k - 1] * AU[v];
      }
      float term = AU[v] + Dv[k] * prev[(size_t)k * B + v];
      if (k < K - 1) {
        term += Bv[k] * alphaC[(size_t)(k + 1) * B + v];
      }
      const float em = e1[k] + e0m1[k] * isZero[v] + e2m0[k] * isTwo[v];
      next[(size_t)k * B + v] = em * term;
    }
  }
}

/* HMM::getPreviousBetaBatched, NO_SSE branch (HMM.cpp:943-1016) */
static void previous_beta(const fo_model* m, int row, int B, const float* last, float* cur, float* vec, float* BU,
                          float* BL, const float* isZero, const float* isTwo, const float* e1, const float* e0m1,
                          const float* e2m0)
{
  const int K = m->K;
  const float* Bv = m->Bt + (size_t)row * K;
  const float* Uv = m->Ut + (size_t)row * K;
  const float* RR = m->RRt + (size_t)row * K;
  const float* Dv = m->Dt + (size_t)row * K;

  for (int k = 0; k < K; ++k) {
    for (int v = 0; v < B; ++v) {
      const float em = e1[k] + e0m1[k] * isZero[v] + e2m0[k] * isTwo[v];
      vec[(size_t)k * B + v] = last[(size_t)k * B + v] * em;
    }
  }
  memset(BU, 0, (size_t)K * B * sizeof(float));
  for (int k = K - 2; k >= 0; --k) {
    for (int v = 0; v < B; ++v) {
      BU[(size_t)k * B + v] = Uv[k] * vec[(size_t)(k + 1) * B + v] + RR[k] * BU[(size_t)(k + 1) * B + v];
    }
  }
  for (int v = 0; v < B; ++v) {
    BL[v] = 0.f;
  }
  for (int k = 0; k < K; ++k) {
    for (int v = 0; v < B; ++v) {
      if (k) {
        BL[v] += Bv[k - 1] * vec[(size_t)(k - 1) * B + v];
      }
      /* (BL + D*vec) + BU : the NO_SSE association, HMM.cpp:1014 */
      cur[(size_t)k * B + v] = BL[v] + Dv[k] * vec[(size_t)k * B + v] + BU[(size_t)k * B + v];
    }
  }
}

void fo_decode_batch(const fo_model* m, const uint8_t* obsBits, const uint8_t* homMinorBits, int B, unsigned from,
                     unsigned to, float* alpha, float* beta, float* alphaFwd)
{
  const int K = m->K;
  const unsigned len = to - from;
  const size_t KB = (size_t)K * B;
  float* work = (float*)malloc(sizeof(float) * (3 * KB + 8 * (size_t)B));
  float* alphaC = work;      /* [K][B]; reused as vec in the backward pass */
  float* BU = work + KB;     /* [K][B] */
  float* scratch = work + 2 * KB;
  float* vecBuf = scratch;   /* [K][B] */
  float* AU = work + 3 * KB; /* [B]; reused as BL */
  float* sums = AU + B;
  float* scal = sums + B;
  float* isZero = scal + B;
  float* isTwo = isZero + B;

  /* ---- forward (HMM.cpp:725-784) ---- */
  expand_obs(obsBits, homMinorBits, B, len, 0, isZero, isTwo);
  {
    float* a0 = alpha + (size_t)from * KB;
    const float* e1 = m->e1 + (size_t)from * K;
    const float* e0m1 = m->e0m1 + (size_t)from * K;
    const float* e2m0 = m->e2m0 + (size_t)from * K;
    for (int k = 0; k < K; ++k) {
      for (int v = 0; v < B; ++v) {
        const float firstEmission = e1[k] + e0m1[k] * isZero[v] + e2m0[k] * isTwo[v];
        a0[(size_t)k * B + v] = m->pi[k] * firstEmission;
      }
    }
    fo_calculate_scaling_batch(a0, scal, sums, B, K);
    fo_apply_scaling_batch(a0, scal, B, K);
  }
  float* zeros = (float*)calloc((size_t)B, sizeof(float)); /* m_allZeros */
  for (unsigned pos = from + 1; pos < to; ++pos) {
    expand_obs(obsBits, homMinorBits, B, len, pos - from, isZero, isTwo);
    float* prev = alpha + (size_t)(pos - 1) * KB;
    float* next = alpha + (size_t)pos * KB;
    if (m->sequence) {
      /* HMM.cpp:760-770: homozygous stretch between the sites, then the site itself */
      const float* h = m->hom + (size_t)pos * K;
      next_alpha(m, m->gapRowF[pos], B, prev, next, alphaC, AU, zeros, zeros, h, h, h);
      memcpy(prev, next, KB * sizeof(float)); /* previousAlpha = nextAlpha (Eigen::Map assignment copies) */
      next_alpha(m, m->siteRowF[pos], B, prev, next, alphaC, AU, isZero, isTwo, m->e1 + (size_t)pos * K,
                 m->e0m1 + (size_t)pos * K, m->e2m0 + (size_t)pos * K);
    } else {
      next_alpha(m, m->stepRow[pos], B, prev, next, alphaC, AU, isZero, isTwo, m->e1 + (size_t)pos * K,
                 m->e0m1 + (size_t)pos * K, m->e2m0 + (size_t)pos * K);
    }
    /* scalingSkip == 1: every site (HMM.cpp:776-779) */
    fo_calculate_scaling_batch(next, scal, sums, B, K);
    fo_apply_scaling_batch(next, scal, B, K);
  }
  if (alphaFwd) {
    memcpy(alphaFwd + (size_t)from * KB, alpha + (size_t)from * KB, (size_t)len * KB * sizeof(float));
  }

  /* ---- backward (HMM.cpp:882-940) ---- */
  {
    float* bl = beta + (size_t)(to - 1) * KB;
    for (size_t i = 0; i < KB; ++i) {
      bl[i] = 1.0f;
    }
    fo_calculate_scaling_batch(bl, scal, sums, B, K);
    fo_apply_scaling_batch(bl, scal, B, K);
  }
  for (long pos = (long)to - 2; pos >= (long)from; --pos) {
    /* emission and observation of site pos+1; transition key of the step pos -> pos+1 (909, 927-929) */
    expand_obs(obsBits, homMinorBits, B, len, (unsigned)(pos + 1) - from, isZero, isTwo);
    float* cur = beta + (size_t)pos * KB;
    float* last = beta + (size_t)(pos + 1) * KB;
    if (m->sequence) {
      /* HMM.cpp:915-925 */
      const float* h = m->hom + (size_t)(pos + 1) * K;
      previous_beta(m, m->gapRowB[pos + 1], B, last, cur, vecBuf, BU, AU /* as BL */, zeros, zeros, h, h, h);
      memcpy(last, cur, KB * sizeof(float)); /* lastComputedBeta = previousBeta */
      previous_beta(m, m->siteRowB[pos + 1], B, last, cur, vecBuf, BU, AU, isZero, isTwo,
                    m->e1 + (size_t)(pos + 1) * K, m->e0m1 + (size_t)(pos + 1) * K, m->e2m0 + (size_t)(pos + 1) * K);
    } else {
      previous_beta(m, m->stepRow[pos + 1], B, last, cur, vecBuf, BU, AU /* as BL */, isZero, isTwo,
                    m->e1 + (size_t)(pos + 1) * K, m->e0m1 + (size_t)(pos + 1) * K, m->e2m0 + (size_t)(pos + 1) * K);
    }
    fo_calculate_scaling_batch(cur, scal, sums, B, K);
    fo_apply_scaling_batch(cur, scal, B, K);
  }
  free(zeros);

  /* ---- combine and normalise (HMM.cpp:669-692, NO_SSE) ---- */
  for (unsigned pos = from; pos < to; ++pos) {
    float* a = alpha + (size_t)pos * KB;
    const float* b = beta + (size_t)pos * KB;
    for (int v = 0; v < B; ++v) {
      sums[v] = 0.f;
    }
    for (int k = 0; k < K; ++k) {
      for (int v = 0; v < B; ++v) {
        a[(size_t)k * B + v] *= b[(size_t)k * B + v];
        sums[v] += a[(size_t)k * B + v];
      }
    }
    for (int v = 0; v < B; ++v) {
      scal[v] = 1.0f / sums[v];
    }
    for (int k = 0; k < K; ++k) {
      for (int v = 0; v < B; ++v) {
        a[(size_t)k * B + v] *= scal[v];
      }
    }
  }
  free(work);
}

/* ------------------------------------------------------------- the reference's scalar path */

/* HMM::getNextAlpha (HMM.cpp:1611-1633): one pair, plain loops over the states */
static void scalar_next_alpha(const fo_model* m, int row, float* alphaC, const float* previousAlpha, float* nextAlpha,
                              const float* e1, const float* e0m1, const float* e2m0, float obsIsZero,
                              float obsIsHomMinor)
{
  const int K = m->K;
  alphaC[K - 1] = previousAlpha[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    alphaC[k] = alphaC[k + 1] + previousAlpha[k];
  }
  const float* B = m->Bt + (size_t)row * K;
  const float* U = m->Ut + (size_t)row * K;
  const float* D = m->Dt + (size_t)row * K;
  float AUc = 0;
  for (int k = 0; k < K; k++) {
    if (k) {
      AUc = U[k - 1] * previousAlpha[k - 1] + m->colRatios[k - 1] * AUc;
    }
    float term = AUc + D[k] * previousAlpha[k];
    if (k < K - 1) {
      term += B[k] * alphaC[k + 1];
    }
    const float currentEmission_k = e1[k] + e0m1[k] * obsIsZero + e2m0[k] * obsIsHomMinor;
    nextAlpha[k] = currentEmission_k * term;
  }
}

/* HMM::getPreviousBeta (HMM.cpp:1692-1721).  BL and BU live across calls like the reference's vectors (zero
 * initialised once: BL[0] and BU[K-1] are never written) */
static void scalar_previous_beta(const fo_model* m, int row, const float* lastComputedBeta, float* BL, float* BU,
                                 float* currentBeta, float* vec, const float* e1, const float* e0m1, const float* e2m0,
                                 float obsIsZero, float obsIsHomMinor)
{
  const int K = m->K;
  for (int k = 0; k < K; k++) {
    const float currentEmission_k = e1[k] + e0m1[k] * obsIsZero + e2m0[k] * obsIsHomMinor;
    vec[k] = lastComputedBeta[k] * currentEmission_k;
  }
  float sum = 0;
  const float* B = m->Bt + (size_t)row * K;
  for (int k = 1; k < K; k++) {
    sum += B[k - 1] * vec[k - 1];
    BL[k] = sum;
  }
  const float* U = m->Ut + (size_t)row * K;
  const float* RR = m->RRt + (size_t)row * K;
  for (int k = K - 2; k >= 0; k--) {
    BU[k] = vec[k + 1] * U[k] + RR[k] * BU[k + 1];
  }
  const float* D = m->Dt + (size_t)row * K;
  for (int k = 0; k < K; k++) {
    currentBeta[k] = BL[k] + vec[k] * D[k] + BU[k];
  }
}

/* HmmUtils.hpp:132-135 getSumOfVector (std::accumulate from 0) + 147-158 elementWiseMultVectorScalar with 1.f / sum */
static void scalar_normalise(float* v, int K)
{
  float sum = 0;
  for (int k = 0; k < K; k++) {
    sum += v[k];
  }
  const float scaling = 1.f / sum;
  for (int k = 0; k < K; k++) {
    v[k] = v[k] * scaling;
  }
}

void fo_decode_scalar(const fo_model* m, const uint8_t* obsBits, const uint8_t* homMinorBits, unsigned from,
                      unsigned to, float* posterior)
{
  const int K = m->K;
  const size_t S = (size_t)m->S;
  float* alpha = (float*)calloc((size_t)K * S, sizeof(float)); /* [K][S], zero outside [from, to) */
  float* beta = (float*)calloc((size_t)K * S, sizeof(float));
  float* work = (float*)calloc((size_t)7 * K + 1, sizeof(float));
  float* previous = work;         /* previousAlpha / lastComputedBeta */
  float* next = work + K;         /* nextAlpha / currentBeta */
  float* alphaC = work + 2 * K;   /* [K + 1] */
  float* BL = work + 3 * K + 1;
  float* BU = work + 4 * K + 1;
  float* vec = work + 5 * K + 1;

  /* ---- forward (HMM.cpp:1533-1609).  The first site's emission: the reference reads it from the emission tables
   * (getEmission, 1509-1530); restated with the prepared rows of that site, which are those table values to an ulp
   * (e1 exactly; e1 + (e0 - e1) and (e1 + (e0 - e1)) + (e2 - e0) for the homozygous classes, HMM.cpp:181-207) */
  {
    const float z = !obsBits[from] ? 1.0f : 0.0f, t = homMinorBits[from] ? 1.0f : 0.0f;
    for (int k = 0; k < K; k++) {
      const float emission = m->e1[(size_t)from * K + k] + m->e0m1[(size_t)from * K + k] * z +
                             m->e2m0[(size_t)from * K + k] * t;
      previous[k] = m->pi[k] * emission;
    }
    scalar_normalise(previous, K);
    for (int k = 0; k < K; k++) {
      alpha[(size_t)k * S + from] = previous[k];
    }
  }
  for (unsigned pos = from + 1; pos < to; pos++) {
    const float obsIsZero = !obsBits[pos] ? 1.0f : 0.0f;
    const float obsIsHomMinor = homMinorBits[pos] ? 1.0f : 0.0f;
    const float* e1 = m->e1 + (size_t)pos * K;
    const float* e0m1 = m->e0m1 + (size_t)pos * K;
    const float* e2m0 = m->e2m0 + (size_t)pos * K;
    if (m->sequence) { /* HMM.cpp:1577-1588 */
      const float* h = m->hom + (size_t)pos * K;
      scalar_next_alpha(m, m->gapRowF[pos], alphaC, previous, next, h, h, h, 0.0f, 0.0f);
      memcpy(previous, next, (size_t)K * sizeof(float));
      scalar_next_alpha(m, m->siteRowF[pos], alphaC, previous, next, e1, e0m1, e2m0, obsIsZero, obsIsHomMinor);
    } else {
      scalar_next_alpha(m, m->stepRow[pos], alphaC, previous, next, e1, e0m1, e2m0, obsIsZero, obsIsHomMinor);
    }
    scalar_normalise(next, K); /* pos % scalingSkip == 0 with scalingSkip = 1 (HMM.cpp:1593-1598) */
    for (int k = 0; k < K; k++) {
      alpha[(size_t)k * S + pos] = next[k];
      previous[k] = next[k];
    }
  }

  /* ---- backward (HMM.cpp:1636-1690) ---- */
  for (int k = 0; k < K; k++) {
    previous[k] = 1.f;
  }
  scalar_normalise(previous, K);
  for (int k = 0; k < K; k++) {
    beta[(size_t)k * S + (to - 1)] = previous[k];
    BL[k] = 0.f;
    BU[k] = 0.f;
  }
  for (long pos = (long)to - 2; pos >= (long)from; pos--) {
    const float obsIsZero = !obsBits[pos + 1] ? 1.0f : 0.0f;
    const float obsIsHomMinor = homMinorBits[pos + 1] ? 1.0f : 0.0f;
    const float* e1 = m->e1 + (size_t)(pos + 1) * K;
    const float* e0m1 = m->e0m1 + (size_t)(pos + 1) * K;
    const float* e2m0 = m->e2m0 + (size_t)(pos + 1) * K;
    if (m->sequence) { /* HMM.cpp:1663-1675 */
      const float* h = m->hom + (size_t)(pos + 1) * K;
      scalar_previous_beta(m, m->gapRowB[pos + 1], previous, BL, BU, next, vec, h, h, h, 0.0f, 0.0f);
      memcpy(previous, next, (size_t)K * sizeof(float));
      scalar_previous_beta(m, m->siteRowB[pos + 1], previous, BL, BU, next, vec, e1, e0m1, e2m0, obsIsZero,
                           obsIsHomMinor);
    } else {
      scalar_previous_beta(m, m->stepRow[pos + 1], previous, BL, BU, next, vec, e1, e0m1, e2m0, obsIsZero,
                           obsIsHomMinor);
    }
    scalar_normalise(next, K);
    for (int k = 0; k < K; k++) {
      beta[(size_t)k * S + pos] = next[k];
      previous[k] = next[k];
    }
  }

  /* ---- posterior (HMM.cpp:1481-1482): element-wise product, then every column divided by its sum
   * (normalizeMatrixColumns, HmmUtils.hpp:218-236: a DIVISION per element, where the batched path multiplies by a
   * reciprocal).  Columns outside [from, to) are 0 / 0 in the reference; they are left 0 here. */
  for (size_t pos = from; pos < to; pos++) {
    float sum = 0;
    for (int k = 0; k < K; k++) {
      posterior[(size_t)k * S + pos] = alpha[(size_t)k * S + pos] * beta[(size_t)k * S + pos];
      sum += posterior[(size_t)k * S + pos];
    }
    for (int k = 0; k < K; k++) {
      posterior[(size_t)k * S + pos] = posterior[(size_t)k * S + pos] / sum;
    }
  }
  free(work);
  free(beta);
  free(alpha);
}

/* ------------------------------------------------------------- consumers */

/* HMM.cpp:1044-1085; note the loop covers all S sites */
void fo_augment_sum_over_pairs(const fo_model* m, const float* post, int actualB, int paddedB,
                               const uint8_t* obsBits, const uint8_t* homMinorBits, int doSums, int doMajorMinor,
                               float* sum, float* sum00, float* sum01, float* sum11)
{
  const int K = m->K;
  const int S = m->S;
  if (!doSums && !doMajorMinor) {
    return;
  }
  for (int pos = 0; pos < S; ++pos) {
    for (int k = 0; k < K; ++k) {
      float s = 0, s00 = 0, s01 = 0, s11 = 0;
      for (int v = 0; v < actualB; ++v) {
        const float p = post[((size_t)pos * K + k) * paddedB + v];
        if (doSums) {
          s += p;
        }
        if (doMajorMinor) {
          if (homMinorBits[(size_t)v * S + pos] == 1) {
            s11 += p;
          } else if (obsBits[(size_t)v * S + pos] == 0) {
            s00 += p;
          } else {
            s01 += p;
          }
        }
      }
      const size_t o = (size_t)pos * K + k;
      if (doSums) {
        sum[o] += s;
      }
      if (doMajorMinor) {
        sum00[o] += s00;
        sum01[o] += s01;
        sum11[o] += s11;
      }
    }
  }
}

/* HMM.cpp:1360-1410 */
void fo_per_pair_output(const fo_model* m, const float* post, int actualB, int paddedB, const float* expCoalTimes,
                        float* meanPost, int32_t* MAP, float* perPairPost, float* sumOfPost)
{
  const int K = m->K;
  const int S = m->S;
  if (meanPost) {
    memset(meanPost, 0, sizeof(float) * (size_t)actualB * S);
    for (int pos = 0; pos < S; ++pos) {
      for (int k = 0; k < K; ++k) {
        for (int b = 0; b < actualB; ++b) {
          const float p = post[((size_t)pos * K + k) * paddedB + b];
          const float postValue = p * expCoalTimes[k];
          meanPost[(size_t)b * S + pos] += postValue;
          if (perPairPost) {
            perPairPost[((size_t)b * K + k) * S + pos] = postValue;
          }
          if (sumOfPost) {
            sumOfPost[(size_t)k * S + pos] += postValue;
          }
        }
      }
    }
  }
  if (MAP) {
    float* curMax = (float*)malloc(sizeof(float) * (size_t)actualB);
    memset(MAP, 0, sizeof(int32_t) * (size_t)actualB * S);
    for (int pos = 0; pos < S; ++pos) {
      for (int b = 0; b < actualB; ++b) {
        curMax[b] = 0.f;
      }
      for (int k = 0; k < K; ++k) {
        for (int b = 0; b < actualB; ++b) {
          const float p = post[((size_t)pos * K + k) * paddedB + b];
          if (curMax[b] < p) {
            MAP[(size_t)b * S + pos] = k;
            curMax[b] = p;
          }
        }
      }
    }
    free(curMax);
  }
}

/* HMM.cpp:1087-1097 */
float fo_posterior_mean(const fo_model* m, const float* vec, int n)
{
  float acc = 0.f;
  for (int k = 0; k < n; ++k) {
    acc += vec[k];
  }
  const float normalization = 1.f / acc;
  float mean = 0.f;
  for (int k = 0; k < n; ++k) {
    mean += normalization * vec[k] * m->expTimes[k];
  }
  return mean;
}

/* HMM.cpp:1099-1107: first maximum of posterior/prior */
float fo_map(const fo_model* m, const float* vec, int n)
{
  int best = 0;
  float bestVal = 0.f;
  for (int k = 0; k < n; ++k) {
    const float r = vec[k] / m->pi[k];
    if (k == 0 || bestVal < r) {
      best = k;
      bestVal = r;
    }
  }
  return m->expTimes[best];
}

/* HMM.cpp:1179-1357 for one pair.  The reference keeps four (flag,start) pairs, one per
 * threshold level; at most one flag is ever set.  They are kept here as arrays. */
int fo_ibd_scan_pair(const fo_model* m, const float* post, int paddedB, int v, unsigned fromV, unsigned toV,
                     unsigned stateThreshold, unsigned ageThreshold, float probabilityThreshold, int wantMean,
                     int wantMAP, uint32_t pairOrdinal, fo_ibd_record* out, int cap)
{
  const int K = m->K;
  const int track = wantMean || wantMAP;
  const unsigned nAge = track ? ageThreshold : 0u;
  int open[4] = {0, 0, 0, 0};
  unsigned start[4] = {0, 0, 0, 0};
  float posteriorIBD = 0;
  int n = 0;
  float* cur = (float*)calloc(3 * (size_t)(nAge ? nAge : 1), sizeof(float));
  float* sumPS = cur + (nAge ? nAge : 1);
  float* prevPS = sumPS + (nAge ? nAge : 1);
  /* int * float products evaluated in fp32 (HMM.cpp:1226,1254,1281,1308) */
  const float thr[4] = {1000 * probabilityThreshold, 100 * probabilityThreshold, 10 * probabilityThreshold,
                        probabilityThreshold};

#define FO_EMIT(S0, E0, VEC)                                                                                           \
  do {                                                                                                                 \
    if (n < cap) {                                                                                                     \
      out[n].pair = pairOrdinal;                                                                                       \
      out[n].start = (int32_t)(S0);                                                                                    \
      out[n].end = (int32_t)(E0);                                                                                      \
      out[n].prob = posteriorIBD;                                                                                      \
      out[n].postMean = wantMean ? fo_posterior_mean(m, (VEC), (int)nAge) : 0.f;                                       \
      out[n].map = wantMAP ? fo_map(m, (VEC), (int)nAge) : 0.f;                                                        \
    }                                                                                                                  \
    n++;                                                                                                               \
  } while (0)

  for (unsigned pos = fromV; pos < toV; ++pos) {
    float sum = 0;
    if (track) {
      for (unsigned k = 0; k < nAge; ++k) {
        const float p = post[((size_t)pos * K + k) * paddedB + v];
        cur[k] = p;
        prevPS[k] = sumPS[k];
        sumPS[k] += p;
        if (k < stateThreshold) {
          sum += p;
        }
      }
    } else {
      for (unsigned k = 0; k < stateThreshold; ++k) {
        sum += post[((size_t)pos * K + k) * paddedB + v];
      }
    }
    int level = 4;
    for (int l = 0; l < 4; ++l) {
      if (sum >= thr[l]) {
        level = l;
        break;
      }
    }
    if (level < 4) {
      if (!open[level]) {
        start[level] = pos;
        memcpy(sumPS, cur, sizeof(float) * nAge);
        if (pos > fromV) {
          for (int l = 0; l < 4; ++l) {
            if (l != level && open[l]) {
              FO_EMIT(start[l], pos - 1, prevPS);
              break;
            }
          }
        }
        posteriorIBD = sum;
      } else {
        posteriorIBD += sum;
      }
      if (pos == toV - 1) {
        FO_EMIT(start[level], toV - 1, sumPS);
        posteriorIBD = 0;
      }
      open[0] = open[1] = open[2] = open[3] = 0;
      open[level] = 1;
    } else {
      for (int l = 0; l < 4; ++l) {
        if (open[l]) {
          FO_EMIT(start[l], pos - 1, prevPS);
          posteriorIBD = 0;
          break;
        }
      }
      open[0] = open[1] = open[2] = open[3] = 0;
    }
  }
#undef FO_EMIT
  free(cur);
  return n;
}
