/*
 * fastsmc_hip.h -- C ABI of the MI355X (gfx950) pairwise-HMM decode library, libfastsmc_hip.so.
 *
 * This is the drop-in seam for ONE path of PalamaraLab/FastSMC: the batched pairwise HMM
 * decode (forward, backward, posterior combine) and its posterior consumers.  The reference
 * has no FFI for this path -- it is private methods of class HMM -- so each entry point
 * below names the reference call it stands in for (paths relative to ASMC_SRC/SRC):
 *
 *   fsmc_model_create      <- what HMM::HMM leaves behind for the path: DecodingQuantities
 *                             vectors + prepareEmissions rows        (HMM.cpp:65-127, 159-256)
 *   fsmc_haps_upload       <- Individual::genotype1/2 bit vectors consumed by makeBits
 *                                                                    (HMM.cpp:147-157)
 *   fsmc_decode_ibd        <- decodeBatch + writePerPairOutputFastSMC for every batch of a
 *                             work list                              (HMM.cpp:575-584, 624-633,
 *                                                                     639-1041, 1179-1357)
 *   fsmc_decode_posteriors <- decodeBatch; result = m_alphaBuffer     (HMM.cpp:639-722)
 *   fsmc_decode_per_pair   <- decodeBatch + writePerPairOutput        (HMM.cpp:1360-1458)
 *   fsmc_decode_sums       <- decodeBatch + augmentSumOverPairs       (HMM.cpp:1044-1085)
 *
 * Conventions: plain C types; host buffers are caller-owned, device buffers library-owned;
 * every function returns 0 on success or a negative FSMC_E* code and never exits or throws;
 * fsmc_last_error() describes the last failure of the context (or of context creation when
 * ctx == NULL).  One context per device; contexts are independent and may be driven from
 * different host threads/processes (one process per GPU).  There is NO CPU fallback: without
 * a usable HIP device every call fails with FSMC_ENODEVICE.
 */
#ifndef FASTSMC_HIP_H
#define FASTSMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSMC_OK 0
#define FSMC_EINVAL (-1)    /* bad argument */
#define FSMC_ENODEVICE (-2) /* no HIP device / HIP runtime failure at start-up */
#define FSMC_EHIP (-3)      /* a HIP call failed; see fsmc_last_error */
#define FSMC_ENOMEM (-4)    /* device or host allocation failed */
#define FSMC_ESTATE (-5)    /* call sequence error (e.g. decode before upload) */
#define FSMC_EOVERFLOW (-6) /* caller's output buffer too small; *n_out holds the needed count */
#define FSMC_EUNSUPPORTED (-7)
#define FSMC_ERUNTIME (-8)  /* the device is there but this process cannot launch on it: two HIP runtimes loaded */

typedef struct fsmc_ctx fsmc_ctx;
typedef struct fsmc_model fsmc_model;

/* Constant inputs of the path.  All pointers are host memory, copied by fsmc_model_create. */
typedef struct {
  int32_t K;                /* states                          (DecodingQuantities::states) */
  int32_t S;                /* sites                           (Data::sites) */
  const float* pi;          /* [K] initialStateProb */
  const float* col_ratios;  /* [K] columnRatios, zero padded   (DecodingQuantities.cpp:299-303) */
  const float* exp_times;   /* [K] expectedTimes */
  int32_t n_rows;           /* rows of the four transition tables */
  const float* D;           /* [n_rows][K] Dvectors */
  const float* B;           /* [n_rows][K] Bvectors   (column K-1 unused) */
  const float* U;           /* [n_rows][K] Uvectors   (column K-1 unused) */
  const float* RR;          /* [n_rows][K] rowRatioVectors (column K-1 unused) */
  const int32_t* step_row;  /* [S] table row of key roundMorgans(gen[p]-gen[p-1]) for p>=1 (HMM.cpp:755,909) */
  const float* e1;          /* [S][K] emission1AtSite */
  const float* e0m1;        /* [S][K] emission0minus1AtSite */
  const float* e2m0;        /* [S][K] emission2minus0AtSite */
  uint32_t state_threshold; /* HMM::stateThreshold      (HMM.cpp:504-513) */
  uint32_t age_threshold;   /* HMM::ageThreshold        (HMM.cpp:101-105) */
  float probability_threshold; /* HMM::probabilityThreshold (HMM.cpp:96-99) */
  /* Sequence mode (DecodingParams::decodingSequence; HMM.cpp:760-770 forward, 915-925 backward): two transition
   * steps per site -- across the homozygous stretch since the previous site, then the site itself.  All zero /
   * NULL in array mode (step_row is then the only row index).  Arrays are [S], indexed by the later site q of
   * the gap (q-1, q); entry 0 unused:
   *   gap_row_f[q]  row of key roundMorgans(recDist_q - rate[q]),   site_row_f[q]  row of key rate[q]
   *   gap_row_b[q]  row of key roundMorgans(recDist_q - rate[q-1]), site_row_b[q]  row of key rate[q-1]
   * (recDist_q = roundMorgans(gen[q]-gen[q-1]), rate[p] = roundMorgans(recRateAtMarker[p])), and
   *   hom[q][K] = homozygousEmissionMap[roundPhysical(phys[q]-phys[q-1]-1)]. */
  int32_t sequence;
  const int32_t* gap_row_f;
  const int32_t* site_row_f;
  const int32_t* gap_row_b;
  const int32_t* site_row_b;
  const float* hom;
} fsmc_model_desc;

/* One haplotype pair: rows of the uploaded bit matrix.  Row 2*ind + (hap-1). */
typedef struct {
  uint32_t hap_a; /* the record's first haplotype  (PairObservations iInd/iHap) */
  uint32_t hap_b; /* the record's second haplotype (PairObservations jInd/jHap) */
} fsmc_pair;

/* A batch of <= 64 consecutive pairs that share one decode window -- the reference's batch
 * (HMM.cpp:555-636).  [from,to) is the padded decode window handed to decodeBatch;
 * [scan_from,scan_to) is the window the IBD scan covers (HMM.cpp:1199-1206).  Non-hashing
 * mode: from = scan_from = 0, to = scan_to = S. */
typedef struct {
  uint32_t first_pair; /* index into the pair list */
  uint32_t n_pairs;    /* 1..64 */
  uint32_t from, to;
  uint32_t scan_from, scan_to;
} fsmc_group;

/* One IBD segment as handed to HMM::writePairIBD (HMM.cpp:1110-1177). */
typedef struct {
  uint32_t pair;   /* index into the pair list */
  int32_t start;   /* first site */
  int32_t end;     /* last site, inclusive */
  float prob;      /* cumulative posterior ("posteriorIBD"); ibd_score = prob / (end-start+1) */
  float post_mean; /* getPosteriorMean of the per-state sums (0 if not requested) */
  float map;       /* getMAP of the per-state sums (0 if not requested) */
} fsmc_ibd_record;

#define FSMC_WANT_MEAN 1u /* DecodingParams::doPerPairPosteriorMean */
#define FSMC_WANT_MAP 2u  /* DecodingParams::doPerPairMAP */
#define FSMC_WANT_SUMS 4u /* DecodingParams::doPosteriorSums (fsmc_decode_sums) */
#define FSMC_WANT_MAJOR_MINOR_SUMS 8u /* doMajorMinorPosteriorSums */

/* ---- context ---- */
/* stream: an existing hipStream_t to launch on (e.g. torch's current stream), or NULL for a private one. */
int fsmc_ctx_create(int device_id, void* stream, fsmc_ctx** out);
void fsmc_ctx_destroy(fsmc_ctx* ctx);
const char* fsmc_last_error(const fsmc_ctx* ctx);
/* Device properties as seen by the library: CU count, and the number of resident decode waves it launches. */
int fsmc_ctx_info(const fsmc_ctx* ctx, int32_t* n_cu, int32_t* n_slots, uint64_t* hbm_bytes);
/* Workspace for the alpha/beta streaming (bytes).  A limit set here is the caller's statement about the job: a plan may
 * use all of it at once -- windows kept whole instead of chunked, resident chunks, long windows in the paired kernel.
 * 0 (default): the library's own policy.  The rows a decode cannot do without may take up to 80 % of the card (at least
 * 40 %); everything beyond them is EARNED: hipMalloc costs about 40 ms per GB on this driver, so a context starts with a
 * free allowance of 24 GB and every launch adds what an upgraded plan is expected to save of it (6 % of its estimated
 * kernel time, at the allocation rate) -- a run of seconds does not spend them allocating.  Growth is
 * amortised and every byte is paid for once: a bigger buffer is a new allocation of its whole size, so the plan is
 * upgraded only when the credit covers twice the buffer held (or the whole budget), and an allocation is debited from
 * the credit.  The buffer is kept for the life of the context. */
int fsmc_ctx_set_workspace_limit(fsmc_ctx* ctx, uint64_t bytes);
/* The caller announces the work the context's coming launches will decode -- pair-sites (pairs x sites of their decode
 * windows) of a model of `states` states: what the reference's HMM::decodeAll knows when it starts (its job's pair
 * range, HMM.cpp:310-321).  Under the library's own workspace policy (no limit set) the credit of the whole job is
 * then there at the first launch: a job long enough to pay for the card allocates it once, at its start, instead of
 * growing into it; a short job stays small.  The announced launches earn nothing again.  No effect with a limit set.
 * pair_sites = 0 ends the announced job (HMM::finishDecoding, HMM.cpp:515-524): what is left of the announcement -- an
 * estimate that was too high, a job that stopped early -- is forgotten and its unspent credit taken back.  The credit
 * of announcements is capped at the device's memory. */
int fsmc_ctx_expect_work(fsmc_ctx* ctx, double pair_sites, int32_t states);
/* Tuning: sites between beta checkpoints when a decode window does not fit the workspace (0 = automatic:
 * max(512, ceil(sqrt(window))), 2048 for the wave-group kernel, rounded up to 16).  Results do not depend on it. */
int fsmc_ctx_set_chunk_sites(fsmc_ctx* ctx, uint32_t sites);
/* Tuning: beta stride of the IBD decode and of the sums over pairs.  1 = every beta row of a chunk goes through HBM
 * (8K bytes per pair-site); 2 = every second row does and the alpha sweep recomputes the others from their successor
 * (4K bytes per pair-site, half a sweep more arithmetic); 0 = automatic: 2 where that kernel exists (array mode,
 * K <= 128) -- for the sums only when the launch has at least as many batches as the chip holds waves (a wave alone on
 * its SIMD only gets the recomputed half sweep on top) --, else 1.  Results do not depend on it.
 * fsmc_ctx_last_beta_stride reports what the last IBD or sums launch used. */
int fsmc_ctx_set_beta_stride(fsmc_ctx* ctx, uint32_t stride);
int fsmc_ctx_last_beta_stride(const fsmc_ctx* ctx, int32_t* stride);
/* How the last launch was laid out: sites per chunk (= the longest window when every beta row fitted), chunks per
 * window, resident waves. */
int fsmc_ctx_last_plan(const fsmc_ctx* ctx, int32_t* chunk_sites, int32_t* max_chunks, int32_t* n_slots);
/* Chunked windows (longer than a wave's workspace holds) rebuild every chunk's beta rows from a checkpoint -- one of the
 * decode's 3.5 sweeps -- except for the window's first chunks, whose rows the backward pass can leave in the workspace
 * ("resident chunks").  chunks = -1 (default): as many as the workspace allows (the limit if one is set, otherwise what
 * the context has earned: see fsmc_ctx_set_workspace_limit); 0: none; n: at most n.  Results do not depend on it. */
int fsmc_ctx_set_resident_chunks(fsmc_ctx* ctx, int32_t chunks);
int fsmc_ctx_last_resident_chunks(const fsmc_ctx* ctx, int32_t* chunks);
/* Two half-groups per wavefront.  A group of at most 32 pairs (a hashing-mode batch of the reference's default size)
 * fills half a wave; with pairing = 1 (default) the IBD decode puts two such groups with nearby windows on one wave,
 * each lane still decoded over its own group's windows (results do not depend on it); half-full groups that find no
 * partner ride in the same kernel as items of their own.  0 = never.
 * fsmc_ctx_last_items: wave work items of the last IBD launch when it paired groups, 0 when it ran them as uploaded. */
int fsmc_ctx_set_pairing(fsmc_ctx* ctx, uint32_t mode);
int fsmc_ctx_last_items(const fsmc_ctx* ctx, int32_t* n_items);
/* Two waves per decode window.  The consumers without state across sites (fsmc_decode_posteriors, fsmc_decode_per_pair,
 * fsmc_decode_sums*) of a model of at most 128 states in array mode: a launch of at most half as many groups (batches of
 * the sums) as the chip holds waves gives every group a workgroup of two waves -- alpha runs up from the window's first
 * site in one while beta runs down from its last in the other, each stores its rows as far as the middle and combines
 * with the other's beyond it (HMM.cpp:672-691, 725-1041: one alpha step, one beta step and one combine per site, the
 * same operations in the same order: results do not depend on it) -- provided the whole windows' rows fit the
 * workspace.  0 (default) = automatic, 1 = never.  fsmc_ctx_last_waves_per_window: 2 if the last such launch did. */
int fsmc_ctx_set_two_wave_windows(fsmc_ctx* ctx, uint32_t mode);
int fsmc_ctx_last_waves_per_window(const fsmc_ctx* ctx, int32_t* waves);
/* 1 when the last IBD launch kept the open segments' per-state posterior sums (FSMC_WANT_MEAN / FSMC_WANT_MAP:
 * HMM.cpp:1212-1229) in LDS instead of the workspace: a launch of fewer wavefronts than the chip's LDS can give
 * (K/4 + 1) KiB each beside the kernel's own -- a small job, whose waves would wait out every round trip of those sums
 * to L2.  The results do not depend on it. */
int fsmc_ctx_last_segment_sums_in_lds(const fsmc_ctx* ctx, int32_t* in_lds);
/* Which kernel the last launch ran: 16 ... 128 = the lane-per-pair kernel compiled for that many states (the exact
 * members 69, 50, 100, or the padded members 16, 32, 48, 64, 80, 96, 112, 128); the wave-group kernel (128 < K <= 1024):
 * 1048 / 1064 / 1080 = four waves per group of 48 / 64 / 80 states (K <= 192 / 256 / 320), 6064 / 7064 / 8064 = six /
 * seven / eight waves of 64 states (K <= 384 / 448 / 512), 8080 / 8096 / 8128 = eight waves of 80 / 96 / 128 states
 * (K <= 640 / 768 / 1024); 0 = the any-K kernel (1024 < K <= 4096: a pair's K-vectors live in the workspace instead of
 * registers -- the same results, far from the roofline). */
int fsmc_ctx_last_kernel(const fsmc_ctx* ctx, int32_t* member);

/* ---- resident inputs ---- */
int fsmc_model_create(fsmc_ctx* ctx, const fsmc_model_desc* desc, fsmc_model** out);
void fsmc_model_destroy(fsmc_model* m);
/* bits: [n_haps][ceil(n_sites/64)] little-endian words, site s at bit (s % 64) of word s / 64. */
int fsmc_haps_upload(fsmc_ctx* ctx, const uint64_t* bits, uint32_t n_haps, uint32_t n_sites);
/* The work list: pairs and the groups (batches) that partition them in order. */
int fsmc_worklist_upload(fsmc_ctx* ctx, const fsmc_pair* pairs, size_t n_pairs, const fsmc_group* groups,
                         size_t n_groups);

/* ---- the hot path, split so that timing can exclude transfers ---- */
/* Launch the IBD decode of the resident work list (asynchronous on the context's stream). */
int fsmc_decode_ibd_launch(fsmc_ctx* ctx, const fsmc_model* m, uint32_t flags);
/* Wait, copy back and order the records like the reference writes them (batch, pair in batch, site).
 * If cap is too small returns FSMC_EOVERFLOW with *n_out = needed. */
int fsmc_decode_ibd_fetch(fsmc_ctx* ctx, fsmc_ibd_record* out, size_t cap, size_t* n_out);
/* Block until the stream is idle. */
/* ---- identification step (scope row f1; replaces the word loop of FastSMC::run, FastSMC.cpp:118-235, with
 * HASHING/SeedHash.hpp:29-136, ExtendHash.hpp:26-128, Match.hpp:29-83, Utils.cpp:22-34) ----
 * Which haplotype pairs of the job share 64-site words over at least min_m centimorgans.  A pair's matching words
 * are merged into one interval while no more than `gap` words in a row are missing; a word whose number of distinct
 * values / n_haps is not above `skip` extends every open interval instead of being compared. */
typedef struct {
  uint32_t window_size;  /* Data::windowSize (haplotypes per side of a job's square), Data.cpp:62-80 */
  uint32_t w_i, w_j;     /* 1-based window numbers of the job */
  int32_t last_job;      /* jobInd == jobs (SeedHash.hpp:97) */
  int32_t j_above_diag;  /* Data::is_j_above_diag */
} fsmc_job_window;

typedef struct {
  uint32_t hap_a, hap_b; /* rows of the word matrix, hap_a < hap_b */
  uint32_t from, to;     /* first site of the first matching word, last site of the last one (Match.hpp:42-52) */
  uint32_t flush_word;   /* word at which the reference's ExtendHash would have reported it (n_words: at the end) */
} fsmc_candidate;

/* words: [n_haps][n_words] host, word w of haplotype h (bit s%64 of word s/64 = allele of site s); global_ids:
 * [n_haps] haplotype numbers in the whole file (2 * sample line + 0/1); gen_pos: [n_sites] Morgans.  Fills `out` with
 * the candidates ordered by (flush_word, hap_a * n_haps + hap_b) -- the order in which fastsmc_amd hands them to
 * HMM::decodeFromHashing.  FSMC_EOVERFLOW: cap too small, *n_out = the number of candidates. */
int fsmc_identify(fsmc_ctx* ctx, const uint64_t* words, uint32_t n_haps, uint32_t n_words, const uint32_t* global_ids,
                  const fsmc_job_window* job, const float* gen_pos, uint32_t n_sites, int32_t gap, float skip,
                  float min_m, fsmc_candidate* out, size_t cap, size_t* n_out);
/* The other knobs of the reference's identification step (DecodingParams.hpp: hashingWordSize, haploid, max_seeds,
 * constReadAhead); fsmc_identify is fsmc_identify_ex with {64, 1, 0, 10}.
 *   word_size   sites per word, 1..64: word w of a haplotype holds sites w*word_size .. +word_size-1 in its LOW bits
 *               (Individuals.hpp:39-50); from/to and the centimorgan test count in these words.
 *   haploid     0: matches are keyed by INDIVIDUAL pairs (ExtendHash.hpp:47-70: haplotype ids rounded down to the
 *               individual, rows 2k and 2k+1 of the matrix): any of the (up to four) haplotype pairs of the job extends
 *               the pair's one interval, the two haplotypes of one individual form a pair, and the candidate names
 *               rows (2 * ind_a, 2 * ind_b), ind_a <= ind_b (locationToPair).  n_haps must be even.
 *   max_seeds   != 0: a seed with more than max_seeds haplotypes is split by the NEXT word, and again, while the words
 *               read ahead last (SeedHash.hpp:41-85): only the pairs that also share those words are extended, to the
 *               last word looked at.
 *   read_ahead  words buffered ahead of the current one, 1..32 (FastSMC.cpp:186-195: while word c is processed
 *               min(n_words, c + read_ahead) words have been read); bounds the splitting above. */
typedef struct {
  uint32_t word_size;
  uint32_t haploid;
  int32_t max_seeds;
  uint32_t read_ahead;
} fsmc_identify_opts;
int fsmc_identify_ex(fsmc_ctx* ctx, const uint64_t* words, uint32_t n_haps, uint32_t n_words,
                     const uint32_t* global_ids, const fsmc_job_window* job, const float* gen_pos, uint32_t n_sites,
                     int32_t gap, float skip, float min_m, const fsmc_identify_opts* opts, fsmc_candidate* out,
                     size_t cap, size_t* n_out);
/* After an fsmc_identify that returned FSMC_EOVERFLOW (*n_out = the count): the complete candidate list of that call, in
 * emission order -- it was finished and kept on the device, so the caller allocates *n_out records and fetches them
 * instead of running the identification a second time.  The kept list is released by the fetch. */
int fsmc_identify_fetch(fsmc_ctx* ctx, fsmc_candidate* out, size_t cap, size_t* n_out);

int fsmc_sync(fsmc_ctx* ctx);
/* Device time (ms, hipEvent) of the last decode call's kernel(s) -- a call of several launches (the sums of more batches
 * than fit one launch) from its first launch to the end of its last, the plane additions in between included; valid
 * after a sync/fetch. */
int fsmc_last_kernel_ms(fsmc_ctx* ctx, float* ms);

/* Diagnostic: shader-clock cycles summed over waves since the last call, {pass B, beta rebuild, alpha sweep,
 * groups}; all zero unless the library was built with -DFSMC_PHASE_STAMPS (never the shipped build). */
int fsmc_phase_cycles(fsmc_ctx* ctx, uint64_t* out, size_t n);

/* Convenience: upload work list + launch + fetch. */
int fsmc_decode_ibd(fsmc_ctx* ctx, const fsmc_model* m, const fsmc_pair* pairs, size_t n_pairs,
                    const fsmc_group* groups, size_t n_groups, uint32_t flags, fsmc_ibd_record* out, size_t cap,
                    size_t* n_out);

/* Posterior of every pair of the resident work list over its group's window, in the reference's batch layout
 * per group: out[group][pos - from][k][lane 0..63], i.e. group g starts at out + offsets[g] floats where
 * offsets[g] = 64*K*sum_{h<g}(to_h - from_h).  Lanes >= n_pairs are zero.  out_floats = capacity of out. */
int fsmc_decode_posteriors(fsmc_ctx* ctx, const fsmc_model* m, float* out, size_t out_floats);

/* writePerPairOutput: mean[n_pairs][S] = sum_k post*exp_times[k]; map[n_pairs][S] = first argmax_k post.
 * Either may be NULL.  Requires whole-sequence groups (from = 0, to = S), as in the reference (HMM.cpp:1378). */
int fsmc_decode_per_pair(fsmc_ctx* ctx, const fsmc_model* m, const float* exp_coal_times, float* mean, int32_t* map);

/* augmentSumOverPairs: sums[S][K] += sum over the pairs of the work list of the posterior
 * (and the 00/01/11 split when the pointers are non-NULL).  Whole-sequence groups only. */
int fsmc_decode_sums(fsmc_ctx* ctx, const fsmc_model* m, float* sums, float* sums00, float* sums01, float* sums11);
/* The same for reference batches of more than 64 pairs (DecodingParams.cpp:301 allows any multiple of 8): the reference
 * sums a whole batch over its pairs, in order, and then adds it (HMM.cpp:1054-1073).  Batch b is the consecutive groups
 * batch_first_group[b] .. batch_first_group[b+1]-1 (n_batches + 1 entries, the first 0, the last the number of groups):
 * they are decoded in turn and share one running sum, so the result is the reference's bit for bit.
 * fsmc_decode_sums treats every group as a batch of its own. */
int fsmc_decode_sums_batches(fsmc_ctx* ctx, const fsmc_model* m, const uint32_t* batch_first_group, size_t n_batches,
                             float* sums, float* sums00, float* sums01, float* sums11);

#ifdef __cplusplus
}
#endif
#endif
