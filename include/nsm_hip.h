/*
 * nsm_hip.h -- C ABI of libnsm_hip.so, the MI355X (gfx950) all-pairs scorer that stands in for
 * the per-pair loop of BIH-CEI/napkon-string-matching.
 *
 * Nothing like this exists upstream (the reference is pure Python); each entry point names the
 * reference code it replaces (paths relative to the reference checkout):
 *
 *   nsm_jaccard_raw_grid     score_func `intersection_vs_union` applied to one operand per item
 *                            (compare/score_functions.py:6-13; the 1xM use in terminology/mesh.py:207-210)
 *   nsm_indel_raw_grid       score_func `fuzzy_match` = rapidfuzz QRatio/100 on one string per item
 *                            (compare/score_functions.py:20-27)
 *   nsm_jaccard_levels_grid  the hot loop of gen_comparable: compare_terms over suffix-nested levels
 *                            with `intersection_vs_union` (types/comparable_data.py:223-232, :248-265),
 *                            the category predicate (:464-490) and the `>= score_threshold` filter (:243)
 *   nsm_indel_levels_grid    the same loop with `fuzzy_match`
 *   nsm_sort_hits            Comparable.sort_by_score (types/comparable.py:69-70), made deterministic:
 *                            (score descending, i ascending, j ascending)
 *
 * Conventions
 *   - every pointer marked "device" is HBM memory owned by the caller (the Python host keeps them
 *     as torch tensors); the library keeps no pointer after a call returns and allocates no device memory of
 *     its own that outlives a call (the builders' scratch is stream-ordered and freed inside the call; the one
 *     grid that wants scratch -- nsm_indel_levels_grid -- takes it from the caller as `workspace`);
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: 0 on success, otherwise a hipError_t or one of the NSM_E_* codes below;
 *     nsm_last_error() gives a thread-local message;
 *   - data errors that the reference raises as Python exceptions (empty-vs-empty Jaccard ->
 *     ZeroDivisionError, zero-level item against an item with levels -> IndexError) depend only on
 *     per-item properties and are detected by the host BEFORE the launch; the kernels define those pairs
 *     as "no hit".  Two items WITHOUT levels score 0 in the reference (types/comparable_data.py:255-258):
 *     the Python host scores such pairs itself and never passes zero-level items down, and a caller of the
 *     levels grids must do the same (the one-word kernel and the NSM_FLAG_PARK / NSM_FLAG_WAVE_WIDE variants
 *     stage a row's strings before they look at its depth).  The multi-word fuzzy levels kernel and the
 *     nsm_*_any_grid entries do take them: such a pair is reported with score 0 when 0 >= threshold;
 *   - hits are appended with one atomic counter; *hit_count keeps counting past `capacity`
 *     (records beyond it are dropped) so the caller can re-run with a larger buffer.
 */
#ifndef NSM_HIP_H
#define NSM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSM_ABI_VERSION 5

#define NSM_E_BADARG 10001   /* inconsistent sizes / unsupported width */
#define NSM_E_UNSUPPORTED 10002

/* One above-threshold pair.  16 bytes, the unit of the RCCL all-gather of hits. */
typedef struct nsm_hit {
  double score; /* exactly the double the reference computes */
  int32_t i;    /* caller's left row id  (left_orig[row]) */
  int32_t j;    /* caller's right row id (right_orig[row]) */
} nsm_hit;

/* Token-id set table of one side, rows sorted by `cnt` DESCENDING (the encoder does this).
 *   ids   device int32 [n][width]  unique ids >= 0, unused slots padded (left: -1, right: -2).  RAW tables: ascending
 *                                  within the row (the builder sorts them; the global inverted index relies on it).
 *                                  Levels tables: first-appearance order, see plen
 *   cnt   device int32 [n]         number of ids in the row (RAW) / in the item's largest level
 *   sig   device uint64[n]         signature word: bits 0..57 = OR over the row's ids of
 *                                  1 << ((((id * 0x9E3779B1u) >> 16) & 0xffff) * 58 >> 16); bits 58..63 =
 *                                  the low c of them set, c = cnt - popcount(bits 0..57) (ids that collided
 *                                  inside the row); a row with c > 6 has all 64 bits set.  The kernels force
 *                                  bits 58..63 of the OTHER side's word to one, so that
 *                                  popcount(a & b) = common hash bits + c >= |A n B|
 *   sig2  device uint64[n]         second, independent signature word (multiplier 0xC2B2AE35u);
 *                                  optional (NULL: the prune has one stage)
 *   orig  device int32 [n]         caller's row id reported in hits
 *   size_start device int32[width+2]  rows with cnt == width - c are [size_start[c], size_start[c+1])
 *                                  (size_start[0] = 0, size_start[width+1] = n); required by the RAW grid
 *   -- levels mode only (NULL in RAW mode) --
 *   nlev  device int32 [n]         number of levels L (>= 1)
 *   plen  device uint8 [n][max_levels]  level l = the first plen[l] ids of the row (non-decreasing)
 *   cat   device uint64[n]         category bit mask, NULL when categories are not filtered
 *   seg, seg_start                 category partition exactly as in nsm_level_items (below): both NULL, or
 *                                  set on both sides together with NSM_CAT_INTERSECT; rows sorted by seg,
 *                                  then by cnt descending
 *   filt  device uint32[n][8]      filter record of the row: {sig lo, sig hi, cat lo, cat hi,
 *                                  plen[min(1,L-1)] | cnt << 8 | nlev << 16, sig1 lo, sig1 hi, 0} where sig1
 *                                  is the signature word (same layout as sig) of the row's first
 *                                  plen[min(1,L-1)] ids, i.e. of the set every step of compare_terms contains
 *                                  (one s_load per 2 rows)
 *   -- global inverted index (optional: NULL = none).  nsm_build_set_table fills both columns when the caller provides
 *      them; the Jaccard grids use the RIGHT table's index to generate candidate pairs instead of visiting all N x M.
 *      RAW: prefix filter -- with the ids of every row in one global order (ascending id) two sets can only reach the
 *      threshold if they share an id among their first few; a host that numbers its tokens by INCREASING corpus frequency
 *      gets the shortest posting lists, any numbering is correct.  Levels: a pair scores above 0 only if it shares an id;
 *      a partitioned table keys its postings by (category segment, id) -- key = seg * vocab + id, 64 vocab keys -- so a
 *      probe only meets rows of its own category --
 *   post        device uint64[n * width]    one entry per (row, id) sorted by (id, position p of the id in its row);
 *                                           entry = row | p << 32 | cnt << 40; the unused tail is zero
 *               (post_format 0).  ABI 5, tables with n <= 2^post_row_bits rows, post_row_bits + 2 log2(width) <= 32 (width
 *               16: 16.7 M rows):
 *                 post_format 1  device uint32[n * width]: entry = row | p << post_row_bits | (cnt - 1) << (post_row_bits +
 *                                log2(width)) -- half the bytes; what a levels table's index needs (only the row is read);
 *                 post_format 2  device uint64[n * width]: that 32-bit entry in the low word, and above it the 27-bit FOLD of
 *                                the row's signature word (bit j = OR of sig bits j, j + 27, j + 54).  With c' = cnt -
 *                                popcount(fold), |A n B| <= popcount(foldA & foldB) + min(c'A, c'B): the RAW grid decides
 *                                most candidates from the entry alone instead of gathering their 8-byte signature words
 *                                (64-byte sectors: two thirds of the HBM traffic of a 1M x 1M grid).
 *               The caller picks the format (and sizes the column) before nsm_build_set_table fills it
 *   post_start  device int32 [5 * keys + 1]  (keys = vocab, or 64 vocab for a partitioned levels table) entries of key t
 *                                           with p < 1 / 2 / 4 / 8 / any are [post_start[5 t], post_start[5 t + 1 / 2 / 3 / 4 / 5])
 *   vocab       every id of the table is < vocab (checked by the builder)
 *   post_sq     HOST values written by the builder: post_sq[c] = sum over keys of (number of its entries in
 *               [post_start[5 t], post_start[5 t + c + 1]))^2 -- what the grid estimates its candidate count from
 */
typedef struct nsm_set_table {
  const int32_t* ids;
  const int32_t* cnt;
  const uint64_t* sig;
  const uint64_t* sig2;
  const int32_t* orig;
  const int32_t* size_start;
  const int32_t* nlev;
  const uint8_t* plen;
  const uint64_t* cat;
  const uint32_t* filt;
  const int32_t* seg;
  const int32_t* seg_start;
  int32_t n;
  int32_t width;      /* 16, 32 or 64 */
  int32_t max_levels; /* row stride of plen */
  int32_t vocab;      /* ids are < vocab (only read when post / post_start are given) */
  const uint64_t* post;
  const int32_t* post_start;
  uint64_t post_sq[5];
  int32_t post_row_bits; /* row bits of the compact posting entry (post_format 1 and 2), else 0 */
  int32_t post_format;   /* 0, 1 or 2: see post */
} nsm_set_table;

/* Code-unit string table of one side, rows sorted by `len` DESCENDING.
 *   codes device uint8 [n][stride]  dense alphabet codes (< alphabet), padded with `alphabet`
 *   len   device int32 [n]
 *   orig  device int32 [n]          caller's row id (RAW) / unused in levels mode
 *   len_start device int32[stride+2] rows with len == stride - c are [len_start[c], len_start[c+1])
 *                                   (len_start[0] = 0, len_start[stride+1] = n); required by the RAW grid
 *   hist  device uint8 [n][32]      symbol histogram: hist[r][b] = number of code units c of row r
 *                                   with (c & 31) == b; optional (NULL: no histogram prune)
 *   hist16 device uint8 [n][16]     (ABI 5) the same histogram over 16 buckets: hist16[r][b] = min(255, hist[r][b] +
 *                                   hist[r][b + 16]); optional.  With it on BOTH sides the RAW grid of 64-unit strings
 *                                   filters in two stages: 16 buckets for every pair (half the arithmetic and half the
 *                                   bytes of the 32-bucket test, which can only pass more pairs), 32 buckets for the
 *                                   pairs that pass (csrc/indel_raw_coarse.hpp); the hits are the same either way.
 *                                   16-byte aligned (a row is read as one 128-bit word)
 */
typedef struct nsm_str_table {
  const uint8_t* codes;
  const int32_t* len;
  const int32_t* orig;
  const int32_t* len_start;
  const uint8_t* hist;
  int32_t n;
  int32_t stride;   /* 64, 128, 256 or 512 code units per row (1, 2, 4 or 8 words of the bit-parallel LCS),
                       the same on both sides of a grid; anything else: NSM_E_UNSUPPORTED */
  int32_t alphabet; /* number of distinct code units, <= 255 */
  const uint8_t* hist16;
} nsm_str_table;

/* Items whose levels are rows of a nsm_str_table (levels mode of fuzzy_match).
 *   first  device int32 [n]  row of level 0 in the string table; level l is row first+l
 *   nlev   device int32 [n]
 *   orig   device int32 [n]  caller's item id (an item may appear in several rows, see seg)
 *   cat    device uint64[n] or NULL: the item's full category mask
 *   -- category partition (both NULL, or both set on BOTH sides) --
 *   seg        device int32 [n]   the ONE category (bit index) this row stands for: an item with k
 *                                 categories has k rows; rows are sorted by seg
 *   seg_start  device int32 [65]  rows of category c are [seg_start[c], seg_start[c+1])
 *   With a partition the grid only visits (left row, right row) pairs of the same category and
 *   reports a pair in its LOWEST common category only, i.e. exactly the pairs with
 *   (cat_left & cat_right) != 0, each once.
 */
typedef struct nsm_level_items {
  const int32_t* first;
  const int32_t* nlev;
  const int32_t* orig;
  const uint64_t* cat;
  const int32_t* seg;
  const int32_t* seg_start;
  int32_t n;
} nsm_level_items;

/* Category predicate (types/comparable_data.py:464-476), applied before scoring when both
 * tables carry `cat`:  match = (cl & cr) != 0  ||  (both_empty_match && cl == 0 && cr == 0). */
#define NSM_CAT_NONE 0
#define NSM_CAT_INTERSECT 1            /* scalar x list, list x scalar(bit), scalar x scalar */
#define NSM_CAT_INTERSECT_OR_BOTH_EMPTY 2 /* list x list */

#define NSM_FLAG_PRUNE 1u /* exact signature / length bound before the full comparison */
#define NSM_FLAG_INDEX 4u    /* nsm_jaccard_raw_grid, nsm_jaccard_levels_grid: candidate generation by inverted index (chosen by
                                itself where it pays; this forces it: the right table's global index when it has one, else
                                the per-tile index built in LDS) */
#define NSM_FLAG_NO_INDEX 8u /* the same grids: never use an inverted index (A/B runs, tests) */
#define NSM_FLAG_TILE_INDEX 64u /* nsm_jaccard_raw_grid: with NSM_FLAG_INDEX, the per-tile LDS index even when the right table
                                   carries a global one (A/B runs, tests) */
#define NSM_FLAG_PARK 16u /* nsm_indel_levels_grid: the round-2 kernel alone (one right tile per wavefront, block-shared
                             park, dense finish inside the kernel) -- instead of the shared-tile kernel for strings beyond 64
                             code units, and instead of the split path (scan kernel -> survivor queue -> finish kernel) that
                             strings up to 64 code units take at thresholds >= 0.7; same hits, A/B runs and tests */
#define NSM_FLAG_SPLIT 128u /* nsm_indel_levels_grid, strings up to 64 code units: take the split path (given a workspace) at ANY
                              threshold -- by itself the library only does from 0.7 up, where few pairs outlive step 1 on
                              every corpus it was measured on; a host that has MEASURED the survival rate (NSM_FLAG_PROBE)
                              knows better */
#define NSM_FLAG_TILE 256u  /* the same grids: the shared-tile kernel (level strings resident in LDS, survivors carried on
                              wave-wide) -- what pays when MANY pairs outlive step 1; by itself the library takes it below
                              0.55 */
#define NSM_FLAG_PROBE 512u /* the same grids, with NSM_FLAG_SPLIT and a workspace: run the scan kernel ONLY -- no finish kernel,
                              no fallback, no hit is written; the workspace's queue counters (words 2 ..) then hold the number
                              of pairs that outlive step 1 (they keep counting past the queue's capacity).  A host probes a
                              SAMPLE of the left rows this way and sizes / routes the real call from the count */
#define NSM_FLAG_ONE_STAGE 1024u /* nsm_indel_raw_grid: the 32-bucket histogram test for every pair even when both tables carry
                                    hist16 (A/B runs, tests) */
#define NSM_FLAG_WAVE_WIDE 2u /* nsm_indel_levels_grid: score every step wave-wide (no block-cooperative
                                 parking of the surviving pairs); same hits, kept for A/B runs and tests */

int nsm_abi_version(void);
const char* nsm_last_error(void);

/* RAW Jaccard: every (i, j) with  |A n B| / |A u B|  >= threshold  (IEEE double division and
 * compare, as Python does).  Pairs with both sets empty never hit (host raises beforehand). */
int nsm_jaccard_raw_grid(const nsm_set_table* left, const nsm_set_table* right, double threshold,
                         uint32_t flags, nsm_hit* hits /*device*/, uint64_t capacity,
                         unsigned long long* hit_count /*device, caller zeroes*/, void* stream);

/* Levels-mode Jaccard: score = sum_{s=1..max(Ll,Lr)} 2^-s * J(level min(s,Ll-1), level min(s,Lr-1))
 * accumulated in double in that order; optional category predicate. */
int nsm_jaccard_levels_grid(const nsm_set_table* left, const nsm_set_table* right, double threshold,
                            int32_t category_mode, uint32_t flags, nsm_hit* hits, uint64_t capacity,
                            unsigned long long* hit_count, void* stream);

/* RAW Indel ratio: ((1 - (la+lb-2*LCS)/(la+lb)) * 100) / 100, 0 when either string is empty. */
int nsm_indel_raw_grid(const nsm_str_table* left, const nsm_str_table* right, double threshold,
                       uint32_t flags, nsm_hit* hits, uint64_t capacity,
                       unsigned long long* hit_count, void* stream);

/* Levels-mode Indel ratio over per-level strings.
 *
 * `workspace` (device, caller-owned, may be NULL) / `workspace_bytes`: scratch for the split path -- scan kernel ->
 * survivor queue -> finish kernel -- that grids of strings up to 64 code units take at thresholds >= 0.7 (with
 * NSM_FLAG_PRUNE and histograms).  nsm_indel_levels_workspace_bytes() says how much such a grid can use (0: the grid
 * never takes that path).  Layout: 64 control words of 8 bytes (word 0 = *hit_count at the start of the call, word 1 =
 * overflow flag, words 2.. = one queue counter per round), then two queue halves of (workspace_bytes - 512) / 16 entries
 * of 8 bytes each.  The library reads nothing from it on entry and keeps no pointer to it: the caller may free or reuse
 * it as soon as the work queued on `stream` by this call has completed (for a captured call: when the graph is
 * destroyed).  Less than asked for = more rounds over the left rows; a queue that overflows anyway is detected on the
 * device -- the hit counter is put back and a single-kernel pass, launched behind the rounds and gated on word 1, redoes
 * the grid (same hits, slower) -- so ANY size is safe; NULL or fewer than 1024 bytes = the single-kernel path alone.
 * After the call has completed, word 1 != 0 says that the overflow path ran (diagnostic; tests read it).
 * Besides the workspace the split path uses a side stream and four events per caller stream (the finish kernel of round
 * k runs beside the scan of round k + 1); they hold no device memory, are created at the first such call on a stream
 * and destroyed by nsm_release().  A call on a stream that is being captured uses `stream` alone.
 * Hits are appended behind the records already counted in hit_count, as everywhere.
 *
 * `expected_survivors`: the caller's estimate of the number of pairs that outlive step 1 (<= 0: the library's guess, 2 % of
 * the pairs a grid visits) -- it decides the queue size nsm_indel_levels_workspace_bytes() asks for and the number of
 * rounds the grid makes with the workspace it is given.  A wrong estimate costs time (more rounds, or the overflow
 * path), never a hit. */
uint64_t nsm_indel_levels_workspace_bytes(const nsm_level_items* left, const nsm_str_table* left_strings,
                                          const nsm_level_items* right, const nsm_str_table* right_strings,
                                          double threshold, uint32_t flags, double expected_survivors);
int nsm_indel_levels_grid(const nsm_level_items* left, const nsm_str_table* left_strings,
                          const nsm_level_items* right, const nsm_str_table* right_strings,
                          double threshold, int32_t category_mode, uint32_t flags, nsm_hit* hits,
                          uint64_t capacity, unsigned long long* hit_count, void* workspace,
                          uint64_t workspace_bytes, double expected_survivors, void* stream);

/* Destroy the side stream and events the library created for `stream` on the current device.  The caller makes sure no nsm_indel_levels_grid work is still queued on that stream.  Returns 0. */
int nsm_release(void* stream);
int nsm_release_all(void); /* the same for every stream of every device */

/* ---------------------------------------------------------------------------------------------------------
 * Builders: plain per-item arrays in device memory (in the CALLER's row order) -> a table the grid functions
 * read, derived columns (signatures, filter records, histograms), the sort the grids rely on and the category
 * partition included.  They replace the per-item half of the reference's preparation that the per-pair loop
 * re-does for every pair (set(...), join_sorted: compare/score_functions.py:10-11,16-17,24-25) and what
 * round 1 of this library left to its Python host (napkon_string_matching_amd/tables.py); a host in any
 * language now only supplies ids / code units / lengths / levels / category masks.
 *
 *   `out`  a table struct whose pointer fields point to CALLER-ALLOCATED device arrays (cast away the const)
 *          and whose scalar fields say how they are laid out: n = capacity in rows on entry, rows built on
 *          return; width / max_levels / stride / alphabet as documented on the struct.  A partitioned table
 *          has one row per (item, category of the item): sum of popcount(cat) rows (<= 64 n; in practice a few
 *          per item).  Optional columns may be NULL (sig2, hist); required ones are checked.
 *   They SORT (stable: ties stay in the caller's order), so the grids never see an unsorted table.
 *   They synchronise `stream` before returning (row count and validation verdict live on the device) and
 *   allocate their scratch stream-ordered (hipMallocAsync / hipFreeAsync): nothing is kept after the call.
 *   Data errors (row wider than the table, duplicate id in a LEVELS row, code unit >= alphabet,
 *   category bit 63 in use where "both empty" must become a category) return NSM_E_BADARG.
 */
#define NSM_BUILD_PARTITION 1u /* levels tables: one row per (item, category), rows grouped by category */
#define NSM_BUILD_VALIDATE 2u  /* (ABI v2 callers; no effect since v3: RAW rows always drop repeated ids -- they are sets --
                                  and a repeated id in a levels row is always NSM_E_BADARG) */
#define NSM_BUILD_SORT 4u      /* string tables: sort by length descending and fill len_start (RAW grid);
                                  without it rows stay in input order (levels mode: items index them) */

/* Token-id set table.  ids_in device int32 [n][width_in]: ids >= 0 in any slots, negative = unused.  side: 0 =
 * left (padding -1), 1 = right (padding -2).  Levels mode when nlev_in != NULL: plen_in device uint8
 * [n][out->max_levels] (level l = the first plen[l] ids; entries past the last level are ignored), cat_in device
 * uint64 [n] or NULL, category_mode as for the grid; with NSM_BUILD_PARTITION and NSM_CAT_INTERSECT_OR_BOTH_EMPTY
 * the empty items become category 63 and the grid is then to be called with NSM_CAT_INTERSECT.  orig_in device
 * int32 [n] or NULL (0 .. n-1): the ids reported in hits. */
int nsm_build_set_table(const int32_t* ids_in, int32_t n, int32_t width_in, int32_t side, const int32_t* nlev_in,
                        const uint8_t* plen_in, const uint64_t* cat_in, const int32_t* orig_in, int32_t category_mode,
                        uint32_t flags, nsm_set_table* out, void* stream);

/* Code-unit string table.  codes_in device uint8 [n][out->stride] (positions >= len are ignored and rewritten to
 * the pad code out->alphabet), len_in device int32 [n].  The hist and hist16 columns are filled when out provides them. */
int nsm_build_str_table(const uint8_t* codes_in, const int32_t* len_in, const int32_t* orig_in, int32_t n, uint32_t flags,
                        nsm_str_table* out, void* stream);

/* Items whose levels are rows first .. first + nlev - 1 of an (unsorted) string table: sorted deeper items first,
 * with NSM_BUILD_PARTITION grouped by category (as nsm_build_set_table). */
int nsm_build_level_items(const int32_t* first_in, const int32_t* nlev_in, const uint64_t* cat_in, const int32_t* orig_in,
                          int32_t n, int32_t category_mode, uint32_t flags, nsm_level_items* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Operands beyond the register-resident kernels (ABI 3): token sets of more than 64 ids, strings of more than 512 code
 * units, alphabets of more than 255 symbols.  The reference has no size limits (compare/score_functions.py:10-13 builds
 * Python sets of any size, :27 hands strings of any length to rapidfuzz).  Plain CSR operands in the CALLER's item order,
 * no derived columns, no pruning: every pair that passes the category predicate is scored in full.  Meant for the FEW
 * items that need it -- a host routes (wide items x all) and (all x wide items) through here and the rest through the
 * grids above (napkon_string_matching_amd/wide.py does).  Caps: 4096 code units per string, 1023 symbols, 65535 ids
 * per item; 64 levels in the nested set layout (none in the independent one, none for strings); beyond them
 * NSM_E_UNSUPPORTED.
 */
typedef struct nsm_any_strings {
  const uint16_t* codes;   /* device: code units of all rows, concatenated */
  const int64_t* offset;   /* device int64 [n_rows + 1]: row r = codes[offset[r] .. offset[r + 1]) */
  int32_t n_rows;
  int32_t alphabet;        /* code units are < alphabet (<= 1023), the same on both sides of a grid */
  int32_t max_len;         /* longest row (<= 4096) */
} nsm_any_strings;

typedef struct nsm_any_items {   /* items whose levels are rows first .. first + nlev - 1 of an nsm_any_strings */
  const int32_t* first;
  const int32_t* nlev;
  const int32_t* orig;     /* caller's item id reported in hits */
  const uint64_t* cat;     /* category masks, or NULL with NSM_CAT_NONE */
  int32_t n;
} nsm_any_items;

typedef struct nsm_any_sets {
  const int32_t* ids;      /* device: every item's distinct ids SORTED BY ID, concatenated */
  const uint8_t* lv;       /* device, parallel to ids: the first level of the item that contains the id (RAW: 0) */
  const int64_t* offset;   /* device int64 [n + 1] */
  const int32_t* nlev;     /* device int32 [n] (RAW: 1) */
  const int32_t* plen;     /* device int32 [n][max_levels]: number of ids in level l (entries past the last level unused) */
  const int32_t* orig;
  const uint64_t* cat;
  int32_t n;
  int32_t max_levels;      /* row stride of plen, the same on both sides */
  int32_t max_ids;         /* most ids in one item (<= 65535) */
  const int32_t* first;    /* NULL: the suffix-nested layout above (at most 64 levels).  Otherwise INDEPENDENT levels -- what the
                              reference scores when a tokenizer makes level l differ from "level l - 1 plus more"
                              (types/comparable_data.py:283-299 re-tokenises every suffix), and items of any depth (a
                              `Variable` of 100 characters has 100 levels, :567-574): device int32 [n], level l of item k is
                              ROW first[k] + l of `offset` (int64 [rows + 1]), its ids sorted by id; lv / plen unused */
} nsm_any_sets;

#define NSM_FLAG_RAW_SCORE 32u /* nsm_*_any_grid: score = score_func(level 0, level 0) (the RAW plugin call) instead of
                                  compare_terms over the levels */

int nsm_indel_any_grid(const nsm_any_items* left, const nsm_any_strings* left_strings, const nsm_any_items* right,
                       const nsm_any_strings* right_strings, double threshold, int32_t category_mode, uint32_t flags,
                       nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, void* stream);
int nsm_jaccard_any_grid(const nsm_any_sets* left, const nsm_any_sets* right, double threshold, int32_t category_mode,
                         uint32_t flags, nsm_hit* hits, uint64_t capacity, unsigned long long* hit_count, void* stream);

/* In-place canonical ordering of the first min(*hit_count, capacity) hits:
 * score descending, then i, then j ascending.  `scratch` is a device buffer of the same capacity (may be NULL when at most
 * 8192 records can be live).
 *   n_hint    0, or the caller's promise that at most n_hint records are live (a host that has read the counter knows):
 *             the launch geometry then follows n_hint instead of capacity -- one launch up to 8192 records whatever the
 *             buffer's size.  With more live records than promised the order is unspecified (no record is lost).
 *   id_limit  0, or the caller's promise that every i and j is in [0, id_limit) -- an exclusive bound of the `orig` ids the
 *             tables were built with (the larger cohort's size when they are row numbers): fewer key bits to sort above
 *             8192 records.  The records are rebuilt from the narrowed keys, so an id outside the range comes back
 *             truncated: pass 0 unless the bound is known (negative ids need 0). */
int nsm_sort_hits(nsm_hit* hits, nsm_hit* scratch, uint64_t capacity, const unsigned long long* hit_count,
                  uint64_t n_hint, uint32_t id_limit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSM_HIP_H */
