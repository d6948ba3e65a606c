// Internal declarations shared by the HIP kernel files and the C-ABI host layer (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <cstdlib>
#include <thread>

namespace da {

constexpr int kFrameRate = 210;     // feature frames / s           (describealign.py:546,559,577)
constexpr int kWin = 41;            // correlation window, frames   (:596-597)
constexpr int kTaps = 7;            // hash taps per feature        (:610)
constexpr int kTapStep = 6;         // (:611)
constexpr int kTapStart = 2;        // (:613)
constexpr int kPad = 64;            // zero padding (elements) after every per-frame device row
// bf16 prefilter: guard subtracted from the 1 that rides in the GEMM's spare K slots.
// Both operands are unit-norm windows rounded to bf16 (relative error <= 2^-8 each), products are exact in
// f32, so the computed dot product differs from the exact correlation by at most
// (2^-7 + 2^-16) sum|x_k y_k| <= 2^-7 + 2^-16 (Cauchy-Schwarz, |x| = |y| = 1).  With (1 - guard) in the norm
// slot the accumulator never exceeds the exact (1 - corr): every pair the exact criterion accepts survives
// the prefilter, whatever the data.
constexpr double kBf16Guard = 0.0078125 + 0.00006103515625;      // 2^-7 + 2^-14

// ---- feature kernel tables (built on the host in double, see dalign_api.cpp) ----------------
struct FeatTables {
  float w13[13];          // hann(15)[1:-1] / sum, float32          (:551-552, :563-564)
  float w15[15];          // hann(17)[1:-1] / sum  (5x3 low-pass and the 1x15 band-3 blur)
  float w21[21];          // hann(23)[1:-1] / sum  (7x3 low-pass)
  // separable form of the 630-tap (42x15) and 90-tap (6x15) Hann blurs:
  //   W[i + d*k] = A * (1 - cos(phi*(i+1)) * cos(d*phi*k) + sin(phi*(i+1)) * sin(d*phi*k))
  double a1, cos1[42], sin1[42], ck1[15], sk1[15];
  double a2, cos2[6], sin2[6], ck2[15], sk2[15];
};

struct FeatArgs {
  const int16_t* pcm;
  int64_t n;              // samples per channel
  int64_t stride_c;       // element stride between channels
  int64_t stride_n;       // element stride between consecutive samples of one channel
  int64_t n_energy;       // 105 * floor(n/105): samples the energy row may see
  int64_t n_band;         // 210 * floor(n/210): samples the other rows may see
  int64_t len_energy;     // ceil(floor(n/105)/2)
  int64_t len_other;      // floor(n/210)
  float* out;             // 5 rows
  int64_t row_stride;
};

void launch_features(const FeatArgs& a, int channels, const FeatTables* d_tables, hipStream_t s);

// ---- per-side preparation for matching -----------------------------------------------------
struct PrepArgs {
  const float* feat;      // 5 rows on device
  int64_t row_stride;
  int64_t len[5];         // row lengths (energy may be one longer)
  int64_t lmax;           // max(len)
  int is_video;
  double* ms[5];          // mean-subtracted rows, float64, zero padded by kPad
  double* nrm[5];         // window norms, float64 (len-40 valid entries)
  uint32_t* hash;         // one hash record per frame (HashSlot below): packed base-7 digits, one nibble per tap (audio: | 0x8888888); video also ~(probe-next flags at bit 0 of each nibble)
};
// A frame's hash words of all five features sit together -- audio: 8 words (32 B), video: 16 words (64 B) -- in the order the
// vote reads them: k_verify takes the vote of 8e8 survivors, and with one array per feature every survivor pulled 3-6
// separate cache lines through L2 for 4 bytes each (PMC: 55 GB from the fabric per launch, 68 B per survivor).
//   audio  [0..3] = features 3, 4, 0, 1   [4] = feature 2
//   video  [0..3] = digits of 3, 4, 0, 1  [4..7] = their flags   [8] = digits of 2, [9] = its flags
constexpr int kHashAudioWords = 8, kHashVideoWords = 16;
__host__ __device__ inline int hash_slot(int j) { return j >= 3 ? j - 3 : (j < 2 ? j + 2 : 4); }   // word of feature j's digits in an AUDIO record: features 3, 4, 0, 1, 2
// VIDEO record, in the order the vote reads it: [D3 D4 G3 G4 | D0 D1 G0 G1 | D2 G2 ...] (D digits, G the inverted probe-next flags):
// the first 16 bytes decide "feature 3 or 4 hits", which fails for most pairs -- they never load the second 16 bytes
__host__ __device__ inline int hash_vdigits(int j) { return j == 3 ? 0 : j == 4 ? 1 : j == 0 ? 4 : j == 1 ? 5 : 8; }
__host__ __device__ inline int hash_vflags(int j) { return j == 3 ? 2 : j == 4 ? 3 : j == 0 ? 6 : j == 1 ? 7 : 9; }
void launch_prep(const PrepArgs& a, const double* d_hann41n, hipStream_t s);

// ---- similarity GEMM -----------------------------------------------------------------------
struct MatchArgs {
  const double* msd_v[3]; const double* msd_a[3];             // float64 rows and window norms the operand
  const double* nrmd_v[3]; const double* nrmd_a[3];           //   fragments are built from
  // BOTH operands in MFMA fragment order (scratch, rebuilt per launch), the video rows scaled by -cscale_j / |V|, the
  // audio columns by 1 / |A|.  bf16: [tile][feature 3][step 3][lane 64] x 16 B = 9 KiB per 32 rows / columns;
  // f32: [tile][chunk 16][lane 64] x 16 B = 16 KiB (fragment 21 j + s of a lane at chunk (21 j + s) / 4)
  void* bfv_frag; int64_t bfv_tiles;
  void* bfa_frag; int64_t bfa_tiles;
  float cscale[3];                                            // per-feature scale (a bf16 number) that puts the threshold at 2^30 in the pattern sum
  const int32_t* vlist; int64_t n_v;      // every 4th non-quiet video frame (:629-630)
  const int32_t* alist; int64_t n_a;      // non-quiet audio frames within the requested rows (:657-658)
  unsigned long long* out;                // staged survivor records (see pack_record)
  unsigned long long* out_count;
  unsigned long long capacity;
  int audio_tiles_per_block;
};
constexpr int kBfVideoTileGroup = 96;   // bfv_tiles is a multiple of this: a workgroup of the GEMMs owns 16 (f32), 24 (bf16) or 32 row tiles
constexpr int kBfAudioTilePad = 2;      // bfa_tiles = column tiles + this: the GEMMs request one tile past the one they are working on
constexpr int kBfTileBytes = 9 * 1024, kF32TileBytes = 16 * 1024;
// per-feature scales of the bf16 GEMM for the threshold `thr` on prod_j (1 - corr_j)
void bf16_gemm_scales(double thr, float out[3]);
void launch_match_f32(const MatchArgs& a, hipStream_t s);
void launch_match_bf16(const MatchArgs& a, hipStream_t s);

struct CorrArgs {   // diagnostics: GEMM-precision correlations for explicit pairs
  MatchArgs m; const int32_t* pi; const int32_t* pv; int64_t n; float* corr; int precision;
};
void launch_corr(const CorrArgs& a, hipStream_t s);
// raw MFMA accumulators of one 32 x 32 tile: out [3][32 rows][32 cols], the frames of its rows / columns (-1 = past the end)
void launch_dump_tile(const MatchArgs& a, int64_t vtile, int64_t atile, int bf16, float* d_out, int32_t* d_vframes, int32_t* d_aframes,
                      hipStream_t s);

// ---- exact verification --------------------------------------------------------------------
struct VerifyArgs {
  const unsigned long long* surv; const unsigned long long* n_surv; unsigned long long capacity;
  const double* ms_v[3]; const double* ms_a[3];
  const double* nrm_v[3]; const double* nrm_a[3];
  const uint32_t* hash_v; const uint32_t* hash_a;     // hash records (see PrepArgs::hash)
  int mode;
  const int32_t* alist; int64_t n_a;      // survivor records carry the POSITION in the audio row list and reject bits (bf_emit)

  const int32_t* vlist; int64_t n_v;      // to expand staged records (video tile, row mask)
  unsigned long long* n_pairs;            // number of (i, v) pairs the records expand to
  unsigned long long* keys; double* quals; unsigned long long* n_out; unsigned long long out_capacity;
};
void launch_verify(const VerifyArgs& a, unsigned long long n_surv_host, hipStream_t s);

// split sorted keys (i << 32 | v) into two int32 arrays
void launch_unpack_keys(const unsigned long long* keys, int64_t n, int32_t* out_i, int32_t* out_v, hipStream_t s);

// device radix sort of (key, qual) pairs (hipCUB); returns 0 on success
int sort_pairs(unsigned long long* keys_in, unsigned long long* keys_out, double* vals_in, double* vals_out,
               int64_t n, void* temp, size_t* temp_bytes, int begin_bit, int end_bit, hipStream_t s);

// row lists on the device: frames i in [lo, hi) with energy[i] > 0.5 (optionally every 4th of them) -> out,
// count -> d_count[0] (d_count[1] is scratch); `scratch` holds hi - lo ints when every_fourth
size_t select_rows_temp_bytes(int64_t n);
int select_rows(const float* energy, int64_t lo, int64_t hi, bool every_fourth, int32_t* scratch, int32_t* out, int32_t* d_count,
                void* temp, size_t temp_bytes, hipStream_t s);

// ---- pass 2: banded evaluation -------------------------------------------------------------
struct BandArgs {
  const double* a_scaled; int64_t La;     // [La][3]
  const double* v_scaled; int64_t Lv;     // [Lv][3]
  double offset, slope;
  int64_t lo, hi;                         // audio frames [lo, hi)
  double a_max, v_max;
};
// sums for the sub-frame offset refinement (:916-930): per block 6 doubles
//   [count_valid, sum(dv*err), sum(dv*dv), sum(err*err)] over valid rows
void launch_band_refine(const BandArgs& a, double* d_partials, int n_blocks, hipStream_t s);
// quality along the line (:931-936): writes y (video position) and qual for x in [lo, hi)
void launch_band_quality(const BandArgs& a, double* d_y, double* d_q, hipStream_t s);
void launch_colmax(const double* d, int64_t n, int stride, double* d_out, hipStream_t s);

// pass 2 with every cluster in one launch
struct BandCluster {
  double offset, slope;
  int64_t lo, hi;         // audio frames [lo, hi)
  int64_t first;          // index of this cluster's first point in the concatenated point arrays
  int32_t refine;         // 1: long enough for the sub-frame refinement (:916)
  int32_t pad;
};
void launch_band_refine_all(const BandArgs& base, const BandCluster* d_cl, int n_clusters, double* d_partials, int n_blocks, hipStream_t s);
void launch_band_quality_all(const BandArgs& base, const BandCluster* d_cl, int n_clusters, int64_t n_points, double* d_y, double* d_q,
                             int32_t* d_cl_of, unsigned long long* d_keys, int32_t* d_ids, hipStream_t s);
void launch_band_heads(const unsigned long long* d_sorted_keys, int64_t n, uint8_t* d_head, hipStream_t s);
void launch_band_gather(const int32_t* d_kept, const int32_t* d_n_kept, const double* d_y, const double* d_q, const int32_t* d_cl_of,
                        const unsigned long long* d_keys_unsorted, double* o_j, double* o_q, int32_t* o_i, int32_t* o_cl, hipStream_t s);
// stable radix sort of (key, id) pairs and selection of flagged ids (hipCUB); temp sizing with d_temp = nullptr
int sort_keys_ids(const unsigned long long* keys_in, unsigned long long* keys_out, const int32_t* ids_in, int32_t* ids_out, int64_t n,
                  void* temp, size_t* temp_bytes, hipStream_t s);
int select_flagged_ids(const int32_t* ids, const uint8_t* flags, int32_t* out, int32_t* d_count, int64_t n, void* temp, size_t* temp_bytes,
                       hipStream_t s);

// ---- stage-2 chain DP on the device (dalign_chain.hip) ---------------------------------------
struct ChainArgs {                         // k_chain_forward
  const double* q; const int32_t* rank;    // per match, sorted by (i, v); rank = 1-based video rank
  const int32_t* row_start; const int32_t* d_nrows;   // first match of every audio row; row count (device)
  int64_t n; int64_t n_ranks; int S;       // S lowest tree levels in global memory, the rest in LDS
  uint4* tree_lo;                          // [n_ranks + 1] 16-byte (sum, id + 1) nodes, zeroed
  int32_t* pred;                           // [n] predecessor ids (-1 = none)
  int64_t* meta;                           // [0] id of the heaviest point (-1 = none), [1] path length
  int xcd; int* claim;                     // wanted XCD (-1: one block, no election); claim[0] taken, claim[1] arrivals
};
struct ChainLaunch {
  const unsigned long long* keys; const double* q; int64_t n;   // sorted (i << 32 | v), qualities
  const int32_t* rankmap; int64_t rankmap_len;                  // video frame -> 1-based position in the match stage's row list (0: not listed), or NULL when `rank` is filled in
  const int32_t* dense;                                         // with rankmap: [rankmap_len] frames below v that have a match (the rank is that + 1); NULL: the row-list position is the rank
  int64_t n_ranks;
  int32_t* rank; uint8_t* flags; int32_t* row_start; int32_t* d_nrows; int32_t* err;
  void* temp; size_t temp_bytes;                                // hipCUB select scratch (chain_rows_temp_bytes)
  void* tree_lo; int32_t* pred; int32_t* path_ids; int64_t* meta;
  unsigned long long* bt_ec; int32_t* bt_seg;                   // back-track scratch: [n] 64-bit (count, exit) words, [2 * chain_backtrack_segments(n)] ints
  int32_t* out_i; int32_t* out_v;                               // the path, ascending
  int xcd;                                                      // XCD the persistent DP workgroup should sit on (-1: any)
  int wide;                                                     // 1: four wavefronts per row super-step (rows of > ~100 matches)
};
// Column-pipelined form of the same DP (k_chain_columns): the video ranks are cut into `n_cols` columns of
// `width` ranks, one single-wavefront workgroup per column with its Fenwick tree in LDS; the columns form a
// pipeline over the audio rows, handing one (sum, id) record per row to the right through global memory.
struct ChainColumns {
  int n_cols, width;                       // columns are contiguous rank ranges of about equal weight, none wider than `width`
  int32_t* rank_cum;                       // [2 * (n_ranks + 2)] matches per rank; then their exclusive sums, overwritten by the column of every rank
  int32_t* col_rank0;                      // [n_cols + 1] first rank of every column
  int32_t* rowid1;                         // [n] 1-based dense row ordinal of every match (inclusive scan of the row heads)
  uint16_t* key_in; uint16_t* key_out;     // [n] column of every match, before / after the stable partition
  uint32_t* val_in; uint32_t* val_out;     // [n] match ids, before / after
  uint32_t* c_row; uint16_t* c_lr; double* c_q; uint32_t* c_gid;   // [n + 64] matches in column-major order: row ordinal, local rank, quality, id
  int32_t* col_start;                      // [n_cols + 1]
  unsigned long long* msg; int64_t msg_stride;   // [n_cols][msg_stride rows][3] tagged 8-byte granules handed to the right (stride = rows rounded up to a batch)
  uint32_t salt;                           // 1 .. 4095, distinct from every launch that has written `msg` since it was last zeroed
  uint32_t* ctl;                           // [0] column tickets, [1] abort flag; zeroed per launch
  void* temp; size_t temp_bytes;           // hipCUB scratch (chain_columns_temp_bytes)
};
constexpr int kChainCtlHead = 16;
constexpr int kChainStampWords = 40;   // diagnostic builds: 8 tick counters + 32 timeline stamps (8 bytes each) per column, behind the control words
struct ChainColumnPlan { int n_cols, width; };
// max_cols: upper limit from the caller's memory budget (24 bytes x rows x columns of hand-over records), 0 = none;
// the LDS limit on a column's width still sets a minimum
ChainColumnPlan chain_columns_plan(int64_t n, int64_t n_ranks, int64_t max_cols);
size_t chain_columns_temp_bytes(int64_t n, int64_t n_ranks);
size_t chain_columns_lds_bytes(int width);
int chain_columns_batch_rows();
// partition + forward DP + back-track + gather, all on stream s (prep has run); -1 = out of range
int launch_chain_columns(const ChainLaunch& c, const ChainColumns& cc, hipStream_t s);

int chain_tree_shift(int64_t n_ranks);
int64_t chain_backtrack_segments(int64_t n);
size_t chain_rows_temp_bytes(int64_t n);
// prep = per-match ranks / row-head flags / validation; dp = row starts, forward DP, back-track, gather.
// Both return -1 when the input is out of the kernels' range.
int launch_chain_prep(const ChainLaunch& c, hipStream_t s, bool columns);
int launch_chain_dp(const ChainLaunch& c, hipStream_t s);
void launch_rankmap(const int32_t* vlist, int64_t n_v, int32_t* rankmap, hipStream_t s);
// dense ranks of the video frames that occur in a sorted match list (see dalign_chain.hip); -1 = out of range
size_t dense_ranks_temp_bytes(int64_t lv);
int launch_dense_ranks(const unsigned long long* keys, int64_t n, int64_t lv, int32_t* used, int32_t* excl, void* temp, size_t temp_bytes, hipStream_t s);
// distinct audio rows of a sorted key list, ADDED to *d_count (zero it first)
void launch_count_rows(const unsigned long long* keys, int64_t n, unsigned long long* d_count, hipStream_t s);

// Wait for a stream by POLLING its completion (hipStreamQuery reads the signal in memory) instead of hipStreamSynchronize's
// blocked wait, whose wake-up travels interrupt -> kernel worker -> this thread and arrived up to 13 ms late on a host whose
// cores are busy with the LP workers of a batch (profiles/r05_pipeline_stalls.txt): the GPU-feeding thread then launched
// everything behind that wait late.  The naps back off: 50 us for the first 5 ms (the per-pair waits that evidence is about),
// then 200 us, and from 50 ms on 1 ms -- a multi-second wait (an 8 h pair's GEMM, the stretch path, da_pcm_stream_close) makes
// ~1 000 calls a second instead of 10-20 000, on cores it shares with the runtime's helper threads and (under a container's
// CPU quota) with the LP workers.  DALIGN_BLOCKING_SYNC=1: the runtime's own wait.
inline hipError_t stream_wait(hipStream_t s) {
  static const bool blocking = std::getenv("DALIGN_BLOCKING_SYNC") != nullptr;
  if (blocking) return hipStreamSynchronize(s);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
    const auto waited = std::chrono::steady_clock::now() - t0;
    const int nap_us = waited < std::chrono::milliseconds(5) ? 50 : waited < std::chrono::milliseconds(50) ? 200 : 1000;
    std::this_thread::sleep_for(std::chrono::microseconds(nap_us));     // (a tight spin is WORSE: 94 of 207 stages late -- the naps leave the core to the runtime's own threads)
  }
}

}  // namespace da
