/* dalign.h -- C ABI of libdalign.so, the MI355X (gfx950) alignment core.
 *
 * The reference (julbean/describealign v2.0.8) has no FFI layer: its hot path is the Python
 * call sequence in combine() (describealign.py:1098-1122).  This header is the boundary a
 * maintainer would bind with ctypes in place of those calls; every entry
 * point cites the reference lines it replaces.
 *
 * Conventions: plain C, caller-allocated buffers, return 0 = ok / negative = error
 * (da_last_error gives the text the host raises as RuntimeError).  One da_ctx per thread/GPU;
 * no globals.  There is NO CPU backend: da_create fails when no gfx950 device is usable.
 */
#ifndef DALIGN_H
#define DALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct da_ctx da_ctx;

enum {
  DA_OK = 0,
  DA_ERR_ARG = -1,        /* bad argument */
  DA_ERR_DEVICE = -2,     /* HIP runtime failure / no device */
  DA_ERR_CAPACITY = -3,   /* caller buffer too small; required size is written back */
  DA_ERR_MISMATCH = -4,   /* "Alignment failed, are the input files mismatched?" (:699, :992) */
  DA_ERR_STATE = -5       /* call order violated (e.g. features before pcm upload) */
};

enum { DA_PREC_F32 = 0, DA_PREC_BF16 = 1 };       /* similarity-GEMM input precision */
enum { DA_SIDE_VIDEO = 0, DA_SIDE_AUDIO = 1 };
enum { DA_MATCH_HASHED = 0,  /* reference candidate vote (:649-660) applied to GEMM survivors */
       DA_MATCH_DENSE = 1,   /* every pair under the correlation threshold (no hash vote) */
       DA_MATCH_RESIDENT_ROWS = 0x100 }; /* OR-ed into `mode`: the rows passed are exactly those the last
                                da_features_resident calls produced for both sides; their device copies
                                are used in place and nothing is uploaded.  (Either way the non-quiet
                                frame lists are compacted on the device; the host rows are not read.) */

int  da_create(int device_id, int precision, da_ctx** out);
void da_destroy(da_ctx* ctx);
const char* da_last_error(const da_ctx* ctx);
/* ABI version of this header (bumped on any signature change). */
int  da_abi_version(void);

/* ---- PCM residency -------------------------------------------------------------------------
 * Replaces the array parse_audio_from_file returns (describealign.py:149-157): int16 PCM at
 * 44.1 kHz, channels in {1,2}; planar = (C,N) like the reference's array, else interleaved
 * s16le frames as ffmpeg emits them.  The copy into HBM happens here (host->device). */
int da_pcm_upload(da_ctx* ctx, int side, const int16_t* pcm, int64_t n_samples, int channels,
                  int planar);

/* Overlapped ingest (replaces the blocking pipe read + cast of
 * describealign.py:149-157).  da_host_alloc returns page-locked host memory for the decoder to write the
 * PCM into (no context needed; da_host_free releases it).  da_pcm_upload_async enqueues the host->device
 * copy on the context's copy stream and returns immediately; the next da_features_resident of that side
 * waits for the copy ON THE DEVICE.  The source buffer must stay untouched until that
 * da_features_resident call has returned.  So the decode of pair k+1 and its PCIe transfer overlap the
 * kernels of pair k.  (From pageable memory the call still works, but the runtime stages the copy.)
 * da_stats().h2d_ms then holds the copy's own duration (HIP events), after that da_features_resident. */
int da_host_alloc(size_t bytes, void** out);
int da_host_free(void* p);
int da_pcm_upload_async(da_ctx* ctx, int side, const int16_t* pcm, int64_t n_samples, int channels, int planar);

/* Streaming ingest: decoder pipe -> page-locked pieces -> HBM, with no whole-file host buffer (replaces the
 * `ffmpeg ... -f s16le -` child + blocking pipe read + cast of describealign.py:123-157; the frames are taken
 * interleaved, as the decoder emits them).  A da_pcm_stream is owned by ONE thread (a decoder thread) and needs
 * no context: da_pcm_stream_open reserves device memory for frames_hint frames (it grows on demand, device to
 * device); da_pcm_stream_piece enqueues the host->device copy of the next n_frames interleaved frames on the
 * stream's own copy queue and returns at once -- the piece's host buffer must stay untouched until
 * da_pcm_stream_sync has returned (all pieces enqueued so far have left their host buffers; a ring of a few
 * page-locked pieces from da_host_alloc is the intended use: read piece k + 2 from the pipe while piece k is in
 * flight).  da_pcm_adopt hands the device buffer over to `side` of a context in place of da_pcm_upload (no copy:
 * the buffers are swapped; that side's next da_features_resident waits for the last piece ON THE DEVICE) and
 * leaves the stream empty, ready for the next file.  da_pcm_stream_close releases everything.
 * Errors: DA_ERR_ARG / DA_ERR_DEVICE; da_pcm_stream_error returns the message of the last failure. */
typedef struct da_pcm_stream da_pcm_stream;
int da_pcm_stream_open(int device_id, int channels, int64_t frames_hint, da_pcm_stream** out);
int da_pcm_stream_piece(da_pcm_stream* st, const int16_t* frames, int64_t n_frames);
int da_pcm_stream_sync(da_pcm_stream* st);
int64_t da_pcm_stream_frames(const da_pcm_stream* st);
const char* da_pcm_stream_error(const da_pcm_stream* st);
int da_pcm_adopt(da_ctx* ctx, int side, da_pcm_stream* st);
/* da_pcm_adopt that leaves the stream holding what the context held (device buffer and frame count) instead of empty: several
 * resident files -- one in the context, the others in streams -- rotate through one context without a copy (bench.py's stream of
 * distinct pairs).  The context's side must be empty or hold interleaved PCM of the stream's channel count (i.e. have come from a
 * stream itself), else DA_ERR_STATE. */
int da_pcm_exchange(da_ctx* ctx, int side, da_pcm_stream* st);
void da_pcm_stream_close(da_pcm_stream* st);

/* ---- features -------------------------------------------------------------------------------
 * get_energy + get_zero_crossings + get_freq_bands (describealign.py:545-593) as one fused
 * kernel over the resident PCM of `side`.  feats receives 5 rows (energy, zero crossings,
 * 3 band energies) with row stride `row_stride` floats; lengths[0] = energy length
 * ceil(floor(N/105)/2), lengths[1] = length of the other four rows floor(N/210).
 * feats may be NULL to leave the rows on the device only. */
int da_features_resident(da_ctx* ctx, int side, float* feats, int64_t row_stride, int64_t lengths[2]);

/* One-shot convenience: upload + features (side slot DA_SIDE_VIDEO is used as scratch). */
int da_features(da_ctx* ctx, const int16_t* pcm, int64_t n_samples, int channels, int planar,
                float* feats, int64_t row_stride, int64_t lengths[2]);

/* ---- stage 1+2: matching --------------------------------------------------------------------
 * align() "memorizing video" + "matching audio" up to the verified match list
 * (describealign.py:595-673): local-mean subtraction, window norms, hash digits, the
 * 3 x 41-tap windowed correlation as an MFMA GEMM over (audio rows x video rows), exact fp64
 * re-verification of the survivors, Naive-Bayes quality.  Feature rows are host pointers
 * (5 rows, stride `*_stride` floats, lengths as returned by da_features*).
 * Output: matches sorted by (audio frame i, video frame v); *n_out in: capacity, out: count
 * (DA_ERR_CAPACITY with the needed count if too small). */
int da_match(da_ctx* ctx,
             const float* vfeat, int64_t v_stride, const int64_t v_lengths[2],
             const float* afeat, int64_t a_stride, const int64_t a_lengths[2],
             int mode, int64_t audio_row_begin, int64_t audio_row_end,
             int32_t* out_i, int32_t* out_v, double* out_q, int64_t* n_out);

/* Split form of da_match for pipelining: da_match_begin uploads the rows, builds the row lists
 * and ENQUEUES preparation + the similarity GEMM without waiting; da_match_finish waits for it,
 * verifies, sorts and returns the match count (results stay resident for da_match_fetch).  The
 * feature row arrays must stay valid until da_match_finish returns.  Between the two calls the
 * caller may run da_match_fetch for the PREVIOUS pair: it copies on a separate stream and so
 * overlaps the GEMM.  da_match == begin + finish + fetch. */
int da_match_begin(da_ctx* ctx,
                   const float* vfeat, int64_t v_stride, const int64_t v_lengths[2],
                   const float* afeat, int64_t a_stride, const int64_t a_lengths[2],
                   int mode, int64_t audio_row_begin, int64_t audio_row_end);
int da_match_finish(da_ctx* ctx, int64_t* n_matches);

/* Copy out the matches of the most recent da_match (they stay resident on the device until the
 * next da_match): lets a caller run da_match with *n_out = 0 to learn the count (it returns
 * DA_ERR_CAPACITY and the count) and then fetch into exactly sized buffers. */
int da_match_fetch(da_ctx* ctx, int32_t* out_i, int32_t* out_v, double* out_q, int64_t n);

/* Device-to-device hand-over of the resident match list, for the single-long-pair mode tiled over
 * several GPUs: every rank matches its block of audio rows, exports its sorted
 * list into device buffers of the communication library (RCCL all-gather over xGMI), and the gathered
 * list -- concatenated in rank order it is sorted by (i, v) -- is imported as "the resident matches" of
 * the context that runs the chain DP (da_chain_begin).  d_keys holds (i << 32 | v), d_q the qualities;
 * both are DEVICE pointers.  da_match_import_device needs a da_match on the same pair first (any row
 * range): the video row list of that match provides the video ranks. */
int da_match_export_device(da_ctx* ctx, uint64_t* d_keys, double* d_q, int64_t n);
int da_match_import_device(da_ctx* ctx, const uint64_t* d_keys, const double* d_q, int64_t n);
/* The same import in two steps, so that a gather can land IN the context's buffers instead of being copied into
 * them: da_match_import_reserve returns device arrays for n matches (keys (i << 32 | v), qualities), the caller
 * fills the first m <= n entries in (i, v) order -- RCCL receives posted straight into them, and
 * da_match_export_device of this rank's own block --, da_match_import_commit(m) makes them the resident match
 * list.  The list resident before stays readable (da_match_export_device) until the commit. */
int da_match_import_reserve(da_ctx* ctx, int64_t n, uint64_t** d_keys, double** d_q);
int da_match_import_commit(da_ctx* ctx, int64_t n);

/* Release the scratch memory of the matching stage (survivor records, unsorted matches, sort / pass-2
 * scratch, idle chain slots); the resident sorted match list, PCM and feature rows stay.  For long pairs
 * (tens of GB of scratch) and for contexts sharing one device.  Buffers grow back on demand. */
int da_trim(da_ctx* ctx);

/* Correlation values of the similarity GEMM for explicit (i, v) pairs, as the selected
 * precision computes them (testing/diagnostics: "similarity values within 1e-3").
 * corr receives [n][3].  Uses the feature rows of the last da_match call. */
int da_match_corr(da_ctx* ctx, const int32_t* i, const int32_t* v, int64_t n, float* corr);

/* The raw matrix-core accumulators of the similarity GEMM for ONE tile of 32 video rows x 32 audio
 * columns of the last da_match (tile indices into its row lists: every 4th non-quiet video frame,
 * the non-quiet audio frames), formed with the production kernel's operands and MFMA sequence in the
 * context's precision: the operand fragment streams the last launch itself read are multiplied, and
 * acc[j][row][col] = 1 - corr_j(row, col) (DA_PREC_F32) or 1 - guard - corr_j(row, col) (DA_PREC_BF16;
 * guard = 2^-7 + 2^-14, the proven bound on the bf16 rounding: never above the exact 1 - corr_j), j = 0..2 --
 * the accumulator divided by the feature's fixed scale.
 * I.e. exactly what the acceptance test of the epilogue sees.  video_frames[32] / audio_frames[32]
 * receive the frame numbers of the rows / columns (-1 past the end of a list).  Testing /
 * diagnostics ("similarity values within 1e-3 relative", north_star). */
int da_match_dump_tile(da_ctx* ctx, int64_t video_tile, int64_t audio_tile, float* acc, int32_t* video_frames,
                       int32_t* audio_frames);

/* ---- stage 2 DP: heaviest chain -------------------------------------------------------------
 * describealign.py:654-656, :674-698: heaviest chain non-decreasing in both coordinates over
 * matches sorted by (i, v).  *n_path in: capacity, out: length.  min_len = the reference's
 * failure bound max(min(Lv,La)/500, 1050) (:698); shorter -> DA_ERR_MISMATCH.
 * With a context the DP runs on the device (the video rows cut into rank columns, one wavefront per
 * column with its Fenwick tree in LDS, the columns pipelined over the audio rows; every sum is the
 * predecessor's sum plus q, one double addition, as in the reference; qualities must be > 0 as the
 * reference's are, :672).  ctx may be NULL: a host-only utility with the same result for CPU
 * tools and tests (then only the return code reports errors). */
int da_chain(da_ctx* ctx, const int32_t* i, const int32_t* v, const double* q, int64_t n,
             double min_len, int32_t* path_i, int32_t* path_v, int64_t* n_path);

/* The same DP on the matches of the most recent da_match / da_match_finish, which are still
 * resident on the device: nothing but the path travels to the host (a 2 h pair has ~5e7 matches,
 * its path ~1.5e6 points).  da_chain_begin hands the resident match list to the DP and ENQUEUES it on
 * a stream of its own -- one persistent workgroup per pair, so the DPs of several pairs run beside
 * the similarity GEMMs of later pairs -- and returns a ticket; the context is immediately free for
 * the next da_match_begin.  da_chain_finish(ticket) waits for that DP and returns its path
 * (*n_path in: capacity, out: length; DA_ERR_CAPACITY keeps the result collectable).  At most 16
 * tickets may be outstanding per context.  da_chain_resident == begin + finish.
 * ORDER: da_chain_begin ranks the matches with the video row list of the match that produced them, and
 * the next da_match_begin overwrites that list -- so call da_chain_begin (or da_match_import_device +
 * da_chain_begin) BEFORE the next da_match_begin on the context; afterwards it fails with DA_ERR_STATE.
 * Limits: fewer than 2^31 matches and 2^24 distinct video rows. */
int da_chain_begin(da_ctx* ctx, uint64_t* ticket);
/* da_chain_begin for a caller that will wait for this DP before it launches anything else on the device (the sequential
 * align() of describealign.py:654-698, rank 0 of a tiled long pair): the DP gets the whole chip.  da_chain_begin's DPs are
 * meant to run BESIDE the similarity GEMMs of the next pairs and are confined to a few compute units per XCD (a column
 * wavefront on a CU keeps a whole GEMM workgroup off it), which makes a DP that runs alone ~1.4x slower than it need be.
 * Same ticket, same da_chain_finish / da_chain_poll. */
int da_chain_begin_exclusive(da_ctx* ctx, uint64_t* ticket);
int da_chain_finish(da_ctx* ctx, uint64_t ticket, double min_len, int32_t* path_i, int32_t* path_v, int64_t* n_path);
/* One pair of a directory batch (describealign.py:1077-1122, the loop body up to the path DP) in ONE call, for the thread that
 * feeds a GPU: da_features_resident of both sides (PCM resident from da_pcm_upload / da_pcm_upload_async / da_pcm_adopt; the
 * rows are downloaded into v_rows / a_rows, [5][stride] each, ideally page-locked -- or NULL: not downloaded), da_match_begin on
 * those resident rows, da_match_finish, da_chain_begin.  Same results, same da_stats (features_ms / features_bytes: both sides
 * together) as the five calls; *ticket goes to da_chain_poll / da_chain_finish.  mode as da_match. */
int da_pair_stage(da_ctx* ctx, float* v_rows, int64_t v_stride, float* a_rows, int64_t a_stride, int mode,
                  int64_t v_lengths[2], int64_t a_lengths[2], int64_t* n_matches, uint64_t* ticket);
int da_chain_resident(da_ctx* ctx, double min_len, int32_t* path_i, int32_t* path_v, int64_t* n_path);
/* 1 when the DP behind `ticket` has completed (da_chain_finish will not block), 0 while it runs,
 * negative on error.  Lets the thread that feeds the context collect finished DPs between its own
 * calls; a context is still used by one thread at a time. */
int da_chain_poll(da_ctx* ctx, uint64_t ticket);
/* 1 when the streams da_chain_begin's DPs run on are confined to their compute-unit mask (a few CUs per XCD, see
 * da_chain_begin_exclusive), 0 when they are ordinary streams -- no mask asked for (DALIGN_CHAIN_CUS = 0 / off / out of
 * range) or the runtime refused it.  A batch pipeline lets unmasked DPs finish before the next similarity GEMM (spread over
 * the chip they cost it +25 %).  Masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags): work an
 * embedding application puts on the legacy null stream serialises with the DPs in flight. */
int da_chain_masked(const da_ctx* ctx);

/* ---- stage 4: banded line extension + second DP ---------------------------------------------
 * describealign.py:895-993.  a_scaled [La][3], v_scaled [Lv][3] (the scaled feature stacks of
 * :733-741); clusters given as first/last audio x of each line cluster plus (offset, slope)
 * from :890-893.  path receives rows (video_idx, audio_idx, cluster, qual, cum_qual) in frames;
 * *n_rows in: capacity, out: rows.  n_points (optional) receives the banded point count. */
int da_refine(da_ctx* ctx, const double* a_scaled, int64_t La, const double* v_scaled, int64_t Lv,
              const double* cl_x0, const double* cl_x1, const double* cl_offset,
              const double* cl_slope, int n_clusters, double min_len,
              double* path, int64_t* n_rows, int64_t* n_points);

/* ---- audio replacement (--stretch_audio) ------------------------------------------------------
 * replace_aligned_segments (describealign.py:230-416): for every interval between consecutive
 * nodes (audio_times[k], video_times[k]) in seconds, the video's audio is replaced by the audio
 * description -- resampled with the chunked quadratic spline of :233-244 when the rate change is
 * inaudible (or no_pitch_correction), otherwise stretched pitch-preservingly by the lag
 * correlation + drift Viterbi + Hann cross-fade splice of :246-385; intervals shorter than 2 s or
 * with |1 - rate| > 0.1 are left alone.  video / audio are IEEE float16 (C, n) planar arrays, the
 * representation the reference holds PCM in (:156); video is updated in place. */
int da_replace_segments(da_ctx* ctx, uint16_t* video_f16, int64_t n_video, const uint16_t* audio_f16,
                        int64_t n_audio, int channels, const double* audio_times,
                        const double* video_times, int n_nodes, int no_pitch_correction);

/* The whole --stretch_audio block of combine() (describealign.py:1135-1153) plus the int16
 * serialisation of :136 on the PCM resident in the two side slots (da_pcm_upload; DA_SIDE_VIDEO is
 * the track that receives the description): loudness matching, da_replace_segments, peak
 * normalisation to +/-32766, int16, interleaved frames as ffmpeg's s16le input wants them.
 * out receives n_video * channels samples; factors (optional) the per-channel loudness ratios. */
int da_stretch_resident(da_ctx* ctx, const double* audio_times, const double* video_times, int n_nodes,
                        int no_pitch_correction, int16_t* out, int64_t out_capacity_frames,
                        double* factors);

/* Jump schedule of the k-th stretched interval of the most recent replace call: pairs
 * (input sample index, signed jump distance) as in best_jumps (:362-367).  *n in: capacity in
 * pairs, out: pairs.  k out of range -> DA_ERR_ARG; da_stretch_schedule(ctx, -1, NULL, n) returns
 * the number of stretched intervals in *n. */
int da_stretch_schedule(da_ctx* ctx, int k, int64_t* pairs, int64_t* n);

/* ---- timing / roofline counters of the most recent calls ------------------------------------ */
typedef struct da_stats_t {
  double features_ms;        /* device time of the feature kernel (HIP events), last call */
  double features_bytes;     /* algorithmic bytes: 2*C*N read + 5 rows written */
  double prep_ms;            /* mean-sub / norms / digits kernels, both sides */
  double gemm_ms;            /* similarity GEMM kernel */
  double gemm_pairs;         /* (audio rows) x (video rows) evaluated */
  double gemm_flops;         /* 246 flop per pair (2*41*3, describealign.py:664-667) */
  double verify_ms;          /* exact re-verification + compaction + sort + the counts behind it (everything between the GEMM and the resident, sorted match list) */
  double survivors;          /* pairs the GEMM passed to verification */
  double matches;            /* verified matches returned */
  double chain_ms;           /* stage-2 chain DP: device time of the last collected DP (host time with a NULL-ctx call) */
  double refine_kernel_ms;   /* banded evaluation kernels */
  double refine_dp_ms;       /* host second DP */
  double refine_points;
  double h2d_ms;             /* last da_pcm_upload */
  /* audio replacement (ABI v2) */
  double resample_ms;        /* spline solve + evaluation kernels */
  double resample_points;    /* output samples resampled (per channel) */
  double resample_bytes;     /* algorithmic bytes: float16 read once + float16 written */
  double correlate_ms;       /* window energy + lag correlation / arg-max kernels */
  double correlate_windows;  /* (window, lag) pairs evaluated */
  double viterbi_ms;         /* drift Viterbi + back-track */
  double splice_ms;          /* run copy + cross-fades */
  double splice_points;
  double stretch_prepare_ms; /* int16 -> float16 + loudness matching (da_stretch_resident) */
  double stretch_finish_ms;  /* peak normalisation + int16 interleave */
  /* chain DP geometry of the last enqueued DP (ABI v4): rank columns and their width; 0 = one-workgroup kernel */
  double chain_columns;
  double chain_column_width;
  double verify_kernel_ms;   /* of verify_ms: the exact re-verification kernel alone (k_verify, last attempt) */
} da_stats_t;

int da_stats(const da_ctx* ctx, da_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif /* DALIGN_H */
