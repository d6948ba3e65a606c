// C-ABI host layer of libdalign.so (see include/dalign.h).  Owns the HIP context state, the
// device buffers, the two host-side dynamic programmes (sequential by nature) and the glue
// between the kernels.  Compiled with hipcc for gfx950; there is no CPU compute backend.
#include "../../include/dalign.h"
#include "dalign_common.h"
#include "dalign_stretch.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

using namespace da;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    static const bool dbg = std::getenv("DALIGN_DEBUG_TIMES") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t had = cap;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    const auto t1 = std::chrono::steady_clock::now();
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (dbg) {
      const auto t2 = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[DevBuf] %zu -> %zu bytes: hipFree %.3f ms, hipMalloc %.3f ms\n", had, want,
                   std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Side {
  DevBuf pcm;  int64_t n = 0; int channels = 0; int planar = 1;
  hipEvent_t up0 = nullptr, up1 = nullptr; bool upload_pending = false;   // async upload: timing + "PCM has landed"
  DevBuf feat; int64_t feat_stride = 0; int64_t len[2] = {0, 0};
  // match-prep buffers
  DevBuf mfeat;                  // 5 rows uploaded by da_match
  DevBuf ms[5], nrm[5], hash;
  int64_t mlen[5] = {0, 0, 0, 0, 0}; int64_t lmax = 0;
  const float* prep_feat = nullptr;   // device rows the last preparation read (resident rows or the uploaded copy)
};

// One set of buffers for a stage-2 chain DP in flight (or for the results of the last match):
// the sorted matches, the DP's working arrays and its own stream, so that the DP of pair k runs
// beside the similarity GEMM of pair k+1 and its buffers are not reused until it has been collected.
struct ChainSlot {
  DevBuf keys, q;                 // matches sorted by (i, v): packed keys and qualities
  DevBuf rank, flags, rows, pred, tree, ids, out_iv, small, temp;
  // column-pipelined DP: row ordinals, partition keys / ids, column-major matches, per-row records, control words
  DevBuf rowid, ckey, cval, c_row, c_lr, c_q, c_gid, col_start, rank_cum, msg, ctl;
  DevBuf seg;                     // back-track: entry id and output offset of every pred[] segment
  DevBuf dense; int64_t dense_len = 0;   // [2][dense_len + 1] ints: frames with a match (flags), then their exclusive sums (rank - 1 of a frame)
  unsigned launches = 0;          // column DPs that have written `msg` since it was last zeroed (tag salt)
  int64_t rows_hint = 0;          // audio rows of the match that filled this slot (upper bound on the rows with matches), 0 = unknown
  int mode = 0;                   // how the DP in flight was launched: 0 = columns, 1 / 4 = one workgroup of 1 / 4 wavefronts
  hipStream_t stream = nullptr;   // the DP's own stream, confined to a few CUs per XCD (create_stream): DPs that run BESIDE the next pairs' GEMMs
  hipStream_t wide = nullptr;     // ... and an unconfined one for the blocking entry points (da_chain, da_chain_resident: nothing of this context overlaps)
  hipStream_t run = nullptr;      // the one the DP in flight was enqueued on
  hipEvent_t e0 = nullptr, e1 = nullptr, ready = nullptr;
  int64_t n = 0, n_ranks = 0;
  int state = 0;                  // 0 free, 1 holds the last finished match, 2 chain DP enqueued, 3 reserved for an import
  unsigned long long ticket = 0;
  long long* h_small = nullptr;   // pinned copy of `small`: [0] rows | err << 32, [1] best id, [2] path length
  void release() {
    for (DevBuf* b : {&keys, &q, &rank, &flags, &rows, &pred, &tree, &ids, &out_iv, &small, &temp,
                      &rowid, &ckey, &cval, &c_row, &c_lr, &c_q, &c_gid, &col_start, &rank_cum, &msg, &ctl, &seg, &dense}) b->release();
    if (stream) (void)hipStreamDestroy(stream);
    if (wide) (void)hipStreamDestroy(wide);
    for (hipEvent_t e : {e0, e1, ready}) if (e) (void)hipEventDestroy(e);
    if (h_small) (void)hipHostFree(h_small);
    stream = wide = run = nullptr; e0 = e1 = ready = nullptr; h_small = nullptr;
  }
};
constexpr int kMaxChainSlots = 16;

// The column DP's per-(column, row) hand-over records are its largest buffer by far (24 bytes x rows with a match x columns:
// 2 h pair, 2.8e5 rows, 1 024 columns: 7 GB; 8 h pair: ~30 GB).  They belong to the context, not to a slot: a slot takes one for the DP it launches and gives it back when the
// DP is collected, so the context holds as many as DPs were ever in flight together (two or three), not one per slot touched,
// and a pair does not pay a multi-GB hipMalloc + memset (measured: 0.3 ms most of the time, 2-5 s every ~15th call, with the
// GPU-feeding thread stalled inside it).  `launches` travels with the buffer (the tag salt of its next DP).
struct HandoverBuf { DevBuf buf; unsigned launches = 0; };
constexpr size_t kHandoverKeep = 3;
constexpr size_t kHandoverKeepBytes = (size_t)64 << 30;

// Page-locked staging for the host <-> device copies of a call whose caller hands in ordinary (pageable) memory (da_refine: the
// scaled feature stacks of a worker process, the points coming back; the path of a chain DP).  A copy straight from / to pageable
// memory is staged by the runtime through the null stream, which waits for every BLOCKING stream of the device -- the CU-masked
// streams of the chain DPs in flight -- and has the pages pinned and unpinned around it; through this arena the copies are plain
// DMA on the context's own stream.  One block per context, grown by replacement; regions are handed out by bumping an offset
// and all given back by reset() at the next call.
struct PinArena {
  char* base = nullptr; size_t cap = 0, used = 0;
  std::vector<void*> retired;                     // superseded blocks: freed by release() (never in the steady state)
  void reset() { used = 0; }
  void* take(size_t bytes) {
    const size_t at = (used + 255) & ~(size_t)255;
    if (at + bytes > cap) {
      const size_t want = std::max<size_t>((at + bytes) * 2, (size_t)1 << 20);
      void* p = nullptr;
      if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      if (base) { std::memcpy(p, base, used); retired.push_back(base); }      // regions already handed out stay valid in the OLD block
      // (callers keep their pointers: the old block lives until release(); only new regions come from the new block)
      base = static_cast<char*>(p); cap = want;
    }
    used = at + bytes;
    return base + at;
  }
  void release() {
    for (void* p : retired) (void)hipHostFree(p);
    retired.clear();
    if (base) (void)hipHostFree(base);
    base = nullptr; cap = used = 0;
  }
};

// ---- pass 2 on the host: points of the banded extension, state of the second DP
struct BandPoint { double j; int32_t i; int32_t cl; double q; };
struct DpEntry { double j; int32_t i; int32_t cl; double q; double cum; int32_t id; uint32_t gen; };
// Host arrays of da_refine, kept between calls.  A 2 h pair needs ~270 MB of them (1.8e6 points, one cache slot per video
// frame, back-pointers, the path): allocated afresh per call they arrive as new mmap'd pages -- zeroed and faulted in under the
// process-wide address-space lock, which also the GPU-feeding thread, the hand-off threads and the other refine threads of a
// pipeline take for their own buffers: measured, da_refine took 0.6 s per 2 h pair of which the DP itself is 0.085 s, and the
// pipeline delivered 2.0 pairs/s with its LP workers 56 % busy.  The cache slots carry a generation number instead of being
// cleared (60 MB) per call.
struct RefineScratch {
  std::vector<double> hj, hq, fmin, pred_cum, out;
  std::vector<int32_t> hi, hcl, pred;
  std::vector<BandPoint> pts;
  std::vector<int64_t> row_start;
  std::vector<DpEntry> cache, frontier, cl_best;
  std::vector<std::pair<int32_t, double>> rev;
  uint32_t gen = 0;
};

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}


struct DbgTimes {
  bool on; double t0, last; const char* who;
  explicit DbgTimes(const char* w) : on(std::getenv("DALIGN_DEBUG_TIMES") != nullptr), t0(now_ms()), last(t0), who(w) {}
  void at(const char* what) { if (on) { const double t = now_ms(); std::fprintf(stderr, "[%s] %-28s +%8.3f ms\n", who, what, t - last); last = t; } }
  ~DbgTimes() { if (on) std::fprintf(stderr, "[%s] total %8.3f ms (entered at %.3f ms)\n", who, now_ms() - t0, std::fmod(t0, 1e6)); }
};


}  // namespace

struct da_ctx {
  int device = 0;
  int precision = DA_PREC_F32;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  Side side[2];
  DevBuf tables, hann41;
  DevBuf vlist, alist, surv, counters, keys0, q0, keys1, q1, sort_tmp, rankmap, rowscratch, bfv, bfa;
  std::vector<ChainSlot*> slots;  // sorted match lists live in slots (see ChainSlot)
  std::vector<HandoverBuf> handover_free;   // hand-over buffers not in use by a DP (see HandoverBuf)
  int res_slot = -1;              // slot holding the results of the last finished match
  int import_slot = -1;           // slot reserved by da_match_import_reserve (state 3) until da_match_import_commit
  unsigned long long next_ticket = 1;
  int64_t res_lv = 0;             // video frames of the last match (rank map size)
  int64_t res_la = 0;             // audio frames of the last match (with res_lv: the key bits the match sort has to look at)
  DevBuf pair_i, pair_v, pair_c;
  DevBuf ascaled, vscaled, band_y, band_q, band_part, band_tab, band_cl, band_keys, band_ids, band_head, band_out, band_tmp;
  RefineScratch refine;
  PinArena pin;                   // page-locked staging of da_refine's copies
  bool match_ready = false;
  unsigned long long n_match_resident = 0;
  // state carried from da_match_begin to da_match_finish
  bool match_pending = false;
  bool fetch_ready = false;       // results of the last finished match are resident (keys0 / the result slot)
  bool rows_of_resident = false;  // vlist / res_lv / pend_nv still describe the RESIDENT match (no da_match_begin since its finish)
  int pend_mode = 0; int64_t pend_nv = 0; size_t pend_cap = 0;
  hipEvent_t gemm_e0 = nullptr, gemm_e1 = nullptr, prep_e0 = nullptr, prep_e1 = nullptr, feat_e0 = nullptr, feat_e1 = nullptr, feat_e2 = nullptr, feat_e3 = nullptr;
  hipStream_t copy_stream = nullptr;
  // Page-locked words for the counts the host reads back (row counts, survivors, matches, rows / frames with a match).
  // Page-locked for two reasons: the copies are queued with hipMemcpyAsync and some error paths return before the stream is
  // synchronised -- a copy into a stack local could land in a dead frame; and a copy to PAGEABLE memory is staged by the
  // runtime through the null stream, which waits for every blocking stream of the device, i.e. for the CU-masked streams of
  // the chain DPs in flight (hipExtStreamCreateWithCUMask takes no flags): the GEMM of pair k + 1 would start after the DP of pair k.
  unsigned long long* h_pin = nullptr;   // [0..1] row counts (4 x int32), [2] survivors, [3] matches, [4] rows with a match, [5] frames with a match
  MatchArgs last_match{};
  da_stats_t st{};
  // audio replacement (--stretch_audio)
  da::StretchState* stretch = nullptr;
  DevBuf st_video, st_audio, st_out;
};

namespace {

int fail(da_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIP_TRY(c, call)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail((c), DA_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// a slot's hand-over buffer returns to the context (see HandoverBuf); beyond kHandoverKeep the smallest one is freed
void give_back_handover(da_ctx* c, ChainSlot& sl) {
  if (!sl.msg.p) return;
  c->handover_free.push_back(HandoverBuf{sl.msg, sl.launches});
  sl.msg = DevBuf{}; sl.launches = 0;
  // at most kHandoverKeep idle buffers and at most kHandoverKeepBytes of them (always one): the buffers of 2 h pairs are 7 GB,
  // those of an 8 h pair 54 GB -- three of them idle in every context would leave no room for a second context on the device
  auto idle_bytes = [&] { size_t b = 0; for (const HandoverBuf& h : c->handover_free) b += h.buf.cap; return b; };
  while (c->handover_free.size() > kHandoverKeep || (c->handover_free.size() > 1 && idle_bytes() > kHandoverKeepBytes)) {
    size_t small = 0;
    for (size_t k = 1; k < c->handover_free.size(); ++k) if (c->handover_free[k].buf.cap < c->handover_free[small].buf.cap) small = k;
    c->handover_free[small].buf.release();
    c->handover_free.erase(c->handover_free.begin() + (long)small);
  }
}

// the slot's sorted match list -> dense ranks of its video frames (enqueued on the main stream; *h_total is valid after
// the caller's synchronisation)
int enqueue_dense_ranks(da_ctx* c, ChainSlot& sl, int64_t n, int64_t lv, int32_t* h_total) {
  const size_t m = (size_t)std::max<int64_t>(0, lv) + 1;
  HIP_TRY(c, sl.dense.ensure(sizeof(int32_t) * 2 * m));
  const size_t tb = da::dense_ranks_temp_bytes(lv);
  HIP_TRY(c, c->sort_tmp.ensure(tb + 256));
  if (da::launch_dense_ranks(sl.keys.as<unsigned long long>(), n, lv, sl.dense.as<int32_t>(), sl.dense.as<int32_t>() + m, c->sort_tmp.p, tb, c->stream) != 0)
    return fail(c, DA_ERR_DEVICE, "da_match: ranking the matched video frames failed");
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(h_total, sl.dense.as<int32_t>() + m + (m - 1), sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  sl.dense_len = (int64_t)m - 1;
  return DA_OK;
}

// Streams with a compute-unit mask (hipExtStreamCreateWithCUMask).  Mask bit b is CU b / 8 of XCD b % 8 (the driver deals the
// bits round-robin over the eight XCDs).  A chain DP is ~1 000 single-wavefront column workgroups: spread over the chip, each
// one keeps a whole CU from taking a GEMM workgroup (four waves of 428 registers need all four SIMDs empty) while it occupies
// one SIMD of it; confined to a few CUs per XCD the columns pack there and the GEMM keeps the rest.
//   DALIGN_CHAIN_CUS = k      the chain DP's streams get the first k CUs of every XCD (default 8; 0: no mask)
//   DALIGN_CHAIN_CUS = xN     ... N whole XCDs (neighbouring columns then hand over through ONE L2)
//   DALIGN_MAIN_CUS  = rest   the context's main stream (GEMM, verify, sort) gets the complement
bool chain_cu_mask(uint32_t (&mask)[8]) {
  const char* e = std::getenv("DALIGN_CHAIN_CUS");
  for (uint32_t& w : mask) w = 0u;
  if (!e || !*e) e = "8";                      // the default (sweep: profiles/r05_cu_mask_sweep.jsonl); "0" = no mask
  if (*e == 'x' || *e == 'X') {
    const int n = std::max(1, std::min(7, std::atoi(e + 1)));
    for (int b = 0; b < 256; ++b) if (b % 8 < n) mask[b / 32] |= 1u << (b % 32);
    return true;
  }
  const int k = std::atoi(e);
  if (k <= 0 || k >= 32) return false;
  for (int b = 0; b < 8 * k; ++b) mask[b / 32] |= 1u << (b % 32);
  return true;
}
// 1 once a chain DP stream has been created WITH its CU mask, 0 once one had to be created without (no mask asked for, or
// the runtime refused it), -1 before the first chain stream exists: what da_chain_masked reports
std::atomic<int> g_chain_masked{-1};
hipError_t create_stream(hipStream_t* s, bool chain) {
  uint32_t mask[8];
  if (chain_cu_mask(mask)) {
    const char* m = std::getenv("DALIGN_MAIN_CUS");
    if (chain) {
      if (hipExtStreamCreateWithCUMask(s, 8, mask) == hipSuccess) { g_chain_masked.store(1); return hipSuccess; }
      (void)hipGetLastError();                                   // a runtime that refuses the mask: an ordinary stream (slower GEMM beside a DP, same results)
      g_chain_masked.store(0);
      return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    }
    if (m && std::strcmp(m, "rest") == 0) {
      for (uint32_t& w : mask) w = ~w;
      return hipExtStreamCreateWithCUMask(s, 8, mask);
    }
  }
  if (chain) g_chain_masked.store(0);
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// a free chain slot (creating one if needed); -1 when all kMaxChainSlots are in flight
int acquire_slot(da_ctx* c) {
  for (size_t k = 0; k < c->slots.size(); ++k)
    if (c->slots[k]->state == 0) return (int)k;
  if ((int)c->slots.size() >= kMaxChainSlots) return -1;
  ChainSlot* sl = new ChainSlot();
  if (create_stream(&sl->stream, true) != hipSuccess || hipStreamCreateWithFlags(&sl->wide, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&sl->e0) != hipSuccess || hipEventCreate(&sl->e1) != hipSuccess ||
      hipEventCreateWithFlags(&sl->ready, hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc((void**)&sl->h_small, 64, hipHostMallocDefault) != hipSuccess) {
    sl->release(); delete sl; return -1;
  }
  c->slots.push_back(sl);
  return (int)c->slots.size() - 1;
}

void hann_inner(int m, std::vector<double>& w) {       // scipy.signal.windows.hann(m+2)[1:-1]
  w.resize(m);
  const double pi = 3.14159265358979323846;
  for (int u = 0; u < m; ++u) w[u] = 0.5 - 0.5 * std::cos(2.0 * pi * (u + 1) / (m + 1));
}

void norm_f32(int m, float* out) {                     // float32 window / float32 sum (:551-552)
  std::vector<double> w; hann_inner(m, w);
  std::vector<float> f(m);
  float s = 0.f;
  for (int u = 0; u < m; ++u) { f[u] = (float)w[u]; }
  // numpy's float32 pairwise sum on <= 21 elements is a plain left-to-right sum of 8-wide
  // partials; the difference to a double sum is below float32 resolution after the divide.
  double sd = 0; for (int u = 0; u < m; ++u) sd += (double)f[u];
  s = (float)sd;
  for (int u = 0; u < m; ++u) out[u] = f[u] / s;
}

void build_tables(FeatTables& T) {
  norm_f32(13, T.w13); norm_f32(15, T.w15); norm_f32(21, T.w21);
  const double pi = 3.14159265358979323846;
  auto sep = [&](int d, int blur, double& A, double* ci, double* si, double* ck, double* sk) {
    const int m = d * blur;
    const double phi = 2.0 * pi / (m + 1);
    // the reference normalises the float32 window by its float32 sum; mirror the float32 taps
    double sum = 0;
    for (int u = 0; u < m; ++u) sum += (double)(float)(0.5 - 0.5 * std::cos(phi * (u + 1)));
    A = 0.5 / sum;
    for (int i = 0; i < d; ++i) { ci[i] = std::cos(phi * (i + 1)); si[i] = std::sin(phi * (i + 1)); }
    for (int k = 0; k < blur; ++k) { ck[k] = std::cos(phi * d * k); sk[k] = std::sin(phi * d * k); }
  };
  sep(42, 15, T.a1, T.cos1, T.sin1, T.ck1, T.sk1);
  sep(6, 15, T.a2, T.cos2, T.sin2, T.ck2, T.sk2);
}

}  // namespace

extern "C" {

int da_abi_version(void) { return 5; }

const char* da_last_error(const da_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int da_create(int device_id, int precision, da_ctx** out) {
  if (!out) return DA_ERR_ARG;
  *out = nullptr;
  if (precision != DA_PREC_F32 && precision != DA_PREC_BF16) return DA_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return DA_ERR_DEVICE;
  da_ctx* c = new da_ctx();
  c->device = device_id;
  c->precision = precision;
  if (hipSetDevice(device_id) != hipSuccess) { delete c; return DA_ERR_DEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete c; return DA_ERR_DEVICE; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete c; return DA_ERR_DEVICE; }
  if (create_stream(&c->stream, false) != hipSuccess) { delete c; return DA_ERR_DEVICE; }
  (void)hipEventCreate(&c->ev0); (void)hipEventCreate(&c->ev1);
  (void)hipEventCreate(&c->gemm_e0); (void)hipEventCreate(&c->gemm_e1);
  (void)hipEventCreate(&c->prep_e0); (void)hipEventCreate(&c->prep_e1);
  (void)hipEventCreate(&c->feat_e0); (void)hipEventCreate(&c->feat_e1); (void)hipEventCreate(&c->feat_e2); (void)hipEventCreate(&c->feat_e3);
  if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return DA_ERR_DEVICE; }
  if (hipHostMalloc((void**)&c->h_pin, 64, hipHostMallocDefault) != hipSuccess) { c->h_pin = nullptr; da_destroy(c); return DA_ERR_DEVICE; }
  FeatTables T; build_tables(T);
  if (c->tables.ensure(sizeof T) != hipSuccess ||
      hipMemcpy(c->tables.p, &T, sizeof T, hipMemcpyHostToDevice) != hipSuccess) { da_destroy(c); return DA_ERR_DEVICE; }
  std::vector<double> w; hann_inner(kWin, w);
  double s = 0; for (double x : w) s += x;
  for (double& x : w) x /= s;
  if (c->hann41.ensure(sizeof(double) * kWin) != hipSuccess ||
      hipMemcpy(c->hann41.p, w.data(), sizeof(double) * kWin, hipMemcpyHostToDevice) != hipSuccess) { da_destroy(c); return DA_ERR_DEVICE; }
  if (c->counters.ensure(64) != hipSuccess) { da_destroy(c); return DA_ERR_DEVICE; }
  *out = c;
  return DA_OK;
}

void da_destroy(da_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (Side& s : c->side) {
    if (s.up0) (void)hipEventDestroy(s.up0);
    if (s.up1) (void)hipEventDestroy(s.up1);
    s.pcm.release(); s.feat.release(); s.mfeat.release();
    for (int j = 0; j < 5; ++j) { s.ms[j].release(); s.nrm[j].release(); }
    s.hash.release();
  }
  DevBuf* all[] = {&c->tables, &c->hann41, &c->vlist, &c->alist, &c->surv, &c->bfv, &c->bfa, &c->counters, &c->keys0,
                   &c->q0, &c->keys1, &c->q1, &c->sort_tmp, &c->rankmap, &c->rowscratch, &c->pair_i, &c->pair_v, &c->pair_c, &c->ascaled, &c->vscaled,
                   &c->band_y, &c->band_q, &c->band_part, &c->band_tab, &c->band_cl, &c->band_keys, &c->band_ids, &c->band_head,
                   &c->band_out, &c->band_tmp};
  for (DevBuf* b : all) b->release();
  for (ChainSlot* sl : c->slots) {
    if (sl->stream) (void)hipStreamSynchronize(sl->stream);
    if (sl->wide) (void)hipStreamSynchronize(sl->wide);
    sl->release(); delete sl;
  }
  for (HandoverBuf& h : c->handover_free) h.buf.release();
  c->slots.clear();
  c->st_video.release(); c->st_audio.release(); c->st_out.release();
  da::stretch_destroy(c->stretch); c->stretch = nullptr;
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : {c->gemm_e0, c->gemm_e1, c->prep_e0, c->prep_e1, c->feat_e0, c->feat_e1, c->feat_e2, c->feat_e3}) if (e) (void)hipEventDestroy(e);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  c->pin.release();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int da_stats(const da_ctx* c, da_stats_t* out) {
  if (!c || !out) return DA_ERR_ARG;
  *out = c->st;
  return DA_OK;
}

// ------------------------------------------------------------------------------------------ PCM
int da_pcm_upload(da_ctx* c, int side, const int16_t* pcm, int64_t n, int channels, int planar) {
  if (!c) return DA_ERR_ARG;
  if (side < 0 || side > 1 || !pcm || n < 0 || (channels != 1 && channels != 2))
    return fail(c, DA_ERR_ARG, "da_pcm_upload: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  Side& s = c->side[side];
  const size_t bytes = sizeof(int16_t) * (size_t)n * channels;
  HIP_TRY(c, s.pcm.ensure(bytes + 64));
  const double t0 = now_ms();
  if (bytes) HIP_TRY(c, hipMemcpyAsync(s.pcm.p, pcm, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, da::stream_wait(c->stream));
  c->st.h2d_ms = now_ms() - t0;
  s.upload_pending = false;
  s.n = n; s.channels = channels; s.planar = planar ? 1 : 0;
  s.len[0] = s.len[1] = 0;
  return DA_OK;
}

// Overlapped ingest: the decoder writes the PCM into page-locked memory from da_host_alloc;
// da_pcm_upload_async enqueues the host->device copy on the copy stream and returns at once; the feature
// kernel of that side waits for the copy on the device, not on the host.
int da_host_alloc(size_t bytes, void** out) {
  if (!out) return DA_ERR_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocDefault) == hipSuccess ? DA_OK : DA_ERR_DEVICE;
}

int da_host_free(void* p) {
  if (!p) return DA_OK;
  return hipHostFree(p) == hipSuccess ? DA_OK : DA_ERR_DEVICE;
}

int da_pcm_upload_async(da_ctx* c, int side, const int16_t* pcm, int64_t n, int channels, int planar) {
  if (!c) return DA_ERR_ARG;
  if (side < 0 || side > 1 || !pcm || n < 0 || (channels != 1 && channels != 2))
    return fail(c, DA_ERR_ARG, "da_pcm_upload_async: bad argument");
  Side& s = c->side[side];
  const size_t bytes = sizeof(int16_t) * (size_t)n * channels;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, s.pcm.ensure(bytes + 64));
  if (!s.up0) { HIP_TRY(c, hipEventCreate(&s.up0)); HIP_TRY(c, hipEventCreate(&s.up1)); }
  HIP_TRY(c, hipEventRecord(s.up0, c->copy_stream));
  if (bytes) HIP_TRY(c, hipMemcpyAsync(s.pcm.p, pcm, bytes, hipMemcpyHostToDevice, c->copy_stream));
  HIP_TRY(c, hipEventRecord(s.up1, c->copy_stream));
  s.upload_pending = true;
  s.n = n; s.channels = channels; s.planar = planar ? 1 : 0;
  s.len[0] = s.len[1] = 0;
  return DA_OK;
}

// Streaming ingest: a decoder thread's handle.  Its device buffer and copy queue are its own; a context only
// ever sees it in da_pcm_adopt, where the buffers are swapped.
struct da_pcm_stream {
  int device = 0, channels = 0;
  int64_t frames = 0;
  DevBuf buf;
  hipStream_t q = nullptr;
  hipEvent_t t0 = nullptr, landed = nullptr;
  bool started = false;
  std::string err;
};

static int stream_fail(da_pcm_stream* st, int code, const char* what, hipError_t e) {
  if (st) st->err = std::string(what) + ": " + hipGetErrorString(e);
  return code;
}
#define STREAM_TRY(st, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return stream_fail((st), DA_ERR_DEVICE, #call, e_); } while (0)

int da_pcm_stream_open(int device_id, int channels, int64_t frames_hint, da_pcm_stream** out) {
  if (!out) return DA_ERR_ARG;
  *out = nullptr;
  if ((channels != 1 && channels != 2) || frames_hint < 0) return DA_ERR_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return DA_ERR_DEVICE;
  da_pcm_stream* st = new da_pcm_stream();
  st->device = device_id; st->channels = channels;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&st->q, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&st->t0) != hipSuccess || hipEventCreate(&st->landed) != hipSuccess ||
      st->buf.ensure(sizeof(int16_t) * (size_t)frames_hint * channels + 64) != hipSuccess) {
    da_pcm_stream_close(st);
    return DA_ERR_DEVICE;
  }
  *out = st;
  return DA_OK;
}

int da_pcm_stream_piece(da_pcm_stream* st, const int16_t* frames, int64_t n_frames) {
  if (!st || n_frames < 0 || (n_frames > 0 && !frames)) return DA_ERR_ARG;
  if (n_frames == 0) return DA_OK;
  STREAM_TRY(st, hipSetDevice(st->device));
  const size_t fb = sizeof(int16_t) * (size_t)st->channels;
  const size_t have = (size_t)st->frames * fb, add = (size_t)n_frames * fb;
  if (have + add + 64 > st->buf.cap) {                       // the hint was short: grow by half, device to device, in queue order
    DevBuf bigger;
    STREAM_TRY(st, bigger.ensure((have + add) + (have + add) / 2 + 64));
    if (have) STREAM_TRY(st, hipMemcpyAsync(bigger.p, st->buf.p, have, hipMemcpyDeviceToDevice, st->q));
    STREAM_TRY(st, da::stream_wait(st->q));
    st->buf.release();
    st->buf = bigger;
  }
  if (!st->started) { STREAM_TRY(st, hipEventRecord(st->t0, st->q)); st->started = true; }
  STREAM_TRY(st, hipMemcpyAsync(static_cast<char*>(st->buf.p) + have, frames, add, hipMemcpyHostToDevice, st->q));
  st->frames += n_frames;
  return DA_OK;
}

int da_pcm_stream_sync(da_pcm_stream* st) {
  if (!st) return DA_ERR_ARG;
  STREAM_TRY(st, hipSetDevice(st->device));
  STREAM_TRY(st, da::stream_wait(st->q));
  return DA_OK;
}

int64_t da_pcm_stream_frames(const da_pcm_stream* st) { return st ? st->frames : -1; }
const char* da_pcm_stream_error(const da_pcm_stream* st) { return st ? st->err.c_str() : "null stream"; }

void da_pcm_stream_close(da_pcm_stream* st) {
  if (!st) return;
  (void)hipSetDevice(st->device);
  if (st->q) { (void)da::stream_wait(st->q); (void)hipStreamDestroy(st->q); }
  for (hipEvent_t e : {st->t0, st->landed}) if (e) (void)hipEventDestroy(e);
  st->buf.release();
  delete st;
}

int da_pcm_adopt(da_ctx* c, int side, da_pcm_stream* st) {
  if (!c) return DA_ERR_ARG;
  if (side < 0 || side > 1 || !st) return fail(c, DA_ERR_ARG, "da_pcm_adopt: bad argument");
  if (st->device != c->device) return fail(c, DA_ERR_ARG, "da_pcm_adopt: the stream lives on device %d, the context on %d", st->device, c->device);
  HIP_TRY(c, hipSetDevice(c->device));
  Side& s = c->side[side];
  if (!s.up0) { HIP_TRY(c, hipEventCreate(&s.up0)); HIP_TRY(c, hipEventCreate(&s.up1)); }
  // the side's kernels may still be reading its old buffer (it becomes the stream's next target): the context
  // owns that order -- every kernel reading PCM is followed by a stream synchronisation inside the call that launched it
  if (!st->started) HIP_TRY(c, hipEventRecord(st->t0, st->q));
  HIP_TRY(c, hipEventRecord(st->landed, st->q));
  std::swap(s.pcm, st->buf);
  std::swap(s.up0, st->t0); std::swap(s.up1, st->landed);    // up0 .. up1 bracket the pieces: da_stats().h2d_ms as for da_pcm_upload_async
  s.upload_pending = true;
  s.n = st->frames; s.channels = st->channels; s.planar = 0;
  s.len[0] = s.len[1] = 0;
  st->frames = 0; st->started = false;
  return DA_OK;
}

// da_pcm_adopt that gives the stream what the context held -- buffer AND frame count -- instead of leaving it empty: a set of
// resident files (one in the context, the others in streams) can be rotated through one context with nothing copied.
int da_pcm_exchange(da_ctx* c, int side, da_pcm_stream* st) {
  if (!c) return DA_ERR_ARG;
  if (side < 0 || side > 1 || !st) return fail(c, DA_ERR_ARG, "da_pcm_exchange: bad argument");
  Side& s = c->side[side];
  if (s.channels != 0 && (s.planar != 0 || s.channels != st->channels))
    return fail(c, DA_ERR_STATE, "da_pcm_exchange: the context's PCM must be interleaved with the stream's channel count (adopt a stream first)");
  const int64_t held = s.channels != 0 ? s.n : 0;
  const int rc = da_pcm_adopt(c, side, st);
  if (rc != DA_OK) return rc;
  st->frames = held; st->started = held > 0;        // its events (the context's former upload bracket) were recorded when that PCM landed
  return DA_OK;
}

// ------------------------------------------------------------------------------------- features
namespace {
// the fused feature kernel + the download of its rows, enqueued on the context's stream; nothing is waited for.  e0 / e1: the
// events that bracket the kernel (null: not recorded -- the caller brackets several launches itself)
int features_enqueue(da_ctx* c, int side, float* feats, int64_t row_stride, int64_t lengths[2], hipEvent_t e0, hipEvent_t e1) {
  if (side < 0 || side > 1 || !lengths) return fail(c, DA_ERR_ARG, "da_features_resident: bad argument");
  Side& s = c->side[side];
  if (s.channels == 0) return fail(c, DA_ERR_STATE, "da_features_resident: no PCM uploaded for side %d", side);
  const int64_t nb = s.n / 105;
  const int64_t le = (nb + 1) / 2, lo = s.n / 210;
  lengths[0] = le; lengths[1] = lo;
  if (feats && row_stride < le) return fail(c, DA_ERR_CAPACITY, "da_features_resident: row_stride %lld < %lld", (long long)row_stride, (long long)le);
  const int64_t dstride = ((le + 63) / 64) * 64 + 64;
  HIP_TRY(c, s.feat.ensure(sizeof(float) * 5 * (size_t)dstride));
  s.feat_stride = dstride; s.len[0] = le; s.len[1] = lo;
  FeatArgs a{};
  a.pcm = s.pcm.as<int16_t>(); a.n = s.n;
  if (s.planar) { a.stride_c = s.n; a.stride_n = 1; } else { a.stride_c = 1; a.stride_n = s.channels; }
  a.n_energy = 105 * nb; a.n_band = 210 * lo; a.len_energy = le; a.len_other = lo;
  a.out = s.feat.as<float>(); a.row_stride = dstride;
  if (s.upload_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, s.up1, 0));       // the PCM copy of da_pcm_upload_async
  HIP_TRY(c, hipMemsetAsync(s.feat.p, 0, sizeof(float) * 5 * (size_t)dstride, c->stream));
  if (e0) HIP_TRY(c, hipEventRecord(e0, c->stream));
  launch_features(a, s.channels, c->tables.as<FeatTables>(), c->stream);
  HIP_TRY(c, hipGetLastError());
  if (e1) HIP_TRY(c, hipEventRecord(e1, c->stream));
  if (feats && le > 0)
    HIP_TRY(c, hipMemcpy2DAsync(feats, sizeof(float) * row_stride, s.feat.p, sizeof(float) * dstride,
                                sizeof(float) * le, 5, hipMemcpyDeviceToHost, c->stream));
  return DA_OK;
}
// behind a synchronisation of the stream: the asynchronous PCM copy (if any) is done
void features_landed(da_ctx* c, int side) {
  Side& s = c->side[side];
  if (s.upload_pending) {
    float up = 0.f; (void)hipEventElapsedTime(&up, s.up0, s.up1); c->st.h2d_ms = up;
    s.upload_pending = false;
  }
}
}  // namespace

int da_features_resident(da_ctx* c, int side, float* feats, int64_t row_stride, int64_t lengths[2]) {
  if (!c) return DA_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = features_enqueue(c, side, feats, row_stride, lengths, c->ev0, c->ev1)) return rc;
  HIP_TRY(c, da::stream_wait(c->stream));
  float ms = 0.f; (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  c->st.features_ms = ms;
  features_landed(c, side);
  c->st.features_bytes = 2.0 * c->side[side].channels * (double)c->side[side].n + 5.0 * 4.0 * (double)lengths[1];
  return DA_OK;
}

int da_features(da_ctx* c, const int16_t* pcm, int64_t n, int channels, int planar, float* feats,
                int64_t row_stride, int64_t lengths[2]) {
  int rc = da_pcm_upload(c, DA_SIDE_VIDEO, pcm, n, channels, planar);
  if (rc != DA_OK) return rc;
  return da_features_resident(c, DA_SIDE_VIDEO, feats, row_stride, lengths);
}

}  // extern "C"

// ------------------------------------------------------------------------------------- matching
namespace {

int upload_and_prep(da_ctx* c, Side& s, const float* feat, int64_t stride, const int64_t lengths[2], int is_video,
                    bool resident) {
  const int64_t le = lengths[0], lo = lengths[1];
  const int64_t lmax = std::max(le, lo);
  s.lmax = lmax;
  s.mlen[0] = le; for (int j = 1; j < 5; ++j) s.mlen[j] = lo;
  int64_t dstride = lmax + kPad;
  PrepArgs p{};
  if (resident && s.feat.p && s.len[0] == le && s.len[1] == lo && s.feat_stride >= lmax + kPad) {
    // DA_MATCH_RESIDENT_ROWS: the rows da_features_resident left on the device (zero padded) are used
    // in place; nothing is uploaded
    dstride = s.feat_stride;
    p.feat = s.feat.as<float>();
  } else {
    HIP_TRY(c, s.mfeat.ensure(sizeof(float) * 5 * (size_t)dstride));
    HIP_TRY(c, hipMemsetAsync(s.mfeat.p, 0, sizeof(float) * 5 * (size_t)dstride, c->stream));
    for (int j = 0; j < 5; ++j) {
      const int64_t L = s.mlen[j];
      if (L > 0)
        HIP_TRY(c, hipMemcpyAsync(s.mfeat.as<float>() + (size_t)j * dstride, feat + (size_t)j * stride, sizeof(float) * L,
                                  hipMemcpyHostToDevice, c->stream));
    }
    p.feat = s.mfeat.as<float>();
  }
  s.prep_feat = p.feat;                                    // row 0 = the energy row the row lists are built from
  p.row_stride = dstride; p.lmax = lmax; p.is_video = is_video;
  const size_t n = (size_t)dstride;
  for (int j = 0; j < 5; ++j) {
    p.len[j] = s.mlen[j];
    HIP_TRY(c, s.ms[j].ensure(sizeof(double) * n)); p.ms[j] = s.ms[j].as<double>();
    HIP_TRY(c, s.nrm[j].ensure(sizeof(double) * n)); p.nrm[j] = s.nrm[j].as<double>();
  }
  HIP_TRY(c, s.hash.ensure(sizeof(uint32_t) * n * (size_t)(is_video ? da::kHashVideoWords : da::kHashAudioWords))); p.hash = s.hash.as<uint32_t>();
  launch_prep(p, c->hann41.as<double>(), c->stream);
  HIP_TRY(c, hipGetLastError());
  return DA_OK;
}

}  // namespace

static int launch_gemm(da_ctx* c, MatchArgs& m, size_t cap) {
  HIP_TRY(c, c->surv.ensure(sizeof(unsigned long long) * cap));
  m.out = c->surv.as<unsigned long long>(); m.capacity = cap;
  {                                             // both operands in MFMA fragment order: 9 KiB (bf16) / 16 KiB (f32) per 32 rows / columns
    const size_t tile_bytes = c->precision == DA_PREC_F32 ? da::kF32TileBytes : da::kBfTileBytes;
    m.bfv_tiles = (((m.n_v + 31) / 32 + da::kBfVideoTileGroup - 1) / da::kBfVideoTileGroup) * da::kBfVideoTileGroup;   // whole row groups
    HIP_TRY(c, c->bfv.ensure((size_t)m.bfv_tiles * tile_bytes));
    m.bfv_frag = c->bfv.p;
    m.bfa_tiles = (m.n_a + 31) / 32 + da::kBfAudioTilePad;
    HIP_TRY(c, c->bfa.ensure((size_t)m.bfa_tiles * tile_bytes));
    m.bfa_frag = c->bfa.p;
  }
  HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, 64, c->stream));
  HIP_TRY(c, hipEventRecord(c->gemm_e0, c->stream));
  if (c->precision == DA_PREC_F32) launch_match_f32(m, c->stream); else launch_match_bf16(m, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->gemm_e1, c->stream));
  c->pend_cap = cap;
  c->last_match = m;
  return DA_OK;
}

extern "C" int da_match_begin(da_ctx* c, const float* vfeat, int64_t v_stride, const int64_t v_lengths[2],
                              const float* afeat, int64_t a_stride, const int64_t a_lengths[2], int mode,
                              int64_t row_begin, int64_t row_end) {
  if (!c) return DA_ERR_ARG;
  const bool resident_rows = (mode & DA_MATCH_RESIDENT_ROWS) != 0;
  mode &= ~DA_MATCH_RESIDENT_ROWS;
  if (!vfeat || !afeat || !v_lengths || !a_lengths || (mode != 0 && mode != 1))
    return fail(c, DA_ERR_ARG, "da_match: bad argument");
  if (v_lengths[0] > v_stride || a_lengths[0] > a_stride || v_lengths[1] > v_lengths[0] || a_lengths[1] > a_lengths[0])
    return fail(c, DA_ERR_ARG, "da_match: inconsistent lengths");
  HIP_TRY(c, hipSetDevice(c->device));
  c->match_ready = false; c->match_pending = false;       // fetch_ready is untouched: the previous results stay fetchable
  c->rows_of_resident = false;                            // ... but the video row list is about to become the next pair's
  c->res_lv = v_lengths[0];
  c->res_la = a_lengths[0];
  Side& V = c->side[0]; Side& A = c->side[1];
  HIP_TRY(c, hipEventRecord(c->prep_e0, c->stream));
  int rc = upload_and_prep(c, V, vfeat, v_stride, v_lengths, 1, resident_rows); if (rc) return rc;
  rc = upload_and_prep(c, A, afeat, a_stride, a_lengths, 0, resident_rows); if (rc) return rc;
  HIP_TRY(c, hipEventRecord(c->prep_e1, c->stream));

  // row lists from the energy rows (describealign.py:629-630, :657-658)
  int64_t n_v = 0, n_a = 0;
  {
    const int64_t nv = std::max<int64_t>(0, v_lengths[0] - kWin);
    const int64_t na = std::max<int64_t>(0, a_lengths[0] - kWin);
    const int64_t b = std::min(na, std::max<int64_t>(0, row_begin)), e = (row_end < 0) ? na : std::max(b, std::min(na, row_end));
    HIP_TRY(c, c->vlist.ensure(sizeof(int32_t) * (size_t)std::max<int64_t>(1, nv)));
    HIP_TRY(c, c->alist.ensure(sizeof(int32_t) * (size_t)std::max<int64_t>(1, na)));
    {
      // compacted on the device from the energy rows the preparation kernels read (resident or just
      // uploaded); only the two counts come back
      const size_t tb = da::select_rows_temp_bytes(std::max(nv, na));
      HIP_TRY(c, c->sort_tmp.ensure(tb + 256));
      HIP_TRY(c, c->rowscratch.ensure(sizeof(int32_t) * (size_t)std::max<int64_t>(1, nv)));
      int32_t* d_cnt32 = c->counters.as<int32_t>() + 8;             // [8..9] video, [10..11] audio (bytes 32..47 of `counters`)
      if (da::select_rows(V.prep_feat, 0, nv, true, c->rowscratch.as<int32_t>(), c->vlist.as<int32_t>(), d_cnt32, c->sort_tmp.p, tb, c->stream) != 0 ||
          da::select_rows(A.prep_feat, b, e, false, nullptr, c->alist.as<int32_t>(), d_cnt32 + 2, c->sort_tmp.p, tb, c->stream) != 0)
        return fail(c, DA_ERR_DEVICE, "da_match: row list compaction failed");
      int32_t* h_cnt = reinterpret_cast<int32_t*>(c->h_pin);
      HIP_TRY(c, hipMemcpyAsync(h_cnt, d_cnt32, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, da::stream_wait(c->stream));
      n_v = h_cnt[0]; n_a = h_cnt[2];
    }
  }
  c->st.gemm_pairs = (double)n_v * (double)n_a;
  c->st.gemm_flops = 246.0 * c->st.gemm_pairs;
  c->st.survivors = 0; c->st.matches = 0; c->st.gemm_ms = 0; c->st.verify_ms = 0; c->st.verify_kernel_ms = 0;

  MatchArgs m{};
  for (int j = 0; j < 3; ++j) {
    m.msd_v[j] = V.ms[j].as<double>(); m.msd_a[j] = A.ms[j].as<double>();
    m.nrmd_v[j] = V.nrm[j].as<double>(); m.nrmd_a[j] = A.nrm[j].as<double>();
  }
  m.vlist = c->vlist.as<int32_t>(); m.n_v = n_v;
  m.alist = c->alist.as<int32_t>(); m.n_a = n_a;
  // The acceptance threshold (1e-8)^(1/2.9) on prod_j (1 - corr_j) lives in the operand scales (dalign_match.hip).
  // Safety margin: bf16 -- the operand rounding is covered by the guard in the norm slot (kBf16Guard: the accumulators
  // never exceed the exact values), 1.001 covers the f32 accumulation; f32 -- 42 products per accumulator, |error| <= 3e-6
  // on a factor, times the other two factors (<= 4): 1.2e-5 on a product compared with 1.74e-3 = 0.7 %: 1.008.
  // Everything is re-verified in float64 afterwards.
  double thr = std::pow(1e-8, 1.0 / 2.9) * (c->precision == DA_PREC_F32 ? 1.008 : 1.001);
  if (const char* dbg = std::getenv("DALIGN_DEBUG_THR_SCALE")) thr *= std::atof(dbg);   // profiling only
  da::bf16_gemm_scales(thr, m.cscale);
  m.out_count = c->counters.as<unsigned long long>();
  c->pend_mode = mode; c->pend_nv = n_v;
  // survivor records: observed 6e-4 .. 7e-4 per row pair (f32 threshold) and 1.6e-3 (bf16, threshold x2);
  // an overflow is detected and the GEMM re-run with the exact size, so the margin is ~1.6-2x, not more:
  // at 8 bytes per record this buffer is the largest of a long pair (8 h pair tiled over 8 GPUs: 23 GB per rank)
  rc = launch_gemm(c, m, (size_t)std::max(1e6, c->st.gemm_pairs * (c->precision == DA_PREC_F32 ? 1.5e-3 : 2.5e-3)));
  if (rc) return rc;
  c->match_pending = true;                      // nothing has been waited for: the GEMM is in flight
  return DA_OK;
}

extern "C" int da_match_finish(da_ctx* c, int64_t* n_out) {
  if (!c) return DA_ERR_ARG;
  if (!c->match_pending || !n_out) return fail(c, DA_ERR_STATE, "da_match_finish: no da_match_begin in flight");
  HIP_TRY(c, hipSetDevice(c->device));
  c->match_pending = false;
  c->fetch_ready = false;                                   // verification is about to overwrite the result buffers
  if (c->res_slot >= 0 && c->slots[c->res_slot]->state == 1) c->slots[c->res_slot]->state = 0;   // never handed to the chain DP
  c->res_slot = -1;
  Side& V = c->side[0]; Side& A = c->side[1];
  unsigned long long* d_cnt = c->counters.as<unsigned long long>();
  const int mode = c->pend_mode; const int64_t n_v = c->pend_nv;
  unsigned long long n_surv = 0;
  size_t cap = c->pend_cap;
  DbgTimes dbgf("match_finish");
  for (int attempt = 0; attempt < 3; ++attempt) {
    HIP_TRY(c, hipMemcpyAsync(c->h_pin + 2, d_cnt, sizeof n_surv, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, da::stream_wait(c->stream));
    dbgf.at("GEMM done (survivor count read)");
    n_surv = c->h_pin[2];
    float ms = 0.f; (void)hipEventElapsedTime(&ms, c->gemm_e0, c->gemm_e1); c->st.gemm_ms = ms;
    if (n_surv <= cap) break;
    if (attempt == 2) return fail(c, DA_ERR_DEVICE, "da_match: survivor list kept overflowing");
    cap = (size_t)(n_surv + n_surv / 16 + 1024);          // rare: rerun with the exact size
    MatchArgs m = c->last_match;
    int rc = launch_gemm(c, m, cap); if (rc) return rc;
  }
  { float ms = 0.f; (void)hipEventElapsedTime(&ms, c->prep_e0, c->prep_e1); c->st.prep_ms = ms; }
  c->st.survivors = (double)n_surv;
  c->match_ready = true;

  // exact verification + sort
  unsigned long long n_match = 0;
  if (n_surv > 0) {
    size_t mcap = (size_t)n_surv / 2 + 4096;      // verified matches: observed 0.14 .. 0.21 per survivor record; overflow -> retried with the exact count
    VerifyArgs v{};
    v.surv = c->surv.as<unsigned long long>(); v.capacity = cap;
    for (int j = 0; j < 3; ++j) {
      v.ms_v[j] = V.ms[j].as<double>(); v.ms_a[j] = A.ms[j].as<double>();
      v.nrm_v[j] = V.nrm[j].as<double>(); v.nrm_a[j] = A.nrm[j].as<double>();
    }
    v.hash_v = V.hash.as<uint32_t>(); v.hash_a = A.hash.as<uint32_t>();
    v.mode = mode;
    v.alist = c->alist.as<int32_t>(); v.n_a = c->last_match.n_a;
    v.vlist = c->vlist.as<int32_t>(); v.n_v = n_v; v.n_pairs = d_cnt + 2;
    v.n_out = d_cnt + 1;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    for (int attempt = 0; attempt < 2; ++attempt) {
      HIP_TRY(c, c->keys0.ensure(sizeof(unsigned long long) * mcap));
      HIP_TRY(c, c->q0.ensure(sizeof(double) * mcap));
      v.keys = c->keys0.as<unsigned long long>(); v.quals = c->q0.as<double>(); v.out_capacity = mcap;
      HIP_TRY(c, hipMemsetAsync(d_cnt + 1, 0, 2 * sizeof(unsigned long long), c->stream));
      HIP_TRY(c, hipEventRecord(c->prep_e0, c->stream));          // (the prep events have been read above: free for the kernel's own bracket)
      launch_verify(v, n_surv, c->stream);
      HIP_TRY(c, hipGetLastError());
      HIP_TRY(c, hipEventRecord(c->prep_e1, c->stream));
      HIP_TRY(c, hipMemcpyAsync(c->h_pin + 3, d_cnt + 1, sizeof n_match, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, da::stream_wait(c->stream));
      dbgf.at("k_verify done (match count read)");
      n_match = c->h_pin[3];
      if (n_match <= mcap) break;
      if (attempt == 1) return fail(c, DA_ERR_DEVICE, "da_match: match list kept overflowing");
      mcap = (size_t)n_match + 1024;
    }
    const int si = acquire_slot(c);
    if (si < 0) return fail(c, DA_ERR_STATE, "da_match: all %d chain slots are in flight; collect them with da_chain_finish", kMaxChainSlots);
    ChainSlot& sl = *c->slots[si];
    HIP_TRY(c, sl.keys.ensure(sizeof(unsigned long long) * std::max<size_t>(1, n_match)));
    HIP_TRY(c, sl.q.ensure(sizeof(double) * std::max<size_t>(1, n_match)));
    sl.n = (int64_t)n_match; sl.n_ranks = n_v; sl.state = 1;
    sl.rows_hint = c->last_match.n_a;
    c->res_slot = si;
    if (n_match > 0) {
      // keys are (audio frame << 32 | video frame): a stable sort on the video field's bits, then one on the audio field's, touches
      // only the bits that can differ -- 3 + 3 eight-bit passes for a 2 h pair (2^21 frames a side) against the 8 of all 64 bits
      auto field_bits = [](int64_t len) { int b = 1; while (b < 32 && (int64_t(1) << b) < len) ++b; return b; };
      const int vbits = field_bits(c->res_lv), ibits = field_bits(c->res_la);
      size_t tmp_bytes = 0;
      if (sort_pairs(nullptr, nullptr, nullptr, nullptr, (int64_t)n_match, nullptr, &tmp_bytes, 0, 64, c->stream) != 0)
        return fail(c, DA_ERR_DEVICE, "da_match: sort sizing failed");
      HIP_TRY(c, c->sort_tmp.ensure(tmp_bytes + 256));
      HIP_TRY(c, c->keys1.ensure(sizeof(unsigned long long) * n_match));
      HIP_TRY(c, c->q1.ensure(sizeof(double) * n_match));
      size_t tb1 = c->sort_tmp.cap, tb2 = c->sort_tmp.cap;
      if (sort_pairs(c->keys0.as<unsigned long long>(), c->keys1.as<unsigned long long>(), c->q0.as<double>(), c->q1.as<double>(),
                     (int64_t)n_match, c->sort_tmp.p, &tb1, 0, vbits, c->stream) != 0 ||
          sort_pairs(c->keys1.as<unsigned long long>(), sl.keys.as<unsigned long long>(), c->q1.as<double>(), sl.q.as<double>(),
                     (int64_t)n_match, c->sort_tmp.p, &tb2, 32, 32 + ibits, c->stream) != 0)
        return fail(c, DA_ERR_DEVICE, "da_match: device sort failed");
      // keys0 is free again after the sort: unpack the sorted keys into two int32 arrays there
      launch_unpack_keys(sl.keys.as<unsigned long long>(), (int64_t)n_match, c->keys0.as<int32_t>(),
                         c->keys0.as<int32_t>() + n_match, c->stream);
      HIP_TRY(c, hipGetLastError());
    }
    // the audio rows that have a match (a 2 h pair: 2.8e5 of 1.56e6 listed): the column DP sizes its per-(column, row)
    // hand-over records by it -- counted here, behind the sort, and read back with the synchronisation that follows anyway
    unsigned long long n_rows = 0;
    unsigned long long* d_rows = c->counters.as<unsigned long long>() + 7;    // bytes 56..63 of `counters`
    HIP_TRY(c, hipMemsetAsync(d_rows, 0, sizeof n_rows, c->stream));
    if (n_match > 0) da::launch_count_rows(sl.keys.as<unsigned long long>(), (int64_t)n_match, d_rows, c->stream);
    HIP_TRY(c, hipMemcpyAsync(c->h_pin + 4, d_rows, sizeof n_rows, hipMemcpyDeviceToHost, c->stream));
    int32_t* n_used = reinterpret_cast<int32_t*>(c->h_pin + 5);   // ... and the video frames that have one: the DP's ranks
    if (int rc = enqueue_dense_ranks(c, sl, (int64_t)n_match, c->res_lv, n_used)) return rc;
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    dbgf.at("sort + counts enqueued");
    HIP_TRY(c, da::stream_wait(c->stream));
    dbgf.at("sort + counts done");
    n_rows = c->h_pin[4];
    sl.rows_hint = (int64_t)n_rows;
    sl.n_ranks = *n_used;
    float ms = 0.f; (void)hipEventElapsedTime(&ms, c->ev0, c->ev1); c->st.verify_ms = ms;
    ms = 0.f; (void)hipEventElapsedTime(&ms, c->prep_e0, c->prep_e1); c->st.verify_kernel_ms = ms;
  }
  c->st.matches = (double)n_match;
  c->n_match_resident = n_match;
  c->fetch_ready = true;
  c->rows_of_resident = true;
  *n_out = (int64_t)n_match;
  return DA_OK;
}

extern "C" int da_match(da_ctx* c, const float* vfeat, int64_t v_stride, const int64_t v_lengths[2],
                        const float* afeat, int64_t a_stride, const int64_t a_lengths[2], int mode,
                        int64_t row_begin, int64_t row_end, int32_t* out_i, int32_t* out_v, double* out_q,
                        int64_t* n_out) {
  if (!c) return DA_ERR_ARG;
  if (!n_out) return fail(c, DA_ERR_ARG, "da_match: bad argument");
  int rc = da_match_begin(c, vfeat, v_stride, v_lengths, afeat, a_stride, a_lengths, mode, row_begin, row_end);
  if (rc) return rc;
  int64_t n_match = 0;
  rc = da_match_finish(c, &n_match);
  if (rc) return rc;
  const int64_t capacity = *n_out;
  *n_out = n_match;
  if (n_match > capacity) return fail(c, DA_ERR_CAPACITY, "da_match: %lld matches exceed the caller's capacity %lld", (long long)n_match, (long long)capacity);
  return da_match_fetch(c, out_i, out_v, out_q, n_match);
}

extern "C" int da_match_fetch(da_ctx* c, int32_t* out_i, int32_t* out_v, double* out_q, int64_t n) {
  if (!c) return DA_ERR_ARG;
  if (!c->fetch_ready) return fail(c, DA_ERR_STATE, "da_match_fetch: no finished match is resident");
  if (n < 0 || (uint64_t)n > c->n_match_resident) return fail(c, DA_ERR_ARG, "da_match_fetch: n out of range");
  if (n == 0) return DA_OK;
  if (!out_i || !out_v || !out_q) return fail(c, DA_ERR_ARG, "da_match_fetch: null output");
  HIP_TRY(c, hipSetDevice(c->device));
  const int32_t* d_i = c->keys0.as<int32_t>();
  // own copy stream: the results are complete (da_match_finish synchronised), so this copy may run
  // while the next pair's da_match_begin work occupies the compute stream
  HIP_TRY(c, hipMemcpyAsync(out_i, d_i, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->copy_stream));
  HIP_TRY(c, hipMemcpyAsync(out_v, d_i + c->n_match_resident, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->copy_stream));
  if (c->res_slot < 0) return fail(c, DA_ERR_STATE, "da_match_fetch: no finished match is resident");
  HIP_TRY(c, hipMemcpyAsync(out_q, c->slots[c->res_slot]->q.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->copy_stream));
  HIP_TRY(c, da::stream_wait(c->copy_stream));
  return DA_OK;
}

// Device-to-device hand-over of the resident match list (multi-GPU long-pair mode: the lists travel
// between ranks by RCCL in device memory, never through the host).
extern "C" int da_match_export_device(da_ctx* c, uint64_t* d_keys, double* d_q, int64_t n) {
  if (!c) return DA_ERR_ARG;
  if (!c->fetch_ready) return fail(c, DA_ERR_STATE, "da_match_export_device: no finished match is resident");
  if (n < 0 || (uint64_t)n > c->n_match_resident) return fail(c, DA_ERR_ARG, "da_match_export_device: n out of range");
  if (n == 0) return DA_OK;
  if (!d_keys || !d_q || c->res_slot < 0) return fail(c, DA_ERR_ARG, "da_match_export_device: null output");
  HIP_TRY(c, hipSetDevice(c->device));
  ChainSlot& sl = *c->slots[c->res_slot];
  HIP_TRY(c, hipMemcpyAsync(d_keys, sl.keys.p, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_q, sl.q.p, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, da::stream_wait(c->stream));
  return DA_OK;
}

// Import of a gathered match list in two steps, so that the gather can land IN the slot (multi-GPU long-pair mode:
// rank 0 posts its receives straight into these buffers): reserve -> fill the returned device arrays -> commit.
extern "C" int da_match_import_reserve(da_ctx* c, int64_t n, uint64_t** d_keys, double** d_q) {
  if (!c) return DA_ERR_ARG;
  if (!c->match_ready || !c->rows_of_resident)
    return fail(c, DA_ERR_STATE, "da_match_import_reserve: call da_match on this context first (its video row list ranks the matches)");
  if (n < 0 || !d_keys || !d_q) return fail(c, DA_ERR_ARG, "da_match_import_reserve: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->import_slot >= 0 && c->slots[c->import_slot]->state == 3) c->slots[c->import_slot]->state = 0;   // an abandoned reservation
  c->import_slot = -1;
  const int si = acquire_slot(c);
  if (si < 0) return fail(c, DA_ERR_STATE, "da_match_import_reserve: all chain slots are in flight");
  ChainSlot& sl = *c->slots[si];
  const size_t nn = (size_t)std::max<int64_t>(1, n);
  HIP_TRY(c, sl.keys.ensure(sizeof(uint64_t) * nn)); HIP_TRY(c, sl.q.ensure(sizeof(double) * nn));
  sl.state = 3;                                             // reserved: not free, not yet a match list
  sl.n = n;
  c->import_slot = si;
  *d_keys = sl.keys.as<uint64_t>(); *d_q = sl.q.as<double>();
  return DA_OK;
}

extern "C" int da_match_import_commit(da_ctx* c, int64_t n) {
  if (!c) return DA_ERR_ARG;
  if (c->import_slot < 0 || c->slots[c->import_slot]->state != 3) return fail(c, DA_ERR_STATE, "da_match_import_commit: no da_match_import_reserve pending");
  ChainSlot& sl = *c->slots[c->import_slot];
  if (n < 0 || n > sl.n) return fail(c, DA_ERR_ARG, "da_match_import_commit: %lld matches, %lld reserved", (long long)n, (long long)sl.n);
  HIP_TRY(c, hipSetDevice(c->device));
  // the list the context held so far (this rank's own block, usually already copied into the new one) goes
  if (c->res_slot >= 0 && c->slots[c->res_slot]->state == 1) c->slots[c->res_slot]->state = 0;
  c->res_slot = -1;
  const size_t nn = (size_t)std::max<int64_t>(1, n);
  if (n > 0) {
    HIP_TRY(c, c->keys0.ensure(sizeof(uint64_t) * nn));
    launch_unpack_keys(sl.keys.as<unsigned long long>(), n, c->keys0.as<int32_t>(), c->keys0.as<int32_t>() + n, c->stream);
    HIP_TRY(c, hipGetLastError());
  }
  // the audio rows the list covers (the column DP sizes its per-row records by it): counted here, the call synchronises anyway
  unsigned long long n_rows = 0;
  unsigned long long* d_rows = c->counters.as<unsigned long long>() + 7;      // bytes 56..63 of `counters`
  HIP_TRY(c, hipMemsetAsync(d_rows, 0, sizeof n_rows, c->stream));
  if (n > 0) da::launch_count_rows(sl.keys.as<unsigned long long>(), n, d_rows, c->stream);
  HIP_TRY(c, hipMemcpyAsync(c->h_pin + 4, d_rows, sizeof n_rows, hipMemcpyDeviceToHost, c->stream));
  int32_t* n_used = reinterpret_cast<int32_t*>(c->h_pin + 5);
  if (int rc = enqueue_dense_ranks(c, sl, n, c->res_lv, n_used)) return rc;
  HIP_TRY(c, da::stream_wait(c->stream));
  n_rows = c->h_pin[4];
  sl.n = n; sl.n_ranks = *n_used; sl.state = 1;
  sl.rows_hint = (int64_t)n_rows;
  c->res_slot = c->import_slot; c->import_slot = -1;
  c->n_match_resident = (unsigned long long)n;
  c->st.matches = (double)n;
  c->fetch_ready = true;
  return DA_OK;
}

extern "C" int da_match_import_device(da_ctx* c, const uint64_t* d_keys, const double* d_q, int64_t n) {
  if (!c) return DA_ERR_ARG;
  if (n < 0 || (n > 0 && (!d_keys || !d_q))) return fail(c, DA_ERR_ARG, "da_match_import_device: bad argument");
  uint64_t* k = nullptr; double* q = nullptr;
  int rc = da_match_import_reserve(c, n, &k, &q);
  if (rc) return rc;
  if (n > 0) {
    HIP_TRY(c, hipMemcpyAsync(k, d_keys, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(q, d_q, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  }
  return da_match_import_commit(c, n);
}

// Give back the scratch memory of the matching stage (survivor records, unsorted matches, sort and
// pass-2 scratch, idle chain slots): after da_match_finish only the sorted match list is needed.  A long
// pair's scratch runs to tens of GB; contexts that share a device (or a caller about to start the
// chain DP of a 1e9-match list) call this between the stages.  Everything grows back on demand.
extern "C" int da_trim(da_ctx* c) {
  if (!c) return DA_ERR_ARG;
  if (c->match_pending) return fail(c, DA_ERR_STATE, "da_trim: a da_match_begin is in flight");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, da::stream_wait(c->stream));
  HIP_TRY(c, da::stream_wait(c->copy_stream));
  for (DevBuf* b : {&c->surv, &c->bfv, &c->bfa, &c->q0, &c->keys1, &c->q1, &c->sort_tmp, &c->rowscratch, &c->rankmap, &c->band_y, &c->band_q, &c->band_part, &c->band_cl,
                    &c->band_keys, &c->band_ids, &c->band_head, &c->band_out, &c->band_tmp, &c->pair_i, &c->pair_v, &c->pair_c})
    b->release();
  { RefineScratch none; std::swap(c->refine, none); }       // host arrays of da_refine
  // keys0 holds the unpacked (i, v) of the resident matches for da_match_fetch: shrink it to what they need
  c->keys0.release();
  if (c->fetch_ready && c->res_slot >= 0 && c->n_match_resident > 0) {
    const int64_t n = (int64_t)c->n_match_resident;
    HIP_TRY(c, c->keys0.ensure(sizeof(unsigned long long) * (size_t)n));
    launch_unpack_keys(c->slots[c->res_slot]->keys.as<unsigned long long>(), n, c->keys0.as<int32_t>(), c->keys0.as<int32_t>() + n, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, da::stream_wait(c->stream));
  }
  for (HandoverBuf& h : c->handover_free) h.buf.release();
  c->handover_free.clear();
  for (ChainSlot* sl : c->slots)
    if (sl->state == 0)
      for (DevBuf* b : {&sl->keys, &sl->q, &sl->rank, &sl->flags, &sl->rows, &sl->pred, &sl->tree, &sl->ids, &sl->out_iv, &sl->temp,
                        &sl->rowid, &sl->ckey, &sl->cval, &sl->c_row, &sl->c_lr, &sl->c_q, &sl->c_gid, &sl->col_start, &sl->rank_cum, &sl->msg, &sl->ctl, &sl->seg, &sl->dense}) b->release();
  return DA_OK;
}

extern "C" int da_match_corr(da_ctx* c, const int32_t* pi, const int32_t* pv, int64_t n, float* corr) {
  if (!c) return DA_ERR_ARG;
  if (!c->match_ready) return fail(c, DA_ERR_STATE, "da_match_corr: call da_match first");
  if (n < 0 || (n > 0 && (!pi || !pv || !corr))) return fail(c, DA_ERR_ARG, "da_match_corr: bad argument");
  if (n == 0) return DA_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->pair_i.ensure(sizeof(int32_t) * n)); HIP_TRY(c, c->pair_v.ensure(sizeof(int32_t) * n));
  HIP_TRY(c, c->pair_c.ensure(sizeof(float) * 3 * n));
  HIP_TRY(c, hipMemcpyAsync(c->pair_i.p, pi, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->pair_v.p, pv, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  CorrArgs a{c->last_match, c->pair_i.as<int32_t>(), c->pair_v.as<int32_t>(), n, c->pair_c.as<float>(), c->precision};
  launch_corr(a, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(corr, c->pair_c.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, da::stream_wait(c->stream));
  return DA_OK;
}

extern "C" int da_match_dump_tile(da_ctx* c, int64_t video_tile, int64_t audio_tile, float* acc, int32_t* video_frames,
                                  int32_t* audio_frames) {
  if (!c) return DA_ERR_ARG;
  if (!c->match_ready) return fail(c, DA_ERR_STATE, "da_match_dump_tile: call da_match first");
  const MatchArgs& m = c->last_match;
  if (!acc || !video_frames || !audio_frames || video_tile < 0 || audio_tile < 0 || video_tile * 32 >= m.n_v || audio_tile * 32 >= m.n_a)
    return fail(c, DA_ERR_ARG, "da_match_dump_tile: tile (%lld, %lld) outside the %lld x %lld row lists", (long long)video_tile,
                (long long)audio_tile, (long long)((m.n_v + 31) / 32), (long long)((m.n_a + 31) / 32));
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->pair_c.ensure(sizeof(float) * 3 * 32 * 32)); HIP_TRY(c, c->pair_i.ensure(sizeof(int32_t) * 64));
  launch_dump_tile(m, video_tile, audio_tile, c->precision == DA_PREC_BF16, c->pair_c.as<float>(), c->pair_i.as<int32_t>(),
                   c->pair_i.as<int32_t>() + 32, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(acc, c->pair_c.p, sizeof(float) * 3 * 32 * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(video_frames, c->pair_i.p, sizeof(int32_t) * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(audio_frames, c->pair_i.as<int32_t>() + 32, sizeof(int32_t) * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, da::stream_wait(c->stream));
  return DA_OK;
}

// ------------------------------------------------------------------------------------- chain DP
// Heaviest chain non-decreasing in both coordinates (describealign.py:654-656, :674-697).  With a
// context it runs on the device (dalign_chain.hip); with a NULL context the host utility below is
// used (CPU-only callers: tests, tools) -- same recurrence, prefix-max Fenwick tree over the rank of v.
// Equal cumulative weights resolve to the point added later, which is what the reference's
// staircase frontier does (a new entry evicts entries to its right whose weight is not larger,
// :679-680).
namespace {

int check_sorted(da_ctx* c, const int32_t* pi, const int32_t* pv, int64_t n, int32_t& vmax) {
  vmax = -1;
  for (int64_t k = 0; k < n; ++k) {
    if (pv[k] < 0 || pi[k] < 0) return fail(c, DA_ERR_ARG, "da_chain: negative frame index");
    vmax = std::max(vmax, pv[k]);
    if (k > 0 && (pi[k] < pi[k - 1] || (pi[k] == pi[k - 1] && pv[k] <= pv[k - 1])))
      return fail(c, DA_ERR_ARG, "da_chain: input not sorted by (i, v)");
  }
  return DA_OK;
}

int chain_host(da_ctx* c, const int32_t* pi, const int32_t* pv, const double* pq, int64_t n, double min_len,
               int32_t* path_i, int32_t* path_v, int64_t* n_path) {
  const double t0 = now_ms();
  int32_t vmax = -1;
  if (int rc = check_sorted(c, pi, pv, n, vmax)) return rc;
  std::vector<int32_t> rank((size_t)vmax + 2, 0);
  for (int64_t k = 0; k < n; ++k) rank[pv[k]] = 1;
  int32_t nr = 0;
  for (size_t x = 0; x < rank.size(); ++x) { if (rank[x]) rank[x] = ++nr; }
  struct Node { double cum; int32_t id; };
  std::vector<Node> tree((size_t)nr + 1, Node{0.0, -1});
  std::vector<int32_t> pred((size_t)n);
  Node best_all{0.0, -1};
  for (int64_t k = 0; k < n; ++k) {
    const int32_t r = rank[pv[k]];
    Node b{0.0, -1};
    for (int32_t x = r; x > 0; x -= x & -x) {
      const Node& t = tree[x];
      if (t.cum > b.cum || (t.cum == b.cum && t.id > b.id)) b = t;
    }
    pred[k] = b.id;
    const Node me{b.cum + pq[k], (int32_t)k};
    for (int32_t x = r; x <= nr; x += x & -x) {
      Node& t = tree[x];
      if (me.cum > t.cum || (me.cum == t.cum && me.id > t.id)) t = me;
    }
    if (me.cum > best_all.cum || (me.cum == best_all.cum && me.id > best_all.id)) best_all = me;
  }
  std::vector<int32_t> chain;
  for (int32_t p = best_all.id; p >= 0; p = pred[p]) chain.push_back(p);
  std::reverse(chain.begin(), chain.end());
  if (c) c->st.chain_ms = now_ms() - t0;
  const int64_t capacity = *n_path;
  *n_path = (int64_t)chain.size();
  if ((double)chain.size() < min_len) return fail(c, DA_ERR_MISMATCH, "Alignment failed, are the input files mismatched?");
  if ((int64_t)chain.size() > capacity) return fail(c, DA_ERR_CAPACITY, "da_chain: path of %zu exceeds capacity", chain.size());
  for (size_t k = 0; k < chain.size(); ++k) { path_i[k] = pi[chain[k]]; path_v[k] = pv[chain[k]]; }
  return DA_OK;
}

// enqueue the device DP of slot `sl` (sorted keys / q resident).  The per-match ranks come either
// from the video row list of the match (rank_from_vlist) or have been uploaded into sl.rank.
// DALIGN_DEBUG_TIMES=1: wall-clock stamps of the host side of chain_enqueue on stderr (where does the calling thread wait?)
int chain_enqueue(da_ctx* c, ChainSlot& sl, bool rank_from_vlist, bool wide = false) {
  DbgTimes dbg("chain_enqueue");
  const int64_t n = sl.n;
  const size_t nn = (size_t)std::max<int64_t>(1, n);
  HIP_TRY(c, sl.rank.ensure(sizeof(int32_t) * nn)); HIP_TRY(c, sl.flags.ensure(nn));
  // Which kernel: the column pipeline (default), or ONE workgroup of one / four wavefronts walking the rows
  // (DALIGN_CHAIN_KERNEL=rows, then DALIGN_CHAIN_WAVES=1|4: the round-2 kernels, kept as a cross-check).
  {
    const char* kern = std::getenv("DALIGN_CHAIN_KERNEL");
    const char* force = std::getenv("DALIGN_CHAIN_WAVES");
    sl.mode = 0;
    if (kern && std::strcmp(kern, "rows") == 0) sl.mode = force ? (std::atoi(force) >= 4 ? 4 : 1) : (n >= 4096 ? 4 : 1);
  }
  HIP_TRY(c, sl.pred.ensure(sizeof(int32_t) * nn));
  HIP_TRY(c, sl.ids.ensure(sizeof(int32_t) * nn)); HIP_TRY(c, sl.out_iv.ensure(sizeof(int32_t) * 2 * nn));
  // back-track scratch: the column-major quality array (8 B per match) is free once the forward pass is over
  HIP_TRY(c, sl.c_q.ensure(sizeof(double) * nn));
  HIP_TRY(c, sl.seg.ensure(sizeof(int32_t) * 2 * (size_t)std::max<int64_t>(1, da::chain_backtrack_segments(n))));
  HIP_TRY(c, sl.small.ensure(128));
  const size_t tb = sl.mode == 0 ? 0 : da::chain_rows_temp_bytes(n);
  if (sl.mode != 0) {                                                       // the one-workgroup kernels' row starts and tree
    HIP_TRY(c, sl.rows.ensure(sizeof(int32_t) * (nn + 1)));
    HIP_TRY(c, sl.tree.ensure(16 * ((size_t)sl.n_ranks + 2 + 256)));       // + one scrap record per thread
    HIP_TRY(c, sl.temp.ensure(tb + 256));
  }
  dbg.at("ensure (common)");
  hipStream_t st = wide ? sl.wide : sl.stream;
  sl.run = st;
  // `small`: int32 [0] rows, [1] err; int64 [1] best id, [2] path length
  ChainLaunch L{};
  L.keys = sl.keys.as<unsigned long long>(); L.q = sl.q.as<double>(); L.n = n;
  L.n_ranks = sl.n_ranks; L.rank = sl.rank.as<int32_t>(); L.flags = sl.flags.as<uint8_t>();
  L.row_start = sl.rows.as<int32_t>(); L.d_nrows = sl.small.as<int32_t>(); L.err = sl.small.as<int32_t>() + 1;
  L.temp = sl.temp.p; L.temp_bytes = tb;
  L.tree_lo = sl.tree.p; L.pred = sl.pred.as<int32_t>(); L.path_ids = sl.ids.as<int32_t>();
  L.bt_ec = sl.c_q.as<unsigned long long>(); L.bt_seg = sl.seg.as<int32_t>();
  L.xcd = (int)(c->next_ticket % 8);       // successive DPs go to successive XCDs
  L.meta = sl.small.as<int64_t>() + 1; L.out_i = sl.out_iv.as<int32_t>(); L.out_v = sl.out_iv.as<int32_t>() + nn;
  if (rank_from_vlist) {
    // the rank map is built from this match's video row list on the MAIN stream (the next
    // da_match_begin overwrites that list); the slot's stream waits for it
    HIP_TRY(c, c->rankmap.ensure(sizeof(int32_t) * (size_t)std::max<int64_t>(1, c->res_lv)));
    HIP_TRY(c, hipMemsetAsync(c->rankmap.p, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(1, c->res_lv), c->stream));
    da::launch_rankmap(c->vlist.as<int32_t>(), c->pend_nv, c->rankmap.as<int32_t>(), c->stream);
    L.rankmap = c->rankmap.as<int32_t>(); L.rankmap_len = c->res_lv;
    // the row list only says which frames a match may name; the DP's ranks are the frames that do have one (da_match_finish)
    if (n > 0) {
      if (!sl.dense.p || sl.dense_len != c->res_lv)     // sl.n_ranks counts exactly these frames: without the table the ranks would not fit the trees
        return fail(c, DA_ERR_STATE, "da_chain: the resident match list has no rank table (internal error)");
      L.dense = sl.dense.as<int32_t>() + (sl.dense_len + 1);
    }
  }
  HIP_TRY(c, hipMemsetAsync(sl.small.p, 0, 128, c->stream));
  dbg.at("rank map");
  L.wide = sl.mode == 4;
  da::ChainColumns K{};
  if (sl.mode == 0 && n > 0) {
    const int64_t rows_bound = std::min<int64_t>(n, sl.rows_hint > 0 ? sl.rows_hint : n);
    const int64_t br = da::chain_columns_batch_rows();
    K.msg_stride = (rows_bound + br - 1) / br * br;
    // the hand-over records (24 bytes x rows x columns) may take half of what is free on the device (counting a buffer this
    // context already holds): an 8 h pair at 2 048 columns wants 61 GB, which a device shared with other contexts may not have --
    // fewer columns are slower, never wrong
    int64_t max_cols = 0;
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        size_t held = sl.msg.cap;
        for (const HandoverBuf& h : c->handover_free) held = std::max(held, h.buf.cap);
        const double per_col = 24.0 * (double)K.msg_stride * 1.125 + 4096.0;
        max_cols = std::max<int64_t>(1, (int64_t)(0.5 * (double)(free_b + held) / per_col));
      }
    }
    const size_t ctb = da::chain_columns_temp_bytes(n, sl.n_ranks);
    HIP_TRY(c, sl.rowid.ensure(sizeof(int32_t) * nn));
    HIP_TRY(c, sl.ckey.ensure(sizeof(uint16_t) * 2 * nn)); HIP_TRY(c, sl.cval.ensure(sizeof(uint32_t) * 2 * nn));
    HIP_TRY(c, sl.c_row.ensure(sizeof(uint32_t) * nn)); HIP_TRY(c, sl.c_lr.ensure(sizeof(uint16_t) * nn));
    HIP_TRY(c, sl.c_q.ensure(sizeof(double) * nn)); HIP_TRY(c, sl.c_gid.ensure(sizeof(uint32_t) * nn));
    HIP_TRY(c, sl.rank_cum.ensure(sizeof(int32_t) * 2 * ((size_t)sl.n_ranks + 2)));
    // the hand-over buffer last, and with fewer columns if the device does not have the memory after all
    for (;;) {
      const da::ChainColumnPlan plan = da::chain_columns_plan(n, sl.n_ranks, max_cols);
      K.n_cols = plan.n_cols; K.width = plan.width;
      // granule tags must never match what an earlier launch left behind: fresh memory and every 4095th
      // launch are zeroed, in between the salt distinguishes the launches
      const size_t need = 24 * (size_t)K.msg_stride * (size_t)K.n_cols;
      if (sl.msg.cap < need) {
        give_back_handover(c, sl);
        size_t best = c->handover_free.size();
        for (size_t k = 0; k < c->handover_free.size(); ++k)
          if (c->handover_free[k].buf.cap >= need && (best == c->handover_free.size() || c->handover_free[k].buf.cap < c->handover_free[best].buf.cap)) best = k;
        if (best != c->handover_free.size()) {
          sl.msg = c->handover_free[best].buf; sl.launches = c->handover_free[best].launches;
          c->handover_free.erase(c->handover_free.begin() + (long)best);
        } else if (!c->handover_free.empty()) {            // none fits: the largest one is regrown rather than kept beside a new one
          size_t big = 0;
          for (size_t k = 1; k < c->handover_free.size(); ++k) if (c->handover_free[k].buf.cap > c->handover_free[big].buf.cap) big = k;
          c->handover_free[big].buf.release();
          c->handover_free.erase(c->handover_free.begin() + (long)big);
        }
      }
      const size_t cap_before = sl.msg.cap;
      hipError_t e = hipSuccess;
#ifdef DA_TEST_HOOKS   // libdalign_dbg.so only (make dbg): pretend the device has no more than this many bytes for the buffer
      if (const char* lim = std::getenv("DALIGN_CHAIN_HANDOVER_LIMIT")) {
        if (need > (size_t)std::strtoull(lim, nullptr, 10)) e = hipErrorOutOfMemory;
      }
#endif
      if (e == hipSuccess) e = sl.msg.ensure(need);
      if (e == hipErrorOutOfMemory && K.n_cols > da::chain_columns_plan(n, sl.n_ranks, 1).n_cols) {
        (void)hipGetLastError();                           // the estimate above was too generous: half the columns
        max_cols = std::max<int64_t>(1, K.n_cols / 2);
        continue;
      }
      HIP_TRY(c, e);
      if (sl.msg.cap != cap_before || sl.launches >= 4095) {
        HIP_TRY(c, hipMemsetAsync(sl.msg.p, 0, sl.msg.cap, c->stream));
        sl.launches = 0;
      }
      K.salt = ++sl.launches;
      break;
    }
    HIP_TRY(c, sl.col_start.ensure(sizeof(int32_t) * 2 * ((size_t)K.n_cols + 1)));      // + the columns' first ranks
    HIP_TRY(c, sl.ctl.ensure(sizeof(uint32_t) * ((size_t)da::kChainCtlHead + (size_t)K.n_cols + 2) + 8 * (size_t)da::kChainStampWords * (size_t)K.n_cols));   // + diagnostic stamps
    HIP_TRY(c, sl.temp.ensure(std::max(tb, ctb) + 256));
    K.rowid1 = sl.rowid.as<int32_t>();
    K.key_in = sl.ckey.as<uint16_t>(); K.key_out = sl.ckey.as<uint16_t>() + nn;
    K.val_in = sl.cval.as<uint32_t>(); K.val_out = sl.cval.as<uint32_t>() + nn;
    K.c_row = sl.c_row.as<uint32_t>(); K.c_lr = sl.c_lr.as<uint16_t>(); K.c_q = sl.c_q.as<double>(); K.c_gid = sl.c_gid.as<uint32_t>();
    K.col_rank0 = sl.col_start.as<int32_t>() + (K.n_cols + 1); K.rank_cum = sl.rank_cum.as<int32_t>();
    K.col_start = sl.col_start.as<int32_t>(); K.msg = sl.msg.as<unsigned long long>(); K.ctl = sl.ctl.as<uint32_t>();
    K.temp = sl.temp.p; K.temp_bytes = ctb;
    L.temp = sl.temp.p;
    c->st.chain_columns = (double)K.n_cols; c->st.chain_column_width = (double)K.width;
    dbg.at("ensure (columns) + msg");
  } else {
    if (sl.mode != 0) HIP_TRY(c, hipMemsetAsync(sl.tree.p, 0, 16 * ((size_t)sl.n_ranks + 2), c->stream));
    c->st.chain_columns = 0; c->st.chain_column_width = 0;
  }
  if (da::launch_chain_prep(L, c->stream, sl.mode == 0) != 0) return fail(c, DA_ERR_ARG, "da_chain: %lld matches / %lld video rows exceed the kernel's range", (long long)n, (long long)sl.n_ranks);
  HIP_TRY(c, hipGetLastError());
  dbg.at("prep launch");
  HIP_TRY(c, hipEventRecord(sl.ready, c->stream));
  HIP_TRY(c, hipStreamWaitEvent(st, sl.ready, 0));
  HIP_TRY(c, hipEventRecord(sl.e0, st));
  dbg.at("events");
  if (sl.mode == 0) {
    if (const int lrc = da::launch_chain_columns(L, K, st))
      return fail(c, DA_ERR_DEVICE, "da_chain: launch failed (step %d: %lld matches, %lld ranks, %d columns of at most %d ranks; %s)", -lrc, (long long)n,
                  (long long)sl.n_ranks, K.n_cols, K.width, hipGetErrorString(hipGetLastError()));
  } else if (da::launch_chain_dp(L, st) != 0) return fail(c, DA_ERR_DEVICE, "da_chain: launch failed");
  HIP_TRY(c, hipGetLastError());
  dbg.at("DP launches");
  HIP_TRY(c, hipEventRecord(sl.e1, st));
  HIP_TRY(c, hipMemcpyAsync(sl.h_small, sl.small.p, 32, hipMemcpyDeviceToHost, st));
  sl.h_small[4] = 0;
  if (sl.mode == 0 && n > 0) HIP_TRY(c, hipMemcpyAsync(sl.h_small + 4, sl.ctl.p, 8, hipMemcpyDeviceToHost, st));   // [1] = abort flag
  dbg.at("result copies");
  sl.state = 2;
  sl.ticket = c->next_ticket++;
  return DA_OK;
}

// wait for a slot's DP and hand the path out; frees the slot unless the caller's buffers were too small
int chain_collect(da_ctx* c, ChainSlot& sl, double min_len, int32_t* path_i, int32_t* path_v, int64_t* n_path) {
  HIP_TRY(c, da::stream_wait(sl.run));
  float ms = 0.f; (void)hipEventElapsedTime(&ms, sl.e0, sl.e1); c->st.chain_ms = ms;
  give_back_handover(c, sl);
  if (std::getenv("DALIGN_DEBUG_STAMPS") && sl.mode == 0 && sl.n > 0) {   // diagnostic builds (-DDA_CHAIN_STAMPS) only
    const int nc = (int)c->st.chain_columns;
    constexpr int W = da::kChainStampWords;
    std::vector<unsigned long long> st((size_t)nc * W);
    const uint32_t* base = sl.ctl.as<uint32_t>() + da::kChainCtlHead + ((nc + 1) & ~1);
    if (hipMemcpy(st.data(), base, st.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < nc; ++k) for (int j = 0; j < 8; ++j) sum[j] += (double)st[(size_t)k * W + j];
      const char* path = std::getenv("DALIGN_DEBUG_STAMPS");
      if (path[0] == '/' || path[0] == '.') {             // a file name: the timeline, one line per column (us since the first column's start)
        if (FILE* f = std::fopen(path, "w")) {
          const unsigned long long t0 = st[8 + 31];
          for (int k = 0; k < nc; ++k) {
            std::fprintf(f, "%d %.2f", k, (double)(st[(size_t)k * W + 8 + 31] - t0) * 1e-2);
            for (int j = 0; j < 31; ++j) std::fprintf(f, " %.2f", st[(size_t)k * W + 8 + j] ? (double)(st[(size_t)k * W + 8 + j] - t0) * 1e-2 : -1.0);
            for (int j = 0; j < 8; ++j) std::fprintf(f, " %.2f", (double)st[(size_t)k * W + j] * 1e-2);
            std::fprintf(f, "\n");
          }
          std::fclose(f);
        }
      }
      const char* names[7] = {"publish-drain", "input-wait", "window-pre+query", "sweep", "finals+update", "window-scan", "batch-end"};
      std::fprintf(stderr, "chain column stamps (mean per column, ms; %d columns, %.0f windows per column, kernel+rest %.2f ms):", nc, sum[7] / nc, ms);
      for (int j = 0; j < 7; ++j) std::fprintf(stderr, " %s %.2f", names[j], sum[j] / nc * 1e-5);
      std::fprintf(stderr, "\n");
    }
  } else if (std::getenv("DALIGN_DEBUG_STAMPS")) {        // one-workgroup kernels
    long long st[6];
    if (hipMemcpy(st, sl.small.as<int64_t>() + 3, sizeof st, hipMemcpyDeviceToHost) == hipSuccess) {
      double tot = 0; for (long long v : st) tot += (double)v;
      if (tot > 0) std::fprintf(stderr, "chain stamps: setup %.1f%% issue-loads %.1f%% wait+reduce %.1f%% jacobi %.1f%% update %.1f%% (total %.0f ticks, %.2f ms)\n",
                                100 * st[0] / tot, 100 * st[1] / tot, 100 * st[2] / tot, 100 * st[3] / tot, 100 * st[4] / tot, tot, ms);
    }
  }
  const int err = (int)((unsigned long long)sl.h_small[0] >> 32);
  const int64_t L = sl.n > 0 ? (int64_t)sl.h_small[2] : 0;
  if (sl.mode == 0 && ((unsigned long long)sl.h_small[4] >> 32) != 0) {
    sl.state = 0;
    return fail(c, DA_ERR_DEVICE, "da_chain: a column of the chain DP waited more than 20 s for its neighbour (device oversubscribed?)");
  }
  if (err) {
    sl.state = 0;
    return fail(c, DA_ERR_ARG, err & 1 ? "da_chain: qualities must be finite and positive (describealign.py:672 yields (0, 50])"
                                       : (err & 2 ? "da_chain: input not sorted by (i, v)" : "da_chain: a match names a video frame outside the matched rows"));
  }
  const int64_t capacity = *n_path;
  *n_path = L;
  if ((double)L < min_len) { sl.state = 0; return fail(c, DA_ERR_MISMATCH, "Alignment failed, are the input files mismatched?"); }
  if (L > capacity) return fail(c, DA_ERR_CAPACITY, "da_chain: path of %lld exceeds capacity %lld", (long long)L, (long long)capacity);
  if (L > 0) {
    if (!path_i || !path_v) return fail(c, DA_ERR_ARG, "da_chain: null output");
    const size_t nn = (size_t)std::max<int64_t>(1, sl.n);
    // through page-locked staging (PinArena): the caller's arrays are ordinary memory
    c->pin.reset();
    int32_t* hi_ = static_cast<int32_t*>(c->pin.take(sizeof(int32_t) * L));
    int32_t* hv_ = static_cast<int32_t*>(c->pin.take(sizeof(int32_t) * L));
    if (!hi_ || !hv_) return fail(c, DA_ERR_DEVICE, "da_chain: no page-locked staging memory");
    HIP_TRY(c, hipMemcpyAsync(hi_, sl.out_iv.as<int32_t>(), sizeof(int32_t) * L, hipMemcpyDeviceToHost, sl.run));
    HIP_TRY(c, hipMemcpyAsync(hv_, sl.out_iv.as<int32_t>() + nn, sizeof(int32_t) * L, hipMemcpyDeviceToHost, sl.run));
    HIP_TRY(c, da::stream_wait(sl.run));
    std::memcpy(path_i, hi_, sizeof(int32_t) * L); std::memcpy(path_v, hv_, sizeof(int32_t) * L);
  }
  sl.state = 0;
  return DA_OK;
}

}  // namespace

extern "C" int da_chain(da_ctx* c, const int32_t* pi, const int32_t* pv, const double* pq, int64_t n, double min_len,
                        int32_t* path_i, int32_t* path_v, int64_t* n_path) {
  if (n < 0 || !n_path || (n > 0 && (!pi || !pv || !pq))) return fail(c, DA_ERR_ARG, "da_chain: bad argument");
  if (!c) return chain_host(nullptr, pi, pv, pq, n, min_len, path_i, path_v, n_path);
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t vmax = -1;
  if (int rc = check_sorted(c, pi, pv, n, vmax)) return rc;
  const int si = acquire_slot(c);
  if (si < 0) return fail(c, DA_ERR_STATE, "da_chain: all chain slots are in flight");
  ChainSlot& sl = *c->slots[si];
  // dense 1-based ranks of the video frames that occur
  std::vector<int32_t> rk((size_t)vmax + 2, 0);
  for (int64_t k = 0; k < n; ++k) rk[pv[k]] = 1;
  int32_t nr = 0;
  for (size_t x = 0; x < rk.size(); ++x) if (rk[x]) rk[x] = ++nr;
  std::vector<unsigned long long> keys((size_t)n);
  std::vector<int32_t> ranks((size_t)n);
  for (int64_t k = 0; k < n; ++k) { keys[k] = ((unsigned long long)(uint32_t)pi[k] << 32) | (uint32_t)pv[k]; ranks[k] = rk[pv[k]]; }
  const size_t nn = (size_t)std::max<int64_t>(1, n);
  HIP_TRY(c, sl.keys.ensure(sizeof(unsigned long long) * nn)); HIP_TRY(c, sl.q.ensure(sizeof(double) * nn));
  HIP_TRY(c, sl.rank.ensure(sizeof(int32_t) * nn));
  if (n > 0) {
    HIP_TRY(c, hipMemcpyAsync(sl.keys.p, keys.data(), sizeof(unsigned long long) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sl.q.p, pq, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sl.rank.p, ranks.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  }
  sl.n = n; sl.n_ranks = nr;
  { int64_t rows = 0; for (int64_t k = 0; k < n; ++k) rows += (k == 0 || pi[k] != pi[k - 1]); sl.rows_hint = rows; }
  int rc = chain_enqueue(c, sl, false, true);
  if (rc) { (void)da::stream_wait(c->stream); sl.state = 0; return rc; }
  HIP_TRY(c, da::stream_wait(c->stream));          // the host staging vectors go out of scope
  rc = chain_collect(c, sl, min_len, path_i, path_v, n_path);
  if (rc == DA_ERR_CAPACITY) sl.state = 0;              // one-shot call: nothing stays resident
  return rc;
}

namespace {
int chain_begin_on(da_ctx* c, uint64_t* ticket, bool wide) {
  if (!c) return DA_ERR_ARG;
  if (!ticket) return fail(c, DA_ERR_ARG, "da_chain_begin: null ticket");
  if (!c->fetch_ready) return fail(c, DA_ERR_STATE, "da_chain_begin: no finished match is resident");
  if (!c->rows_of_resident)
    return fail(c, DA_ERR_STATE, "da_chain_begin: da_match_begin has been called since the resident match finished; its video row list "
                                 "(the ranks of the matches) is gone -- call da_chain_begin before the next da_match_begin");
  HIP_TRY(c, hipSetDevice(c->device));
  int si = c->res_slot;
  if (si < 0) {                                         // the match produced nothing: an empty DP
    si = acquire_slot(c);
    if (si < 0) return fail(c, DA_ERR_STATE, "da_chain_begin: all chain slots are in flight");
    c->slots[si]->n = 0; c->slots[si]->n_ranks = 0; c->slots[si]->rows_hint = 0;
  }
  ChainSlot& sl = *c->slots[si];
  if (sl.state == 2) return fail(c, DA_ERR_STATE, "da_chain_begin: the chain DP of this match is already in flight");
  const int rc = chain_enqueue(c, sl, true, wide);
  if (rc) return rc;
  c->res_slot = si;                                     // still the resident match (da_match_fetch keeps working)
  *ticket = sl.ticket;
  return DA_OK;
}
}  // namespace

extern "C" int da_chain_begin(da_ctx* c, uint64_t* ticket) { return chain_begin_on(c, ticket, false); }
extern "C" int da_chain_begin_exclusive(da_ctx* c, uint64_t* ticket) { return chain_begin_on(c, ticket, true); }

// The whole device stage of one pair of a batch in ONE call: features of both resident PCM sides, preparation, similarity
// GEMM, verification + sort, and the chain DP enqueued on its own stream.  What it saves over the five calls it replaces is
// not device work but the caller between them: a Python thread that feeds the GPU from inside a busy pipeline re-acquires
// the interpreter lock after every call, and the device waited 8 of every 51 ms for it (profiles/r05_trace_cfg1_gaps.json).
extern "C" int da_pair_stage(da_ctx* c, float* v_rows, int64_t v_stride, float* a_rows, int64_t a_stride, int mode,
                             int64_t v_lengths[2], int64_t a_lengths[2], int64_t* n_matches, uint64_t* ticket) {
  if (!c) return DA_ERR_ARG;
  if (!v_lengths || !a_lengths || !n_matches || !ticket) return fail(c, DA_ERR_ARG, "da_pair_stage: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  DbgTimes dbg("pair_stage");
  // each side's feature kernel in its own event bracket (the row downloads lie outside them)
  if (int rc = features_enqueue(c, DA_SIDE_VIDEO, v_rows, v_stride, v_lengths, c->feat_e0, c->feat_e1)) return rc;
  if (int rc = features_enqueue(c, DA_SIDE_AUDIO, a_rows, a_stride, a_lengths, c->feat_e2, c->feat_e3)) return rc;
  // the rows stay on the device: da_match_begin's host pointers are only read when the rows are NOT resident
  if (int rc = da_match_begin(c, v_rows ? v_rows : reinterpret_cast<float*>(c->h_pin), v_stride > 0 ? v_stride : v_lengths[0], v_lengths,
                              a_rows ? a_rows : reinterpret_cast<float*>(c->h_pin), a_stride > 0 ? a_stride : a_lengths[0], a_lengths,
                              mode | DA_MATCH_RESIDENT_ROWS, 0, -1)) return rc;
  dbg.at("features enqueued + match_begin (sync on the row counts, GEMM launched)");
  // da_match_begin has synchronised the stream (row counts): features and downloads are complete
  { float m0 = 0.f, m1 = 0.f; (void)hipEventElapsedTime(&m0, c->feat_e0, c->feat_e1); (void)hipEventElapsedTime(&m1, c->feat_e2, c->feat_e3); c->st.features_ms = m0 + m1; }
  features_landed(c, DA_SIDE_VIDEO); features_landed(c, DA_SIDE_AUDIO);
  c->st.features_bytes = 2.0 * c->side[0].channels * (double)c->side[0].n + 5.0 * 4.0 * (double)v_lengths[1] +
                         2.0 * c->side[1].channels * (double)c->side[1].n + 5.0 * 4.0 * (double)a_lengths[1];
  if (int rc = da_match_finish(c, n_matches)) return rc;
  dbg.at("match_finish");
  if (dbg.on) std::fprintf(stderr, "[pair_stage] kernels: features %.3f prep %.3f gemm %.3f verify %.3f (k_verify %.3f) ms; result slot %d\n",
                           c->st.features_ms, c->st.prep_ms, c->st.gemm_ms, c->st.verify_ms, c->st.verify_kernel_ms, c->res_slot);
  const int rc = da_chain_begin(c, ticket);
  dbg.at("chain_begin");
  return rc;
}

extern "C" int da_chain_finish(da_ctx* c, uint64_t ticket, double min_len, int32_t* path_i, int32_t* path_v, int64_t* n_path) {
  if (!c) return DA_ERR_ARG;
  if (!n_path) return fail(c, DA_ERR_ARG, "da_chain_finish: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  for (size_t k = 0; k < c->slots.size(); ++k) {
    ChainSlot& sl = *c->slots[k];
    if (sl.state == 2 && sl.ticket == ticket) {
      const int rc = chain_collect(c, sl, min_len, path_i, path_v, n_path);
      if (sl.state == 0 && c->res_slot == (int)k) { c->res_slot = -1; c->fetch_ready = false; }
      return rc;
    }
  }
  return fail(c, DA_ERR_STATE, "da_chain_finish: unknown ticket %llu", (unsigned long long)ticket);
}

extern "C" int da_chain_poll(da_ctx* c, uint64_t ticket) {
  if (!c) return DA_ERR_ARG;
  for (ChainSlot* sl : c->slots)
    if (sl->state == 2 && sl->ticket == ticket) {
      const hipError_t e = hipStreamQuery(sl->run);
      if (e == hipSuccess) return 1;
      if (e == hipErrorNotReady) return 0;
      return fail(c, DA_ERR_DEVICE, "da_chain_poll: %s", hipGetErrorString(e));
    }
  return fail(c, DA_ERR_STATE, "da_chain_poll: unknown ticket %llu", (unsigned long long)ticket);
}

extern "C" int da_chain_masked(const da_ctx* c) {
  (void)c;
  const int seen = g_chain_masked.load();
  if (seen >= 0) return seen;
  uint32_t mask[8];
  return chain_cu_mask(mask) ? 1 : 0;          // no chain stream yet: what the environment asks for
}

extern "C" int da_chain_resident(da_ctx* c, double min_len, int32_t* path_i, int32_t* path_v, int64_t* n_path) {
  uint64_t t = 0;
  int rc = chain_begin_on(c, &t, true);                 // blocking call: the DP has the chip to itself
  if (rc) return rc;
  return da_chain_finish(c, t, min_len, path_i, path_v, n_path);
}

// ------------------------------------------------------------------------------------- refine
namespace {

void x_limits(double x_first, double x_last, double offset, double slope, int64_t La, int64_t Lv, int64_t extend,
              int64_t& lo, int64_t& hi) {                       // describealign.py:895-900
  const int64_t margin = 4;
  lo = std::max<int64_t>((int64_t)x_first - extend, 0);
  hi = std::min<int64_t>((int64_t)x_last + extend, La - 1);
  lo = std::max<int64_t>(lo, (int64_t)std::ceil(((double)margin - offset) / slope));
  hi = std::min<int64_t>(hi, (int64_t)std::floor(((double)(Lv - margin) - offset) / slope));
}

// Second DP (describealign.py:946-990) over points sorted by (i, j, cluster, qual).
// Returns rows (j, i, cluster, qual, cum-as-used-by-the-successor) in S.out.
void second_dp(RefineScratch& S, int64_t La, int64_t Lv, int n_clusters) {
  using Entry = DpEntry;
  const std::vector<BandPoint>& pts = S.pts;
  const std::vector<int64_t>& row_start = S.row_start;
  std::vector<double>& path_rows = S.out;
  if (++S.gen == 0u) { S.cache.clear(); S.gen = 1u; }           // generation wrap: forget every slot
  const uint32_t gen = S.gen;
  std::vector<Entry>& frontier = S.frontier;                    // sorted by j, cum strictly increasing
  frontier.clear();
  frontier.push_back(Entry{0.0, 0, -1, 0.0, 0.0, -1, gen});
  std::vector<Entry>& cl_best = S.cl_best;
  cl_best.assign((size_t)n_clusters, Entry{0.0, 0, 0, 0.0, -1000.0, -1, gen});
  std::vector<Entry>& cache = S.cache;                          // one slot per video frame; a slot of another generation is empty
  if (cache.size() < (size_t)Lv) cache.resize((size_t)Lv, Entry{0.0, 0, 0, 0.0, 0.0, -2, 0u});
  cache[0] = Entry{0.0, 0, -1, 0.0, 0.0, -1, gen};
  const size_t np = pts.size();
  std::vector<int32_t>& pred = S.pred; pred.assign(np, -1);
  std::vector<double>& pred_cum = S.pred_cum; pred_cum.assign(np, 0.0);
  // forward_min[i]: smallest j among rows >= i
  std::vector<double>& fmin = S.fmin;
  fmin.assign((size_t)La + 1, std::numeric_limits<double>::infinity());
  for (int64_t i = La - 1; i >= 0; --i) {
    double m = fmin[i + 1];
    if (row_start[i + 1] > row_start[i]) m = std::min(m, pts[row_start[i]].j);
    fmin[i] = m;
  }
  for (int64_t i = 0; i < La; ++i) {
    for (int64_t p = row_start[i]; p < row_start[i + 1]; ++p) {
      const BandPoint& pt = pts[p];
      const double j = pt.j;
      // bisect_right on the key j
      size_t pos = std::upper_bound(frontier.begin(), frontier.end(), j,
                                    [](double key, const Entry& e) { return key < e.j; }) - frontier.begin();
      const Entry& fe = frontier[pos - 1];
      int32_t pid = fe.id; double best = fe.cum;
      const Entry last = cl_best[pt.cl];
      if (last.cum >= best) { pid = last.id; best = last.cum; }
      const int64_t jj = (int64_t)j;
      for (int64_t t = std::max<int64_t>(0, jj - 2); t <= jj; ++t) {
        const Entry& nd = cache[t];
        if (nd.gen != gen) continue;                 // -inf slot: every comparison is false
        double cum = nd.cum;
        if (pt.cl != nd.cl) {
          const double skew = (j - nd.j) - ((double)i - (double)nd.i);
          cum -= 100.0 + 100.0 * skew * skew;
        }
        if ((double)nd.i >= (double)(i - 2) && nd.j <= j && cum >= best) { pid = nd.id; best = cum; }
      }
      const double cum = best + pt.q;
      cache[jj] = Entry{j, (int32_t)i, pt.cl, pt.q, cum, (int32_t)p, gen};
      const double cjump = cum - 1000.0;
      if (frontier[pos - 1].cum < cjump) {
        size_t end = pos;
        while (end < frontier.size() && frontier[end].cum <= cjump) ++end;
        frontier.erase(frontier.begin() + pos, frontier.begin() + end);
        frontier.insert(frontier.begin() + pos, Entry{j, (int32_t)i, pt.cl, pt.q, cjump, (int32_t)p, gen});
      }
      if (fmin[i] == j && pos > 1) frontier.erase(frontier.begin(), frontier.begin() + (pos - 1));
      const double ccl = cum - 50.0;
      if (last.cum < ccl) cl_best[pt.cl] = Entry{j, (int32_t)i, pt.cl, pt.q, ccl, (int32_t)p, gen};
      pred[p] = pid; pred_cum[p] = best;
    }
  }
  // backtrack from the last frontier entry (:985-989)
  path_rows.clear();
  std::vector<std::pair<int32_t, double>>& rev = S.rev;         // (point id, cum as recorded)
  rev.clear();
  const Entry& lastf = frontier.back();
  if (lastf.id >= 0) {
    rev.emplace_back(lastf.id, lastf.cum);
    while (true) {
      const int32_t p = rev.back().first;
      const int32_t q = pred[p];
      if (q < 0) break;
      rev.emplace_back(q, pred_cum[p]);
    }
  }
  path_rows.reserve(rev.size() * 5);
  for (auto it = rev.rbegin(); it != rev.rend(); ++it) {
    const BandPoint& pt = pts[it->first];
    path_rows.push_back(pt.j); path_rows.push_back((double)pt.i); path_rows.push_back((double)pt.cl);
    path_rows.push_back(pt.q); path_rows.push_back(it->second);
  }
}

}  // namespace

extern "C" int da_refine(da_ctx* c, const double* a_scaled, int64_t La, const double* v_scaled, int64_t Lv,
                         const double* cl_x0, const double* cl_x1, const double* cl_offset, const double* cl_slope,
                         int n_clusters, double min_len, double* path, int64_t* n_rows, int64_t* n_points) {
  if (!c) return DA_ERR_ARG;
  if (!a_scaled || !v_scaled || La < 2 || Lv < 2 || n_clusters < 0 || !n_rows ||
      (n_clusters > 0 && (!cl_x0 || !cl_x1 || !cl_offset || !cl_slope)))
    return fail(c, DA_ERR_ARG, "da_refine: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->ascaled.ensure(sizeof(double) * 3 * La)); HIP_TRY(c, c->vscaled.ensure(sizeof(double) * 3 * Lv));
  // every copy of this call goes through page-locked staging (see PinArena)
  c->pin.reset();
  auto staged = [&](const void* src, size_t bytes) -> void* {
    void* p = c->pin.take(bytes);
    if (p && src) std::memcpy(p, src, bytes);
    return p;
  };
  {
    void* sa = staged(a_scaled, sizeof(double) * 3 * La);
    void* sv = staged(v_scaled, sizeof(double) * 3 * Lv);
    if (!sa || !sv) return fail(c, DA_ERR_DEVICE, "da_refine: no page-locked staging memory");
    HIP_TRY(c, hipMemcpyAsync(c->ascaled.p, sa, sizeof(double) * 3 * La, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->vscaled.p, sv, sizeof(double) * 3 * Lv, hipMemcpyHostToDevice, c->stream));
  }
  double a_max = -1e300, v_max = -1e300;                                  // (:908-909)
  for (int64_t i = 0; i < La; ++i) a_max = std::max(a_max, a_scaled[3 * i]);
  for (int64_t i = 0; i < Lv; ++i) v_max = std::max(v_max, v_scaled[3 * i]);
  BandArgs base{};
  base.a_scaled = c->ascaled.as<double>(); base.La = La; base.v_scaled = c->vscaled.as<double>(); base.Lv = Lv;
  base.a_max = a_max; base.v_max = v_max;
  const int64_t extend = (int64_t)kFrameRate * 30;
  const int kRefBlocks = 64;

  // ---- every cluster's line, one launch per step (no per-cluster round trips)
  struct Job { int ci; double offset, slope, xf, xl; int64_t lo, hi; bool refine; };
  std::vector<Job> jobs;
  for (int ci = 0; ci < n_clusters; ++ci) {
    Job j{ci, cl_offset[ci], cl_slope[ci], cl_x0[ci], cl_x1[ci], 0, 0, false};
    x_limits(j.xf, j.xl, j.offset, j.slope, La, Lv, 0, j.lo, j.hi);
    if (j.hi < j.lo + 5) continue;                                        // (:914-915)
    j.refine = j.hi > j.lo + 100;                                         // (:916)
    jobs.push_back(j);
  }
  const int nj = (int)jobs.size();
  std::vector<BandCluster> tab((size_t)std::max(1, nj));
  HIP_TRY(c, c->band_tab.ensure(sizeof(BandCluster) * tab.size()));
  float ms = 0.f; double kernel_ms = 0.0;
  if (nj > 0) {
    for (int k = 0; k < nj; ++k) tab[k] = BandCluster{jobs[k].offset, jobs[k].slope, jobs[k].lo, jobs[k].hi, 0, jobs[k].refine ? 1 : 0, 0};
    void* stab = staged(tab.data(), sizeof(BandCluster) * nj);
    const size_t n_part = (size_t)4 * kRefBlocks * nj;
    double* part = static_cast<double*>(staged(nullptr, sizeof(double) * n_part));
    if (!stab || !part) return fail(c, DA_ERR_DEVICE, "da_refine: no page-locked staging memory");
    HIP_TRY(c, hipMemcpyAsync(c->band_tab.p, stab, sizeof(BandCluster) * nj, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, c->band_part.ensure(sizeof(double) * 4 * kRefBlocks * nj));
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    launch_band_refine_all(base, c->band_tab.as<BandCluster>(), nj, c->band_part.as<double>(), kRefBlocks, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipMemcpyAsync(part, c->band_part.p, sizeof(double) * n_part, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, da::stream_wait(c->stream));
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1); kernel_ms += ms;
    for (int k = 0; k < nj; ++k) {
      Job& j = jobs[k];
      if (j.refine) {                                                     // (:916-930)
        double cnt = 0, sde = 0, sdd = 0, see = 0;
        for (int b = 0; b < kRefBlocks; ++b) {
          const double* p4 = &part[4 * ((size_t)k * kRefBlocks + b)];
          cnt += p4[0]; sde += p4[1]; sdd += p4[2]; see += p4[3];
        }
        if (cnt > 50 && sdd > 0 && see > 0) {
          const double sol = sde / sdd;
          const double resid = see - sol * sde;
          const double explained = 1.0 - resid / see;
          const double z = std::sqrt(explained * 3.0 * cnt) - 1.0;
          if (z > 8 && std::fabs(sol) < 2) j.offset += sol;
        }
        j.xf = (double)j.lo; j.xl = (double)(j.hi - 1);                   // the reference rebinds x to arange(lo, hi) here (:917)
      }
      x_limits(j.xf, j.xl, j.offset, j.slope, La, Lv, extend, j.lo, j.hi);
    }
  }
  int64_t n_all = 0;
  int nk = 0;
  for (int k = 0; k < nj; ++k) {
    const Job& j = jobs[k];
    if (j.hi <= j.lo) continue;
    tab[nk] = BandCluster{j.offset, j.slope, j.lo, j.hi, n_all, 0, j.ci};          // pad carries the caller's cluster index
    n_all += j.hi - j.lo; ++nk;
  }
  // ---- quality of every banded point, de-duplication on (audio frame, int(video position)) keeping
  // the first cluster's point (:937-941), in (audio frame, video position) order -- all on the device
  RefineScratch& S = c->refine;
  std::vector<BandPoint>& pts = S.pts;
  pts.clear();
  int64_t total_points = 0;
  if (n_all > 0x7fffffffLL) return fail(c, DA_ERR_ARG, "da_refine: %lld banded points exceed the kernels' range", (long long)n_all);
  if (n_all > 0) {
    const size_t n = (size_t)n_all;
    HIP_TRY(c, c->band_y.ensure(sizeof(double) * n)); HIP_TRY(c, c->band_q.ensure(sizeof(double) * n));
    HIP_TRY(c, c->band_cl.ensure(sizeof(int32_t) * n)); HIP_TRY(c, c->band_keys.ensure(sizeof(unsigned long long) * 2 * n));
    HIP_TRY(c, c->band_ids.ensure(sizeof(int32_t) * 3 * n + 64)); HIP_TRY(c, c->band_head.ensure(n));
    HIP_TRY(c, c->band_out.ensure((sizeof(double) * 2 + sizeof(int32_t) * 2) * n));
    unsigned long long* keys = c->band_keys.as<unsigned long long>(); unsigned long long* keys_s = keys + n;
    int32_t* ids = c->band_ids.as<int32_t>(); int32_t* ids_s = ids + n; int32_t* kept = ids + 2 * n;
    int32_t* d_cnt = c->counters.as<int32_t>() + 12;                      // bytes 48..51 of `counters`
    size_t t1 = 0, t2 = 0;
    if (sort_keys_ids(nullptr, nullptr, nullptr, nullptr, n_all, nullptr, &t1, c->stream) != 0 ||
        select_flagged_ids(nullptr, nullptr, nullptr, nullptr, n_all, nullptr, &t2, c->stream) != 0)
      return fail(c, DA_ERR_DEVICE, "da_refine: scratch sizing failed");
    size_t tb = std::max(t1, t2);
    HIP_TRY(c, c->band_tmp.ensure(tb + 256));
    void* stab2 = staged(tab.data(), sizeof(BandCluster) * std::max(1, nk));
    if (!stab2) return fail(c, DA_ERR_DEVICE, "da_refine: no page-locked staging memory");
    HIP_TRY(c, hipMemcpyAsync(c->band_tab.p, stab2, sizeof(BandCluster) * nk, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    launch_band_quality_all(base, c->band_tab.as<BandCluster>(), nk, n_all, c->band_y.as<double>(), c->band_q.as<double>(),
                            c->band_cl.as<int32_t>(), keys, ids, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (sort_keys_ids(keys, keys_s, ids, ids_s, n_all, c->band_tmp.p, &tb, c->stream) != 0) return fail(c, DA_ERR_DEVICE, "da_refine: sort failed");
    launch_band_heads(keys_s, n_all, c->band_head.as<uint8_t>(), c->stream);
    tb = std::max(t1, t2);
    if (select_flagged_ids(ids_s, c->band_head.as<uint8_t>(), kept, d_cnt, n_all, c->band_tmp.p, &tb, c->stream) != 0)
      return fail(c, DA_ERR_DEVICE, "da_refine: selection failed");
    double* o_j = c->band_out.as<double>(); double* o_q = o_j + n;
    int32_t* o_i = reinterpret_cast<int32_t*>(o_q + n); int32_t* o_cl = o_i + n;
    launch_band_gather(kept, d_cnt, c->band_y.as<double>(), c->band_q.as<double>(), c->band_cl.as<int32_t>(), keys, o_j, o_q, o_i, o_cl, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    int32_t* h_kept = reinterpret_cast<int32_t*>(c->h_pin + 6);
    HIP_TRY(c, hipMemcpyAsync(h_kept, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, da::stream_wait(c->stream));
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1); kernel_ms += ms;
    const int32_t n_kept = *h_kept;
    total_points = n_kept;
    pts.resize((size_t)std::max(0, n_kept));
    if (n_kept > 0) {
      double* hj = static_cast<double*>(staged(nullptr, sizeof(double) * n_kept));
      double* hq = static_cast<double*>(staged(nullptr, sizeof(double) * n_kept));
      int32_t* hi_ = static_cast<int32_t*>(staged(nullptr, sizeof(int32_t) * n_kept));
      int32_t* hcl = static_cast<int32_t*>(staged(nullptr, sizeof(int32_t) * n_kept));
      if (!hj || !hq || !hi_ || !hcl) return fail(c, DA_ERR_DEVICE, "da_refine: no page-locked staging memory");
      HIP_TRY(c, hipMemcpyAsync(hj, o_j, sizeof(double) * n_kept, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipMemcpyAsync(hq, o_q, sizeof(double) * n_kept, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipMemcpyAsync(hi_, o_i, sizeof(int32_t) * n_kept, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipMemcpyAsync(hcl, o_cl, sizeof(int32_t) * n_kept, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, da::stream_wait(c->stream));
      for (int32_t t = 0; t < n_kept; ++t) pts[t] = BandPoint{hj[t], hi_[t], tab[hcl[t]].pad, hq[t]};
    }
  }
  c->st.refine_kernel_ms = kernel_ms;
  c->st.refine_points = (double)total_points;
  if (n_points) *n_points = total_points;
  const double t0 = now_ms();
  std::vector<int64_t>& row_start = S.row_start;
  row_start.assign((size_t)La + 1, 0);
  {
    size_t p = 0;
    for (int64_t i = 0; i <= La; ++i) {
      while (p < pts.size() && pts[p].i < i) ++p;
      row_start[i] = (int64_t)p;
    }
  }
  second_dp(S, La, Lv, n_clusters);
  std::vector<double>& out = S.out;
  c->st.refine_dp_ms = now_ms() - t0;
  const int64_t rows_out = (int64_t)(out.size() / 5);
  const int64_t capacity = *n_rows;
  *n_rows = rows_out;
  if ((double)rows_out < min_len) return fail(c, DA_ERR_MISMATCH, "Alignment failed, are the input files mismatched?");
  if (rows_out > capacity) return fail(c, DA_ERR_CAPACITY, "da_refine: %lld rows exceed capacity %lld", (long long)rows_out, (long long)capacity);
  if (rows_out && !path) return fail(c, DA_ERR_ARG, "da_refine: null path");
  std::memcpy(path, out.data(), sizeof(double) * out.size());
  return DA_OK;
}


// ------------------------------------------------------------------------ audio replacement
namespace {

void stretch_stats(da_ctx* c, const da::StretchTimes& t) {
  c->st.resample_ms = t.resample_ms; c->st.resample_points = t.resample_points; c->st.resample_bytes = t.resample_bytes;
  c->st.correlate_ms = t.correlate_ms; c->st.correlate_windows = t.correlate_windows;
  c->st.viterbi_ms = t.viterbi_ms; c->st.splice_ms = t.splice_ms; c->st.splice_points = t.splice_points;
}

int check_nodes(da_ctx* c, const double* at, const double* vt, int n_nodes, const char* who) {
  if (!at || !vt || n_nodes < 2) return fail(c, DA_ERR_ARG, "%s: need at least two nodes", who);
  for (int k = 0; k < n_nodes; ++k)
    if (!std::isfinite(at[k]) || !std::isfinite(vt[k]) || std::fabs(at[k]) > 1e9 || std::fabs(vt[k]) > 1e9)
      return fail(c, DA_ERR_ARG, "%s: node %d is not a finite time in seconds", who, k);
  return DA_OK;
}

}  // namespace

extern "C" int da_replace_segments(da_ctx* c, uint16_t* video, int64_t n_video, const uint16_t* audio, int64_t n_audio,
                                   int channels, const double* audio_times, const double* video_times, int n_nodes,
                                   int no_pitch_correction) {
  if (!c) return DA_ERR_ARG;
  if (!video || !audio || n_video <= 0 || n_audio <= 0 || (channels != 1 && channels != 2))
    return fail(c, DA_ERR_ARG, "da_replace_segments: bad argument");
  if (int rc = check_nodes(c, audio_times, video_times, n_nodes, "da_replace_segments")) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->stretch) c->stretch = da::stretch_create();
  const int64_t vs = da::stretch_channel_stride(n_video), as = da::stretch_channel_stride(n_audio);
  HIP_TRY(c, c->st_video.ensure(sizeof(uint16_t) * (size_t)vs * channels));
  HIP_TRY(c, c->st_audio.ensure(sizeof(uint16_t) * (size_t)as * channels));
  HIP_TRY(c, hipMemcpy2DAsync(c->st_video.p, sizeof(uint16_t) * vs, video, sizeof(uint16_t) * n_video, sizeof(uint16_t) * n_video,
                              channels, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpy2DAsync(c->st_audio.p, sizeof(uint16_t) * as, audio, sizeof(uint16_t) * n_audio, sizeof(uint16_t) * n_audio,
                              channels, hipMemcpyHostToDevice, c->stream));
  da::StretchTimes t;
  std::string err;
  const int rc = da::stretch_replace(c->stretch, c->stream, c->st_video.as<uint16_t>(), n_video, c->st_audio.as<uint16_t>(),
                                     n_audio, channels, audio_times, video_times, n_nodes, no_pitch_correction != 0, t, err);
  if (rc) { (void)da::stream_wait(c->stream); return fail(c, rc, "%s", err.c_str()); }
  stretch_stats(c, t);
  HIP_TRY(c, hipMemcpy2DAsync(video, sizeof(uint16_t) * n_video, c->st_video.p, sizeof(uint16_t) * vs, sizeof(uint16_t) * n_video,
                              channels, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, da::stream_wait(c->stream));
  return DA_OK;
}

extern "C" int da_stretch_resident(da_ctx* c, const double* audio_times, const double* video_times, int n_nodes,
                                   int no_pitch_correction, int16_t* out, int64_t out_capacity_frames, double* factors) {
  if (!c) return DA_ERR_ARG;
  if (int rc = check_nodes(c, audio_times, video_times, n_nodes, "da_stretch_resident")) return rc;
  Side& sv = c->side[DA_SIDE_VIDEO];
  Side& sa = c->side[DA_SIDE_AUDIO];
  if (sv.channels == 0 || sa.channels == 0) return fail(c, DA_ERR_STATE, "da_stretch_resident: upload PCM for both sides first");
  if (sv.channels != sa.channels) return fail(c, DA_ERR_ARG, "da_stretch_resident: channel counts differ (%d vs %d)", sv.channels, sa.channels);
  if (!out || out_capacity_frames < sv.n) return fail(c, DA_ERR_CAPACITY, "da_stretch_resident: output needs %lld frames", (long long)sv.n);
  if (sv.n <= 0 || sa.n <= 0) return fail(c, DA_ERR_ARG, "da_stretch_resident: empty PCM");
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->stretch) c->stretch = da::stretch_create();
  const int C = sv.channels;
  HIP_TRY(c, c->st_video.ensure(sizeof(uint16_t) * (size_t)da::stretch_channel_stride(sv.n) * C));
  HIP_TRY(c, c->st_audio.ensure(sizeof(uint16_t) * (size_t)da::stretch_channel_stride(sa.n) * C));
  HIP_TRY(c, c->st_out.ensure(sizeof(int16_t) * (size_t)sv.n * C));
  std::string err;
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  double f[2] = {0, 0};
  int rc = da::stretch_prepare(c->stretch, c->stream, sv.pcm.as<int16_t>(), sv.n, sv.planar, sa.pcm.as<int16_t>(), sa.n,
                               sa.planar, C, c->st_video.as<uint16_t>(), c->st_audio.as<uint16_t>(), f, err);
  if (rc) return fail(c, rc, "%s", err.c_str());
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  if (factors) for (int k = 0; k < C; ++k) factors[k] = f[k];
  da::StretchTimes t;
  rc = da::stretch_replace(c->stretch, c->stream, c->st_video.as<uint16_t>(), sv.n, c->st_audio.as<uint16_t>(), sa.n, C,
                           audio_times, video_times, n_nodes, no_pitch_correction != 0, t, err);
  if (rc) { (void)da::stream_wait(c->stream); return fail(c, rc, "%s", err.c_str()); }
  stretch_stats(c, t);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, c->ev0, c->ev1); c->st.stretch_prepare_ms = ms;
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  rc = da::stretch_finish(c->stretch, c->stream, c->st_video.as<uint16_t>(), sv.n, C, c->st_out.as<int16_t>(), err);
  if (rc) return fail(c, rc, "%s", err.c_str());
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  HIP_TRY(c, hipMemcpyAsync(out, c->st_out.p, sizeof(int16_t) * (size_t)sv.n * C, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, da::stream_wait(c->stream));
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1); c->st.stretch_finish_ms = ms;
  return DA_OK;
}

extern "C" int da_stretch_schedule(da_ctx* c, int k, int64_t* pairs, int64_t* n) {
  if (!c || !n) return DA_ERR_ARG;
  const int count = da::stretch_schedule_count(c->stretch);
  if (k == -1) { *n = count; return DA_OK; }
  const std::vector<int64_t>* s = da::stretch_schedule(c->stretch, k);
  if (!s) return fail(c, DA_ERR_ARG, "da_stretch_schedule: interval %d of %d", k, count);
  const int64_t need = (int64_t)s->size() / 2;
  if (*n < need || (need && !pairs)) { *n = need; return fail(c, DA_ERR_CAPACITY, "da_stretch_schedule: %lld pairs", (long long)need); }
  if (need) std::memcpy(pairs, s->data(), sizeof(int64_t) * s->size());
  *n = need;
  return DA_OK;
}
