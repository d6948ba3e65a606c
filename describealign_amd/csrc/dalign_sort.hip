// Device radix sort of the verified matches by (audio frame, video frame) -- hipCUB/rocPRIM.
// Kept in its own translation unit: the library templates dominate compile time.
#include "dalign_common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>

namespace da {

// stable LSD radix sort on key bits [begin_bit, end_bit): two calls -- video bits first, audio bits second -- sort (i << 32 | v)
// keys in as many 8-bit passes as the two fields need (2 h pair: 3 + 3) instead of the 8 of a full 64-bit sort
int sort_pairs(unsigned long long* keys_in, unsigned long long* keys_out, double* vals_in, double* vals_out,
               int64_t n, void* temp, size_t* temp_bytes, int begin_bit, int end_bit, hipStream_t s) {
  if (n > 0x7fffffffLL) return -1;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                    (int)n, begin_bit, end_bit, s);
  return e == hipSuccess ? 0 : -1;
}

int sort_keys_ids(const unsigned long long* keys_in, unsigned long long* keys_out, const int32_t* ids_in, int32_t* ids_out, int64_t n,
                  void* temp, size_t* temp_bytes, hipStream_t s) {
  if (n > 0x7fffffffLL) return -1;
  return hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, ids_in, ids_out, (int)n, 0, 64, s) == hipSuccess ? 0 : -1;
}

int select_flagged_ids(const int32_t* ids, const uint8_t* flags, int32_t* out, int32_t* d_count, int64_t n, void* temp, size_t* temp_bytes,
                       hipStream_t s) {
  if (n > 0x7fffffffLL) return -1;
  return hipcub::DeviceSelect::Flagged(temp, *temp_bytes, ids, flags, out, d_count, (int)n, s) == hipSuccess ? 0 : -1;
}

// ---- row lists of the matching stage, built on the device from a resident energy row ----------
// describealign.py:629-630 (every 4th non-quiet video frame) and :657-658 (non-quiet audio frames
// inside the requested row range): frames i in [lo, hi) with energy[i] > 0.5, in order.
namespace {
struct NonQuiet {
  const float* e;
  __device__ bool operator()(const int32_t& i) const { return e[i] > 0.5f; }
};
__global__ void k_every_fourth(const int32_t* __restrict__ all, const int32_t* __restrict__ n_all, int32_t* __restrict__ out,
                               int32_t* __restrict__ n_out) {
  const int32_t n = *n_all;
  const int32_t m = (n + 3) / 4;
  for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) out[k] = all[4 * k];
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = m;
}
}  // namespace

size_t select_rows_temp_bytes(int64_t n) {
  size_t bytes = 0;
  hipcub::CountingInputIterator<int32_t> ids(0);
  (void)hipcub::DeviceSelect::If(nullptr, bytes, ids, (int32_t*)nullptr, (int32_t*)nullptr, (int)std::max<int64_t>(1, n), NonQuiet{nullptr});
  return bytes;
}

int select_rows(const float* energy, int64_t lo, int64_t hi, bool every_fourth, int32_t* scratch, int32_t* out, int32_t* d_count,
                void* temp, size_t temp_bytes, hipStream_t s) {
  const int64_t n = hi - lo;
  if (n <= 0) return hipMemsetAsync(d_count, 0, sizeof(int32_t), s) == hipSuccess ? 0 : -1;
  if (n > 0x7fffffffLL) return -1;
  hipcub::CountingInputIterator<int32_t> ids((int32_t)lo);
  int32_t* first = every_fourth ? scratch : out;
  int32_t* cnt = every_fourth ? d_count + 1 : d_count;
  if (hipcub::DeviceSelect::If(temp, temp_bytes, ids, first, cnt, (int)n, NonQuiet{energy}, s) != hipSuccess) return -1;
  if (every_fourth) hipLaunchKernelGGL(k_every_fourth, dim3(256), dim3(256), 0, s, scratch, cnt, out, d_count);
  return 0;
}

}  // namespace da
