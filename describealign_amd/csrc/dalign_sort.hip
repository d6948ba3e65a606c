// Device radix sort of the verified matches by (audio frame, video frame) -- hipCUB/rocPRIM.
// Kept in its own translation unit: the library templates dominate compile time.
#include "dalign_common.h"
#include <hipcub/hipcub.hpp>

namespace da {

int sort_pairs(unsigned long long* keys_in, unsigned long long* keys_out, double* vals_in, double* vals_out,
               int64_t n, void* temp, size_t* temp_bytes, hipStream_t s) {
  if (n > 0x7fffffffLL) return -1;
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                    (int)n, 0, 64, s);
  return e == hipSuccess ? 0 : -1;
}

}  // namespace da
