// sort.hip -- the one library call of the device path: rocPRIM's radix sort (through hipCUB), used to order the surfel
// pool by position before the surfel pass traces it (kernels.hip, k_surfel_keys / k_surfel_trace). Kept in its own
// translation unit so the traversal kernels do not compile against the rocPRIM templates.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstddef>
#include <cstdint>

namespace dust {

// tmp == nullptr: only reports the temporary storage the sort needs
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, uint32_t key_bits, hipStream_t s) {
  return hipcub::DeviceRadixSort::SortPairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, (int)key_bits, s);
}

}  // namespace dust
