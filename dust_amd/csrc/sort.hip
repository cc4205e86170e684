// sort.hip -- the one library call of the device path: rocPRIM's radix sort (through hipCUB), used to order the surfel
// pool by position before the surfel pass traces it (kernels.hip, k_surfel_keys / k_surfel_trace). Kept in its own
// translation unit so the traversal kernels do not compile against the rocPRIM templates.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstddef>
#include <cstdint>

namespace dust {

// tmp == nullptr: only reports the temporary storage the sort needs.
// The keys are 16 bits wide ON PURPOSE: for keys of up to two bytes rocPRIM sorts this many items with Onesweep (a histogram,
// a scan and one launch per 8-bit digit: 4 launches), for wider keys with a merge sort whose 21 launches cost 0.12 ms whatever
// the number of significant bits -- more than the finer order of a 30-bit key saves in the trace.
hipError_t sort_pairs_u16(void* tmp, size_t* tmp_bytes, const uint16_t* keys_in, uint16_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, hipStream_t s) {
  return hipcub::DeviceRadixSort::SortPairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, 16, s);
}

}  // namespace dust
