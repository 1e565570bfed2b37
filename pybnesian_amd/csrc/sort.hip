// Stable device sort of (key, value) pairs on the low `bits` of 32-bit keys (rocPRIM radix sort); the temporary storage
// lives in a grow-only arena of the caller.  Used by the spatially sorted packs of the pruned KDE sweeps.
#include <rocprim/device/device_radix_sort.hpp>

#include "kde_kernels.hpp"

namespace pbn {

void sort_keys(dev_buf<char>& tmp, const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
               int bits, hipStream_t st) {
    if (n == 0) return;
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits, st));
    tmp.reserve(bytes);
    HIP_CHECK(rocprim::radix_sort_pairs((void*)tmp.p, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, bits, st));
}

}  // namespace pbn
