// Addressing of the paged KV cache, shared by the attention kernels.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

template <typename T>
__device__ __forceinline__ T* kv_ptr(const KVView& kv, int layer, int row, int pos, int which, int g) {
    // without a table every row owns max_pages consecutive pages (saves a dependent load in front of every K/V access)
    const int pidx = row * kv.max_pages + ((pos / kv.page_tokens) & kv.page_mask);
    const int page = kv.page_table ? kv.page_table[pidx] : pidx;
    const int slot = pos % kv.page_tokens;
    return (T*)kv.base + (size_t)layer * kv.layer_stride +
           ((((size_t)page * 2 + which) * kv.nkv + g) * kv.page_tokens + slot) * kv.d;
}

}  // namespace qtts
