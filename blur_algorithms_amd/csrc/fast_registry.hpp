// fast_registry.hpp -- the FFT lengths that have compile-time specialised kernels.
#pragma once
#include "fast_kernels.hpp"
namespace blur_amd {
const FastEntry* fast_entry_4000();   // 4K  sigma 20 rows  (3840 + 130 -> 4000)
const FastEntry* fast_entry_2304();   // 4K  sigma 20 cols, 1080p sigma 20 rows
const FastEntry* fast_entry_1280();   // 1080p sigma 20 cols
const FastEntry* fast_entry_2560();   // 4K  sigma 50 cols
const FastEntry* fast_entry_4320();   // 4K  sigma 50 rows
inline const FastEntry* find_fast_entry(int n)
{
    switch (n) {
    case 4000: return fast_entry_4000();
    case 2304: return fast_entry_2304();
    case 1280: return fast_entry_1280();
    case 2560: return fast_entry_2560();
    case 4320: return fast_entry_4320();
    default: return nullptr;
    }
}
}  // namespace blur_amd
