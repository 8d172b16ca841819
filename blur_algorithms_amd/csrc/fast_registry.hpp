// fast_registry.hpp -- the (FFT length, role) pairs that have compile-time specialised kernels.
#pragma once
#include "fast_kernels.hpp"
namespace blur_amd {
// rows: the row pass of an image whose cols + 2 pad rounds up to N; cols: likewise from rows
const FastEntry* fast_row_entry_4000();   // 4K  sigma 20 (3840 + 130 -> 4000)
const FastEntry* fast_col_entry_2304();   // 4K  sigma 20 (2160 + 130 -> 2304)
const FastEntry* fast_row_entry_2304();   // 1080p sigma 20 rows
const FastEntry* fast_col_entry_1280();   // 1080p sigma 20 cols
const FastEntry* fast_row_entry_4320();   // 4K  sigma 50 rows
const FastEntry* fast_col_entry_2560();   // 4K  sigma 50 cols
inline const FastEntry* find_fast_entry(int n, bool column_role)
{
    if (column_role) {
        switch (n) {
        case 2304: return fast_col_entry_2304();
        case 1280: return fast_col_entry_1280();
        case 2560: return fast_col_entry_2560();
        default: return nullptr;
        }
    }
    switch (n) {
    case 4000: return fast_row_entry_4000();
    case 2304: return fast_row_entry_2304();
    case 4320: return fast_row_entry_4320();
    default: return nullptr;
    }
}
}  // namespace blur_amd
