// wr_registry.hpp -- the (R0, role) pairs that have wave-resident kernels (wr_kernels.hpp); N = 256 * R0.
#pragma once
#include "wr_kernels.hpp"
namespace blur_amd {
const WrEntry* wr_col_entry_9();    // 4K sigma 20 columns: 2160 + 130 -> 2304
const WrEntry* wr_row_entry_16();   // 4K sigma 20 rows:    3840 + 130 -> 4096
// smallest supported transform that holds `need` points (nullptr: none)
inline const WrEntry* find_wr_entry(int need, bool column_role)
{
    const WrEntry* cand[] = { column_role ? wr_col_entry_9() : nullptr, column_role ? nullptr : wr_row_entry_16() };
    const WrEntry* best = nullptr;
    for (const WrEntry* e : cand)
        if (e && e->r0 * kWrS >= need && (!best || e->r0 < best->r0)) best = e;
    return best;
}
}  // namespace blur_amd
