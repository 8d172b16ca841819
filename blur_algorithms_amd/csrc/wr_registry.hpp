// wr_registry.hpp -- the (R0, role) pairs that have wave-resident kernels (wr_kernels.hpp); N = 256 * R0.
// One translation unit each (wr_col_<R0>.hip, wr_row_<R0>.hip).  Column role: 4 complex lines (8 columns) per task, so
// N <= 2560 fits LDS beside the byte stage -- R0 = 12, 15, 16 take 2 lines (4 columns: WrEntry::g = 4) and reach 4096; row role:
// the 3 channel lines of a row pair, N <= 4096.
#pragma once
#include "wr_kernels.hpp"
namespace blur_amd {
#define BLUR_WR_DECL_COL(R0_) const WrEntry* wr_col_entry_##R0_();
#define BLUR_WR_DECL_ROW(R0_) const WrEntry* wr_row_entry_##R0_();
BLUR_WR_DECL_COL(3) BLUR_WR_DECL_COL(4) BLUR_WR_DECL_COL(5) BLUR_WR_DECL_COL(6) BLUR_WR_DECL_COL(8) BLUR_WR_DECL_COL(9) BLUR_WR_DECL_COL(10)
BLUR_WR_DECL_COL(12) BLUR_WR_DECL_COL(15) BLUR_WR_DECL_COL(16)      // C = 2: strips of 4 columns
BLUR_WR_DECL_ROW(3) BLUR_WR_DECL_ROW(4) BLUR_WR_DECL_ROW(5) BLUR_WR_DECL_ROW(6) BLUR_WR_DECL_ROW(8) BLUR_WR_DECL_ROW(9) BLUR_WR_DECL_ROW(10)
BLUR_WR_DECL_ROW(12) BLUR_WR_DECL_ROW(15) BLUR_WR_DECL_ROW(16)
#undef BLUR_WR_DECL_COL
#undef BLUR_WR_DECL_ROW
// smallest supported transform that holds `need` points (nullptr: none)
inline const WrEntry* find_wr_entry(int need, bool column_role)
{
    static const WrEntry* const cols[] = { wr_col_entry_3(), wr_col_entry_4(), wr_col_entry_5(), wr_col_entry_6(), wr_col_entry_8(), wr_col_entry_9(),
                                           wr_col_entry_10(), wr_col_entry_12(), wr_col_entry_15(), wr_col_entry_16() };
    static const WrEntry* const rows[] = { wr_row_entry_3(), wr_row_entry_4(), wr_row_entry_5(), wr_row_entry_6(), wr_row_entry_8(), wr_row_entry_9(),
                                           wr_row_entry_10(), wr_row_entry_12(), wr_row_entry_15(), wr_row_entry_16() };
    const WrEntry* const* list = column_role ? cols : rows;
    const int n = column_role ? static_cast<int>(sizeof cols / sizeof cols[0]) : static_cast<int>(sizeof rows / sizeof rows[0]);
    for (int i = 0; i < n; ++i)                 // ascending R0
        if (list[i]->r0 * kWrS >= need) return list[i];
    return nullptr;
}
}  // namespace blur_amd
