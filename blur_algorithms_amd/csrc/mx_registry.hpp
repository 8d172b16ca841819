// mx_registry.hpp -- the window sizes (NKB blocks of 16) the matrix-core engine is instantiated for, one translation
// unit each (mx_conv_<NKB>.hip).  pad <= 8 (NKB - 2).
#pragma once
#include "mx_kernels.hpp"
namespace blur_amd {
#define BLUR_MX_DECL(NKB_) const MxEntry* mx_entry_##NKB_();
BLUR_MX_DECL(11)
#undef BLUR_MX_DECL
// smallest instantiated window that holds the taps (nullptr: none)
inline const MxEntry* find_mx_entry(int pad)
{
    static const MxEntry* const list[] = { mx_entry_11() };
    for (const MxEntry* e : list)
        if (8 * (e->nkb - 2) >= pad) return e;
    return nullptr;
}
}  // namespace blur_amd
