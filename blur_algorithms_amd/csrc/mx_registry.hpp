// mx_registry.hpp -- the window sizes (NKB blocks of 16 positions) the matrix-core engine is instantiated for, one
// translation unit each (mx_conv_<NKB>.hip).  A kernel serves every pad <= 8 (NKB - 2); the smallest that fits is used.
#pragma once
#include "mx_kernels.hpp"
namespace blur_amd {
#define BLUR_MX_DECL(NKB_) const MxEntry* mx_entry_##NKB_();
BLUR_MX_DECL(3) BLUR_MX_DECL(5) BLUR_MX_DECL(7) BLUR_MX_DECL(9) BLUR_MX_DECL(11) BLUR_MX_DECL(13) BLUR_MX_DECL(15) BLUR_MX_DECL(17) BLUR_MX_DECL(19) BLUR_MX_DECL(21) BLUR_MX_DECL(23)
#undef BLUR_MX_DECL
// smallest instantiated window that holds the taps (nullptr: none -- the FFT kernels take over)
inline const MxEntry* find_mx_entry(int pad)
{
    static const MxEntry* const list[] = { mx_entry_3(), mx_entry_5(), mx_entry_7(), mx_entry_9(), mx_entry_11(), mx_entry_13(), mx_entry_15(), mx_entry_17(), mx_entry_19(), mx_entry_21(), mx_entry_23() };
    for (const MxEntry* e : list)
        if (8 * (e->nkb - 2) >= pad) return e;
    return nullptr;
}
}  // namespace blur_amd
