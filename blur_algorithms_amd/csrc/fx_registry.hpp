// fx_registry.hpp -- the window sizes (NKB blocks of 16 positions) the fused matrix-core kernel (fx_kernels.hpp) is
// instantiated for, one translation unit each (fx_conv_<NKB>.hip).  A kernel serves every pad <= 8 (NKB - 2).
#pragma once
#include "fx_kernels.hpp"
namespace blur_amd {
#define BLUR_FX_DECL(NKB_) const FxEntry* fx_entry_##NKB_();
BLUR_FX_DECL(3) BLUR_FX_DECL(5) BLUR_FX_DECL(7) BLUR_FX_DECL(9) BLUR_FX_DECL(11)
#undef BLUR_FX_DECL
inline const FxEntry* find_fx_entry(int pad)
{
    static const FxEntry* const list[] = { fx_entry_3(), fx_entry_5(), fx_entry_7(), fx_entry_9(), fx_entry_11() };
    for (const FxEntry* e : list)
        if (8 * (e->nkb - 2) >= pad) return e;
    return nullptr;
}
}  // namespace blur_amd
