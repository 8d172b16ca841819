// fx_registry.hpp -- the window sizes (NKB blocks of 16 positions) the fused matrix-core kernels are instantiated for, one
// translation unit each: fx_kernels.hpp (three channels per workgroup) for NKB <= 11 (fx_conv_<NKB>.hip), fw_kernels.hpp (one
// channel per workgroup) for 13 .. 23 (fw_conv_<NKB>.hip).  A kernel serves every pad <= 8 (NKB - 2).
#pragma once
#include "fx_kernels.hpp"
namespace blur_amd {
#define BLUR_FX_DECL(NKB_) const FxEntry* fx_entry_##NKB_();
BLUR_FX_DECL(3) BLUR_FX_DECL(5) BLUR_FX_DECL(7) BLUR_FX_DECL(9) BLUR_FX_DECL(11)
BLUR_FX_DECL(13) BLUR_FX_DECL(15) BLUR_FX_DECL(17) BLUR_FX_DECL(19) BLUR_FX_DECL(21) BLUR_FX_DECL(23)
#undef BLUR_FX_DECL
inline const FxEntry* find_fx_entry(int pad)
{
    static const FxEntry* const list[] = { fx_entry_3(), fx_entry_5(), fx_entry_7(), fx_entry_9(), fx_entry_11(),
                                           fx_entry_13(), fx_entry_15(), fx_entry_17(), fx_entry_19(), fx_entry_21(), fx_entry_23() };
    for (const FxEntry* e : list)
        if (8 * (e->nkb - 2) >= pad) return e;
    return nullptr;
}
}  // namespace blur_amd
