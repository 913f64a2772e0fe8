// Small host-side helpers shared by the plan builders (lf_erfnet.hip, lf_convchain.hip).
#pragma once
#include <string.h>

#include "lf_conv.h"

#define LF_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != 0) return rc_; \
    } while (0)

struct LfBump {   // workspace layout: float offsets, 256-byte aligned
    long cur = 0;
    long take(long nfloats) { long o = cur; cur += (nfloats + 63) / 64 * 64; return o; }
};

inline long lf_maxl(long a, long b) { return a > b ? a : b; }

inline LfTapArgs lf_no_args() { LfTapArgs a; memset(&a, 0, sizeof(a)); return a; }

// logical grid == source grid == destination grid, unit strides, no taps yet
inline LfTapGeom lf_base_geom(int N, int Hl, int Wl, int Hs, int Ws, int Cs_pix, int Hd, int Wd, int Cd_pix, int Cs, int Cd) {
    LfTapGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Hl = Hl; g.Wl = Wl;
    g.Hs = Hs; g.Ws = Ws; g.s_pix = Cs_pix; g.s_choff = 0; g.ssh = 1; g.ssw = 1;
    g.Hd = Hd; g.Wd = Wd; g.d_pix = Cd_pix; g.d_choff = 0; g.dsh = 1; g.dsw = 1; g.dah = 0; g.daw = 0;
    g.Cs = Cs; g.Cd = Cd; g.ntaps = 0;
    return g;
}
