// Stream descriptors of the GRU sequence kernels (gru_seq.hip: batch-tile-persistent; gru_coop.hip: column-split for small
// batches) and their decoding from the C-ABI tables of include/vame_hip.h.
#pragma once
#include "vame_common.h"

struct GruFwdStream {
    const float* gi; int64_t gi_row, gi_t;
    const float* wp; const float* bhn;
    const float* h0; int64_t h0_row;
    float* y; int64_t y_row, y_t;
    float* hn; int64_t hn_row;
    float* stash;
    int64_t T, reverse, pad;
    const float* wpx; const float* bgi; int64_t xf;      // fused input projection (xf = features, 0 = gi is precomputed)
};
struct GruFwdParams { GruFwdStream s[8]; int nstreams; int B; int ntiles; int tile_off = 0; int kernel = 0, pace_cp = -1, pace_ld = -1; };   // tile_off: first 32-row tile of this launch (gru_coop)

struct GruBwdStream {
    const float* stash; const float* y; int64_t y_row, y_t;
    const float* h0; int64_t h0_row;
    const float* wpt;
    const float* dy; int64_t dy_row, dy_t;
    const float* dhn; int64_t dhn_row;
    float* dg;
    float* dh0; int64_t dh0_row;
    float* dbias;
    int64_t T, reverse, pad;
};
struct GruBwdParams { GruBwdStream s[8]; int nstreams; int B; int ntiles; int tile_off = 0; int kernel = 0, pace_cp = -1, pace_ld = -1; };   // pace_*: see gru_ws_bwd_kernel

// In the 32x32 accumulator layout register r of lane l holds row CR(r) + 4*(l>>5), column l&31.
#define CR(r) (((r) & 3) + 8 * ((r) >> 2))

#ifdef VAME_EMU
#define UNIFORM(x) (x)
#else
#define UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif


// GF_OPT / GB_OPT of stream 0 (include/vame_hip.h, VAME_GRU_OPT): kernel choice and pacing are arguments, not process state
// (kernel in bits 0-7, pacing values + 1 in bits 8-15 / 16-23; anything above bit 23, or a kernel number with bits above 3, is a caller bug: refused)
static inline bool gru_parse_opt(int64_t opt, int& kernel, int& pace_cp, int& pace_ld) {
    kernel = (int)(opt & 255);
    pace_cp = (int)((opt >> 8) & 255) - 1;          // -1 = default
    pace_ld = (int)((opt >> 16) & 255) - 1;
    return (opt >> 24) == 0 && kernel < 16;
}

static inline int gru_parse_fwd(const int64_t* desc, int nstreams, int B, GruFwdParams& P) {
    P.nstreams = nstreams; P.B = B; P.ntiles = (int)cdiv64(B, 32);
    VAME_CHECK_ARG(gru_parse_opt(desc[GF_OPT], P.kernel, P.pace_cp, P.pace_ld), VAME_E_BADARG, "gru_seq_fwd: malformed option word %lld", (long long)desc[GF_OPT]);
    VAME_CHECK_ARG(P.kernel <= VAME_GRU_KERNEL_SKEWED, VAME_E_BADARG, "gru_seq_fwd: unknown kernel option %d", P.kernel);
    for (int i = 0; i < nstreams; ++i) {
        const int64_t* d = desc + (int64_t)i * VAME_GRU_FWD_FIELDS;
        GruFwdStream& s = P.s[i];
        s.gi = (const float*)d[GF_GI]; s.gi_row = d[GF_GI_ROW]; s.gi_t = d[GF_GI_T];
        s.wp = (const float*)d[GF_WP]; s.bhn = (const float*)d[GF_BHN];
        s.h0 = (const float*)d[GF_H0]; s.h0_row = d[GF_H0_ROW];
        s.y = (float*)d[GF_Y]; s.y_row = d[GF_Y_ROW]; s.y_t = d[GF_Y_T];
        s.hn = (float*)d[GF_HN]; s.hn_row = d[GF_HN_ROW];
        s.stash = (float*)d[GF_STASH];
        s.T = d[GF_T]; s.reverse = d[GF_REVERSE]; s.pad = d[GF_PAD];
        s.wpx = (const float*)d[GF_WPX]; s.bgi = (const float*)d[GF_BGI]; s.xf = d[GF_XF];
        VAME_CHECK_ARG(s.gi && s.wp && s.bhn, VAME_E_BADARG, "gru_seq_fwd: stream %d: gi/wp/bhn null", i);
        VAME_CHECK_ARG((s.xf > 0) == (P.s[0].xf > 0), VAME_E_BADARG, "gru_seq_fwd: fused-input and gi streams cannot share a launch");
        VAME_CHECK_ARG(s.xf == 0 || (s.wpx && s.bgi && s.xf <= 32 && s.xf % 4 == 0 && s.gi_row % 4 == 0 && s.gi_t % 4 == 0 &&
                                     (uintptr_t)s.gi % 16 == 0), VAME_E_SHAPE,
                       "gru_seq_fwd: stream %d: fused input needs F <= 32, F %% 4 == 0 and 16-byte aligned rows", i);
        VAME_CHECK_ARG(s.T >= 1, VAME_E_SHAPE, "gru_seq_fwd: stream %d: T=%lld", i, (long long)s.T);
    }
    return VAME_OK;
}

static inline int gru_parse_bwd(const int64_t* desc, int nstreams, int B, GruBwdParams& P) {
    P.nstreams = nstreams; P.B = B; P.ntiles = (int)cdiv64(B, 32);
    VAME_CHECK_ARG(gru_parse_opt(desc[GB_OPT], P.kernel, P.pace_cp, P.pace_ld), VAME_E_BADARG, "gru_seq_bwd: malformed option word %lld", (long long)desc[GB_OPT]);
    VAME_CHECK_ARG(P.kernel <= VAME_GRU_KERNEL_SKEWED, VAME_E_BADARG, "gru_seq_bwd: unknown kernel option %d", P.kernel);
    for (int i = 0; i < nstreams; ++i) {
        const int64_t* d = desc + (int64_t)i * VAME_GRU_BWD_FIELDS;
        GruBwdStream& s = P.s[i];
        s.stash = (const float*)d[GB_STASH]; s.y = (const float*)d[GB_Y]; s.y_row = d[GB_Y_ROW]; s.y_t = d[GB_Y_T];
        s.h0 = (const float*)d[GB_H0]; s.h0_row = d[GB_H0_ROW];
        s.wpt = (const float*)d[GB_WPT];
        s.dy = (const float*)d[GB_DY]; s.dy_row = d[GB_DY_ROW]; s.dy_t = d[GB_DY_T];
        s.dhn = (const float*)d[GB_DHN]; s.dhn_row = d[GB_DHN_ROW];
        s.dg = (float*)d[GB_DG];
        s.dh0 = (float*)d[GB_DH0]; s.dh0_row = d[GB_DH0_ROW];
        s.dbias = (float*)d[GB_DBIAS];
        s.T = d[GB_T]; s.reverse = d[GB_REVERSE]; s.pad = d[GB_PAD];
        VAME_CHECK_ARG(s.stash && s.y && s.wpt && s.dg, VAME_E_BADARG, "gru_seq_bwd: stream %d: stash/y/wpt/dg null", i);
        VAME_CHECK_ARG(s.T >= 1, VAME_E_SHAPE, "gru_seq_bwd: stream %d: T=%lld", i, (long long)s.T);
    }
    return VAME_OK;
}
