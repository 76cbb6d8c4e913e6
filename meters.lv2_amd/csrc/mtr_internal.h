/* mtr_internal.h — shared between the engine host code and the HIP kernel TUs. Not installed. */
#ifndef MTR_INTERNAL_H
#define MTR_INTERNAL_H

#include <stdint.h>

#include "mtr_engine.h"

#define MTR_FIR_HALO   47          /* 2*hl - 1 frames of history the 48-tap window needs */
#define MTR_WARM_SEC   0.075f      /* K-filter warm-up for mid-stream segments of the wave-per-segment kernels, in whole tiles (two of ~2500
                                    * frames at 48 kHz: |lambda|^4992 = 1.5e-11; see MTR_SEG_WARM_SEC.  Was 0.2 s = four tiles: 1e-21) */
#ifndef MTR_SEG_WARM_SEC
#define MTR_SEG_WARM_SEC 0.075f    /* the segments of the lane = segment kernel (layout 7): |lambda|^(0.075 fs) = 1.6e-8 of the state a segment
                                    * starts without — under f32's own resolution of it, 6e-8, which 0.0694 s reach (any rate: the slowest
                                    * pole is the 38 Hz high-pass).  Measured: 0.2 -> 0.1 s -2.9 % kernel time, 0.1 -> 0.075 s -0.9 % */
#endif

typedef struct mtr_stream_state mtr_stream_state;
typedef struct mtr_fused_args mtr_fused_args;
typedef struct mtr_gate_args mtr_gate_args;
typedef struct mtr_bank_args mtr_bank_args;

/* Per-stream persistent state (device). */
struct mtr_stream_state {
	float    kz[8];            /* K-filter states z1..z4, channel-interleaved: z1L z1R z2L z2R ... */
	float    frpwr;            /* partial power of the fragment in progress (Ebu_r128_proc::_frpwr) */
	float    ring[64];         /* _power[], chronological: ring[63] is the newest fragment */
	int32_t  div1, div2;
	float    loud_M, max_M, loud_S, max_S;
	float    integ, integ_thr, rmin, rmax, rthr;
	int32_t  cnt_M, cnt_S, err_M, err_S;
	uint32_t tp_call[2];       /* float bits, atomicMax target of the fused kernel; zero between calls */
	float    tp_last[2];       /* peak of the most recent call (process_max + read) */
	float    tp_hold[2];       /* max since reset */
	float    tpb_z1[2], tpb_z2[2], tpb_m[2], tpb_p[2];   /* TPBALLIST */
};

/* Arguments of the fused K-weighting + true-peak kernel (by value in the kernarg segment). */
struct mtr_fused_args {
	const float*    audio;        /* [S][stride][2] */
	uint64_t        stride;       /* frames */
	const float*    hist;         /* [S][47][2]: the 47 frames before frame 0 of this call */
	const uint32_t* tile_start;   /* [n_tiles + 1] frame offsets; tile j = [start[j], start[j+1]) */
	const uint32_t* seg_tile;     /* [n_segs + 1] first tile of each time segment */
	const float*    scan_m;       /* [6][16] (A^K)^(2^d), row-major, for the wave scan */
	mtr_stream_state* state;      /* [S] */
	float*          tile_power;   /* [S][n_tiles] channel-weighted sum of y^2 over the tile */
	uint32_t        n_streams, n_segs, n_tiles, warm_tiles;
	uint64_t        n_frames;     /* frames per stream in this call (bounds for staging) */
	uint32_t        buf_slots;    /* wave-specialised kernel: 8-byte slots per LDS buffer (multiple of 128) */
	uint32_t        fir_form;     /* 0 = mirror-symmetric form (120 ops/frame), 1 = dense 3 x 48 taps (144) */
	uint32_t        rotate;       /* wave-specialised kernel: rotate the loader / K-filter role over the four waves */
	uint32_t        prune;        /* exact peak pruning: skip the interpolator where L1 * max|x| cannot beat the running peak */
	uint32_t*       prune_stats;  /* [4] tile passes considered / skipped, channel-blocks screened / completed (device counters), may be NULL */
	const uint16_t* mfma_a;       /* layout 6: [12][64][8] hi / lo A fragments of the matrix-pipe interpolator (mtr_mfma16_fir.h) */
	float           a0, a1, a2, b1, b2, c3, c4;
	float           gain_l, gain_r;
};

/* Arguments of the lane = time segment kernel (mtr_seg.hip, layout 7).  The launch covers the whole fragments of a call,
 * from `head` frames into it (the rest of a fragment the call started in, k_kwtp16's); every tile is tile_frames long — any
 * length of at least four steps: a tile that is not a multiple of MTR_SEG_STEP ends inside a step.  Segment q of a stream
 * answers for seg_base + (q < seg_rem) consecutive tiles, and every lane processes n_main = seg_base + (seg_rem > 0) of
 * them, starting one tile early where its own segment is the shorter kind. */
#define MTR_SEG_STEP 16            /* frames per lane and step: one column of the block-Toeplitz product */
typedef struct mtr_seg_args {
	const float*    audio;        /* [S][stride][2]: the call's buffer (8-byte aligned: a segment may start on any frame) */
	uint64_t        stride;       /* frames */
	const float*    hist;         /* [S][47][2]: the 47 frames before frame 0 of this call */
	uint32_t        head;         /* frames of this call in front of the launch's first tile (the rest of a fragment the call started in) */
	uint32_t        tile0;        /* ... and how many tiles of the plan they are: tile_power index of the launch's first tile */
	mtr_stream_state* state;      /* [S] */
	float*          tile_power;   /* [S][n_tiles] */
	const uint16_t* mfma_a;       /* [12][64][8] hi / lo A fragments (mtr_mfma16_fir.h) */
	uint32_t        n_streams, n_segs, n_tiles, tile_frames;
	uint32_t        seg_base, seg_rem, n_main;
	uint32_t        warm_steps;   /* K-filter warm-up in front of a segment that does not start the call: steps of 16 frames, multiple of 4 */
	int64_t         p0_end;       /* phase 0 (|x[n - 24]|) of this call covers the frames below n_frames - 24: that frame, counted from the first tile */
	float           a0, a1, a2, b1, b2, c3, c4;
	float           gain_l, gain_r;
} mtr_seg_args;

struct mtr_gate_args {
	mtr_stream_state* state;      /* [S] */
	int32_t*        hist;         /* [S][2][751] */
	const float*    tile_power;   /* [S][n_tiles] */
	const uint32_t* frag_tile;    /* [n_frag + 1] first tile of each fragment that ENDS in this call */
	float*          frag_power;   /* [S][n_frag] out (mean power per fragment), may be NULL */
	const float*    bin_power;    /* [100] 10^(j/100) as powf gives it on the host */
	uint32_t        n_streams, n_tiles, n_frag;
	uint32_t        tail_tile;    /* first tile after the last complete fragment (tiles of the open fragment) */
	float           fragm;        /* frames per fragment, as float */
	int32_t         integr;       /* integration on? */
	int32_t*        max_scratch;  /* [S][2] max-hold of M / S as sortable ints, for the multi-workgroup path */
	uint32_t        polite_grid;  /* 0: one workgroup per stream; else at most this many workgroups, each walking several streams (deferred gate) */
	int32_t         fold_tp;      /* 1: fold tp_call into tp_last / tp_hold here (the serial order); 0: k_history has done it (deferred gate) */
};

struct mtr_bank_args {
	const float*    audio;
	uint64_t        stride;
	uint64_t        n_frames;
	const double*   coef;         /* [30][6][5]: b0 b1 b2 a1 a2 per section */
	double*         z;            /* [S][30][12] section states */
	float*          val;          /* [S][30] */
	float*          mx;           /* [S][30] */
	const int32_t*  ac_in;        /* [S] dither toggle parity at the start of the call (shared by the 30 bands of a stream) */
	int32_t*        ac_out;       /* [S] ... and after it: ANOTHER buffer (a workgroup that starts late must still read the old one) */
	uint32_t        n_streams, n_channels;
	float           omega;
};

/* per-stream state of the integer paths */
typedef struct mtr_bitstats_state {
	int32_t hist[MTR_BIM_LAST];      /* src/uris.h:53-60 layout */
	int32_t n_zero, n_pos, n_nan, n_inf, n_den;
	float   vmin, vmax;              /* bim_min (init +inf), bim_max (init 0) */
} mtr_bitstats_state;

typedef struct mtr_sigdist_state {
	int32_t  bins[MTR_DIST_BIN];
	int32_t  peak_cnt, peak_bin;
	double   avg, var_m, var_s;      /* hist_avgS (sum), hist_tmpS (mean), hist_varS (M2) */
	int64_t  count, n_binned;
	unsigned long long last[MTR_DIST_BIN];   /* 1-based index of the last sample per bin (peak tie-break) */
} mtr_sigdist_state;

/* DR-14 per-stream state (src/dr14.c LV2dr14: rms_sum, peak_cur, peak_hist, m_rms, m_peak, num_fragments) */
typedef struct mtr_dr14_state {
	float    rms_sum[2], peak_cur[2], peak_hist[2][2], m_rms[2], m_peak[2];
	uint32_t num_fragments;
} mtr_dr14_state;

typedef struct mtr_dr14_args {
	const float*    audio;        /* [S][stride][C] */
	uint64_t        stride, n_frames;
	uint64_t        window;       /* n_sample_cnt + 1 samples close a window (dr14.c:404) */
	uint64_t        e0;           /* call frame at which the window open on entry closes */
	uint32_t        n_streams, n_channels, n_pieces, n_windows;
	mtr_dr14_state* state;        /* [S] */
	uint32_t*       hist;         /* [S][C][8000] */
	double*         piece_sum;    /* [S][n_pieces][2] */
	float*          piece_peak;   /* [S][n_pieces][2] */
} mtr_dr14_args;

/* Kmeterdsp per (stream, channel) state (jmeters/kmeterdsp.h) */
typedef struct mtr_kmeter_state {
	float    z1, z2, rms, peak;
	int32_t  cnt, flag;
} mtr_kmeter_state;

typedef struct mtr_kmeter_args {
	const float*    audio;        /* [S][stride][C] */
	uint64_t        stride, n_groups;   /* n_frames / 4 (kmeterdsp.cc:71) */
	uint32_t        n_streams, n_channels, n_pieces, fpp;
	int32_t         hold;
	float           omega, fall;
	double          pw1[3];       /* A = [[a, 0], [c, b]] per group of four samples */
	mtr_kmeter_state* state;      /* [S][2] */
	double*         piece_state;  /* [S][n_pieces][4]: each chunk's weighted sums (z1, z2 per channel), already carried to the call's end */
	float*          piece_max;    /* [S][n_pieces][2] */
} mtr_kmeter_args;

typedef struct mtr_tpb_args mtr_tpb_args;
struct mtr_tpb_args {
	const float*    audio;        /* [S][stride][C] */
	uint64_t        stride, n_frames;
	const float*    hist;         /* [S][47][2] */
	const uint16_t* mfma_a;       /* A fragments of the matrix-pipe interpolator (mtr_mfma16_fir.h) */
	mtr_stream_state* state;
	uint32_t        n_streams, n_channels;
	float           w1, w2, w3, g;   /* truepeakdsp.cc:154-157 */
};

#ifdef __cplusplus
extern "C" {
#endif
/* host-side setup math (mtr_setup.c, plain C, no FMA contraction) */
void mtr_setup_kweight (float fsamp, float* out7);
void mtr_setup_fir_table (float* out120);
void mtr_setup_band (double rate, uint32_t band, double* out36);
void mtr_setup_bin_power (float* out100);
void mtr_setup_kweight_matrix (const float* k7, double* A16, double* B4);
void mtr_setup_hist_loudness (const int32_t* hist_M, const int32_t* hist_S, float* integ, float* integ_thr,
                              float* rmin, float* rmax, float* rthr);
#ifdef __cplusplus
}

/* kernel launchers (one per HIP TU) */
int  mtr_launch_fused2 (int run, bool ebu, bool tp, const mtr_fused_args& a, uint32_t n_units, void* stream);
int  mtr_launch_kw (int run, const mtr_fused_args& a, uint32_t n_units, void* stream);
int  mtr_launch_kwtp16 (int run, bool ebu, const mtr_fused_args& a, uint32_t n_units, void* stream);
int  mtr_launch_seg (bool ebu, const mtr_seg_args& a, uint32_t n_waves, void* stream);
size_t mtr_seg_lds_bytes (void);
int  mtr_launch_dr14 (const mtr_dr14_args& a, void* stream);
int  mtr_launch_kmeter (const mtr_kmeter_args& a, void* stream);
void mtr_kmeter_powers (float omega, double* pw1 /* [3] */);
uint32_t mtr_kmeter_pieces (uint64_t n_groups);
int  mtr_fused2_upload_taps (const float* g144);
int  mtr_launch_history (const float* audio, uint64_t stride, uint64_t n_frames, const float* hist_in,
                         float* hist_out, uint32_t n_streams, mtr_stream_state* fold_state /* NULL: k_gate folds the peaks */, void* stream);
int  mtr_launch_gate (const mtr_gate_args& a, void* stream);
int  mtr_launch_delay (uint32_t us, void* stream);
int  mtr_launch_state_init (mtr_stream_state* st, int32_t* hist, uint32_t n_streams, int what, void* stream);
int  mtr_launch_bank (const mtr_bank_args& a, void* stream);
int  mtr_launch_tpb (const mtr_tpb_args& a, void* stream);
int  mtr_launch_bitstats (const float* audio, uint64_t stride, uint64_t n_frames, mtr_bitstats_state* out,
                          uint32_t n_streams, void* stream);
int  mtr_launch_sigdist (const float* audio, uint64_t stride, uint64_t n_frames, mtr_sigdist_state* out,
                         uint32_t n_streams, void* stream);
int  mtr_launch_history_mono (const float* audio, uint64_t stride, uint64_t n_frames, const float* hist_in,
                              float* hist_out, uint32_t n_streams, void* stream);
int  mtr_launch_aggregate (const mtr_stream_state* st, const int32_t* hist, uint32_t n_streams, int32_t* d_hist, float* d_max, void* stream);
int  mtr_launch_synth (float* d_audio, uint32_t n_streams, uint64_t n_frames, uint64_t stride,
                       uint32_t seed, float fs, int kind, void* stream);
#endif

#ifdef __HIPCC__
/* TruePeakdsp::read () as the LV2 glue uses it (src/ebulv2.cc:361-365): the call's peak becomes the value a getter sees and
 * enters the max-hold; tp_call is zero again for the next call's atomicMax.  One lane per stream. */
__device__ __forceinline__ void mtr_fold_truepeak (mtr_stream_state* st)
{
	const float cl = __uint_as_float (st->tp_call[0]), cr = __uint_as_float (st->tp_call[1]);
	st->tp_last[0] = cl; st->tp_last[1] = cr;
	if (cl > st->tp_hold[0]) st->tp_hold[0] = cl;
	if (cr > st->tp_hold[1]) st->tp_hold[1] = cr;
	st->tp_call[0] = 0; st->tp_call[1] = 0;
}
#endif

#define MTR_INIT_ALL     0
#define MTR_INIT_INTEGR  1
#define MTR_INIT_TP      2

#endif
