/* mtr_oracle.h — CPU restatement of the meters.lv2 per-sample DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP engine, not
 * a product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under meters.lv2_amd/ links, imports
 * or calls it.
 *
 * Every function restates (from the equations, not by copying) one routine of
 * the reference (x42/meters.lv2 v0.9.28); the citation next to each prototype
 * is the reference file:line it follows.  Built with the reference's release
 * flags (Makefile:35: -O3 -msse -msse2 -mfpmath=sse -fno-finite-math-only) plus
 * -ffp-contract=off, the restatement is checked bit-for-bit against the
 * reference objects themselves (oracle/_ref, built from /root/reference by
 * oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the golden
 * vectors in tests/golden/ generated from that reference build.
 *
 * The reference ships no tests and no golden vectors of its own,
 * so the pin is: reference objects run here + known-answer values.
 */
#ifndef MTR_ORACLE_H
#define MTR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MO_MAXCH      5      /* ebumeter/ebu_r128_proc.h:26 */
#define MO_HIST_LEN   751    /* src/uris.h:45, ebu_r128_proc.cc:38 */
#define MO_NBANDS     30     /* src/spectrumlv2.c:33 */
#define MO_NSECT      6      /* src/spectr.c:55 */
#define MO_TP_HL      24     /* jmeters/truepeakdsp.cc:150 (hlen) */
#define MO_TP_NP      4      /* 4x oversampling */
#define MO_BIM_LAST   584    /* src/uris.h:60 */
#define MO_DIST_BIN   361    /* src/uris.h:47 */

/* ---- EBU R128 (ebumeter/ebu_r128_proc.{h,cc}) ------------------------- */

typedef struct {
	float a0, a1, a2, b1, b2, c3, c4;        /* ebu_r128_proc.h:121-123 */
} mo_kw_coef;

typedef struct {
	int histc[MO_HIST_LEN];                  /* ebu_r128_proc.h:58 */
	int count, error;
} mo_hist;

typedef struct {
	int   integr;                            /* ebu_r128_proc.h:102-127 */
	int   nchan;
	float fsamp;
	int   fragm, frcnt;
	float frpwr;
	float power[64];
	int   wrind, div1, div2;
	float loudness_M, maxloudn_M, loudness_S, maxloudn_S;
	float integrated, integ_thr, range_min, range_max, range_thr;
	mo_kw_coef k;
	float z[MO_MAXCH][4];
	mo_hist hist_M, hist_S;
} mo_ebu;

/* Ebu_r128_proc::detect_init  ebu_r128_proc.cc:263-293 */
void  mo_kw_init (mo_kw_coef* k, float fsamp);
/* Ebu_r128_proc::init / reset / integr_*  ebu_r128_proc.cc:166-204, .h:77-79 */
void  mo_ebu_init (mo_ebu* e, int nchan, float fsamp);
void  mo_ebu_reset (mo_ebu* e);
void  mo_ebu_integr_reset (mo_ebu* e);
void  mo_ebu_integr_start (mo_ebu* e);
void  mo_ebu_integr_pause (mo_ebu* e);
/* Ebu_r128_proc::process  ebu_r128_proc.cc:207-248 (planar input, as the reference) */
void  mo_ebu_process (mo_ebu* e, int nfram, const float* const* input);
/* Ebu_r128_proc::detect_process  ebu_r128_proc.cc:302-337 */
float mo_ebu_detect (mo_ebu* e, int nfram, const float* const* input);
/* Ebu_r128_hist::addpoint/integrate/calc_integ/calc_range  ebu_r128_proc.cc:66-150 */
void  mo_hist_reset (mo_hist* h);
void  mo_hist_addpoint (mo_hist* h, float v);
float mo_hist_integrate (const mo_hist* h, int ind);
void  mo_hist_calc_integ (const mo_hist* h, float* vi, float* th);
void  mo_hist_calc_range (const mo_hist* h, float* v0, float* v1, float* th);

/* ---- 4x true peak (jmeters/truepeakdsp.cc + zita-resampler) ----------- */

typedef struct {
	float m, p, z1, z2;                      /* truepeakdsp.h:48-60 */
	int   res;
	float w1, w2, w3, g;
	float win[2 * MO_TP_HL];                 /* the resampler's 48-sample window, oldest first */
} mo_tp;

/* Resampler_table ctor  zita-resampler/resampler-table.cc:52-75 (fr=1, hl=24, np=4) */
const float* mo_tp_table (void);             /* float[(np+1)*hl] = 120 */
/* TruePeakdsp::init  truepeakdsp.cc:148-169 */
void  mo_tp_init (mo_tp* t, float fsamp);
/* Resampler::process  resampler.cc:171-262, for the state TruePeakdsp leaves it in:
 * n inputs -> 4n outputs */
void  mo_tp_resample (mo_tp* t, const float* in, int n, float* out4n);
/* TruePeakdsp::process_max  truepeakdsp.cc:101-124 */
void  mo_tp_process_max (mo_tp* t, const float* in, int n);
/* TruePeakdsp::process  truepeakdsp.cc:41-99 */
void  mo_tp_process (mo_tp* t, const float* in, int n);
/* TruePeakdsp::read / reset  truepeakdsp.cc:127-145 */
float mo_tp_read (mo_tp* t);
void  mo_tp_read2 (mo_tp* t, float* m, float* p);
void  mo_tp_reset (mo_tp* t);

/* ---- 30-band 1/3-octave bank (src/spectr.c, src/spectrumlv2.c) -------- */

typedef struct {
	double W[6];                             /* a0 a1 a2 b0 b1 b2  spectr.c:51,57-60 */
	double z[2];
} mo_biquad;

typedef struct {
	mo_biquad f[MO_NSECT];                   /* spectr.c:62-66 */
	uint32_t  stages;
	int       ac;
} mo_band;

typedef struct {
	uint32_t nchannels;                      /* spectrumlv2.c:46-66 */
	double   rate;
	float    omega;
	float    val_f[MO_NBANDS], max_f[MO_NBANDS];
	mo_band  flt[MO_NBANDS];
	float    spec_db[MO_NBANDS], max_db[MO_NBANDS];   /* what run() writes to ports 0-29 / 30-59 */
} mo_spectr;

/* bandpass_setup  spectr.c:89-206 */
void  mo_band_setup (mo_band* fb, double rate, double freq, double band, int order);
/* bandpass_process / proc_one  spectr.c:68-87 */
float mo_band_process (mo_band* fb, float in);
/* spectrum_instantiate  spectrumlv2.c:73-121 */
void  mo_spectr_init (mo_spectr* s, uint32_t nchannels, double rate);
/* speed port change  spectrumlv2.c:170-177 */
void  mo_spectr_set_speed (mo_spectr* s, float v);
/* spectrum_run inner loop + epilogue  spectrumlv2.c:208-248 (no port handshakes) */
void  mo_spectr_run (mo_spectr* s, const float* inL, const float* inR, uint32_t n);
void  mo_spectr_reset_peak (mo_spectr* s);

/* ---- VU (jmeters/vumeterdsp.cc) — config 0 plumbing ------------------- */

typedef struct { float z1, z2, m; int res; float w, g; } mo_vu;
void  mo_vu_init (mo_vu* v, float fsamp);                  /* vumeterdsp.cc:82-86 */
void  mo_vu_process (mo_vu* v, const float* p, int n);     /* vumeterdsp.cc:45-73 */
float mo_vu_read (mo_vu* v);                               /* vumeterdsp.cc:75-79 */

/* ---- the other needle meters (CPU plumbing like VU) ---------------------- */

/* Iec1ppmdsp (DIN / Nordic) and Iec2ppmdsp (BBC / EBU): the same loop, different constants */
typedef struct { float z1, z2, m; int res; float w1, w2, w3, g; } mo_ppm;
void  mo_ppm_init_iec1 (mo_ppm* p, float fsamp);           /* iec1ppmdsp.cc:89-95 */
void  mo_ppm_init_iec2 (mo_ppm* p, float fsamp);           /* iec2ppmdsp.cc:89-95 */
void  mo_ppm_process (mo_ppm* p, const float* in, int n);  /* iec1ppmdsp.cc:47-79 = iec2ppmdsp.cc:45-79 */
float mo_ppm_read (mo_ppm* p);                             /* iec1ppmdsp.cc:82-86 */
/* Msppmdsp: the IEC2 loop on L+R (M) or L-R (S), scaled by 10^(dB/20) */
typedef struct { mo_ppm p; float db, mv; } mo_msppm;
void  mo_msppm_init (mo_msppm* p, float fsamp, float mdb); /* msppmdsp.cc:34-43, 131-137 */
void  mo_msppm_set_gain (mo_msppm* p, float db);           /* msppmdsp.cc:140-148 */
void  mo_msppm_process (mo_msppm* p, const float* l, const float* r, int n, int side);   /* :50-81 (M), :83-114 (S) */
float mo_msppm_read (mo_msppm* p);                         /* msppmdsp.cc:117-121 */
/* Stcorrdsp: stereo phase correlation */
typedef struct { float zl, zr, zlr, zll, zrr, w1, w2; } mo_stcorr;
void  mo_stcorr_init (mo_stcorr* c, int fsamp, float flp, float tcf);   /* stcorrdsp.cc:84-93 */
void  mo_stcorr_process (mo_stcorr* c, const float* l, const float* r, int n);   /* stcorrdsp.cc:47-76 */
float mo_stcorr_read (const mo_stcorr* c);                 /* stcorrdsp.cc:79-82 */
/* Kmeterdsp: RMS with ballistics + digital peak with hold and fallback */
typedef struct { float z1, z2, rms, peak; int cnt, fpp; float fall; int flag; int hold; float fsamp, omega; } mo_kmeter;
void  mo_kmeter_init (mo_kmeter* k, float fsamp);          /* kmeterdsp.cc:47-54 */
void  mo_kmeter_process (mo_kmeter* k, const float* p, int n);   /* kmeterdsp.cc:56-138 */
void  mo_kmeter_read (mo_kmeter* k, float* rms, float* peak);    /* kmeterdsp.cc:148-153 */
void  mo_kmeter_reset (mo_kmeter* k);                      /* kmeterdsp.cc:155-160 */

/* ---- DR-14 / TP+RMS (src/dr14.c; needs the LV2 headers, so this part is parity-unpinned by the
 *      reference build: restated from the source, the DSP objects underneath are pinned) -------- */
#define MO_DR_HISTBINS 8000
typedef struct {
	int      n_channels, dr_mode;
	double   rate;
	uint64_t n_sample_cnt, sample_count, num_fragments;
	float    m_dbtp[2], m_peak[2], m_rms[2];
	float    rms_sum[2], peak_cur[2], peak_hist[2][2];
	mo_kmeter km[2];
	mo_tp     tp[2];
	uint32_t  hist[2][MO_DR_HISTBINS];
} mo_dr14;
/* the port values dr14_run leaves behind (src/dr14.c:413-451) */
typedef struct { float v_rms[2], v_peak[2], m_rms[2], m_peak[2], dr[2], dr_total, block_count; } mo_dr14_ports;
void  mo_dr14_init (mo_dr14* d, int n_channels, int dr_mode, double rate);   /* dr14.c:108-166 */
void  mo_dr14_reset (mo_dr14* d);                                            /* reset_peaks :245-260 */
void  mo_dr14_run (mo_dr14* d, const float* const* in, uint32_t n, mo_dr14_ports* out);   /* :354-453 */

/* ---- integer paths ---------------------------------------------------- */

typedef struct {
	int32_t hist[MO_BIM_LAST];               /* src/uris.h:53-60 layout */
	int32_t n_zero, n_pos, n_nan, n_inf, n_den;
	float   vmin, vmax;                      /* bim_min (init +inf), bim_max (init 0) */
} mo_bitstats;
void  mo_bitstats_reset (mo_bitstats* b);                  /* bitmeter.c:47-60 */
void  mo_bitstats_run (mo_bitstats* b, const float* x, uint32_t n); /* float_stats bitmeter.c:63-105 */

typedef struct {
	int32_t bins[MO_DIST_BIN];
	int32_t peak_cnt, peak_bin;
	double  avg, var_m, var_s;
	int64_t count;
} mo_sigdist;
void  mo_sigdist_reset (mo_sigdist* d);
void  mo_sigdist_run (mo_sigdist* d, const float* x, uint32_t n);   /* sigdistlv2.c:303-318 */

/* ---- batch helpers used by tests / bench (oracle-side glue only) ------ */

/* One interleaved stereo stream [T][2] -> all EBU results; integration on from
 * frame 0; host block size `block` (the reference is block-size sensitive at
 * the 1e-6 dB level). out9 = M, maxM, S, maxS, I, I_thr, Rmin, Rmax, R_thr.
 * frag_power (may be NULL) receives the per-fragment mean powers, n_frag = T / fragm. */
void  mo_batch_ebu (const float* interleaved, uint32_t T, float fsamp, uint32_t block,
                    float* out9, int32_t* hist_M, int32_t* hist_S, int32_t* counts2,
                    float* frag_power);
/* Both channels through process_max in blocks; peak[2] = max over blocks. */
void  mo_batch_tp (const float* interleaved, uint32_t T, float fsamp, uint32_t block, float* peak2);
/* Stereo down-mix through the bank in blocks; val/max [30] raw + dB. */
void  mo_batch_spectr (const float* interleaved, uint32_t T, double rate, uint32_t block,
                       float* val30, float* max30, float* valdb30, float* maxdb30);
/* Fill an interleaved stereo stream with the repo's LCG noise:
 * s <- 1664525 s + 1013904223; u = ((s>>8) - 2^23) / 2^23; gain applied as a power of two. */
void  mo_fill_lcg (float* interleaved, uint32_t T, uint32_t seed, float gain);

#ifdef __cplusplus
}
#endif
#endif
