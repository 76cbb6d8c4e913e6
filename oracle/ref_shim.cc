/* ref_shim.cc — C wrappers around the REFERENCE's own DSP objects.
 *
 * TEST INFRASTRUCTURE ONLY.  Compiled by oracle/Makefile together with the
 * reference sources where they lie under /root/reference (never copied into
 * this repo) into oracle/_ref/libmeters_ref.so, which is git-ignored.  It pins
 * the CPU restatement (mtr_oracle.c) and generates tests/golden/.
 *
 * Buildable from the reference's own files with g++ alone:
 *   ebumeter/ebu_r128_proc.cc, jmeters/truepeakdsp.cc, jmeters/vumeterdsp.cc,
 *   zita-resampler/resampler.cc, zita-resampler/resampler-table.cc, src/spectr.c
 * NOT buildable here (need the LV2 SDK headers, which this image lacks, and
 * stand-in headers are not allowed): src/meters.cc, src/ebulv2.cc,
 * src/spectrumlv2.c, src/bitmeter.c, src/sigdistlv2.c.  For those the loop
 * around the reference kernels (e.g. spectrum_run's per-sample loop around
 * bandpass_process) is restated below / in mtr_oracle.c and pinned by
 * known-answer tests only.
 *
 * This TU is compiled with -fno-access-control so it can read the reference
 * classes' private state (filter coefficients, fragment-power ring, FIR table)
 * without touching the reference headers.
 */
#include <assert.h>
#include <complex>
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ebumeter/ebu_r128_proc.h"
#include "jmeters/truepeakdsp.h"
#include "jmeters/vumeterdsp.h"
#include "jmeters/iec1ppmdsp.h"
#include "jmeters/iec2ppmdsp.h"
#include "jmeters/msppmdsp.h"
#include "jmeters/stcorrdsp.h"
#include "jmeters/kmeterdsp.h"

namespace refspectr {
#include "src/spectr.c"
}

using namespace LV2M;

extern "C" {

/* ---- EBU R128 --------------------------------------------------------- */

void* ref_ebu_new (int nchan, float fsamp)
{
	Ebu_r128_proc* e = new Ebu_r128_proc ();
	e->init (nchan, fsamp);
	return e;
}
void ref_ebu_free (void* h) { delete (Ebu_r128_proc*) h; }
void ref_ebu_reset (void* h) { ((Ebu_r128_proc*) h)->reset (); }
void ref_ebu_integr_reset (void* h) { ((Ebu_r128_proc*) h)->integr_reset (); }
void ref_ebu_integr_start (void* h) { ((Ebu_r128_proc*) h)->integr_start (); }
void ref_ebu_integr_pause (void* h) { ((Ebu_r128_proc*) h)->integr_pause (); }

void ref_ebu_process (void* h, int nfram, float* L, float* R)
{
	float* in[2] = { L, R };
	((Ebu_r128_proc*) h)->process (nfram, in);
}

void ref_ebu_coef (void* h, float* out7)
{
	Ebu_r128_proc* e = (Ebu_r128_proc*) h;
	out7[0] = e->_a0; out7[1] = e->_a1; out7[2] = e->_a2;
	out7[3] = e->_b1; out7[4] = e->_b2; out7[5] = e->_c3; out7[6] = e->_c4;
}

void ref_ebu_get (void* h, float* out9, int32_t* hist_M, int32_t* hist_S, int32_t* counts2)
{
	Ebu_r128_proc* e = (Ebu_r128_proc*) h;
	out9[0] = e->loudness_M (); out9[1] = e->maxloudn_M ();
	out9[2] = e->loudness_S (); out9[3] = e->maxloudn_S ();
	out9[4] = e->integrated (); out9[5] = e->integ_thr ();
	out9[6] = e->range_min ();  out9[7] = e->range_max (); out9[8] = e->range_thr ();
	if (hist_M) memcpy (hist_M, e->histogram_M (), 751 * sizeof (int));
	if (hist_S) memcpy (hist_S, e->histogram_S (), 751 * sizeof (int));
	if (counts2) { counts2[0] = e->hist_M_count (); counts2[1] = e->hist_S_count (); }
}

/* ring of fragment powers + write index, for per-fragment comparisons */
void ref_ebu_ring (void* h, float* power64, int* wrind, int* frcnt, float* frpwr)
{
	Ebu_r128_proc* e = (Ebu_r128_proc*) h;
	memcpy (power64, e->_power, 64 * sizeof (float));
	*wrind = e->_wrind;
	*frcnt = e->_frcnt;
	*frpwr = e->_frpwr;
}

void ref_ebu_state (void* h, float* z /* [nchan][4] */)
{
	Ebu_r128_proc* e = (Ebu_r128_proc*) h;
	for (int c = 0; c < e->_nchan; ++c) {
		z[4 * c + 0] = e->_fst[c]._z1; z[4 * c + 1] = e->_fst[c]._z2;
		z[4 * c + 2] = e->_fst[c]._z3; z[4 * c + 3] = e->_fst[c]._z4;
	}
}

/* whole interleaved stream in host blocks; mirrors mo_batch_ebu */
void ref_batch_ebu (const float* x, uint32_t T, float fsamp, uint32_t block,
                    float* out9, int32_t* hist_M, int32_t* hist_S, int32_t* counts2,
                    float* frag_power)
{
	Ebu_r128_proc* e = new Ebu_r128_proc ();
	e->init (2, fsamp);
	e->integr_start ();
	float* L = (float*) malloc (sizeof (float) * block);
	float* R = (float*) malloc (sizeof (float) * block);
	uint32_t nf = 0;
	for (uint32_t pos = 0; pos < T; pos += block) {
		uint32_t n = (T - pos < block) ? T - pos : block;
		for (uint32_t i = 0; i < n; ++i) { L[i] = x[2 * (size_t)(pos + i)]; R[i] = x[2 * (size_t)(pos + i) + 1]; }
		if (frag_power) {
			uint32_t done = 0;
			while (done < n) {
				uint32_t k = (uint32_t) e->_frcnt < n - done ? (uint32_t) e->_frcnt : n - done;
				float* in[2] = { L + done, R + done };
				int w0 = e->_wrind;
				e->process ((int) k, in);
				if (e->_wrind != w0) frag_power[nf++] = e->_power[w0];
				done += k;
			}
		} else {
			float* in[2] = { L, R };
			e->process ((int) n, in);
		}
	}
	ref_ebu_get (e, out9, hist_M, hist_S, counts2);
	free (L); free (R);
	delete e;
}

/* ---- true peak -------------------------------------------------------- */

void* ref_tp_new (float fsamp)
{
	TruePeakdsp* t = new TruePeakdsp ();
	t->init (fsamp);
	return t;
}
void  ref_tp_free (void* h) { delete (TruePeakdsp*) h; }
void  ref_tp_process (void* h, float* p, int n) { ((TruePeakdsp*) h)->process (p, n); }
void  ref_tp_process_max (void* h, float* p, int n) { ((TruePeakdsp*) h)->process_max (p, n); }
float ref_tp_read (void* h) { return ((TruePeakdsp*) h)->read (); }
void  ref_tp_read2 (void* h, float* m, float* p) { ((TruePeakdsp*) h)->read (*m, *p); }
void  ref_tp_reset (void* h) { ((TruePeakdsp*) h)->reset (); }

/* the 4n oversampled outputs of the last process/process_max call */
void ref_tp_lastbuf (void* h, float* out, int n4) { memcpy (out, ((TruePeakdsp*) h)->_buf, n4 * sizeof (float)); }

void ref_tp_table (void* h, float* out120)
{
	TruePeakdsp* t = (TruePeakdsp*) h;
	Resampler_table* T = t->_src._table;
	memcpy (out120, T->_ctab, T->_hl * (T->_np + 1) * sizeof (float));
}

void ref_tp_resampler_state (void* h, unsigned* nread, unsigned* phase, unsigned* index)
{
	TruePeakdsp* t = (TruePeakdsp*) h;
	*nread = t->_src._nread; *phase = t->_src._phase; *index = t->_src._index;
}

void ref_tp_consts (void* h, float* out4)
{
	TruePeakdsp* t = (TruePeakdsp*) h;
	out4[0] = t->_w1; out4[1] = t->_w2; out4[2] = t->_w3; out4[3] = t->_g;
}

void ref_batch_tp (const float* x, uint32_t T, float fsamp, uint32_t block, float* peak2)
{
	TruePeakdsp tl, tr;
	if (block > 8192) block = 8192;
	tl.init (fsamp); tr.init (fsamp);
	float* L = (float*) malloc (sizeof (float) * block);
	float* R = (float*) malloc (sizeof (float) * block);
	peak2[0] = peak2[1] = 0;
	for (uint32_t pos = 0; pos < T; pos += block) {
		uint32_t n = (T - pos < block) ? T - pos : block;
		for (uint32_t i = 0; i < n; ++i) { L[i] = x[2 * (size_t)(pos + i)]; R[i] = x[2 * (size_t)(pos + i) + 1]; }
		tl.process_max (L, (int) n);
		tr.process_max (R, (int) n);
		float a = tl.read (), b = tr.read ();
		if (a > peak2[0]) peak2[0] = a;
		if (b > peak2[1]) peak2[1] = b;
	}
	free (L); free (R);
}

/* ---- filter bank (reference kernels bandpass_setup / bandpass_process) -- */

void* ref_band_new (double rate, double freq, double band, int order)
{
	refspectr::FilterBank* fb = (refspectr::FilterBank*) calloc (1, sizeof (refspectr::FilterBank));
	refspectr::bandpass_setup (fb, rate, freq, band, order);
	return fb;
}
void  ref_band_free (void* h) { free (h); }
float ref_band_process (void* h, float in) { return refspectr::bandpass_process ((refspectr::FilterBank*) h, in); }
void  ref_band_coef (void* h, double* out36)
{
	refspectr::FilterBank* fb = (refspectr::FilterBank*) h;
	for (int i = 0; i < 6; ++i) memcpy (out36 + 6 * i, fb->f[i].W, 6 * sizeof (double));
}

/* The 30-band layout and per-sample loop of spectrum_run (src/spectrumlv2.c:90-118,
 * 208-248) around the reference's own bandpass_* — spectrumlv2.c itself needs LV2
 * headers and cannot be compiled here. */
void ref_batch_spectr (const float* x, uint32_t T, double rate, uint32_t block,
                       float* val30, float* max30, float* valdb30, float* maxdb30)
{
	refspectr::FilterBank* flt = (refspectr::FilterBank*) calloc (30, sizeof (refspectr::FilterBank));
	float val_f[30] = { 0 }, max_f[30] = { 0 }, sval[30] = { 0 }, smax[30] = { 0 };
	const float omega = 1.0f - expf (-2.0 * M_PI * 1.0 / rate);
	const double f1f = pow (2, -1. / 6.), f2f = pow (2, 1. / 6.);
	for (int i = 0; i < 30; ++i) {
		const double f_m = pow (2, (i - 16) / 3.) * 1000;
		refspectr::bandpass_setup (&flt[i], rate, f_m, f_m * f2f - f_m * f1f, 6);
	}
	for (uint32_t pos = 0; pos < T; pos += block) {
		uint32_t n = (T - pos < block) ? T - pos : block;
		for (int i = 0; i < 30; ++i) { val_f[i] = sval[i]; max_f[i] = smax[i]; }
		for (uint32_t j = 0; j < n; ++j) {
			const float L = x[2 * (size_t)(pos + j)], R = x[2 * (size_t)(pos + j) + 1];
			const float in = (L + R) / 2.0f;
			for (int i = 0; i < 30; ++i) {
				const float v = refspectr::bandpass_process (&flt[i], in);
				const float s = v * v;
				val_f[i] += omega * (s - val_f[i]);
				if (val_f[i] > max_f[i]) max_f[i] = val_f[i];
			}
		}
		for (int i = 0; i < 30; ++i) {
			if (!std::isfinite (val_f[i])) val_f[i] = 0;
			if (!std::isfinite (max_f[i])) max_f[i] = 0;
			sval[i] = val_f[i] + 1e-20f;
			smax[i] = max_f[i];
			const float vs = sqrtf (2. * val_f[i]);
			const float mx = sqrtf (2. * max_f[i]);
			if (valdb30) valdb30[i] = vs > .00001f ? 20.0 * log10f (vs) : -100.0;
			if (maxdb30) maxdb30[i] = mx > .00001f ? 20.0 * log10f (mx) : -100.0;
		}
	}
	if (val30) memcpy (val30, sval, sizeof (sval));
	if (max30) memcpy (max30, smax, sizeof (smax));
	free (flt);
}

/* ---- VU --------------------------------------------------------------- */

void* ref_vu_new (float fsamp)
{
	Vumeterdsp* v = new Vumeterdsp ();
	Vumeterdsp::init (fsamp);
	return v;
}
void  ref_vu_free (void* h) { delete (Vumeterdsp*) h; }
void  ref_vu_process (void* h, float* p, int n) { ((Vumeterdsp*) h)->process (p, n); }
float ref_vu_read (void* h) { return ((Vumeterdsp*) h)->read (); }

/* ---- the other needle meters: kind 1 = Iec1ppmdsp, 2 = Iec2ppmdsp ----------------------------- */

void* ref_ppm_new (int kind, float fsamp)
{
	if (kind == 1) { Iec1ppmdsp* p = new Iec1ppmdsp (); Iec1ppmdsp::init (fsamp); return p; }
	Iec2ppmdsp* p = new Iec2ppmdsp (); Iec2ppmdsp::init (fsamp); return p;
}
void  ref_ppm_free (void* h) { delete (JmeterDSP*) h; }
void  ref_ppm_process (void* h, float* p, int n) { ((JmeterDSP*) h)->process (p, n); }
float ref_ppm_read (void* h) { return ((JmeterDSP*) h)->read (); }

void* ref_msppm_new (float fsamp, float mdb) { Msppmdsp* p = new Msppmdsp (mdb); Msppmdsp::init (fsamp); return p; }
void  ref_msppm_free (void* h) { delete (Msppmdsp*) h; }
void  ref_msppm_set_gain (void* h, float db) { ((Msppmdsp*) h)->set_gain (db); }
void  ref_msppm_process (void* h, float* l, float* r, int n, int side)
{
	if (side) ((Msppmdsp*) h)->processS (l, r, n); else ((Msppmdsp*) h)->processM (l, r, n);
}
float ref_msppm_read (void* h) { return ((Msppmdsp*) h)->read (); }

void* ref_stcorr_new (int fsamp, float flp, float tcf) { Stcorrdsp* c = new Stcorrdsp (); c->init (fsamp, flp, tcf); return c; }
void  ref_stcorr_free (void* h) { delete (Stcorrdsp*) h; }
void  ref_stcorr_process (void* h, float* l, float* r, int n) { ((Stcorrdsp*) h)->process (l, r, n); }
float ref_stcorr_read (void* h) { return ((Stcorrdsp*) h)->read (); }

void* ref_kmeter_new (float fsamp) { Kmeterdsp* k = new Kmeterdsp (); k->init (fsamp); return k; }
void  ref_kmeter_free (void* h) { delete (Kmeterdsp*) h; }
void  ref_kmeter_process (void* h, float* p, int n) { ((Kmeterdsp*) h)->process (p, n); }
void  ref_kmeter_read (void* h, float* rms, float* peak) { ((Kmeterdsp*) h)->read (*rms, *peak); }
void  ref_kmeter_reset (void* h) { ((Kmeterdsp*) h)->reset (); }

} /* extern "C" */
