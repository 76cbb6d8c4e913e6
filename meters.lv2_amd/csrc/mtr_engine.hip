// mtr_engine.hip — host side of libmtr_engine.so: the C ABI of include/mtr_engine.h.
//
// Owns device state for `n_streams` lock-step streams, turns each process call into a tiling
// plan (tiles never cross 50 ms fragment boundaries; time segments give the fused kernel enough
// independent waves when the batch is small) and launches the HIP kernels on the caller's
// stream.  There is no CPU fallback anywhere in this file: without a HIP device create() fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mtr_internal.h"
#include "mtr_mfma16_fir.h"

static thread_local std::string g_err;

static int fail (int code, const char* what, hipError_t he = hipSuccess)
{
	char buf[256];
	if (he != hipSuccess) snprintf (buf, sizeof (buf), "%s: %s", what, hipGetErrorString (he));
	else                  snprintf (buf, sizeof (buf), "%s", what);
	g_err = buf;
	return code;
}

#define HIPCHK(call) do { hipError_t he_ = (call); if (he_ != hipSuccess) return fail (MTR_ERR_HIP, #call, he_); } while (0)

template <typename T> struct DevBuf {
	T*     p = nullptr;
	size_t n = 0;
	int reserve (size_t want) {
		if (want <= n) return 0;
		if (p) (void) hipFree (p);
		p = nullptr; n = 0;
		if (hipMalloc ((void**) &p, want * sizeof (T)) != hipSuccess) return -1;
		n = want;
		return 0;
	}
	void release () { if (p) (void) hipFree (p); p = nullptr; n = 0; }
};

// page-locked host memory (staging of the n_streams = 1 host path, plan uploads, result snapshots)
template <typename T> struct PinBuf {
	T*     p = nullptr;
	size_t n = 0;
	int reserve (size_t want) {
		if (want <= n) return 0;
		if (p) (void) hipHostFree (p);
		p = nullptr; n = 0;
		if (hipHostMalloc ((void**) &p, want * sizeof (T), hipHostMallocDefault) != hipSuccess) return -1;
		n = want;
		return 0;
	}
	void release () { if (p) (void) hipHostFree (p); p = nullptr; n = 0; }
};

// The tiling plan of a call lives in one of PLAN_SLOTS device buffers, uploaded from page-locked memory ON THE CALL'S
// STREAM: a call whose (n_frames, fragment phase) differs from the previous one — every call, for 1024-frame blocks at
// 48 kHz — never overwrites arrays that kernels of an earlier call may still be reading, and never blocks the host.
constexpr int PLAN_SLOTS = 4;
struct PlanSlot {
	DevBuf<uint32_t> dev;       // [tile_start (n_tiles + 1) | seg_tile (n_segs + 1) | frag_tile (n_frag + 1)]
	PinBuf<uint32_t> pin;
	hipEvent_t       done = nullptr;   // recorded behind the last kernel that reads `dev`
	bool             pending = false;
};

struct Plan {
	uint64_t n_frames = 0;
	uint32_t frcnt_in = 0;      // frames left in the open fragment when the call starts
	uint32_t frcnt_out = 0;
	uint32_t n_tiles = 0, n_frag = 0, n_segs = 0, tail_tile = 0, buf_slots = 0, kw_slots = 0;
	uint32_t body_tiles = 0;    // whole-fragment tiles (the lane = segment kernel's part of the call), 0 = none
	uint32_t head_tiles = 0;    // ... and the tiles in front of them (the rest of a fragment the call started in)
	bool     valid = false;
};

struct mtr_engine {
	mtr_config cfg;
	int      run = 39;            // K: frames per lane run
	int      layout = 6;          // 3 = exact-f32 VALU interpolator (mtr_fused2.hip), 4 = k_kw, 6 = k_kwtp16 (+ 7: k_seg for the calls it fits)
	bool     seg_ok = false;      // layout 7: calls that fit go through k_seg (mtr_seg.hip), the rest through k_kwtp16
	uint32_t seg_slots = 1024;    // resident k_seg waves: one per SIMD
	uint64_t seg_calls = 0, seg_frames = 0;
	uint32_t fragm = 0;           // frames per 50 ms fragment
	uint32_t frcnt = 0;           // frames remaining in the open fragment (all streams in lock step)
	bool     integr = false;
	bool     advanced = false;    // a process call has run since create / reset: the lock-step cursors are no longer a fresh engine's
	float    kw[7];
	float    omega = 0.f;
	hipStream_t last_stream = nullptr;

	DevBuf<mtr_stream_state> state;
	DevBuf<int32_t>  hist;
	DevBuf<int32_t>  gate_max;      // [S][2] max-hold scratch of the multi-workgroup gate path
	DevBuf<float>    fir_hist[2];   // ping-pong 47-frame history
	int              hist_cur = 0;
	DevBuf<float>    scan_m, bin_power, tile_power[2], frag_power, stage;
	// The call's tail — k_gate, then the job's reduction (k_aggregate + the RCCL all-reduce) — DEFERRED to an engine-owned side
	// stream: the fused kernel of call i + 1 needs only what the fused kernel and k_history of call i wrote (K-filter state, FIR
	// history), never the gate's bookkeeping, so the tail of call i runs beside it instead of in front of it.  tile_power is
	// double-buffered (the gate of call i reads one while the fused kernel of call i + 1 fills the other); the true-peak fold
	// moves from the gate into k_history on the caller's stream (tp_call is already being raised by call i + 1).  Results are
	// bit for bit those of the serial order: same kernels, same inputs, the fragment inserts in fragment order (gates follow
	// one another on the side stream; ebumeter/ebu_r128_proc.cc:217-244).
	int              tail_mode = 0;          // 0 auto (a k_seg batch of >= TAIL_AUTO_STREAMS streams and >= TAIL_AUTO_FRAMES stream-frames in an EBU / TRUEPEAK engine), 1 never, 2 always
	hipStream_t      tail_stream = nullptr;
	hipEvent_t       ev_fused = nullptr;     // caller's stream -> side: the call's fused kernels are done
	hipEvent_t       ev_gate[2] = { nullptr, nullptr };   // side -> caller's: the gate that read tile_power[b] is done
	bool             gate_pending[2] = { false, false };
	hipEvent_t       ev_red = nullptr;       // side -> caller's: the reduction that read the peak holds is done (the next fold waits for it)
	bool             red_pending = false;
	hipEvent_t       ev_main = nullptr;      // caller's -> side: everything the reduction reads from the caller's stream (the fold) is done
	hipEvent_t       ev_join = nullptr;
	bool             tail_pending = false;   // the side stream holds work nobody has waited for yet
	bool             last_deferred = false;  // the most recent process call deferred its tail: mtr_engine_reduce follows it there
	int              tp_cur = 0;             // tile_power buffer of the most recent call
	uint64_t         deferred_calls = 0;
	uint32_t         tail_gate_grid = 512;   // workgroups of a deferred gate: two per CU (set from the device's CU count)
	uint32_t         tail_delay_us = 100;    // see process_device: the deferred gate must not be dispatched together with the next fused kernel
	PlanSlot         plan_slot[PLAN_SLOTS];
	int              plan_cur = 0;
	const uint32_t*  tile_start = nullptr;   // into plan_slot[plan_cur].dev
	const uint32_t*  seg_tile = nullptr;
	const uint32_t*  frag_tile = nullptr;
	const uint32_t*  head_seg = nullptr;     // {0, first tile of the k_seg body}: the one segment of the k_kwtp16 launch in front of it (if any)
	const uint32_t*  tail_seg = nullptr;     // {first tile behind the k_seg body, n_tiles}: the one segment of the k_kwtp16 launch that finishes such a call
	// n_streams = 1 host path (the shape of an LV2 run ()): own stream, page-locked staging, and ONE synchronisation per
	// block — the state (and the bank's levels) come back with the same wait and serve the result getters
	hipStream_t      own_stream = nullptr;
	PinBuf<float>    pin_in;
	PinBuf<mtr_stream_state> pin_state;
	PinBuf<float>    pin_bank;               // [2][30] val, max
	bool             snap_valid = false;
	bool             queued = false;         // something has been launched on last_stream
	hipEvent_t       xs_event = nullptr;     // orders a new stream behind the previous one
	DevBuf<double>   bank_coef, bank_z;
	DevBuf<float>    bank_val, bank_max;
	DevBuf<int32_t>  bank_ac[2];     // ping-pong: k_bank reads one, writes the other
	int              bank_ac_cur = 0;
	DevBuf<mtr_bitstats_state> bim;
	DevBuf<mtr_sigdist_state>  sdh;
	DevBuf<mtr_dr14_state>     dr_state;
	DevBuf<uint32_t>           dr_hist;       // [S][C][8000]
	DevBuf<double>             dr_sum;        // [S][pieces][2]
	DevBuf<float>              dr_peak;
	uint64_t                   dr_scnt = 0;   // samples in the open window (all streams run in lock step)
	DevBuf<mtr_kmeter_state>   km_state;      // [S][2]
	DevBuf<double>             km_piece;
	DevBuf<float>              km_max;
	double                     km_pw1[3];
	uint32_t                   km_fpp = 0;
	float                      km_fall = 0.f;
	DevBuf<float>    fir_g;         // [3][48] taps in device memory
	DevBuf<uint16_t> m16_a;         // layouts 6, 7: hi / lo A fragments of the f32-grade MFMA interpolator (mtr_mfma16_fir.h)
	DevBuf<uint32_t> prune_cnt;     // [4] interpolator tile passes considered / skipped, channel-blocks screened / completed
	uint64_t         prune_tot[4] = { 0, 0, 0, 0 };
	float            tpb_w[4];      // w1 w2 w3 g of TruePeakdsp::init
	Plan             plan;
	uint32_t         last_n_frag = 0;

	// A process call may cover a VIEW of the batch: streams [v_off, v_off + v_cnt) (v_cnt = 0: all of them).  The chunked host
	// path (mtr_engine_process_host) walks the batch view by view — every per-stream array is indexed from v_off, the host-side
	// cursors (fragment phase, ping-pong indices, open DR-14 window) move when the last view has been queued.
	uint32_t v_off = 0, v_cnt = 0;
	size_t           host_chunk_bytes = (size_t) 256 << 20;
	hipStream_t      copy_stream = nullptr;
	hipEvent_t       ev_copied[2] = { nullptr, nullptr }, ev_computed[2] = { nullptr, nullptr };

	bool timing = false;
	std::vector<hipEvent_t> ev;     // groups of EV_PER_CALL: start, fused end, gate begin, gate end (those two on the stream the gate ran on), rest begin, end
	uint32_t timed_calls = 0;
};

constexpr int EV_PER_CALL = 6;
constexpr uint32_t TAIL_AUTO_STREAMS = 4096;       // ... and streams per call
constexpr uint64_t TAIL_AUTO_FRAMES = 1ull << 24;   // stream-frames per call (134 MB of stereo f32: ~40 us of the fused kernel) from which the tail is deferred

static void mat4_mul (const double* a, const double* b, double* c)
{
	double t[16];
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j) {
			double s = 0;
			for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
			t[i * 4 + j] = s;
		}
	memcpy (c, t, sizeof (t));
}

static int upload_consts (mtr_engine* e)
{
	// (A^K)^(2^d), d = 0..5, in double, rounded once to float
	double A[16], B[4], P[16];
	mtr_setup_kweight_matrix (e->kw, A, B);
	for (int i = 0; i < 16; ++i) P[i] = (i % 5 == 0) ? 1.0 : 0.0;
	for (int i = 0; i < e->run; ++i) mat4_mul (A, P, P);
	const int K = e->run;
	std::vector<float> m (96 + 4 * K + 4 + 32 * 16);
	{
		// M^1 .. M^32 (M = A^K) for the row-broadcast steps of the DPP scan (mtr_wave.h), after the functionals
		double Q[16];
		memcpy (Q, P, sizeof (Q));
		for (int p = 0; p < 32; ++p) {
			for (int i = 0; i < 16; ++i) m[96 + 4 * K + 4 + 16 * p + i] = (float) Q[i];
			mat4_mul (P, Q, Q);
		}
	}
	for (int d = 0; d < 6; ++d) {
		for (int i = 0; i < 16; ++i) m[d * 16 + i] = (float) P[i];
		mat4_mul (P, P, P);
	}
	// end-state functionals of a K-frame run from zero state: F[n] = A^(K-1-n) B, n = 0..K-1, and the
	// constant response to the +1e-15f bias, e0 = (sum_n A^(K-1-n) B) * 1e-15
	{
		double v[4] = { B[0], B[1], B[2], B[3] }, acc[4] = { 0, 0, 0, 0 };
		for (int n = K - 1; n >= 0; --n) {
			for (int j = 0; j < 4; ++j) { m[96 + 4 * n + j] = (float) v[j]; acc[j] += v[j]; }
			double w[4];
			for (int i = 0; i < 4; ++i) w[i] = A[i * 4] * v[0] + A[i * 4 + 1] * v[1] + A[i * 4 + 2] * v[2] + A[i * 4 + 3] * v[3];
			memcpy (v, w, sizeof (v));
		}
		for (int j = 0; j < 4; ++j) m[96 + 4 * K + j] = (float) (acc[j] * (double) 1e-15f);
	}
	if (e->scan_m.reserve (m.size ())) return fail (MTR_ERR_NOMEM, "hipMalloc scan_m");
	HIPCHK (hipMemcpy (e->scan_m.p, m.data (), m.size () * sizeof (float), hipMemcpyHostToDevice));

	float bp[100];
	mtr_setup_bin_power (bp);
	if (e->bin_power.reserve (100)) return fail (MTR_ERR_NOMEM, "hipMalloc bin_power");
	HIPCHK (hipMemcpy (e->bin_power.p, bp, sizeof (bp), hipMemcpyHostToDevice));

	// 48-tap kernels of phases 1..3 from the 5x24 table (resampler.cc:216-227)
	float tab[120], g[3][48];
	mtr_setup_fir_table (tab);
	for (int ph = 1; ph <= 3; ++ph)
		for (int i = 0; i < 48; ++i)
			g[ph - 1][i] = (i < 24) ? tab[24 * ph + i] : tab[24 * (4 - ph) + (47 - i)];
	if (mtr_fused2_upload_taps (&g[0][0])) return fail (MTR_ERR_HIP, "hipMemcpyToSymbol c_fir");
	if (e->fir_g.reserve (144) || e->prune_cnt.reserve (4)) return fail (MTR_ERR_NOMEM, "hipMalloc fir_g");
	HIPCHK (hipMemset (e->prune_cnt.p, 0, 16));
	HIPCHK (hipMemcpy (e->fir_g.p, g, sizeof (g), hipMemcpyHostToDevice));
	{
		// and as A fragments of the matrix-pipe interpolator
		std::vector<uint16_t> a16 (MTR_M16_A_HALVES);
		mtr_m16_build_a (&g[0][0], a16.data ());
		if (e->m16_a.reserve (a16.size ())) return fail (MTR_ERR_NOMEM, "hipMalloc m16_a");
		HIPCHK (hipMemcpy (e->m16_a.p, a16.data (), a16.size () * sizeof (uint16_t), hipMemcpyHostToDevice));
	}
	// TruePeakdsp::init, jmeters/truepeakdsp.cc:154-157 — float / float / double, stored as float
	const float fs = e->cfg.sample_rate;
	e->tpb_w[0] = 4000.0f / fs / 4.0;
	e->tpb_w[1] = 17200.0f / fs / 4.0;
	e->tpb_w[2] = 1.0f - 7.0f / fs / 4.0;
	e->tpb_w[3] = 0.502f;
	return MTR_OK;
}

// `st` waits for everything the side stream holds (a serial gate, a reset, the caller's own aggregate behind deferred gates)
static int join_tail (mtr_engine* e, hipStream_t st)
{
	if (!e->tail_pending || !e->tail_stream) return MTR_OK;
	if (!e->ev_join) HIPCHK (hipEventCreateWithFlags (&e->ev_join, hipEventDisableTiming));
	HIPCHK (hipEventRecord (e->ev_join, e->tail_stream));
	HIPCHK (hipStreamWaitEvent (st, e->ev_join, 0));
	// (only the engine's own stream carries the later calls and the host's waits: a join onto any other stream settles nothing for them)
	if (st == e->last_stream) { e->tail_pending = false; e->gate_pending[0] = e->gate_pending[1] = false; e->red_pending = false; }
	return MTR_OK;
}

// the host waits for the caller's stream and the side stream
static int sync_all (mtr_engine* e)
{
	HIPCHK (hipStreamSynchronize (e->last_stream));
	if (e->tail_stream && e->tail_pending) {
		HIPCHK (hipStreamSynchronize (e->tail_stream));
		e->tail_pending = false; e->gate_pending[0] = e->gate_pending[1] = false; e->red_pending = false;
	}
	return MTR_OK;
}

static int tail_setup (mtr_engine* e)
{
	if (e->tail_stream) return MTR_OK;
	HIPCHK (hipStreamCreateWithFlags (&e->tail_stream, hipStreamNonBlocking));
	hipEvent_t* evs[] = { &e->ev_fused, &e->ev_gate[0], &e->ev_gate[1], &e->ev_red, &e->ev_main };
	for (hipEvent_t* v : evs) if (!*v) HIPCHK (hipEventCreateWithFlags (v, hipEventDisableTiming));
	return MTR_OK;
}

static int state_init (mtr_engine* e, int what, hipStream_t st)
{
	{ const int jrc = join_tail (e, st); if (jrc) return jrc; }       // (a deferred gate may still be writing what this clears)
	e->queued = true;                // (work on last_stream: a caller that moves to another stream must be ordered behind it)
	if (mtr_launch_state_init (e->state.p, e->hist.p, e->cfg.n_streams, what, st)) return fail (MTR_ERR_HIP, "k_state_init");
	return MTR_OK;
}

extern "C" {

const char* mtr_last_error (void) { return g_err.c_str (); }
// A library built with MTR_TIMING_ONLY_BUILD may carry kernels with a role switched off (tools/: elimination runs that price a
// part of a kernel — WRONG RESULTS by construction): it says so here, and meters.lv2_amd/engine.py refuses to load it outside tools/.
#ifdef MTR_TIMING_ONLY_BUILD
const char* mtr_version (void) { return "meters.lv2_amd 0.1 (gfx950) TIMING-ONLY BUILD: results are wrong by construction"; }
#else
const char* mtr_version (void) { return "meters.lv2_amd 0.1 (gfx950)"; }
#endif
int mtr_abi_version (void) { return MTR_ABI_VERSION; }

int mtr_kweight_coef (float sample_rate, float* out7)
{
	if (!out7 || !(sample_rate > 0)) return fail (MTR_ERR_ARG, "mtr_kweight_coef");
	mtr_setup_kweight (sample_rate, out7);
	return MTR_OK;
}

int mtr_fir_table (float* out120)
{
	if (!out120) return fail (MTR_ERR_ARG, "mtr_fir_table");
	mtr_setup_fir_table (out120);
	return MTR_OK;
}

int mtr_band_coef (double rate, uint32_t band, double* out36)
{
	if (!out36 || band >= MTR_NBANDS || !(rate > 0)) return fail (MTR_ERR_ARG, "mtr_band_coef");
	mtr_setup_band (rate, band, out36);
	return MTR_OK;
}

void mtr_hist_loudness (const int32_t* hm, const int32_t* hs, float* integ, float* integ_thr,
                        float* rmin, float* rmax, float* rthr)
{
	float d[5];
	mtr_setup_hist_loudness (hm, hs, &d[0], &d[1], &d[2], &d[3], &d[4]);
	if (integ) *integ = d[0];
	if (integ_thr) *integ_thr = d[1];
	if (rmin) *rmin = d[2];
	if (rmax) *rmax = d[3];
	if (rthr) *rthr = d[4];
}

// Which kernels serve a configuration (pure: mtr_engine_create and mtr_plan_query share it).  Returns what is wrong with it, or NULL.
static const char* resolve_layout (const mtr_config* cfg, int* layout, int* run, bool* seg_ok)
{
	if (cfg->tune_run != 0 && cfg->tune_run != 19 && cfg->tune_run != 38 && cfg->tune_run != 39) return "tune_run must be 0, 19, 38 or 39";
	if (cfg->tune_layout != 0 && cfg->tune_layout != 3 && cfg->tune_layout != 4 && cfg->tune_layout != 6 && cfg->tune_layout != 7)
		return "tune_layout must be 0, 3, 4, 6 or 7 (layouts 1, 2 and 5 of earlier versions are gone)";
	if (cfg->tune_fir > 1) return "tune_fir must be 0 or 1";
	const bool kw_only = (cfg->meters & MTR_METER_EBU) && !(cfg->meters & MTR_METER_TRUEPEAK);
	const bool has_tp = cfg->meters & MTR_METER_TRUEPEAK;
	int lay = cfg->tune_layout ? (int) cfg->tune_layout : kw_only ? 4 : (has_tp && (cfg->tune_run == 0 || cfg->tune_run == 38)) ? 7 : 3;
	*seg_ok = lay == 7 && cfg->tune_prune == 0;
	if (lay == 7) lay = 6;
	*layout = lay;
	*run = cfg->tune_run ? (int) cfg->tune_run : (lay == 6 ? 38 : 39);
	if ((lay == 6) != (*run == 38)) return "layouts 6 and 7 run 38-frame lane runs, and only they do";
	if (lay == 6 && !has_tp) return "layouts 6 and 7 are true-peak kernels: need TRUEPEAK";
	if (lay == 4 && !kw_only) return "layout 4 is the EBU-only kernel";
	if (lay == 3 && *run != 39) return "layout 3 needs tune_run 39";
	if (lay == 4 && *run != 39 && *run != 19) return "layout 4 needs tune_run 19 or 39";
	return nullptr;
}

int mtr_engine_create (const mtr_config* cfg, mtr_engine** out)
{
	if (!cfg || !out || cfg->struct_size != sizeof (mtr_config)) return fail (MTR_ERR_ARG, "mtr_engine_create: bad config");
	*out = nullptr;
	if (cfg->n_streams == 0 || !(cfg->sample_rate >= 8000.f) || cfg->meters == 0) return fail (MTR_ERR_ARG, "mtr_engine_create: n_streams / sample_rate / meters");
	if (cfg->n_channels != 1 && cfg->n_channels != 2) return fail (MTR_ERR_ARG, "n_channels must be 1 or 2");
	if (cfg->meters & ~(uint32_t) (MTR_METER_EBU | MTR_METER_TRUEPEAK | MTR_METER_SPECTR30 | MTR_METER_TPBALLIST | MTR_METER_BITSTATS
	                               | MTR_METER_SIGDIST | MTR_METER_DR14 | MTR_METER_KMETER))
		return fail (MTR_ERR_ARG, "unknown bits in the meters mask");
	if (cfg->n_channels == 1 && (cfg->meters & (MTR_METER_EBU | MTR_METER_TRUEPEAK)))
		return fail (MTR_ERR_UNSUPPORTED, "EBU / TRUEPEAK need stereo frames (the reference's EBUr128 plugin is stereo only)");
	if ((cfg->meters & (MTR_METER_BITSTATS | MTR_METER_SIGDIST)) && cfg->n_channels != 1)
		return fail (MTR_ERR_UNSUPPORTED, "BITSTATS / SIGDIST take mono streams (the reference's bitmeter / SigDistHist are mono plugins)");
	{
		int l_, r_; bool k_;
		if (const char* why = resolve_layout (cfg, &l_, &r_, &k_)) return fail (MTR_ERR_ARG, why);
	}

	int ndev = 0;
	if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0)
		return fail (MTR_ERR_NODEVICE, "no HIP device: the engine has no CPU path");
	if (cfg->device < 0 || cfg->device >= ndev) return fail (MTR_ERR_ARG, "device ordinal out of range");
	HIPCHK (hipSetDevice (cfg->device));

	mtr_engine* e = new (std::nothrow) mtr_engine ();
	if (!e) return fail (MTR_ERR_NOMEM, "new mtr_engine");
	e->cfg = *cfg;
	// layout 4 = k_kw, the K-weighting-only kernel (mtr_kw.hip): the default when no true peak is asked for;
	// layout 6 = k_kwtp16 (mtr_fused4.hip): wherever a true peak is asked for — the interpolator on the matrix pipe at f32
	//            grade, one wave per (stream, time segment);
	// layout 7 (the default with a true peak) = layout 6 plus k_seg (mtr_seg.hip, lane = time segment) for every call that
	//            fits it: a big batch that starts on a fragment boundary (seg_plan below);
	// layout 3 = the exact-f32 VALU interpolator (mtr_fused2.hip), kept as the bit-for-bit cross-check of the matrix-pipe paths.
	{
		const char* why = resolve_layout (cfg, &e->layout, &e->run, &e->seg_ok);
		if (why) { delete e; return fail (MTR_ERR_ARG, why); }
	}
	{
		hipDeviceProp_t pr;
		if (hipGetDeviceProperties (&pr, cfg->device) == hipSuccess && pr.multiProcessorCount > 0) { e->seg_slots = 4u * (uint32_t) pr.multiProcessorCount; e->tail_gate_grid = 2u * (uint32_t) pr.multiProcessorCount; }
	}
	// (test knobs: the whole -m gpu suite and the fuzzers run green with MTR_TAIL_MODE=2 — every call of every test with its tail on the side stream)
	if (const char* v = getenv ("MTR_TAIL_MODE")) { const int m = atoi (v); if (m >= 0 && m <= 2) e->tail_mode = m; }
	if (const char* v = getenv ("MTR_TAIL_DELAY_US")) e->tail_delay_us = (uint32_t) atoi (v);
	if (const char* v = getenv ("MTR_TAIL_GATE_GRID")) e->tail_gate_grid = (uint32_t) atoi (v);   // (tools/r06_tail_probe.py: the experiment behind the default)
	e->fragm = (uint32_t) ((int) cfg->sample_rate / 20);     // ebu_r128_proc.cc:170
	e->frcnt = e->fragm;
	mtr_setup_kweight (cfg->sample_rate, e->kw);
	e->omega = 1.0f - expf (-2.0 * M_PI * 1.0 / (double) cfg->sample_rate);   // spectrumlv2.c:98

	const uint32_t S = cfg->n_streams;
	int rc = MTR_OK;
	if (e->state.reserve (S) || e->hist.reserve ((size_t) S * 2 * MTR_HIST_LEN)
	    || e->fir_hist[0].reserve ((size_t) S * MTR_FIR_HALO * 2) || e->fir_hist[1].reserve ((size_t) S * MTR_FIR_HALO * 2))
		rc = fail (MTR_ERR_NOMEM, "hipMalloc stream state");
	if (rc == MTR_OK) rc = upload_consts (e);
	if (rc == MTR_OK) {
		// max-hold scratch of the multi-workgroup gate: "minus infinity" as a sortable int (mtr_gate.hip)
		std::vector<int32_t> m ((size_t) S * 2, (int32_t) 0x807fffff);
		if (e->gate_max.reserve (m.size ())) rc = fail (MTR_ERR_NOMEM, "hipMalloc gate scratch");
		else if (hipMemcpy (e->gate_max.p, m.data (), m.size () * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail (MTR_ERR_HIP, "hipMemcpy gate scratch");
	}
	if (rc == MTR_OK && (cfg->meters & MTR_METER_SPECTR30)) {
		std::vector<double> c (MTR_NBANDS * 6 * 5);
		for (uint32_t b = 0; b < MTR_NBANDS; ++b) {
			double w[36];
			mtr_setup_band ((double) cfg->sample_rate, b, w);
			for (int i = 0; i < 6; ++i) {
				double* o = &c[(b * 6 + i) * 5];
				o[0] = w[i * 6 + 3]; o[1] = w[i * 6 + 4]; o[2] = w[i * 6 + 5];   // b0 b1 b2
				o[3] = w[i * 6 + 1]; o[4] = w[i * 6 + 2];                         // a1 a2
			}
		}
		if (e->bank_coef.reserve (c.size ()) || e->bank_z.reserve ((size_t) S * MTR_NBANDS * 12)
		    || e->bank_val.reserve ((size_t) S * MTR_NBANDS) || e->bank_max.reserve ((size_t) S * MTR_NBANDS)
		    || e->bank_ac[0].reserve (S) || e->bank_ac[1].reserve (S))
			rc = fail (MTR_ERR_NOMEM, "hipMalloc bank state");
		else if (hipMemcpy (e->bank_coef.p, c.data (), c.size () * sizeof (double), hipMemcpyHostToDevice) != hipSuccess)
			rc = fail (MTR_ERR_HIP, "hipMemcpy bank_coef");
	}
	if (rc != MTR_OK) { mtr_engine_destroy (e); return rc; }
	rc = mtr_engine_reset (e);
	if (rc != MTR_OK) { mtr_engine_destroy (e); return rc; }     // never an error code together with a live handle
	*out = e;
	return MTR_OK;
}

void mtr_engine_destroy (mtr_engine* e)
{
	if (!e) return;
	(void) hipSetDevice (e->cfg.device);
	(void) hipDeviceSynchronize ();
	for (hipEvent_t ev : e->ev) (void) hipEventDestroy (ev);
	{
		hipEvent_t evs[] = { e->ev_fused, e->ev_gate[0], e->ev_gate[1], e->ev_red, e->ev_main, e->ev_join };
		for (hipEvent_t v : evs) if (v) (void) hipEventDestroy (v);
		if (e->tail_stream) (void) hipStreamDestroy (e->tail_stream);
	}
	e->state.release (); e->hist.release (); e->fir_hist[0].release (); e->fir_hist[1].release ();
	e->scan_m.release (); e->bin_power.release (); e->tile_power[0].release (); e->tile_power[1].release (); e->frag_power.release ();
	e->stage.release ();
	for (PlanSlot& ps : e->plan_slot) { ps.dev.release (); ps.pin.release (); if (ps.done) (void) hipEventDestroy (ps.done); }
	e->pin_in.release (); e->pin_state.release (); e->pin_bank.release ();
	if (e->own_stream) (void) hipStreamDestroy (e->own_stream);
	if (e->copy_stream) (void) hipStreamDestroy (e->copy_stream);
	for (int b = 0; b < 2; ++b) { if (e->ev_copied[b]) (void) hipEventDestroy (e->ev_copied[b]); if (e->ev_computed[b]) (void) hipEventDestroy (e->ev_computed[b]); }
	if (e->xs_event) (void) hipEventDestroy (e->xs_event);
	e->bank_coef.release (); e->bank_z.release (); e->bank_val.release (); e->bank_max.release (); e->bank_ac[0].release (); e->bank_ac[1].release ();
	e->fir_g.release (); e->m16_a.release ();
	e->bim.release (); e->sdh.release (); e->prune_cnt.release ();
	e->dr_state.release (); e->dr_hist.release (); e->dr_sum.release (); e->dr_peak.release ();
	e->km_state.release (); e->km_piece.release (); e->km_max.release ();
	delete e;
}

int mtr_engine_reset (mtr_engine* e)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	hipStream_t st = e->last_stream;
	int rc = state_init (e, MTR_INIT_ALL, st);
	if (rc) return rc;
	const size_t hb = (size_t) e->cfg.n_streams * MTR_FIR_HALO * 2 * sizeof (float);
	HIPCHK (hipMemsetAsync (e->fir_hist[0].p, 0, hb, st));
	HIPCHK (hipMemsetAsync (e->fir_hist[1].p, 0, hb, st));
	if (e->cfg.meters & MTR_METER_SPECTR30) {
		HIPCHK (hipMemsetAsync (e->bank_z.p, 0, e->bank_z.n * sizeof (double), st));
		HIPCHK (hipMemsetAsync (e->bank_val.p, 0, e->bank_val.n * sizeof (float), st));
		HIPCHK (hipMemsetAsync (e->bank_max.p, 0, e->bank_max.n * sizeof (float), st));
		HIPCHK (hipMemsetAsync (e->bank_ac[0].p, 0, e->bank_ac[0].n * sizeof (int32_t), st));
		HIPCHK (hipMemsetAsync (e->bank_ac[1].p, 0, e->bank_ac[1].n * sizeof (int32_t), st));
		e->bank_ac_cur = 0;
	}
	e->frcnt = e->fragm;
	e->integr = false;
	e->advanced = false;
	e->hist_cur = 0;
	e->last_n_frag = 0;
	e->last_deferred = false;
	if (e->cfg.meters & MTR_METER_DR14) { const int drc = mtr_engine_dr14_reset (e); if (drc) return drc; }
	if (e->cfg.meters & MTR_METER_KMETER) { const int krc = mtr_engine_kmeter_reset (e); if (krc) return krc; e->km_fpp = 0; e->km_fall = 0.f; }
	if (e->cfg.meters & (MTR_METER_BITSTATS | MTR_METER_SIGDIST)) return mtr_engine_intstat_reset (e);
	return MTR_OK;
}

int mtr_engine_kmeter_reset (mtr_engine* e)
{
	if (!e || !(e->cfg.meters & MTR_METER_KMETER)) return fail (MTR_ERR_ARG, "no KMETER in this engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	const size_t n = (size_t) e->cfg.n_streams * 2;
	if (e->km_state.reserve (n)) return fail (MTR_ERR_NOMEM, "hipMalloc KMETER state");
	mtr_kmeter_powers (9.72f / e->cfg.sample_rate, e->km_pw1);           // kmeterdsp.cc:52
	HIPCHK (hipStreamSynchronize (e->last_stream));
	HIPCHK (hipMemset (e->km_state.p, 0, n * sizeof (mtr_kmeter_state)));   // :142-146
	return MTR_OK;
}

int mtr_engine_kmeter_read (mtr_engine* e, uint32_t first, uint32_t count, float* rms, float* peak)
{
	if (!e || !rms || !peak || !(e->cfg.meters & MTR_METER_KMETER)) return fail (MTR_ERR_ARG, "no KMETER in this engine");
	if ((uint64_t) first + count > e->cfg.n_streams) return fail (MTR_ERR_ARG, "stream range");
	HIPCHK (hipSetDevice (e->cfg.device));
	HIPCHK (hipStreamSynchronize (e->last_stream));
	std::vector<mtr_kmeter_state> h ((size_t) count * 2);
	HIPCHK (hipMemcpy (h.data (), e->km_state.p + (size_t) first * 2, h.size () * sizeof (mtr_kmeter_state), hipMemcpyDeviceToHost));
	for (size_t i = 0; i < h.size (); ++i) { rms[i] = h[i].rms; peak[i] = h[i].peak; h[i].flag = 1; }
	HIPCHK (hipMemcpy (e->km_state.p + (size_t) first * 2, h.data (), h.size () * sizeof (mtr_kmeter_state), hipMemcpyHostToDevice));
	return MTR_OK;
}

int mtr_engine_dr14_reset (mtr_engine* e)
{
	if (!e || !(e->cfg.meters & MTR_METER_DR14)) return fail (MTR_ERR_ARG, "no DR14 in this engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	const uint32_t S = e->cfg.n_streams;
	if (e->dr_state.reserve (S) || e->dr_hist.reserve ((size_t) S * e->cfg.n_channels * MTR_DR_HISTBINS))
		return fail (MTR_ERR_NOMEM, "hipMalloc DR14 state");
	std::vector<mtr_dr14_state> h (S);
	memset (h.data (), 0, S * sizeof (mtr_dr14_state));
	for (auto& v : h) for (int c = 0; c < 2; ++c) { v.m_rms[c] = -81.f; v.m_peak[c] = -81.f; }   // dr14.c:247-248
	HIPCHK (hipStreamSynchronize (e->last_stream));
	HIPCHK (hipMemcpy (e->dr_state.p, h.data (), S * sizeof (mtr_dr14_state), hipMemcpyHostToDevice));
	HIPCHK (hipMemset (e->dr_hist.p, 0, (size_t) S * e->cfg.n_channels * MTR_DR_HISTBINS * sizeof (uint32_t)));
	e->dr_scnt = 0;
	return MTR_OK;
}

int mtr_engine_dr14_results (mtr_engine* e, uint32_t first, uint32_t count, mtr_dr14_result* out)
{
	if (!e || !out || !(e->cfg.meters & MTR_METER_DR14)) return fail (MTR_ERR_ARG, "no DR14 in this engine");
	if ((uint64_t) first + count > e->cfg.n_streams) return fail (MTR_ERR_ARG, "stream range");
	HIPCHK (hipSetDevice (e->cfg.device));
	HIPCHK (hipStreamSynchronize (e->last_stream));
	std::vector<mtr_dr14_state> h (count);
	HIPCHK (hipMemcpy (h.data (), e->dr_state.p + first, count * sizeof (mtr_dr14_state), hipMemcpyDeviceToHost));
	const int C = (int) e->cfg.n_channels;
	for (uint32_t i = 0; i < count; ++i) {
		mtr_dr14_result& r = out[i];
		memset (&r, 0, sizeof (r));
		float total = 0.f;
		int valid = 0;
		for (int c = 0; c < C; ++c) {                          // dr14.c:430-441
			const float rdb = h[i].m_rms[c], pdb = h[i].m_peak[c];
			const float dr = (0.f < pdb ? 0.f : pdb) - rdb;
			const bool ok = rdb > -80.f && pdb > -80.f;
			if (ok) { total += dr; ++valid; }
			const float cl = 20.f < dr ? 20.f : dr;
			r.dr[c] = ok ? (1.f > cl ? 1.f : cl) : 21.f;
			r.m_rms[c] = rdb; r.m_peak[c] = pdb;
		}
		if (C > 1) {                                           // :443-450
			if (valid > 0) { const float m = total / (float) valid; const float cl = 20.f < m ? 20.f : m; r.dr_total = 1.f > cl ? 1.f : cl; }
			else r.dr_total = 21.f;
		}
		r.block_count = 3.0f * (float) h[i].num_fragments;
	}
	return MTR_OK;
}

int mtr_engine_intstat_reset (mtr_engine* e)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	const uint32_t S = e->cfg.n_streams;
	HIPCHK (hipStreamSynchronize (e->last_stream));
	if (e->cfg.meters & MTR_METER_BITSTATS) {
		std::vector<mtr_bitstats_state> h (S);
		memset (h.data (), 0, S * sizeof (mtr_bitstats_state));
		for (auto& b : h) { b.vmin = INFINITY; b.vmax = 0; }          // bim_clear, src/bitmeter.c:47-55
		if (e->bim.reserve (S)) return fail (MTR_ERR_NOMEM, "hipMalloc bitstats state");
		HIPCHK (hipMemcpy (e->bim.p, h.data (), S * sizeof (mtr_bitstats_state), hipMemcpyHostToDevice));
	}
	if (e->cfg.meters & MTR_METER_SIGDIST) {
		if (e->sdh.reserve (S)) return fail (MTR_ERR_NOMEM, "hipMalloc sigdist state");
		std::vector<mtr_sigdist_state> h (S);
		memset (h.data (), 0, S * sizeof (mtr_sigdist_state));
		for (auto& d : h) d.peak_bin = -1;                            // sdh_reset, src/sigdistlv2.c:54: no peak yet
		HIPCHK (hipMemcpy (e->sdh.p, h.data (), S * sizeof (mtr_sigdist_state), hipMemcpyHostToDevice));
	}
	return MTR_OK;
}

int mtr_engine_bitstats (mtr_engine* e, uint32_t first, uint32_t count, int32_t* hist, int32_t* counters, float* minmax)
{
	if (!e || !(e->cfg.meters & MTR_METER_BITSTATS)) return fail (MTR_ERR_ARG, "no BITSTATS in this engine");
	if ((uint64_t) first + count > e->cfg.n_streams) return fail (MTR_ERR_ARG, "stream range out of bounds");
	if (count == 0) return MTR_OK;
	int rc = mtr_engine_sync (e);
	if (rc) return rc;
	std::vector<mtr_bitstats_state> h (count);
	HIPCHK (hipMemcpy (h.data (), e->bim.p + first, count * sizeof (mtr_bitstats_state), hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < count; ++i) {
		if (hist) memcpy (hist + (size_t) i * MTR_BIM_LAST, h[i].hist, sizeof (h[i].hist));
		if (counters) { int32_t* c = counters + (size_t) i * 5; c[0] = h[i].n_zero; c[1] = h[i].n_pos; c[2] = h[i].n_nan; c[3] = h[i].n_inf; c[4] = h[i].n_den; }
		if (minmax) { minmax[2 * i] = h[i].vmin; minmax[2 * i + 1] = h[i].vmax; }
	}
	return MTR_OK;
}

int mtr_engine_sigdist (mtr_engine* e, uint32_t first, uint32_t count, int32_t* bins, int32_t* peak, double* moments, int64_t* n)
{
	if (!e || !(e->cfg.meters & MTR_METER_SIGDIST)) return fail (MTR_ERR_ARG, "no SIGDIST in this engine");
	if ((uint64_t) first + count > e->cfg.n_streams) return fail (MTR_ERR_ARG, "stream range out of bounds");
	if (count == 0) return MTR_OK;
	int rc = mtr_engine_sync (e);
	if (rc) return rc;
	std::vector<mtr_sigdist_state> h (count);
	HIPCHK (hipMemcpy (h.data (), e->sdh.p + first, count * sizeof (mtr_sigdist_state), hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < count; ++i) {
		if (bins) memcpy (bins + (size_t) i * MTR_DIST_BIN, h[i].bins, sizeof (h[i].bins));
		if (peak) { peak[2 * i] = h[i].peak_cnt; peak[2 * i + 1] = h[i].peak_bin; }
		if (moments) { moments[3 * i] = h[i].avg; moments[3 * i + 1] = h[i].var_m; moments[3 * i + 2] = h[i].var_s; }
		if (n) n[i] = h[i].count;
	}
	return MTR_OK;
}

int mtr_engine_integr_start (mtr_engine* e) { if (!e) return fail (MTR_ERR_ARG, "null engine"); e->integr = true;  return MTR_OK; }
int mtr_engine_integr_pause (mtr_engine* e) { if (!e) return fail (MTR_ERR_ARG, "null engine"); e->integr = false; return MTR_OK; }
int mtr_engine_integr_reset (mtr_engine* e)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	return state_init (e, MTR_INIT_INTEGR, e->last_stream);
}
int mtr_engine_truepeak_reset (mtr_engine* e)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	return state_init (e, MTR_INIT_TP, e->last_stream);
}

int mtr_engine_spectr_set_speed (mtr_engine* e, float v)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	if (v < 0.01) v = 0.01;                                    // spectrumlv2.c:172-175
	if (v > 15.0) v = 15.0;
	e->omega = 1.0f - expf (-2.0 * M_PI * v / (double) e->cfg.sample_rate);
	return MTR_OK;
}

int mtr_engine_spectr_reset_peak (mtr_engine* e)
{
	if (!e || !(e->cfg.meters & MTR_METER_SPECTR30)) return fail (MTR_ERR_ARG, "no SPECTR30 in this engine");
	e->snap_valid = false;
	HIPCHK (hipSetDevice (e->cfg.device));
	HIPCHK (hipMemsetAsync (e->bank_max.p, 0, e->bank_max.n * sizeof (float), e->last_stream));
	e->queued = true;
	return MTR_OK;
}

// The lane = time segment kernel's share of a call (mtr_seg.hip, layout 7): `tiles` whole fragments from frame 0, cut into
// n_segs segments per stream of base (+ 1 for the first rem) tiles; every lane walks n_main of them.
struct SegPlan {
	bool     use = false;
	uint32_t head = 0;          // frames of the call in front of the first whole fragment (the rest of the open one)
	uint32_t tiles = 0, n_segs = 0, base = 0, rem = 0, n_main = 0, warm_steps = 0;
};

// Does this call go through k_seg?  It must hold at least one whole fragment behind the one it may start in, and it must be
// a BATCH: the kernel's unit of parallelism is a lane, so it needs ~64 x the units of
// k_kwtp16 to fill the chip.  The number of segments per stream is the one that minimises the modelled time — rounds of
// resident waves x steps per wave, a warm-up step (K-filter only) at 0.3 of a full one — and the call takes this path
// when that beats the model of k_kwtp16 (1.2 x the time per frame when both fill the machine, measured: 11.7 vs 9.75 ms,
// profiles/r03*; k_kwtp16's waves are a stream-tile each, so it fills the machine with any batch).
// (everything the planning needs from an engine: mtr_plan_query builds one from a configuration alone, without a device)
struct PlanCtx {
	mtr_config cfg;
	bool       seg_ok;
	int        layout, run;
	uint32_t   fragm, frcnt, seg_slots;
};
static PlanCtx plan_ctx (const mtr_engine* e)
{
	PlanCtx c;
	c.cfg = e->cfg; c.seg_ok = e->seg_ok; c.layout = e->layout; c.run = e->run; c.fragm = e->fragm; c.frcnt = e->frcnt; c.seg_slots = e->seg_slots;
	return c;
}

static SegPlan seg_plan (const PlanCtx* e, const float* d_audio, uint64_t N, uint64_t stride)
{
	SegPlan sp;
	(void) stride;
	const bool ebu = e->cfg.meters & MTR_METER_EBU;
	if (!e->seg_ok || e->layout != 6 || e->cfg.n_channels != 2) return sp;
	if (e->fragm < 4 * MTR_SEG_STEP) return sp;
	if (reinterpret_cast<uintptr_t> (d_audio) & 7) return sp;
	// (a segment may start on any frame — odd strides, 2205-frame fragments, a call that starts inside a fragment: the kernel's
	// loads only assume a frame's 8 bytes.)  The rest of an open fragment in front (`head`) and what is left behind the last whole
	// fragment go to k_kwtp16, in stream order.  A tile that is not a whole number of steps (44.1 / 88.2 kHz) lets a lane read up
	// to 15 frames past its last tile — the next segment's; the lanes of a stream's last segment stop at the frame (mtr_seg.hip).
	const uint64_t head = e->frcnt != e->fragm ? e->frcnt : 0;
	if (head >= N) return sp;
	const uint64_t Nb = N - head;
	const uint64_t tiles = Nb / e->fragm;
	if (tiles == 0 || tiles > 0x7fffffffull / (e->fragm / MTR_SEG_STEP + 1)) return sp;
	sp.head = (uint32_t) head;
	const double spt = (double) e->fragm / MTR_SEG_STEP;
	const uint32_t warm_steps = ebu ? ((uint32_t) std::ceil (MTR_SEG_WARM_SEC * e->cfg.sample_rate / (float) MTR_SEG_STEP) + 3) / 4 * 4 : 0;
	const uint32_t warm_tiles = (warm_steps * MTR_SEG_STEP + e->fragm - 1) / e->fragm;
	const uint64_t S = e->cfg.n_streams;
	uint64_t gmax = ebu ? tiles / (warm_tiles + 2) : tiles;
	if (gmax < 1) gmax = 1;
	if (e->cfg.tune_segments) gmax = std::min<uint64_t> (gmax, e->cfg.tune_segments);
	const uint64_t g0 = e->cfg.tune_segments ? gmax : 1;
	// Candidates: for a given number of tiles per lane n = ceil (tiles / g) the smallest g has the fewest rounds, so only the g
	// at which n changes are evaluated — O (sqrt (tiles)) of them instead of every g up to 65536 (this runs on every process
	// call, in the caller's thread: one stream x one hour cost 0.17 ms here before it was told to take k_kwtp16).
	double best = 0; uint64_t bg = 0;
	for (uint64_t g = g0; g <= gmax && g <= 65536; ) {
		const uint64_t waves = (S * g + 63) / 64, rounds = (waves + e->seg_slots - 1) / e->seg_slots;
		const uint64_t n_main = tiles / g + (tiles % g ? 1 : 0);
		const double t = (double) rounds * ((double) n_main * spt + (g > 1 ? 0.3 * warm_steps : 0.0));
		if (!bg || t < best) { best = t; bg = g; }
		if (n_main <= 1) break;
		g = std::max<uint64_t> (g + 1, (tiles + n_main - 2) / (n_main - 1));      // the smallest g with fewer tiles per lane
	}
	const double t6 = 1.2 * (double) S * (double) tiles * spt / (64.0 * e->seg_slots);
	if (!e->cfg.tune_segments && best > t6) return sp;
	sp.use = true;
	sp.tiles = (uint32_t) tiles; sp.n_segs = (uint32_t) bg;
	sp.base = (uint32_t) (tiles / bg); sp.rem = (uint32_t) (tiles % bg); sp.n_main = sp.base + (sp.rem ? 1 : 0);
	sp.warm_steps = bg > 1 ? warm_steps : 0;
	return sp;
}

// A launch behind build_plan's upload failed: the slot stays busy until the stream has passed this point, the plan is not reused.
static void plan_abort (mtr_engine* e, hipStream_t st)
{
	PlanSlot& ps = e->plan_slot[e->plan_cur];
	if (ps.done && hipEventRecord (ps.done, st) == hipSuccess) ps.pending = true;
	e->plan.valid = false;
}

// Tiling of a call of N frames that starts with `frcnt` frames left in the open fragment (pure).  body_tiles > 0: behind the
// `head` frames that finish the open fragment (0 if the call starts on a boundary) body_tiles tiles are whole fragments
// (k_seg's part), whatever their length.
struct Tiling {
	std::vector<uint32_t> ts, ft, sg;       // tile starts (+ N), first tile of every fragment that ends in the call, segment starts
	uint32_t n_tiles = 0, n_frag = 0, tail = 0, head_tiles = 0, n_segs = 0, frcnt_out = 0, maxlen = 0;
};
static const char* plan_tiling (const PlanCtx* e, uint64_t N, uint32_t head, uint32_t body_tiles, Tiling& t)
{
	const uint32_t LT = 64u * (uint32_t) e->run;
	std::vector<uint32_t>& ts = t.ts;
	std::vector<uint32_t>& ft = t.ft;
	ts.clear (); ft.clear ();
	ts.reserve ((size_t) (N / LT + N / e->fragm + 4));
	uint64_t pos = 0;
	uint32_t left = e->frcnt;
	uint32_t head_tiles = 0, body_done = 0;
	ft.push_back (0);
	while (pos < N) {
		const bool body = body_done < body_tiles && pos >= head;
		if (body && body_done == 0) head_tiles = (uint32_t) ts.size ();
		body_done += body;
		const uint32_t piece = body ? e->fragm : (uint32_t) std::min<uint64_t> (std::min<uint64_t> (LT, left), N - pos);
		ts.push_back ((uint32_t) pos);
		pos += piece;
		left -= piece;
		if (left == 0) { ft.push_back ((uint32_t) ts.size ()); left = e->fragm; }
	}
	ts.push_back ((uint32_t) N);
	const uint32_t n_tiles = (uint32_t) ts.size () - 1;
	const uint32_t n_frag  = (uint32_t) ft.size () - 1;      // fragments that end inside this call
	const uint32_t tail    = ft.back ();
	ft.resize ((size_t) n_frag + 1);

	// time segments: enough (stream, segment) waves to fill the chip; each at least 4 warm-up spans long
	const uint32_t warm_tiles = (uint32_t) std::ceil (MTR_WARM_SEC * e->cfg.sample_rate / (float) LT);
	const uint64_t min_seg_frames = (uint64_t) 4 * warm_tiles * LT;
	uint32_t n_segs = e->cfg.tune_segments;
	if (n_segs == 0) {
		// one-wave workgroups (layouts 4-6): eight per CU are resident, 2048 in all — exactly one round of them keeps the
		// warm-up overhead of the segments smallest (one stream x 3600 s: 0.29 ms with 2048 segments, 0.37 with 8192);
		// the four-wave workgroups of layouts 1-3 want more, smaller units
		const uint32_t target_units = e->layout >= 4 ? 2048 : 8192;
		n_segs = (target_units + e->cfg.n_streams - 1) / e->cfg.n_streams;
	}
	const uint64_t max_segs = std::max<uint64_t> (1, N / std::max<uint64_t> (min_seg_frames, 1));
	n_segs = (uint32_t) std::min<uint64_t> (n_segs, max_segs);
	if (body_tiles) n_segs = 1;                                 // k_kwtp16 only starts / finishes such a call: one segment each (head_seg, tail_seg)
	n_segs = std::max<uint32_t> (1, std::min<uint32_t> (n_segs, n_tiles));
	t.sg.assign (n_segs + 1, 0);
	for (uint32_t q = 0; q <= n_segs; ++q) t.sg[q] = (uint32_t) ((uint64_t) q * n_tiles / n_segs);
	for (uint32_t q = 1; q < n_segs; ++q)
		if ((uint64_t) ts[t.sg[q]] < (uint64_t) warm_tiles * LT) return "internal: segment shorter than its warm-up";
	t.n_tiles = n_tiles; t.n_frag = n_frag; t.tail = tail; t.head_tiles = head_tiles; t.n_segs = n_segs; t.frcnt_out = left;
	t.maxlen = n_segs > 1 ? LT : 0;                             // warm-up tiles are full tiles
	for (uint32_t j = 0; j < n_tiles; ++j)
		if (j < head_tiles || j >= head_tiles + body_tiles) t.maxlen = std::max (t.maxlen, ts[j + 1] - ts[j]);
	return nullptr;
}

// The plan of a call on the device: the tiling above in the next slot of the plan ring.
static int build_plan (mtr_engine* e, uint64_t N, uint32_t head, uint32_t body_tiles, hipStream_t st)
{
	Plan& pl = e->plan;
	if (pl.valid && pl.n_frames == N && pl.frcnt_in == e->frcnt && pl.body_tiles == body_tiles) return MTR_OK;
	pl.valid = false;
	const PlanCtx ctx = plan_ctx (e);
	Tiling til;
	if (const char* why = plan_tiling (&ctx, N, head, body_tiles, til)) return fail (MTR_ERR_ARG, why);
	const std::vector<uint32_t>& ts = til.ts;
	const std::vector<uint32_t>& ft = til.ft;
	const std::vector<uint32_t>& sg = til.sg;
	const uint32_t n_tiles = til.n_tiles, n_frag = til.n_frag, tail = til.tail, head_tiles = til.head_tiles, n_segs = til.n_segs, left = til.frcnt_out;

	// (with head-room: an LV2 host's blocks see a fragment end in some calls and none in others, and a buffer that grows
	// by one word then is a hipMalloc — 0.3 ms — in the audio thread)
	// (a buffer that grows is freed first, and hipFree waits for the device: no deferred gate still reads it)
	if (e->tile_power[0].reserve ((size_t) e->cfg.n_streams * std::max<uint32_t> (n_tiles, 16))
	    || e->tile_power[1].reserve ((size_t) e->cfg.n_streams * std::max<uint32_t> (n_tiles, 16))
	    || e->frag_power.reserve ((size_t) e->cfg.n_streams * std::max<uint32_t> (n_frag, 16)))
		return fail (MTR_ERR_NOMEM, "hipMalloc plan buffers");
	// the next slot of the ring; its previous contents were last read PLAN_SLOTS plans ago
	const int slot = (e->plan_cur + 1) % PLAN_SLOTS;
	PlanSlot& ps = e->plan_slot[slot];
	if (ps.pending) { HIPCHK (hipEventSynchronize (ps.done)); ps.pending = false; }
	const uint32_t tseg[4] = { 0, head_tiles, head_tiles + body_tiles, n_tiles };
	const size_t words = ts.size () + sg.size () + ft.size () + 4;
	if (ps.dev.reserve (std::max<size_t> (words, 256)) || ps.pin.reserve (std::max<size_t> (words, 256))) return fail (MTR_ERR_NOMEM, "plan slot");
	if (!ps.done) HIPCHK (hipEventCreateWithFlags (&ps.done, hipEventDisableTiming));
	memcpy (ps.pin.p, ts.data (), ts.size () * 4);
	memcpy (ps.pin.p + ts.size (), sg.data (), sg.size () * 4);
	memcpy (ps.pin.p + ts.size () + sg.size (), ft.data (), ft.size () * 4);
	memcpy (ps.pin.p + ts.size () + sg.size () + ft.size (), tseg, 16);
	HIPCHK (hipMemcpyAsync (ps.dev.p, ps.pin.p, words * 4, hipMemcpyHostToDevice, st));
	// (the slot is busy from here on: its `done` event is recorded behind the plan's last reader, k_gate — or by plan_abort ()
	// if a launch behind this copy fails, so that the slot is never handed out again under a pending upload: ADVICE r2.
	// One hipEventRecord per call, not two: it is 2-3 us of an LV2 run ().)
	e->plan_cur = slot;
	e->tile_start = ps.dev.p; e->seg_tile = ps.dev.p + ts.size (); e->frag_tile = ps.dev.p + ts.size () + sg.size ();
	e->head_seg = e->frag_tile + ft.size ();
	e->tail_seg = e->head_seg + 2;

	pl.n_frames = N; pl.frcnt_in = e->frcnt; pl.frcnt_out = left;
	pl.n_tiles = n_tiles; pl.n_frag = n_frag; pl.n_segs = n_segs; pl.tail_tile = tail; pl.body_tiles = body_tiles; pl.head_tiles = head_tiles;
	const uint32_t maxlen = til.maxlen;
	// + look-ahead frames of the FIR register tile + 4 slots for the carried K-filter state (layout 3)
	pl.buf_slots = (maxlen + 48 + 13 + 4 + 127) / 128 * 128;
	pl.kw_slots = (maxlen + 1 + 127) / 128 * 128;               // k_kw: the tile + one frame of alignment slack
	pl.valid = true;
	return MTR_OK;
}

static hipEvent_t next_event (mtr_engine* e, size_t idx)
{
	while (e->ev.size () <= idx) {
		hipEvent_t v;
		if (hipEventCreate (&v) != hipSuccess) return nullptr;
		e->ev.push_back (v);
	}
	return e->ev[idx];
}

int mtr_engine_process_device (mtr_engine* e, const float* d_audio, uint64_t n_frames,
                               uint64_t stride, void* hip_stream)
{
	if (!e || !d_audio) return fail (MTR_ERR_ARG, "mtr_engine_process_device: null argument");
	if (n_frames == 0) return MTR_OK;
	if (stride < n_frames) return fail (MTR_ERR_ARG, "stream_stride_frames < n_frames");
	HIPCHK (hipSetDevice (e->cfg.device));
	hipStream_t st = (hipStream_t) hip_stream;
	// every per-meter limit is checked before anything is launched or any host-side counter moves: a call that
	// returns an error has not advanced the engine
	if ((e->cfg.meters & (MTR_METER_BITSTATS | MTR_METER_SIGDIST)) && n_frames >= 0x7fffffffull)
		return fail (MTR_ERR_ARG, "BITSTATS / SIGDIST: n_frames per call must be < 2^31 - 1");
	if ((e->cfg.meters & MTR_METER_KMETER) && n_frames >= 0x7fffffffull)
		return fail (MTR_ERR_ARG, "KMETER: n_frames per call must be < 2^31 - 1 (the reference's int n)");
	if ((e->cfg.meters & MTR_METER_TPBALLIST) && n_frames >= 0x7ffff000ull)
		return fail (MTR_ERR_ARG, "TPBALLIST: n_frames per call must be < 2^31 - 4096");
	if ((e->cfg.meters & (MTR_METER_EBU | MTR_METER_TRUEPEAK)) && n_frames >= 0xFFFFFFFFull)
		return fail (MTR_ERR_ARG, "n_frames per call must be < 2^32 - 1");
	if (st != e->last_stream && e->queued) {
		// the caller moved to another stream: order it behind what the previous one still has to do
		if (!e->xs_event) HIPCHK (hipEventCreateWithFlags (&e->xs_event, hipEventDisableTiming));
		HIPCHK (hipEventRecord (e->xs_event, e->last_stream));
		HIPCHK (hipStreamWaitEvent (st, e->xs_event, 0));
	}
	e->last_stream = st;
	e->queued = true;
	e->snap_valid = false;
	e->advanced = true;
	const uint32_t S = e->v_cnt ? e->v_cnt : e->cfg.n_streams;     // the streams of this call's view ...
	const size_t vo = e->v_cnt ? e->v_off : 0;                       // ... and where they start in every per-stream array
	const bool ebu = e->cfg.meters & MTR_METER_EBU, tp = e->cfg.meters & MTR_METER_TRUEPEAK;
	const bool bank = e->cfg.meters & MTR_METER_SPECTR30;

	const bool tm = e->timing && e->timed_calls < 4096;
	const size_t ev0 = (size_t) e->timed_calls * EV_PER_CALL;
	if (tm) { hipEvent_t v = next_event (e, ev0); if (v) HIPCHK (hipEventRecord (v, st)); }

	// The tail of this call (k_gate; the job's reduction if mtr_engine_reduce follows) on the side stream?  Auto: a batch whose
	// whole fragments go through k_seg, in an engine that meters nothing else — measured (profiles/r06_tail.md): beside k_seg (issue-
	// bound, one wave per SIMD, registers and LDS to spare) the step is 0.04 - 1.4 % shorter than with the gate in front of it; beside k_kw (HBM-bound,
	// eight waves per CU) it costs 8 % MORE; behind k_bank it would start exactly when the next k_seg does; the chunks of a host
	// call are link-bound anyway.
	SegPlan sp;
	if (ebu || tp) { const PlanCtx pctx = plan_ctx (e); sp = seg_plan (&pctx, d_audio, n_frames, stride); }
	const bool only_fused = (e->cfg.meters & ~(uint32_t) (MTR_METER_EBU | MTR_METER_TRUEPEAK)) == 0;
	// (and a batch of thousands of streams: the gate's serial time grows with the streams, what deferring it costs does not — at 1024 streams x 60 s
	// the serial order is 0.5 % FASTER, at 8192 x 10 s the deferred one by 0.4 - 1.4 % across boxes)
	const bool defer = (ebu || tp) && (e->tail_mode == 2 || (e->tail_mode == 0 && e->v_cnt == 0 && sp.use && only_fused && S >= TAIL_AUTO_STREAMS
	                                                          && (uint64_t) S * n_frames >= TAIL_AUTO_FRAMES));
	if (defer) { const int trc = tail_setup (e); if (trc) return trc; }
	else if (ebu || tp) { const int jrc = join_tail (e, st); if (jrc) return jrc; }   // a serial gate follows the deferred ones
	e->last_deferred = defer;
	bool fold_in_history = false;

	if (ebu || tp) {
		int rc = build_plan (e, n_frames, sp.use ? sp.head : 0, sp.use ? sp.tiles : 0, st);
		if (rc) return rc;
		const Plan& pl = e->plan;
		mtr_fused_args fa;
		fa.audio = d_audio; fa.stride = stride;
		fa.hist = e->fir_hist[e->hist_cur].p + vo * MTR_FIR_HALO * 2;
		fa.tile_start = e->tile_start; fa.seg_tile = e->seg_tile; fa.scan_m = e->scan_m.p;
		// (deferred: the other tile_power buffer than the previous call's, whose gate may still be reading; the gate that read
		// this one two calls ago must be through — it has been for a whole call)
		const int tb = defer ? (e->tp_cur ^ 1) : 0;
		if (defer && e->gate_pending[tb]) { HIPCHK (hipStreamWaitEvent (st, e->ev_gate[tb], 0)); e->gate_pending[tb] = false; }
		e->tp_cur = tb;
		fa.state = e->state.p + vo; fa.tile_power = e->tile_power[tb].p + vo * pl.n_tiles;
		fa.n_streams = S; fa.n_segs = pl.n_segs; fa.n_tiles = pl.n_tiles;
		fa.warm_tiles = (uint32_t) std::ceil (MTR_WARM_SEC * e->cfg.sample_rate / (float) (64 * e->run));
		fa.a0 = e->kw[0]; fa.a1 = e->kw[1]; fa.a2 = e->kw[2]; fa.b1 = e->kw[3]; fa.b2 = e->kw[4];
		fa.c3 = e->kw[5]; fa.c4 = e->kw[6];
		fa.gain_l = 1.0f; fa.gain_r = 1.0f;                  // _chan_gain[0..1], ebu_r128_proc.cc:29
		fa.n_frames = n_frames;
		fa.buf_slots = e->layout >= 4 ? pl.kw_slots : pl.buf_slots;
		fa.mfma_a = e->m16_a.p;
		fa.fir_form = e->cfg.tune_fir;
		fa.rotate = e->layout == 3;
		fa.prune = e->cfg.tune_prune > 2 ? 2 : (int) e->cfg.tune_prune;
		fa.prune_stats = e->prune_cnt.p;
		int lrc = 0;
		if (sp.use) {
			// the batch path: whole fragments through k_seg; the rest of an open fragment in front of them and what is left of
			// the call behind them (less than a fragment each) through k_kwtp16, each as ONE segment that picks the K-filter
			// state up where its predecessor in the stream left it
			if (pl.head_tiles) {
				fa.seg_tile = e->head_seg; fa.n_segs = 1;
				lrc = mtr_launch_kwtp16 (e->run, ebu, fa, S, st);
			}
			mtr_seg_args sa;
			sa.audio = d_audio; sa.stride = stride; sa.hist = fa.hist; sa.state = fa.state; sa.tile_power = fa.tile_power;
			sa.head = sp.head; sa.tile0 = pl.head_tiles;
			sa.mfma_a = e->m16_a.p;
			sa.n_streams = S; sa.n_segs = sp.n_segs; sa.n_tiles = pl.n_tiles; sa.tile_frames = e->fragm;
			sa.seg_base = sp.base; sa.seg_rem = sp.rem; sa.n_main = sp.n_main; sa.warm_steps = sp.warm_steps;
			sa.p0_end = (int64_t) n_frames - 24 - (int64_t) sp.head;
			sa.a0 = fa.a0; sa.a1 = fa.a1; sa.a2 = fa.a2; sa.b1 = fa.b1; sa.b2 = fa.b2; sa.c3 = fa.c3; sa.c4 = fa.c4;
			sa.gain_l = fa.gain_l; sa.gain_r = fa.gain_r;
			const uint64_t units = (uint64_t) S * sp.n_segs;
			if (!lrc) lrc = mtr_launch_seg (ebu, sa, (uint32_t) ((units + 63) / 64), st);
			if (!lrc && pl.n_tiles > pl.head_tiles + sp.tiles) {
				fa.seg_tile = e->tail_seg; fa.n_segs = 1;
				lrc = mtr_launch_kwtp16 (e->run, ebu, fa, S, st);
			}
			e->seg_calls += 1; e->seg_frames += (uint64_t) sp.tiles * e->fragm;
		} else {
			lrc = e->layout == 6 ? mtr_launch_kwtp16 (e->run, ebu, fa, S * pl.n_segs, st)
			    : e->layout == 4 ? mtr_launch_kw (e->run, fa, S * pl.n_segs, st)
			                     : mtr_launch_fused2 (e->run, ebu, tp, fa, S * pl.n_segs, st);
		}
		if (lrc) { plan_abort (e, st); return fail (MTR_ERR_HIP, "k_fused launch", hipGetLastError ()); }
		if (tm) { hipEvent_t v = next_event (e, ev0 + 1); if (v) HIPCHK (hipEventRecord (v, st)); }

		hipStream_t gst = st;                                          // the stream the gate runs on
		if (defer) {
			gst = e->tail_stream;
			HIPCHK (hipEventRecord (e->ev_fused, st));
			HIPCHK (hipStreamWaitEvent (gst, e->ev_fused, 0));
			// The gate becomes runnable at the very moment the NEXT call's fused kernel does (both wait for this call's), and
			// its 8192 workgroups would flood the CUs while k_seg's 1024 one-wave workgroups are being placed, one per SIMD:
			// measured, k_seg then takes 15.4 ms instead of 9.5 (profiles/r06_tail.md) — the placement of a persistent kernel
			// is for good.  So the side stream first idles for tail_delay_us: by then k_seg is resident (its dispatch takes
			// ~10 us) and the gate's waves (72 VGPRs) fill in beside it (344 of 512).  Off the critical path by construction.
			if (e->tail_delay_us && mtr_launch_delay (e->tail_delay_us, gst)) { plan_abort (e, st); return fail (MTR_ERR_HIP, "k_delay launch"); }
			e->tail_pending = true;
			e->deferred_calls++;
			fold_in_history = tp;
		}
		if (tm) { hipEvent_t v = next_event (e, ev0 + 2); if (v) HIPCHK (hipEventRecord (v, gst)); }
		mtr_gate_args ga;
		ga.state = e->state.p + vo; ga.hist = e->hist.p + vo * 2 * MTR_HIST_LEN; ga.tile_power = fa.tile_power;
		ga.frag_tile = e->frag_tile; ga.frag_power = e->frag_power.p + vo * pl.n_frag; ga.bin_power = e->bin_power.p;
		ga.n_streams = S; ga.n_tiles = ebu ? pl.n_tiles : 0; ga.n_frag = ebu ? pl.n_frag : 0;
		ga.tail_tile = ebu ? pl.tail_tile : 0;
		ga.fragm = (float) e->fragm; ga.integr = e->integr ? 1 : 0;
		ga.max_scratch = e->gate_max.p + vo * 2;
		ga.fold_tp = fold_in_history ? 0 : 1;
		ga.polite_grid = defer ? e->tail_gate_grid : 0;
		if (mtr_launch_gate (ga, gst)) { plan_abort (e, gst); return fail (MTR_ERR_HIP, "k_gate launch"); }
		if (tm) { hipEvent_t v = next_event (e, ev0 + 3); if (v) HIPCHK (hipEventRecord (v, gst)); }
		{
			PlanSlot& ps = e->plan_slot[e->plan_cur];                 // k_gate is the plan's last reader
			HIPCHK (hipEventRecord (ps.done, gst));
			ps.pending = true;
		}
		if (defer) { HIPCHK (hipEventRecord (e->ev_gate[tb], gst)); e->gate_pending[tb] = true; }
		e->last_n_frag = ga.n_frag;
		e->frcnt = pl.frcnt_out;
	} else if (tm) {
		for (int k = 1; k <= 3; ++k) { hipEvent_t v = next_event (e, ev0 + k); if (v) HIPCHK (hipEventRecord (v, st)); }
	}
	if (tm) { hipEvent_t v = next_event (e, ev0 + 4); if (v) HIPCHK (hipEventRecord (v, st)); }

	if (bank) {
		mtr_bank_args ba;
		ba.audio = d_audio; ba.stride = stride; ba.n_frames = n_frames;
		ba.coef = e->bank_coef.p; ba.z = e->bank_z.p + vo * MTR_NBANDS * 12; ba.val = e->bank_val.p + vo * MTR_NBANDS; ba.mx = e->bank_max.p + vo * MTR_NBANDS;
		ba.ac_in = e->bank_ac[e->bank_ac_cur].p + vo; ba.ac_out = e->bank_ac[e->bank_ac_cur ^ 1].p + vo;
		ba.n_streams = S; ba.n_channels = e->cfg.n_channels; ba.omega = e->omega;
		if (mtr_launch_bank (ba, st)) return fail (MTR_ERR_HIP, "k_bank launch");
		e->bank_ac_cur ^= 1;
	}
	// (the integer tables are int32, as the reference's, which stops counting at 2^31 - 1 samples; the kernels index
	// a call's samples with 32 bits: checked on entry)
	if (e->cfg.meters & MTR_METER_BITSTATS)
		if (mtr_launch_bitstats (d_audio, stride, n_frames, e->bim.p + vo, S, st)) return fail (MTR_ERR_HIP, "k_bitstats launch");
	if (e->cfg.meters & MTR_METER_SIGDIST)
		if (mtr_launch_sigdist (d_audio, stride, n_frames, e->sdh.p + vo, S, st)) return fail (MTR_ERR_HIP, "k_sigdist launch");
	if (e->cfg.meters & MTR_METER_DR14) {
		mtr_dr14_args da;
		da.audio = d_audio; da.stride = stride; da.n_frames = n_frames;
		da.window = (uint64_t) rintf (e->cfg.sample_rate * 3.0f) + 1;       // dr14.c:155, :404
		da.e0 = da.window - e->dr_scnt;
		const uint64_t tot = e->dr_scnt + n_frames;
		da.n_windows = (uint32_t) (tot / da.window);
		da.n_pieces = da.n_windows + (tot % da.window ? 1 : 0);
		da.n_streams = S; da.n_channels = e->cfg.n_channels;
		if (e->dr_sum.reserve ((size_t) e->cfg.n_streams * da.n_pieces * 2) || e->dr_peak.reserve ((size_t) e->cfg.n_streams * da.n_pieces * 2))
			return fail (MTR_ERR_NOMEM, "hipMalloc DR14 pieces");
		da.state = e->dr_state.p + vo; da.hist = e->dr_hist.p + vo * e->cfg.n_channels * MTR_DR_HISTBINS;
		da.piece_sum = e->dr_sum.p + vo * da.n_pieces * 2; da.piece_peak = e->dr_peak.p + vo * da.n_pieces * 2;
		if (mtr_launch_dr14 (da, st)) return fail (MTR_ERR_HIP, "k_dr14 launch");
		e->dr_scnt = tot % da.window;
	}
	if (e->cfg.meters & MTR_METER_KMETER) {
		mtr_kmeter_args ka;
		ka.audio = d_audio; ka.stride = stride; ka.n_groups = n_frames / 4;
		ka.n_streams = S; ka.n_channels = e->cfg.n_channels;
		ka.n_pieces = mtr_kmeter_pieces (ka.n_groups);
		if (e->km_fpp != (uint32_t) n_frames) {                              // kmeterdsp.cc:60-65
			e->km_fall = powf (10.0f, -0.05f * 15.0f * ((float) n_frames / e->cfg.sample_rate));
			e->km_fpp = (uint32_t) n_frames;
		}
		ka.fpp = e->km_fpp; ka.fall = e->km_fall;
		ka.hold = (int32_t) (0.5f * e->cfg.sample_rate + 0.5f);             // :51
		ka.omega = 9.72f / e->cfg.sample_rate;
		memcpy (ka.pw1, e->km_pw1, sizeof (ka.pw1));
		ka.state = e->km_state.p + vo * 2;
		if (e->km_piece.reserve ((size_t) e->cfg.n_streams * std::max<uint32_t> (ka.n_pieces, 1) * 4) || e->km_max.reserve ((size_t) e->cfg.n_streams * std::max<uint32_t> (ka.n_pieces, 1) * 2))
			return fail (MTR_ERR_NOMEM, "hipMalloc KMETER pieces");
		ka.piece_state = e->km_piece.p + vo * ka.n_pieces * 4; ka.piece_max = e->km_max.p + vo * ka.n_pieces * 2;
		if (mtr_launch_kmeter (ka, st)) return fail (MTR_ERR_HIP, "k_kmeter launch");
	}
	const bool tpb = e->cfg.meters & MTR_METER_TPBALLIST;
	if (tpb) {
		mtr_tpb_args ta;
		ta.audio = d_audio; ta.stride = stride; ta.n_frames = n_frames;
		ta.hist = e->fir_hist[e->hist_cur].p + vo * MTR_FIR_HALO * 2; ta.mfma_a = e->m16_a.p; ta.state = e->state.p + vo;
		ta.n_streams = S; ta.n_channels = e->cfg.n_channels;
		ta.w1 = e->tpb_w[0]; ta.w2 = e->tpb_w[1]; ta.w3 = e->tpb_w[2]; ta.g = e->tpb_w[3];
		if (mtr_launch_tpb (ta, st)) return fail (MTR_ERR_HIP, "k_tpb launch");
	}
	if (tp || tpb) {
		// the 47 frames before the next call; after every consumer of the current history
		// (deferred: the fold of this call's peaks rides here — behind the reduction of the previous call, which reads the holds)
		if (fold_in_history && e->red_pending) { HIPCHK (hipStreamWaitEvent (st, e->ev_red, 0)); e->red_pending = false; }
		const int hrc = e->cfg.n_channels == 2
			? mtr_launch_history (d_audio, stride, n_frames, e->fir_hist[e->hist_cur].p + vo * MTR_FIR_HALO * 2, e->fir_hist[e->hist_cur ^ 1].p + vo * MTR_FIR_HALO * 2, S,
			                      fold_in_history ? e->state.p + vo : nullptr, st)
			: mtr_launch_history_mono (d_audio, stride, n_frames, e->fir_hist[e->hist_cur].p + vo * MTR_FIR_HALO * 2, e->fir_hist[e->hist_cur ^ 1].p + vo * MTR_FIR_HALO * 2, S, st);
		if (hrc) return fail (MTR_ERR_HIP, "k_history launch");
		e->hist_cur ^= 1;
	}
	if (tm) {
		hipEvent_t v = next_event (e, ev0 + 5);
		if (v) { HIPCHK (hipEventRecord (v, st)); e->timed_calls++; }      // (a call without all of its events is not a timed call)
	}
	return MTR_OK;
}

static int host_stream (mtr_engine* e, hipStream_t* st)
{
	if (!e->own_stream) HIPCHK (hipStreamCreateWithFlags (&e->own_stream, hipStreamNonBlocking));
	*st = e->own_stream;
	return MTR_OK;
}

// One view of the batch through mtr_engine_process_device; the host-side cursors move only with the last one.
static int process_view (mtr_engine* e, const float* d_audio, uint64_t n_frames, uint64_t stride, hipStream_t st,
                         uint32_t off, uint32_t cnt, bool last)
{
	const uint32_t frcnt = e->frcnt;
	const int hist_cur = e->hist_cur, ac_cur = e->bank_ac_cur;
	const uint64_t dr_scnt = e->dr_scnt, seg_calls = e->seg_calls, seg_frames = e->seg_frames;
	e->v_off = off; e->v_cnt = cnt;
	const int rc = mtr_engine_process_device (e, d_audio, n_frames, stride, st);
	e->v_off = 0; e->v_cnt = 0;
	if (rc || !last) {
		e->frcnt = frcnt; e->hist_cur = hist_cur; e->bank_ac_cur = ac_cur; e->dr_scnt = dr_scnt;
		e->seg_calls = seg_calls; e->seg_frames = seg_frames;
	}
	return rc;
}

int mtr_engine_set_host_chunk_bytes (mtr_engine* e, uint64_t bytes)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	e->host_chunk_bytes = bytes ? (size_t) bytes : (size_t) 256 << 20;
	return MTR_OK;
}

// Host memory in, CHUNKED by streams (results are per stream: chunking is exact, and the routing of a call — which kernel,
// how many time segments — is decided for the whole batch, so every stream sees the arithmetic it would see resident):
// chunk k + 1 crosses the host link on a copy stream while the kernels of chunk k run on the engine's own; two device
// buffers of one chunk each instead of a copy of the whole batch.  End to end the call runs at the link's rate
// (bench.py: extra.end_to_end_host).
int mtr_engine_process_host (mtr_engine* e, const float* h_audio, uint64_t n_frames, uint64_t stride)
{
	if (!e || !h_audio) return fail (MTR_ERR_ARG, "mtr_engine_process_host: null argument");
	if (n_frames == 0) return MTR_OK;
	if (stride < n_frames) return fail (MTR_ERR_ARG, "stream_stride_frames < n_frames");
	HIPCHK (hipSetDevice (e->cfg.device));
	const size_t C = e->cfg.n_channels;
	const uint32_t S = e->cfg.n_streams;
	// (streams start on 16 bytes in the staging buffers — an even stride of stereo frames, a multiple of four mono ones — so that
	// every layout can take the call and k_tpb's LDS-DMA its source)
	const uint64_t dstride = C == 1 ? (n_frames + 3) & ~(uint64_t) 3 : (n_frames + 1) & ~(uint64_t) 1;
	const size_t row = (size_t) dstride * C;                                  // floats per staged stream
	uint32_t cs = (uint32_t) std::min<uint64_t> (S, std::max<uint64_t> (1, e->host_chunk_bytes / (row * sizeof (float))));
	const uint32_t n_chunks = (S + cs - 1) / cs;
	cs = (S + n_chunks - 1) / n_chunks;                                         // even chunks
	const size_t buf_floats = ((size_t) cs * row + 63) & ~(size_t) 63;         // the second buffer starts on 256 bytes
	hipStream_t st;
	int rc = host_stream (e, &st);
	if (rc) return rc;
	if (!e->copy_stream) HIPCHK (hipStreamCreateWithFlags (&e->copy_stream, hipStreamNonBlocking));
	for (int b = 0; b < 2; ++b) {
		if (!e->ev_copied[b]) HIPCHK (hipEventCreateWithFlags (&e->ev_copied[b], hipEventDisableTiming));
		if (!e->ev_computed[b]) HIPCHK (hipEventCreateWithFlags (&e->ev_computed[b], hipEventDisableTiming));
	}
	// the staging buffers may still be read by the previous call (on whatever stream that ran)
	HIPCHK (hipStreamSynchronize (e->last_stream));
	if (e->stage.reserve (buf_floats * (n_chunks > 1 ? 2 : 1))) return fail (MTR_ERR_NOMEM, "hipMalloc staging buffers");
	// (every exit behind the first copy goes through ONE place that waits for the copy stream: the source is pageable caller
	// memory and the copies are truly asynchronous — the caller may free or reuse it as soon as we return, error or not)
	hipError_t he = hipSuccess;
	const char* what = nullptr;
#define HOSTCHK(call) do { he = (call); if (he != hipSuccess) { what = #call; goto done; } } while (0)
	for (uint32_t k = 0, off = 0; k < n_chunks; ++k, off += cs) {
		const uint32_t cnt = std::min (cs, S - off);
		const int b = (int) (k & 1);
		float* const dst = e->stage.p + (size_t) b * buf_floats;
		if (k >= 2) HOSTCHK (hipStreamWaitEvent (e->copy_stream, e->ev_computed[b], 0));   // the kernels of chunk k - 2 have read this buffer
		HOSTCHK (hipMemcpy2DAsync (dst, row * sizeof (float), h_audio + (size_t) off * stride * C, stride * C * sizeof (float),
		                           n_frames * C * sizeof (float), cnt, hipMemcpyHostToDevice, e->copy_stream));
		HOSTCHK (hipEventRecord (e->ev_copied[b], e->copy_stream));
		HOSTCHK (hipStreamWaitEvent (st, e->ev_copied[b], 0));
		rc = process_view (e, dst, n_frames, dstride, st, off, cnt, k + 1 == n_chunks);
		if (rc) goto done;
		HOSTCHK (hipEventRecord (e->ev_computed[b], st));
	}
#undef HOSTCHK
done:
	{
		// wait for the copies (not for the kernels)
		const hipError_t hs = hipStreamSynchronize (e->copy_stream);
		if (what) return fail (MTR_ERR_HIP, what, he);
		if (rc) return rc;
		if (hs != hipSuccess) return fail (MTR_ERR_HIP, "hipStreamSynchronize (copy stream)", hs);
	}
	return MTR_OK;
}

// One LV2 block: interleave into page-locked memory, one H2D copy, the kernels, one D2H copy of the stream's state
// (and the bank's levels), ONE wait.  No allocation after the first block of a given size.
int mtr_engine_process_planar_host (mtr_engine* e, const float* const* ch, uint32_t n_frames)
{
	if (!e || !ch || !ch[0]) return fail (MTR_ERR_ARG, "mtr_engine_process_planar_host: null argument");
	if (e->cfg.n_streams != 1) return fail (MTR_ERR_ARG, "planar host input is the n_streams == 1 (LV2) path");
	if (n_frames == 0) return MTR_OK;
	const uint32_t C = e->cfg.n_channels;
	if (C == 2 && !ch[1]) return fail (MTR_ERR_ARG, "missing right channel");
	HIPCHK (hipSetDevice (e->cfg.device));
	hipStream_t st;
	int rc = host_stream (e, &st);
	if (rc) return rc;
	if (e->last_stream != st) HIPCHK (hipStreamSynchronize (e->last_stream));   // resets queued before the first block
	const size_t total = (size_t) n_frames * C;
	if (e->pin_in.n < total || e->stage.n < total) {
		HIPCHK (hipStreamSynchronize (st));
		if (e->pin_in.reserve (total) || e->stage.reserve (total)) return fail (MTR_ERR_NOMEM, "staging buffers");
	}
	if (e->pin_state.reserve (1) || e->pin_bank.reserve (2 * MTR_NBANDS)) return fail (MTR_ERR_NOMEM, "hipHostMalloc snapshot");
	float* const il = e->pin_in.p;                 // free: the previous block ended with a wait
	if (C == 2) for (uint32_t i = 0; i < n_frames; ++i) { il[2 * i] = ch[0][i]; il[2 * i + 1] = ch[1][i]; }
	else        memcpy (il, ch[0], (size_t) n_frames * sizeof (float));
	HIPCHK (hipMemcpyAsync (e->stage.p, il, total * sizeof (float), hipMemcpyHostToDevice, st));
	rc = mtr_engine_process_device (e, e->stage.p, n_frames, n_frames, st);
	if (rc) return rc;
	{ const int jrc = join_tail (e, st); if (jrc) return jrc; }     // (a deferred gate — tail mode 2 only, at this size — writes the state copied next)
	HIPCHK (hipMemcpyAsync (e->pin_state.p, e->state.p, sizeof (mtr_stream_state), hipMemcpyDeviceToHost, st));
	if (e->cfg.meters & MTR_METER_SPECTR30) {
		HIPCHK (hipMemcpyAsync (e->pin_bank.p, e->bank_val.p, MTR_NBANDS * sizeof (float), hipMemcpyDeviceToHost, st));
		HIPCHK (hipMemcpyAsync (e->pin_bank.p + MTR_NBANDS, e->bank_max.p, MTR_NBANDS * sizeof (float), hipMemcpyDeviceToHost, st));
	}
	HIPCHK (hipStreamSynchronize (st));
	e->snap_valid = true;
	return MTR_OK;
}

int mtr_engine_prepare_host (mtr_engine* e, uint32_t max_block_frames)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	if (e->cfg.n_streams != 1) return fail (MTR_ERR_ARG, "mtr_engine_prepare_host is for the n_streams == 1 (LV2) path");
	if (max_block_frames == 0) return MTR_OK;
	// one silent block of the largest size: page-locked staging, device buffers, the engine's stream and every kernel's
	// code object exist afterwards (a first launch loads the module: milliseconds); then back to the state of a new engine
	std::vector<float> zeros (max_block_frames, 0.f);
	const float* ch[2] = { zeros.data (), zeros.data () };
	int rc = mtr_engine_process_planar_host (e, ch, max_block_frames);
	if (rc) return rc;
	return mtr_engine_reset (e);
}

int mtr_engine_sync (mtr_engine* e)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	HIPCHK (hipSetDevice (e->cfg.device));
	return sync_all (e);
}

int mtr_engine_set_deferred_tail (mtr_engine* e, int mode)
{
	if (!e || mode < 0 || mode > 2) return fail (MTR_ERR_ARG, "mtr_engine_set_deferred_tail: mode must be 0 (auto), 1 (never) or 2 (always)");
	e->tail_mode = mode;
	return MTR_OK;
}

int mtr_engine_join (mtr_engine* e, void* hip_stream)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	HIPCHK (hipSetDevice (e->cfg.device));
	return join_tail (e, (hipStream_t) hip_stream);
}

int mtr_engine_deferred_stats (mtr_engine* e, uint64_t* calls)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	if (calls) *calls = e->deferred_calls;
	return MTR_OK;
}

static int check_range (mtr_engine* e, uint32_t first, uint32_t count)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	if ((uint64_t) first + count > e->cfg.n_streams) return fail (MTR_ERR_ARG, "stream range out of bounds");
	return MTR_OK;
}

int mtr_engine_results (mtr_engine* e, uint32_t first, uint32_t count, mtr_stream_result* out)
{
	int rc = check_range (e, first, count);
	if (rc) return rc;
	if (!out) return fail (MTR_ERR_ARG, "null output");
	if (count == 0) return MTR_OK;
	// (the one-stream snapshot path — an LV2 run () — touches no heap: the state came back with the block's own wait)
	std::vector<mtr_stream_state> hv;
	const mtr_stream_state* h = nullptr;
	if (e->snap_valid && e->cfg.n_streams == 1) {
		h = e->pin_state.p;
	} else {
		rc = mtr_engine_sync (e);
		if (rc) return rc;
		hv.resize (count);
		HIPCHK (hipMemcpy (hv.data (), e->state.p + first, count * sizeof (mtr_stream_state), hipMemcpyDeviceToHost));
		h = hv.data ();
	}
	for (uint32_t i = 0; i < count; ++i) {
		const mtr_stream_state& s = h[i];
		mtr_stream_result& r = out[i];
		r.loudness_M = s.loud_M; r.maxloudn_M = s.max_M; r.loudness_S = s.loud_S; r.maxloudn_S = s.max_S;
		r.integrated = s.integ; r.integ_thr = s.integ_thr;
		r.range_min = s.rmin; r.range_max = s.rmax; r.range_thr = s.rthr;
		r.hist_M_count = s.cnt_M; r.hist_S_count = s.cnt_S;
		for (int c = 0; c < 2; ++c) {
			r.truepeak[c] = s.tp_hold[c]; r.truepeak_call[c] = s.tp_last[c];
			r.tpb_level[c] = s.tpb_m[c]; r.tpb_peak[c] = s.tpb_p[c];
		}
	}
	return MTR_OK;
}

int mtr_engine_histograms (mtr_engine* e, uint32_t first, uint32_t count, int32_t* hm, int32_t* hs)
{
	int rc = check_range (e, first, count);
	if (rc) return rc;
	if (count == 0) return MTR_OK;
	rc = mtr_engine_sync (e);
	if (rc) return rc;
	std::vector<int32_t> h ((size_t) count * 2 * MTR_HIST_LEN);
	HIPCHK (hipMemcpy (h.data (), e->hist.p + (size_t) first * 2 * MTR_HIST_LEN, h.size () * 4, hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < count; ++i) {
		if (hm) memcpy (hm + (size_t) i * MTR_HIST_LEN, &h[(size_t) i * 2 * MTR_HIST_LEN], MTR_HIST_LEN * 4);
		if (hs) memcpy (hs + (size_t) i * MTR_HIST_LEN, &h[(size_t) i * 2 * MTR_HIST_LEN + MTR_HIST_LEN], MTR_HIST_LEN * 4);
	}
	return MTR_OK;
}

int mtr_engine_fragment_powers (mtr_engine* e, uint32_t first, uint32_t count, float* out,
                                uint32_t cap, uint32_t* n_frag)
{
	int rc = check_range (e, first, count);
	if (rc) return rc;
	if (n_frag) *n_frag = e->last_n_frag;
	if (!out || count == 0 || e->last_n_frag == 0) return MTR_OK;
	if (cap < e->last_n_frag) return fail (MTR_ERR_ARG, "capacity_per_stream < n_frag");
	rc = mtr_engine_sync (e);
	if (rc) return rc;
	HIPCHK (hipMemcpy2D (out, (size_t) cap * 4, e->frag_power.p + (size_t) first * e->last_n_frag,
	                     (size_t) e->last_n_frag * 4, (size_t) e->last_n_frag * 4, count, hipMemcpyDeviceToHost));
	return MTR_OK;
}

int mtr_engine_spectrum (mtr_engine* e, uint32_t first, uint32_t count, float* val, float* mx, float* val_db, float* max_db)
{
	int rc = check_range (e, first, count);
	if (rc) return rc;
	if (!(e->cfg.meters & MTR_METER_SPECTR30)) return fail (MTR_ERR_ARG, "no SPECTR30 in this engine");
	if (count == 0) return MTR_OK;
	const size_t n = (size_t) count * MTR_NBANDS;
	std::vector<float> hv;
	const float* v = nullptr; const float* m = nullptr;
	if (e->snap_valid && e->cfg.n_streams == 1) {
		v = e->pin_bank.p; m = e->pin_bank.p + MTR_NBANDS;            // came back with the block's own wait: no heap, no copy
	} else {
		rc = mtr_engine_sync (e);
		if (rc) return rc;
		hv.resize (2 * n);
		HIPCHK (hipMemcpy (hv.data (), e->bank_val.p + (size_t) first * MTR_NBANDS, n * 4, hipMemcpyDeviceToHost));
		HIPCHK (hipMemcpy (hv.data () + n, e->bank_max.p + (size_t) first * MTR_NBANDS, n * 4, hipMemcpyDeviceToHost));
		v = hv.data (); m = hv.data () + n;
	}
	for (size_t i = 0; i < n; ++i) {
		// spectrumlv2.c:240-247.  The stored val carries the +1e-20f of :237; above the -100 dB floor
		// (val > 5e-11) that addition does not change the float, so the port value is unaffected.
		const float vs = sqrtf (2. * v[i]);
		const float ms = sqrtf (2. * m[i]);
		if (val) val[i] = v[i];
		if (mx) mx[i] = m[i];
		if (val_db) val_db[i] = vs > .00001f ? 20.0 * log10f (vs) : -100.0;
		if (max_db) max_db[i] = ms > .00001f ? 20.0 * log10f (ms) : -100.0;
	}
	return MTR_OK;
}

int mtr_engine_aggregate_device (mtr_engine* e, int32_t* d_hist, float* d_max, void* hip_stream)
{
	if (!e || !d_hist || !d_max) return fail (MTR_ERR_ARG, "mtr_engine_aggregate_device: null argument");
	HIPCHK (hipSetDevice (e->cfg.device));
	{ const int jrc = join_tail (e, (hipStream_t) hip_stream); if (jrc) return jrc; }   // behind the deferred gates: it reads what they write
	if (mtr_launch_aggregate (e->state.p, e->hist.p, e->cfg.n_streams, d_hist, d_max, hip_stream))
		return fail (MTR_ERR_HIP, "k_aggregate launch");
	return MTR_OK;
}

// ---- the one collective of a multi-GPU job: RCCL behind the C ABI ---------------------------------------------------
struct mtr_comm {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = 0;
	bool nonblocking = false;          // built by mtr_comm_init_timeout: every call on it is polled to a deadline
	uint32_t timeout_ms = 0;
	hipStream_t probe_stream = nullptr;
	int32_t* probe_buf = nullptr;
};

static int nccl_fail (const char* what, ncclResult_t r)
{
	char buf[256];
	snprintf (buf, sizeof (buf), "%s: %s", what, ncclGetErrorString (r));
	g_err = buf;
	return MTR_ERR_HIP;
}

static double ms_since (std::chrono::steady_clock::time_point t0)
{
	return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t0).count ();
}

// the communicator is beyond repair (a deadline passed, an asynchronous error): abort it; only mtr_comm_destroy may follow
static void comm_abort (mtr_comm* c)
{
	if (c->comm) (void) ncclCommAbort (c->comm);
	c->comm = nullptr;
}

// A call on a non-blocking communicator returned ncclInProgress: poll its state until RCCL has taken the call in, it has
// failed, or the deadline passes (then the communicator is aborted).  `t0` = when the call was made.
static int comm_wait (mtr_comm* c, uint32_t timeout_ms, std::chrono::steady_clock::time_point t0, const char* what)
{
	for (;;) {
		ncclResult_t state = ncclSuccess;
		const ncclResult_t r = ncclCommGetAsyncError (c->comm, &state);
		if (r != ncclSuccess) { comm_abort (c); return nccl_fail (what, r); }
		if (state == ncclSuccess) return MTR_OK;
		if (state != ncclInProgress) { comm_abort (c); return nccl_fail (what, state); }
		if (timeout_ms && ms_since (t0) >= (double) timeout_ms) {
			comm_abort (c);
			char buf[160];
			snprintf (buf, sizeof (buf), "%s: no answer from RCCL within %u ms (communicator aborted)", what, timeout_ms);
			return fail (MTR_ERR_TIMEOUT, buf);
		}
		std::this_thread::sleep_for (std::chrono::microseconds (200));
	}
}

int mtr_rccl_version (void)
{
	int v = 0;
	const ncclResult_t r = ncclGetVersion (&v);
	if (r != ncclSuccess) return nccl_fail ("ncclGetVersion", r);
	return v;
}

int mtr_comm_unique_id (void* id128)
{
	static_assert (sizeof (ncclUniqueId) == MTR_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
	if (!id128) return fail (MTR_ERR_ARG, "mtr_comm_unique_id: null argument");
	ncclUniqueId id;
	const ncclResult_t r = ncclGetUniqueId (&id);
	if (r != ncclSuccess) return nccl_fail ("ncclGetUniqueId", r);
	memcpy (id128, &id, sizeof (id));
	return MTR_OK;
}

int mtr_comm_init_timeout (mtr_comm** out, int rank, int world, const void* id128, int device, uint32_t timeout_ms, float* init_ms)
{
	if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail (MTR_ERR_ARG, "mtr_comm_init: bad argument");
	*out = nullptr;
	if (init_ms) *init_ms = 0.f;
	int ndev = 0;
	if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0) return fail (MTR_ERR_NODEVICE, "no HIP device");
	if (device < 0 || device >= ndev) return fail (MTR_ERR_ARG, "device ordinal out of range");
	HIPCHK (hipSetDevice (device));
	mtr_comm* c = new (std::nothrow) mtr_comm ();
	if (!c) return fail (MTR_ERR_NOMEM, "new mtr_comm");
	c->rank = rank; c->world = world; c->device = device;
	c->nonblocking = timeout_ms != 0; c->timeout_ms = timeout_ms;
	ncclUniqueId id;
	memcpy (&id, id128, sizeof (id));
	const auto t0 = std::chrono::steady_clock::now ();
	int rc = MTR_OK;
	if (!c->nonblocking) {
		const ncclResult_t r = ncclCommInitRank (&c->comm, world, id, rank);
		if (r != ncclSuccess) { c->comm = nullptr; rc = nccl_fail ("ncclCommInitRank", r); }
	} else {
		// the header may be newer than the RCCL this process runs (torch ships its own): never claim a newer version than the library's
		ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
		int lib_version = 0;
		if (ncclGetVersion (&lib_version) == ncclSuccess && lib_version > 0 && (unsigned) lib_version < cfg.version) cfg.version = (unsigned) lib_version;
		cfg.blocking = 0;
		const ncclResult_t r = ncclCommInitRankConfig (&c->comm, world, id, rank, &cfg);
		if (r != ncclSuccess && r != ncclInProgress) { c->comm = nullptr; rc = nccl_fail ("ncclCommInitRankConfig", r); }
		else if (!c->comm) rc = fail (MTR_ERR_HIP, "ncclCommInitRankConfig returned no communicator");
		else rc = comm_wait (c, timeout_ms, t0, "ncclCommInitRankConfig");
	}
	if (rc == MTR_OK) {
		if (hipStreamCreateWithFlags (&c->probe_stream, hipStreamNonBlocking) != hipSuccess || hipMalloc ((void**) &c->probe_buf, sizeof (int32_t)) != hipSuccess)
			rc = fail (MTR_ERR_HIP, "mtr_comm_init: probe stream / buffer");
	}
	if (init_ms) *init_ms = (float) ms_since (t0);
	if (rc != MTR_OK) { mtr_comm_destroy (c); return rc; }
	*out = c;
	return MTR_OK;
}

int mtr_comm_init (mtr_comm** out, int rank, int world, const void* id128, int device)
{
	return mtr_comm_init_timeout (out, rank, world, id128, device, 0, nullptr);
}

int mtr_comm_set_timeout (mtr_comm* c, uint32_t timeout_ms)
{
	if (!c) return fail (MTR_ERR_ARG, "mtr_comm_set_timeout: null communicator");
	if (!c->nonblocking && timeout_ms) return fail (MTR_ERR_ARG, "mtr_comm_set_timeout: a communicator built by mtr_comm_init blocks; use mtr_comm_init_timeout");
	c->timeout_ms = timeout_ms;
	return MTR_OK;
}

void mtr_comm_destroy (mtr_comm* c)
{
	if (!c) return;
	(void) hipSetDevice (c->device);
	if (c->comm && !c->nonblocking) (void) ncclCommDestroy (c->comm);
	else if (c->comm) {
		// a healthy communicator with a deadline: finalise (collectives still queued complete), polled to the deadline, then
		// destroy; abort only if that does not happen in time (abort kills a reduce that is still queued)
		const auto t0 = std::chrono::steady_clock::now ();
		const uint32_t limit = c->timeout_ms ? c->timeout_ms : 10000u;
		ncclResult_t r = ncclCommFinalize (c->comm);
		bool ok = r == ncclSuccess || r == ncclInProgress;
		while (ok) {
			ncclResult_t state = ncclSuccess;
			if (ncclCommGetAsyncError (c->comm, &state) != ncclSuccess || (state != ncclSuccess && state != ncclInProgress)) { ok = false; break; }
			if (state == ncclSuccess) break;
			if (ms_since (t0) >= (double) limit) { ok = false; break; }
			std::this_thread::sleep_for (std::chrono::microseconds (200));
		}
		if (ok) (void) ncclCommDestroy (c->comm); else (void) ncclCommAbort (c->comm);
	}
	c->comm = nullptr;
	if (c->probe_buf) (void) hipFree (c->probe_buf);
	if (c->probe_stream) (void) hipStreamDestroy (c->probe_stream);
	delete c;
}

int mtr_comm_probe (mtr_comm* c, uint32_t timeout_ms, float* ms)
{
	if (ms) *ms = 0.f;
	if (!c) return fail (MTR_ERR_ARG, "mtr_comm_probe: null communicator");
	if (!c->comm) return fail (MTR_ERR_ARG, "mtr_comm_probe: the communicator has been aborted");
	// (every failure from here on leaves the communicator aborted, as the header says)
#define PROBECHK(call) do { hipError_t he_ = (call); if (he_ != hipSuccess) { comm_abort (c); return fail (MTR_ERR_HIP, #call, he_); } } while (0)
	PROBECHK (hipSetDevice (c->device));
	const int32_t one = 1;
	PROBECHK (hipMemcpyAsync (c->probe_buf, &one, sizeof (one), hipMemcpyHostToDevice, c->probe_stream));
	PROBECHK (hipStreamSynchronize (c->probe_stream));
	const auto t0 = std::chrono::steady_clock::now ();
	const ncclResult_t r = ncclAllReduce (c->probe_buf, c->probe_buf, 1, ncclInt32, ncclSum, c->comm, c->probe_stream);
	if (r != ncclSuccess && r != ncclInProgress) { comm_abort (c); return nccl_fail ("ncclAllReduce (probe)", r); }
	if (c->nonblocking) { const int rc = comm_wait (c, timeout_ms, t0, "ncclAllReduce (probe)"); if (rc) return rc; }
	// the collective itself: poll the stream, and the communicator for an asynchronous error (a peer that died)
	for (;;) {
		const hipError_t q = hipStreamQuery (c->probe_stream);
		if (q == hipSuccess) break;
		if (q != hipErrorNotReady) { comm_abort (c); return fail (MTR_ERR_HIP, "hipStreamQuery (probe)", q); }
		ncclResult_t state = ncclSuccess;
		if (ncclCommGetAsyncError (c->comm, &state) == ncclSuccess && state != ncclSuccess && state != ncclInProgress) { comm_abort (c); return nccl_fail ("ncclAllReduce (probe)", state); }
		if (timeout_ms && ms_since (t0) >= (double) timeout_ms) {
			comm_abort (c);
			char buf[160];
			snprintf (buf, sizeof (buf), "mtr_comm_probe: the first all-reduce did not finish within %u ms (communicator aborted)", timeout_ms);
			return fail (MTR_ERR_TIMEOUT, buf);
		}
		std::this_thread::sleep_for (std::chrono::microseconds (100));
	}
	if (ms) *ms = (float) ms_since (t0);
	int32_t got = 0;
	PROBECHK (hipMemcpy (&got, c->probe_buf, sizeof (got), hipMemcpyDeviceToHost));
#undef PROBECHK
	if (got != c->world) {
		char buf[160];
		snprintf (buf, sizeof (buf), "mtr_comm_probe: all-reduce of ones over %d ranks gave %d", c->world, (int) got);
		comm_abort (c);
		return fail (MTR_ERR_HIP, buf);
	}
	return MTR_OK;
}

int mtr_comm_nranks (mtr_comm* c)
{
	if (!c) return fail (MTR_ERR_ARG, "mtr_comm_nranks: null communicator");
	if (!c->comm) return fail (MTR_ERR_ARG, "mtr_comm_nranks: the communicator has been aborted");
	int n = 0;
	const ncclResult_t r = ncclCommCount (c->comm, &n);
	if (r != ncclSuccess) return nccl_fail ("ncclCommCount", r);
	return n;
}

int mtr_comm_device (mtr_comm* c)
{
	if (!c) return fail (MTR_ERR_ARG, "mtr_comm_device: null communicator");
	if (!c->comm) return fail (MTR_ERR_ARG, "mtr_comm_device: the communicator has been aborted");
	int d = -1;
	const ncclResult_t r = ncclCommCuDevice (c->comm, &d);
	if (r != ncclSuccess) return nccl_fail ("ncclCommCuDevice", r);
	return d;
}

int mtr_engine_reduce (mtr_engine* e, mtr_comm* c, int32_t* d_hist, float* d_max, void* hip_stream)
{
	if (!e || !c || !d_hist || !d_max) return fail (MTR_ERR_ARG, "mtr_engine_reduce: null argument");
	if (!c->comm) return fail (MTR_ERR_ARG, "mtr_engine_reduce: the communicator has been aborted");
	if (c->device != e->cfg.device) return fail (MTR_ERR_ARG, "mtr_engine_reduce: engine and communicator sit on different devices");
	hipStream_t st = (hipStream_t) hip_stream;
	const bool deferred = e->last_deferred && e->tail_stream;
	if (deferred) {
		// behind the deferred gate, on the side stream: the aggregate reads what the gate wrote there and what the caller's stream
		// has written up to now (the fold of the call's peaks in k_history); d_hist / d_max are valid after mtr_engine_join /
		// mtr_engine_sync.  The next call's fold waits for ev_red.
		HIPCHK (hipSetDevice (e->cfg.device));
		HIPCHK (hipEventRecord (e->ev_main, st));
		HIPCHK (hipStreamWaitEvent (e->tail_stream, e->ev_main, 0));
		st = e->tail_stream;
		e->tail_pending = true;
		if (mtr_launch_aggregate (e->state.p, e->hist.p, e->cfg.n_streams, d_hist, d_max, st)) return fail (MTR_ERR_HIP, "k_aggregate launch");
	} else {
		const int rc = mtr_engine_aggregate_device (e, d_hist, d_max, hip_stream);
		if (rc) return rc;
	}
	// one group: RCCL launches the sum and the max together (6 KB + 16 B: both are pure latency on xGMI)
	const auto t0 = std::chrono::steady_clock::now ();
	ncclResult_t r = ncclGroupStart ();
	if (r != ncclSuccess) return nccl_fail ("ncclGroupStart", r);
	const ncclResult_t r1 = ncclAllReduce (d_hist, d_hist, 2 * MTR_HIST_LEN, ncclInt32, ncclSum, c->comm, st);
	const ncclResult_t r2 = ncclAllReduce (d_max, d_max, 4, ncclFloat32, ncclMax, c->comm, st);
	r = ncclGroupEnd ();
	if (r1 != ncclSuccess && r1 != ncclInProgress) return nccl_fail ("ncclAllReduce (histograms)", r1);
	if (r2 != ncclSuccess && r2 != ncclInProgress) return nccl_fail ("ncclAllReduce (peaks)", r2);
	if (r == ncclInProgress && c->nonblocking) { const int rc = comm_wait (c, c->timeout_ms, t0, "ncclGroupEnd (mtr_engine_reduce)"); if (rc) return rc; }
	else if (r != ncclSuccess) return nccl_fail ("ncclGroupEnd", r);
	if (deferred) { HIPCHK (hipEventRecord (e->ev_red, st)); e->red_pending = true; }
	return MTR_OK;
}

// ---- per-stream state: checkpoint / resume, re-sharding ----------------------------------------------------------------
} // extern "C"
namespace {

struct StateHeader {
	uint32_t magic, version, header_bytes;
	uint32_t meters, n_channels;
	float    sample_rate;
	uint32_t count, per_stream_bytes, stream_state_bytes;
	// the engine's lock-step cursors
	uint32_t frcnt, integr;
	float    omega;
	uint64_t dr_scnt;
	uint64_t payload_fnv;       // FNV-1a (64 bit) of everything behind the header: a checkpoint file that rotted is refused, not trusted
};
constexpr uint32_t STATE_MAGIC = 0x5352544du;   // "MTRS"
constexpr uint32_t STATE_VERSION = 2;           // 2: + payload_fnv

uint64_t fnv1a64 (const unsigned char* p, size_t n)
{
	uint64_t h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
	return h;
}

struct StateSection { const void* base; size_t elem; };   // a per-stream array: `elem` bytes per stream

// every array a stream carries from call to call, in the blob's order (a function of the configuration alone)
std::vector<StateSection> state_sections (const mtr_engine* e)
{
	std::vector<StateSection> v;
	const uint32_t m = e->cfg.meters;
	v.push_back ({ e->state.p, sizeof (mtr_stream_state) });
	v.push_back ({ e->hist.p, (size_t) 2 * MTR_HIST_LEN * sizeof (int32_t) });
	v.push_back ({ e->fir_hist[e->hist_cur].p, (size_t) MTR_FIR_HALO * 2 * sizeof (float) });
	if (m & MTR_METER_SPECTR30) {
		v.push_back ({ e->bank_z.p, (size_t) MTR_NBANDS * 12 * sizeof (double) });
		v.push_back ({ e->bank_val.p, (size_t) MTR_NBANDS * sizeof (float) });
		v.push_back ({ e->bank_max.p, (size_t) MTR_NBANDS * sizeof (float) });
		v.push_back ({ e->bank_ac[e->bank_ac_cur].p, sizeof (int32_t) });
	}
	if (m & MTR_METER_BITSTATS) v.push_back ({ e->bim.p, sizeof (mtr_bitstats_state) });
	if (m & MTR_METER_SIGDIST) v.push_back ({ e->sdh.p, sizeof (mtr_sigdist_state) });
	if (m & MTR_METER_DR14) {
		v.push_back ({ e->dr_state.p, sizeof (mtr_dr14_state) });
		v.push_back ({ e->dr_hist.p, (size_t) e->cfg.n_channels * MTR_DR_HISTBINS * sizeof (uint32_t) });
	}
	if (m & MTR_METER_KMETER) v.push_back ({ e->km_state.p, 2 * sizeof (mtr_kmeter_state) });
	return v;
}

size_t state_per_stream (const mtr_engine* e)
{
	size_t n = 0;
	for (const StateSection& s : state_sections (e)) n += s.elem;
	return n;
}

}  // namespace
extern "C" {

size_t mtr_engine_state_bytes (const mtr_engine* e, uint32_t count)
{
	if (!e) return 0;
	return sizeof (StateHeader) + (size_t) count * state_per_stream (e);
}

uint32_t mtr_state_blob_count (const void* blob, size_t bytes)
{
	StateHeader h;
	if (!blob || bytes < sizeof (h)) return 0;
	memcpy (&h, blob, sizeof (h));
	if (h.magic != STATE_MAGIC || h.version != STATE_VERSION || h.header_bytes != sizeof (h)) return 0;
	if (bytes < sizeof (h) + (size_t) h.count * h.per_stream_bytes) return 0;
	return h.count;
}

int mtr_engine_state_export (mtr_engine* e, uint32_t first, uint32_t count, void* blob, size_t capacity)
{
	int rc = check_range (e, first, count);
	if (rc) return rc;
	if (!blob) return fail (MTR_ERR_ARG, "mtr_engine_state_export: null blob");
	const size_t need = mtr_engine_state_bytes (e, count);
	if (capacity < need) return fail (MTR_ERR_ARG, "mtr_engine_state_export: capacity < mtr_engine_state_bytes ()");
	rc = mtr_engine_sync (e);
	if (rc) return rc;
	StateHeader h;
	memset (&h, 0, sizeof (h));
	h.magic = STATE_MAGIC; h.version = STATE_VERSION; h.header_bytes = sizeof (h);
	h.meters = e->cfg.meters; h.n_channels = e->cfg.n_channels; h.sample_rate = e->cfg.sample_rate;
	h.count = count; h.per_stream_bytes = (uint32_t) state_per_stream (e); h.stream_state_bytes = sizeof (mtr_stream_state);
	h.frcnt = e->frcnt; h.integr = e->integr ? 1u : 0u; h.omega = e->omega; h.dr_scnt = e->dr_scnt;
	unsigned char* const o0 = static_cast<unsigned char*> (blob) + sizeof (h);
	unsigned char* o = o0;
	for (const StateSection& s : state_sections (e)) {
		if (count) HIPCHK (hipMemcpy (o, static_cast<const unsigned char*> (s.base) + (size_t) first * s.elem, (size_t) count * s.elem, hipMemcpyDeviceToHost));
		o += (size_t) count * s.elem;
	}
	h.payload_fnv = fnv1a64 (o0, (size_t) (o - o0));
	memcpy (blob, &h, sizeof (h));
	return MTR_OK;
}

int mtr_engine_state_import (mtr_engine* e, uint32_t first, const void* blob, size_t bytes)
{
	if (!e || !blob) return fail (MTR_ERR_ARG, "mtr_engine_state_import: null argument");
	StateHeader h;
	if (bytes < sizeof (h)) return fail (MTR_ERR_STATE, "mtr_engine_state_import: not a state blob (too short)");
	memcpy (&h, blob, sizeof (h));
	if (h.magic != STATE_MAGIC) return fail (MTR_ERR_STATE, "mtr_engine_state_import: not a state blob (magic)");
	if (h.version != STATE_VERSION || h.header_bytes != sizeof (h)) return fail (MTR_ERR_STATE, "mtr_engine_state_import: blob of another format version");
	if (h.meters != e->cfg.meters || h.n_channels != e->cfg.n_channels || h.sample_rate != e->cfg.sample_rate)
		return fail (MTR_ERR_STATE, "mtr_engine_state_import: the blob comes from another configuration (meters, channels or sample rate)");
	if (h.stream_state_bytes != sizeof (mtr_stream_state) || h.per_stream_bytes != state_per_stream (e))
		return fail (MTR_ERR_STATE, "mtr_engine_state_import: the blob's per-stream layout is not this build's");
	if (bytes < sizeof (h) + (size_t) h.count * h.per_stream_bytes) return fail (MTR_ERR_STATE, "mtr_engine_state_import: truncated blob");
	int rc = check_range (e, first, h.count);
	if (rc) return rc;
	// the cursors go straight into the tiling of the next call, the payload's ring indices and counters into the kernels:
	// nothing of a blob is trusted before it has been checked
	if (h.frcnt == 0 || h.frcnt > e->fragm) return fail (MTR_ERR_STATE, "mtr_engine_state_import: corrupt blob (frames left in the open fragment)");
	if (h.integr > 1 || !(h.omega > 0.f && h.omega < 1.f)) return fail (MTR_ERR_STATE, "mtr_engine_state_import: corrupt blob (integration flag / bank speed)");
	if (h.dr_scnt > (uint64_t) rintf (e->cfg.sample_rate * 3.0f)) return fail (MTR_ERR_STATE, "mtr_engine_state_import: corrupt blob (open DR-14 window)");
	const unsigned char* const i0 = static_cast<const unsigned char*> (blob) + sizeof (h);
	if (fnv1a64 (i0, (size_t) h.count * h.per_stream_bytes) != h.payload_fnv) return fail (MTR_ERR_STATE, "mtr_engine_state_import: corrupt blob (checksum of the payload)");
	// the streams of an engine advance in lock step: a fresh engine takes the blob's cursors — integration on / off and the bank's
	// speed included, whatever integr_start / spectr_set_speed said before: they are part of where the streams stand — any other
	// must stand at the same ones
	const bool fresh = !e->advanced;
	if (!fresh && (e->frcnt != h.frcnt || e->integr != (h.integr != 0) || e->omega != h.omega || e->dr_scnt != h.dr_scnt))
		return fail (MTR_ERR_STATE, "mtr_engine_state_import: the engine does not stand where the blob's streams do (fragment phase, integration, bank speed or DR-14 window)");
	rc = mtr_engine_sync (e);
	if (rc) return rc;
	e->snap_valid = false;
	const unsigned char* i = i0;
	for (const StateSection& s : state_sections (e)) {
		if (h.count) HIPCHK (hipMemcpy (const_cast<unsigned char*> (static_cast<const unsigned char*> (s.base)) + (size_t) first * s.elem, i, (size_t) h.count * s.elem, hipMemcpyHostToDevice));
		i += (size_t) h.count * s.elem;
	}
	if (fresh) {                                                 // (only now: a failed sync or copy has not moved the engine)
		e->frcnt = h.frcnt; e->integr = h.integr != 0; e->omega = h.omega; e->dr_scnt = h.dr_scnt;
		e->plan.valid = false;
		e->advanced = true;
	}
	return MTR_OK;
}

int mtr_engine_layout (const mtr_engine* e) { return e ? (e->seg_ok ? 7 : e->layout) : MTR_ERR_ARG; }

int mtr_plan_query (const mtr_config* cfg, uint32_t frames_left_in_fragment, uint64_t n_frames, uint32_t n_slots, mtr_plan_info* out)
{
	if (!cfg || !out) return fail (MTR_ERR_ARG, "null argument");
	if (cfg->struct_size != sizeof (mtr_config)) return fail (MTR_ERR_ARG, "mtr_plan_query: bad config (struct_size)");
	if (cfg->n_streams == 0 || !(cfg->sample_rate >= 1000.0f) || n_frames == 0 || n_frames > 0xffffffffull) return fail (MTR_ERR_ARG, "mtr_plan_query: streams, rate or frames out of range");
	PlanCtx c;
	c.cfg = *cfg;
	if (const char* why = resolve_layout (cfg, &c.layout, &c.run, &c.seg_ok)) return fail (MTR_ERR_ARG, why);
	c.fragm = (uint32_t) ((int) cfg->sample_rate / 20);
	c.frcnt = frames_left_in_fragment ? frames_left_in_fragment : c.fragm;
	if (c.frcnt > c.fragm) return fail (MTR_ERR_ARG, "mtr_plan_query: more frames left than a fragment has");
	c.seg_slots = n_slots ? n_slots : 1024;
	memset (out, 0, sizeof (*out));
	const bool fused = cfg->meters & (MTR_METER_EBU | MTR_METER_TRUEPEAK);
	out->layout = c.seg_ok ? 7u : (uint32_t) c.layout;
	if (!fused) { out->frames_left_after = c.frcnt; return MTR_OK; }
	const SegPlan sp = seg_plan (&c, nullptr, n_frames, n_frames);
	Tiling t;
	if (const char* why = plan_tiling (&c, n_frames, sp.use ? sp.head : 0, sp.use ? sp.tiles : 0, t)) return fail (MTR_ERR_ARG, why);
	out->uses_seg = sp.use;
	out->head_frames = sp.use ? sp.head : 0; out->body_fragments = sp.use ? sp.tiles : 0; out->segments = sp.use ? sp.n_segs : 0;
	out->fragments_per_lane = sp.use ? sp.n_main : 0; out->warm_steps = sp.use ? sp.warm_steps : 0;
	out->n_tiles = t.n_tiles; out->head_tiles = t.head_tiles; out->n_fragments_ended = t.n_frag;
	out->kw_segments = sp.use ? 0 : t.n_segs;
	out->frames_left_after = t.frcnt_out;
	return MTR_OK;
}

int mtr_engine_seg_stats (mtr_engine* e, uint64_t* calls, uint64_t* frames)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	if (calls) *calls = e->seg_calls;
	if (frames) *frames = e->seg_frames;
	return MTR_OK;
}

static int drain_prune_counters (mtr_engine* e)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	int rc = mtr_engine_sync (e);
	if (rc) return rc;
	uint32_t h[4];
	HIPCHK (hipMemcpy (h, e->prune_cnt.p, 16, hipMemcpyDeviceToHost));
	HIPCHK (hipMemset (e->prune_cnt.p, 0, 16));      // 32-bit device counters are drained into 64-bit totals
	for (int i = 0; i < 4; ++i) e->prune_tot[i] += h[i];
	return MTR_OK;
}

int mtr_engine_prune_stats (mtr_engine* e, uint64_t* considered, uint64_t* skipped)
{
	int rc = drain_prune_counters (e);
	if (rc) return rc;
	if (considered) *considered = e->prune_tot[0];
	if (skipped) *skipped = e->prune_tot[1];
	return MTR_OK;
}

int mtr_engine_refine_stats (mtr_engine* e, uint64_t* screened, uint64_t* completed)
{
	int rc = drain_prune_counters (e);
	if (rc) return rc;
	if (screened) *screened = e->prune_tot[2];
	if (completed) *completed = e->prune_tot[3];
	return MTR_OK;
}

int mtr_engine_timing_enable (mtr_engine* e, int on)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	e->timing = on != 0;
	e->timed_calls = 0;
	return MTR_OK;
}

int mtr_engine_timing_query (mtr_engine* e, float* ms_fused, float* ms_gate, float* ms_bank, uint32_t* calls)
{
	if (!e) return fail (MTR_ERR_ARG, "null engine");
	int rc = mtr_engine_sync (e);
	if (rc) return rc;
	float f = 0, g = 0, b = 0;
	for (uint32_t i = 0; i < e->timed_calls && (size_t) i * EV_PER_CALL + EV_PER_CALL - 1 < e->ev.size (); ++i) {
		float t;
		hipEvent_t* const v = &e->ev[(size_t) i * EV_PER_CALL];
		if (hipEventElapsedTime (&t, v[0], v[1]) == hipSuccess) f += t;
		if (hipEventElapsedTime (&t, v[2], v[3]) == hipSuccess) g += t;
		if (hipEventElapsedTime (&t, v[4], v[5]) == hipSuccess) b += t;
	}
	if (ms_fused) *ms_fused = f;
	if (ms_gate) *ms_gate = g;
	if (ms_bank) *ms_bank = b;
	if (calls) *calls = e->timed_calls;
	e->timed_calls = 0;
	return MTR_OK;
}

int mtr_engine_timing_calls (mtr_engine* e, float* out, uint32_t cap, uint32_t* calls)
{
	if (!e || (!out && cap)) return fail (MTR_ERR_ARG, "mtr_engine_timing_calls: null argument");
	int rc = mtr_engine_sync (e);
	if (rc) return rc;
	if (calls) *calls = e->timed_calls;
	for (uint32_t i = 0; i < e->timed_calls && i < cap && (size_t) i * EV_PER_CALL + EV_PER_CALL - 1 < e->ev.size (); ++i) {
		float* o = out + (size_t) i * 4;
		hipEvent_t* const v = &e->ev[(size_t) i * EV_PER_CALL];
		for (int k = 0; k < 3; ++k)
			if (hipEventElapsedTime (&o[k], v[2 * k], v[2 * k + 1]) != hipSuccess) o[k] = 0.f;
		if (hipEventElapsedTime (&o[3], v[0], v[5]) != hipSuccess) o[3] = 0.f;     // (on the caller's stream: a deferred gate is not in it)
	}
	return MTR_OK;
}

int mtr_synth_fill_device (float* d_audio, uint32_t n_streams, uint64_t n_frames, uint64_t stride,
                           uint32_t seed, float fs, int kind, void* hip_stream)
{
	if (!d_audio || stride < n_frames) return fail (MTR_ERR_ARG, "mtr_synth_fill_device");
	if (n_frames == 0 || n_streams == 0) return MTR_OK;
	if (mtr_launch_synth (d_audio, n_streams, n_frames, stride, seed, fs, kind, hip_stream)) return fail (MTR_ERR_HIP, "k_synth launch");
	return MTR_OK;
}

} // extern "C"
