/* mtr_engine.h — C ABI of the MI355X batch metering engine (libmtr_engine.so).
 *
 * This is the drop-in boundary for the per-sample DSP hot path of x42/meters.lv2.
 * The reference has no FFI of its own: its L1 DSP layer is a set of C++ classes
 * (namespace LV2M) plus the C structs of src/spectr.c, called once per audio block
 * from the LV2 run() functions.  Each entry point below names the reference
 * interface it replaces; INTEGRATION.md shows the binding a maintainer would add
 * in src/ebulv2.cc / src/meters.cc / src/spectrumlv2.c.
 *
 * One engine = `n_streams` independent stereo streams advancing in lock step
 * (the batched many-stream variant of one plugin instance each).  The LV2 shim
 * (lib/meters_amd.so, include/lv2_min.h) is a thin n_streams = 1 client.
 *
 * Plain C: opaque handle, int status returns, plain pointers and sizes, no
 * exceptions and no torch/HIP types in any signature (`hip_stream` is a
 * hipStream_t passed as void*, NULL = the default stream).
 */
#ifndef MTR_ENGINE_H
#define MTR_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: + mtr_engine_set_deferred_tail / _join / _deferred_stats, mtr_comm_nranks / _device (round 6); the entry points of
 *    round 5 (mtr_comm_init_timeout, mtr_comm_probe, mtr_comm_set_timeout, mtr_rccl_version, mtr_engine_state_*,
 *    mtr_state_blob_count, MTR_ERR_TIMEOUT / MTR_ERR_STATE) are what a version-1 library may lack.  A client checks
 *    mtr_abi_version () >= the version it was written against before it binds anything newer. */
#define MTR_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------- */
#define MTR_OK               0
#define MTR_ERR_ARG         -1   /* NULL / out-of-range argument */
#define MTR_ERR_UNSUPPORTED -2   /* configuration this build does not implement */
#define MTR_ERR_NODEVICE    -3   /* no usable HIP device: the engine never falls back to the CPU */
#define MTR_ERR_HIP         -4   /* a HIP runtime call failed; see mtr_last_error() */
#define MTR_ERR_NOMEM       -5
#define MTR_ERR_TIMEOUT     -6   /* a deadline passed (mtr_comm_init_timeout, mtr_comm_probe, mtr_engine_reduce on such a communicator) */
#define MTR_ERR_STATE       -7   /* a state blob that does not fit this engine (mtr_engine_state_import) */

/* ---- meters (bit mask) --------------------------------------------------- */
#define MTR_METER_EBU        0x01u  /* Ebu_r128_proc: K-weighting + gated loudness  (ebumeter/ebu_r128_proc.cc) */
#define MTR_METER_TRUEPEAK   0x02u  /* TruePeakdsp::process_max: 4x true peak       (jmeters/truepeakdsp.cc:101-124) */
#define MTR_METER_SPECTR30   0x04u  /* 30-band 1/3-octave bank                      (src/spectr.c, src/spectrumlv2.c) */
#define MTR_METER_TPBALLIST  0x08u  /* TruePeakdsp::process: PPM-style ballistics   (jmeters/truepeakdsp.cc:41-99) */
#define MTR_METER_BITSTATS   0x10u  /* float_stats                                   (src/bitmeter.c:63-105) */
#define MTR_METER_SIGDIST    0x20u  /* signal distribution histogram                (src/sigdistlv2.c:303-318) */
#define MTR_METER_DR14       0x40u  /* DR-14 dynamic range (dr_operation_mode)      (src/dr14.c:283-352, 394-412) */
#define MTR_METER_KMETER     0x80u  /* Kmeterdsp: RMS + peak with hold / fall-back  (jmeters/kmeterdsp.cc:56-140) */

#define MTR_HIST_LEN   751          /* src/uris.h:45  HIST_LEN */
#define MTR_NBANDS     30           /* src/spectrumlv2.c:33  FILTER_COUNT */
#define MTR_BIM_LAST   584          /* src/uris.h:60 */
#define MTR_DIST_BIN   361          /* src/uris.h:47 */
#define MTR_DR_HISTBINS 8000        /* src/dr14.c:43 */

typedef struct mtr_engine mtr_engine;

typedef struct {
	uint32_t struct_size;    /* = sizeof(mtr_config) */
	uint32_t meters;         /* MTR_METER_* mask */
	uint32_t n_streams;      /* independent streams in the batch (>= 1) */
	uint32_t n_channels;     /* 2 (interleaved stereo frames) or 1 (mono; SPECTR30 / TPBALLIST only) */
	float    sample_rate;    /* Hz; reference: instantiate()'s `rate` (src/meters.cc:194) */
	int32_t  device;         /* HIP device ordinal */
	uint32_t max_frames;     /* largest n_frames a process call will carry (scratch sizing); 0 = grow on demand */
	uint32_t tune_run;       /* frames per lane run of the wave-per-segment kernels: 0 = auto, 39 (layouts 3, 4), 19 (layout 4), 38 (layouts 6, 7) */
	uint32_t tune_segments;  /* time segments per stream per call: 0 = auto (layout 7: also forces the lane = segment kernel
	                          * onto every call it can serve, however small the batch) */
	uint32_t tune_layout;    /* 0 = auto,
	                          * 3 = the exact-f32 VALU interpolator, bit-for-bit an fmaf chain over the reference's taps
	                          *     (round 1's default; the cross-check of the matrix-pipe paths),
	                          * 4 = the K-weighting-only kernel (EBU without TRUEPEAK; auto for that mask),
	                          * 6 = K-weighting + the interpolator on the matrix pipe AT F32 GRADE, one wave per (stream, time
	                          *     segment): samples and taps as two f16 halves each, three partial products, f32
	                          *     accumulation — within 3e-7 relative of a float64 interpolator, like the f32 chain itself;
	                          *     held to the same 2e-6 relative parity bound as layout 3 (tests/test_gpu_layout6.py),
	                          * 7 = (auto wherever TRUEPEAK is asked for) layout 6, and for every call that fits it — a batch
	                          *     big enough to fill the chip's lanes, with at least one whole 50 ms fragment in it; any
	                          *     sample rate, stride and position in the stream — the same arithmetic with LANE = TIME
	                          *     SEGMENT (mtr_seg.hip): whole fragments through that kernel, the rest of a fragment the call
	                          *     started in and what is left behind the last whole one through layout 6.
	                          * Layouts 1, 2 and 5 of earlier versions no longer exist: MTR_ERR_ARG. */
	uint32_t tune_fir;       /* layout 3 only: 0 = mirror-symmetric form (120 ops / frame), 1 = dense 3 x 48 taps */
	uint32_t tune_prune;     /* 1 = exact true-peak pruning (branch and bound on L1 * max|x| per tile): identical result,
	                          * data-dependent speed; off by default so the default timing is the dense one.
	                          * 2 (layout 6) = the same, and inside a tile every 256-frame block is screened with the first of
	                          * the three f16 products and completed only if it can still hold the maximum: also identical.
	                          * (Pruning is a layout 6 feature: with tune_prune set, layout 7 is not used.) */
} mtr_config;

/* Per-stream results.  The first nine floats are Ebu_r128_proc's getters in
 * declaration order (ebumeter/ebu_r128_proc.h:81-89), then the two histogram
 * counts (:93-94). */
typedef struct {
	float   loudness_M, maxloudn_M, loudness_S, maxloudn_S;
	float   integrated, integ_thr, range_min, range_max, range_thr;
	int32_t hist_M_count, hist_S_count;
	float   truepeak[2];       /* max |4x-oversampled sample| since reset, per channel, linear:
	                            * the max-hold of TruePeakdsp::read() the LV2 glue keeps (src/ebulv2.cc:361-365) */
	float   truepeak_call[2];  /* the same over the most recent process call only = process_max() + read() */
	float   tpb_level[2];      /* TPBALLIST: TruePeakdsp::read(m, p) after the most recent call: m (src/meters.cc:491-507) */
	float   tpb_peak[2];       /*            ... and p, the raw true peak of that call */
} mtr_stream_result;

/* ---- lifecycle ----------------------------------------------------------- */

/* replaces: new Ebu_r128_proc + init(2, rate) (src/ebulv2.cc:189-190), new TruePeakdsp + init(rate)
 * x2 (:192-196), bandpass_setup x30 (src/spectrumlv2.c:103-118), per stream. */
int  mtr_engine_create (const mtr_config* cfg, mtr_engine** out);
/* replaces: delete ebu / delete mtr[c] (src/ebulv2.cc:500-512), spectrum_cleanup */
void mtr_engine_destroy (mtr_engine* e);

/* replaces: Ebu_r128_proc::reset (ebu_r128_proc.cc:176-189) + TruePeakdsp::reset + zeroed bank state */
int  mtr_engine_reset (mtr_engine* e);
/* replaces: Ebu_r128_proc::integr_start / integr_pause / integr_reset (ebu_r128_proc.h:77-79, .cc:192-204) */
int  mtr_engine_integr_start (mtr_engine* e);
int  mtr_engine_integr_pause (mtr_engine* e);
int  mtr_engine_integr_reset (mtr_engine* e);
/* replaces: TruePeakdsp::reset on every stream (src/meters.cc:451-456) */
int  mtr_engine_truepeak_reset (mtr_engine* e);
/* replaces: the speed-port handler (src/spectrumlv2.c:170-177) and the peak-hold reset (:191-205) */
int  mtr_engine_spectr_set_speed (mtr_engine* e, float v);
int  mtr_engine_spectr_reset_peak (mtr_engine* e);

/* ---- the hot path --------------------------------------------------------- */

/* Advance every stream by n_frames.  `d_audio` is DEVICE memory, stream s at
 * d_audio + s * stream_stride_frames * n_channels, frames interleaved [L R].
 * Asynchronous on `hip_stream`.
 * replaces, per stream: Ebu_r128_proc::process (ebu_r128_proc.cc:207-248) as called at
 * src/ebulv2.cc:341-342, TruePeakdsp::process_max x2 (:344-347), the per-sample loop of
 * spectrum_run (src/spectrumlv2.c:210-227), TruePeakdsp::process (src/meters.cc:465-475). */
int  mtr_engine_process_device (mtr_engine* e, const float* d_audio, uint64_t n_frames,
                                uint64_t stream_stride_frames, void* hip_stream);
/* Same with HOST memory: the batch crosses the host link in chunks of streams (results are per stream, so chunking is
 * exact — bit for bit what mtr_engine_process_device gives on the same audio), chunk k + 1 on a copy stream under the kernels
 * of chunk k, through two device buffers of one chunk each.  Host-link-bound: end to end at the link's rate (bench.py
 * reports it as extra.end_to_end_host, never as `value`).  Returns when the caller's memory has been read; the kernels
 * may still run (mtr_engine_sync / the result getters wait).  Argument errors are reported before anything is queued; a
 * HIP failure in the middle of a call (MTR_ERR_HIP / MTR_ERR_NOMEM) leaves the chunks already queued metered and the
 * rest not — the streams are no longer in lock step: mtr_engine_reset () before the engine is used again. */
int  mtr_engine_process_host (mtr_engine* e, const float* h_audio, uint64_t n_frames,
                              uint64_t stream_stride_frames);
/* Bytes of audio per chunk of mtr_engine_process_host (0 = the default, 256 MiB; at least one stream per chunk). */
int  mtr_engine_set_host_chunk_bytes (mtr_engine* e, uint64_t bytes);
/* n_streams == 1, planar host channels — the shape an LV2 run() hands over
 * (src/meters.cc:298-299: one float* per port, n_samples frames). */
int  mtr_engine_process_planar_host (mtr_engine* e, const float* const* channels, uint32_t n_frames);
/* Before the first block (an LV2 instantiate / activate): one silent block of `max_block_frames` through the engine and
 * a full reset — staging buffers, the engine's stream and the kernels' code objects then exist, and the first run() of the
 * host's audio thread costs what every later one does (tests/test_lv2_latency.py) instead of several milliseconds. */
int  mtr_engine_prepare_host (mtr_engine* e, uint32_t max_block_frames);

/* Wait for everything queued by process calls (on the caller's stream and on the engine's side stream, below). */
int  mtr_engine_sync (mtr_engine* e);

/* The tail of a process call — the once-per-fragment bookkeeping of Ebu_r128_proc::process (ebu_r128_proc.cc:217-244: k_gate)
 * and, if mtr_engine_reduce follows, the job's reduction — may run DEFERRED on an engine-owned side stream, beside the
 * fused kernel of the NEXT call instead of in front of it: that kernel needs the K-filter state and the interpolator history
 * of this call, never the gate's results.  Same kernels, same inputs, fragments inserted in fragment order: every result is
 * bit for bit that of the serial order (tests/test_gpu_tail.py).  mode 0 = auto (a batch — >= 4096 streams, >= 2^24 stream-frames per call — whose
 * whole fragments go through the lane = time segment kernel, in an engine of EBU / TRUEPEAK only: where it was measured to pay),
 * 1 = never (everything on the caller's stream, as the LV2-sized calls always are), 2 = always.
 * What a caller must know: results are complete when BOTH streams are — every getter, mtr_engine_sync, state export / import
 * and the resets wait for both; a caller that reads d_hist / d_max of mtr_engine_reduce (or any engine buffer) in stream
 * order calls mtr_engine_join (e, stream) first: `stream` then waits for the side stream's work queued so far.
 * No reference counterpart (the reference is one instance on one thread).
 * (Read once by mtr_engine_create, for tests and experiments only: MTR_TAIL_MODE = the initial mode — the -m gpu suite runs green under 2 —,
 * MTR_TAIL_DELAY_US / MTR_TAIL_GATE_GRID = the two constants of the deferred gate, see profiles/r06_tail.md.) */
int  mtr_engine_set_deferred_tail (mtr_engine* e, int mode);
int  mtr_engine_join (mtr_engine* e, void* hip_stream);
/* Process calls whose tail was deferred, since the engine was created. */
int  mtr_engine_deferred_stats (mtr_engine* e, uint64_t* calls);

/* ---- results (synchronise, then copy to host memory) ----------------------- */

/* replaces: loudness_M() ... range_thr(), hist_*_count() (ebu_r128_proc.h:81-94), TruePeakdsp::read */
int  mtr_engine_results (mtr_engine* e, uint32_t first, uint32_t count, mtr_stream_result* out);
/* replaces: histogram_M() / histogram_S() (ebu_r128_proc.h:91-92); out arrays are [count][751] */
int  mtr_engine_histograms (mtr_engine* e, uint32_t first, uint32_t count, int32_t* hist_M, int32_t* hist_S);
/* Per-fragment mean powers of the most recent call ([count][n_frag], n_frag returned), i.e. the
 * values Ebu_r128_proc::process pushes into _power[] (ebu_r128_proc.cc:219). Diagnostic / parity. */
int  mtr_engine_fragment_powers (mtr_engine* e, uint32_t first, uint32_t count, float* out,
                                 uint32_t capacity_per_stream, uint32_t* n_frag);
/* replaces: spectrum_run's epilogue (src/spectrumlv2.c:230-248): raw val_f / max_f and the dB
 * values written to ports 0-29 / 30-59; arrays are [count][30], any may be NULL */
int  mtr_engine_spectrum (mtr_engine* e, uint32_t first, uint32_t count,
                          float* val, float* max, float* val_db, float* max_db);

/* Integer paths (mono engines, n_channels == 1, as the reference's bitmeter / SigDistHist plugins).
 * replaces: float_stats' table and counters (src/bitmeter.c:63-105, layout src/uris.h:53-60):
 *   hist [count][584], counters [count][5] = zero, pos, nan, inf, denormal, minmax [count][2] */
int  mtr_engine_bitstats (mtr_engine* e, uint32_t first, uint32_t count,
                          int32_t* hist, int32_t* counters, float* minmax);
/* replaces: the state sdh_run's loop maintains (src/sigdistlv2.c:296-327): bins [count][361],
 *   peak [count][2] = {count, bin}, moments [count][3] = {sum, mean, M2} (double), n [count].
 *   mean / M2 are the reference's accumulators hist_tmpS / hist_varS: Welford's moments while every sample is binned,
 *   and — once a sample fell outside the 361 bins, after which the reference keeps dividing by the index among ALL
 *   samples (:312-315) and they stop being moments — the very numbers it then reports (1e-12 relative; mtr_intstat.hip);
 *   bins, peak, n and sum are bit-identical in every case. */
int  mtr_engine_sigdist (mtr_engine* e, uint32_t first, uint32_t count,
                         int32_t* bins, int32_t* peak, double* moments, int64_t* n);
/* replaces: bim_reset (src/bitmeter.c:47-60) and the SDH reset */
int  mtr_engine_intstat_reset (mtr_engine* e);

/* DR-14 for a batch of tracks (MTR_METER_DR14; 1 or 2 channels).  Replaces what dr14_run leaves on the dr14
 * plugins' ports in dr_operation_mode (src/dr14.c:413-451): m_rms = the score of dr14_calc_rms_score, m_peak = the
 * second-highest window peak in dB (LV2dr14::m_peak, the one that enters dr — the plugin's m_peak PORT shows the
 * true-peak maximum, which is MTR_METER_TPBALLIST's tpb_peak), dr [c] = clamp (min (0, m_peak) - m_rms, 1, 20) or 21 while undefined,
 * dr_total = the same of the mean over the valid channels, block_count = 3 * windows counted. */
typedef struct mtr_dr14_result {
	float m_rms[2], m_peak[2], dr[2], dr_total, block_count;
} mtr_dr14_result;
int  mtr_engine_dr14_results (mtr_engine* e, uint32_t first, uint32_t count, mtr_dr14_result* out);
/* replaces: reset_peaks (src/dr14.c:245-260) */
int  mtr_engine_dr14_reset (mtr_engine* e);

/* Kmeterdsp for a batch (MTR_METER_KMETER; 1 or 2 channels): every process call is one Kmeterdsp::process () per
 * channel (jmeters/kmeterdsp.cc:56-140; n mod 4 trailing frames are dropped as there).
 * replaces: Kmeterdsp::read (rms, peak) (:148-153) — rms, peak [count][2] linear; arms the "start a new maximum"
 * flag exactly as read () does */
int  mtr_engine_kmeter_read (mtr_engine* e, uint32_t first, uint32_t count, float* rms, float* peak);
/* replaces: Kmeterdsp::reset (:142-146) */
int  mtr_engine_kmeter_reset (mtr_engine* e);

/* ---- multi-GPU aggregate ---------------------------------------------------- */

/* Sum the two loudness histograms over this engine's streams into d_hist[2][751] (int32) and take
 * the max of true peak / max-M / max-S into d_max[4] = {tp_L, tp_R, maxM, maxS} — both DEVICE
 * buffers owned by the caller, so the host can all-reduce them across ranks (RCCL) in place.
 * No reference counterpart (the reference is single-instance). */
int  mtr_engine_aggregate_device (mtr_engine* e, int32_t* d_hist, float* d_max, void* hip_stream);
/* The job's ONE collective, RCCL inside: every rank's streams are independent, so the only
 * exchange is the final reduction of the per-rank aggregates — sum of the two 751-bin histograms, max of the peaks.
 *   rank 0:  mtr_comm_unique_id (id)                      -> ncclGetUniqueId; ship the 128 bytes to every rank
 *   each:    mtr_comm_init (&c, rank, world, id, device)  -> ncclCommInitRank (collective: all ranks call it)
 *   per job: mtr_engine_reduce (e, c, d_hist, d_max, st)  -> mtr_engine_aggregate_device on this rank's streams, then
 *            ncclAllReduce (int32[1502], sum) and ncclAllReduce (float[4], max) in place on `st`;
 *            mtr_hist_loudness () turns the summed histograms into the programme's loudness and range.
 * world = 1 is valid (the reduction is the identity).  No reference counterpart (the reference is single-instance). */
#define MTR_COMM_ID_BYTES 128
typedef struct mtr_comm mtr_comm;
int  mtr_comm_unique_id (void* id128);
int  mtr_comm_init (mtr_comm** out, int rank, int world, const void* id128, int device);
/* The same with a DEADLINE, for a job's first contact with RCCL: ncclCommInitRankConfig (blocking = 0) and
 * ncclCommGetAsyncError polled until the communicator stands, fails, or `timeout_ms` have passed — then ncclCommAbort and
 * MTR_ERR_TIMEOUT, so that the host can fall back (a rank that never arrives would otherwise leave the others inside
 * ncclCommInitRank for good).  timeout_ms = 0: mtr_comm_init (blocking, no deadline).  *init_ms (may be NULL): how long it took.
 * Every later call on such a communicator is bounded the same way (mtr_comm_set_timeout; default: the init's deadline). */
int  mtr_comm_init_timeout (mtr_comm** out, int rank, int world, const void* id128, int device, uint32_t timeout_ms, float* init_ms);
/* One 4-byte all-reduce on the communicator's own stream, polled (hipStreamQuery + ncclCommGetAsyncError) to `timeout_ms`
 * (0: wait for good): the job's first collective, before anything is timed.  MTR_ERR_TIMEOUT / MTR_ERR_HIP leave the
 * communicator ABORTED: only mtr_comm_destroy may follow.  *ms (may be NULL): how long the collective took. */
int  mtr_comm_probe (mtr_comm* c, uint32_t timeout_ms, float* ms);
/* Deadline of the calls mtr_engine_reduce makes on a communicator built by mtr_comm_init_timeout (the enqueue of the
 * collective, not its execution on the stream). */
int  mtr_comm_set_timeout (mtr_comm* c, uint32_t timeout_ms);
/* A communicator built with a deadline is finalised (ncclCommFinalize, polled to that deadline) and destroyed — collectives
 * still queued on a stream complete — and aborted only if that does not happen in time or it had failed before; a blocking
 * one is ncclCommDestroy'd.  Either way: synchronise the streams mtr_engine_reduce was given (mtr_engine_sync) first. */
void mtr_comm_destroy (mtr_comm* c);
/* After a process call whose tail was deferred (above) the aggregate and the all-reduce follow the gate on the engine's side
 * stream (ordered behind what `hip_stream` holds at the time of the call): d_hist / d_max are complete after
 * mtr_engine_join (e, hip_stream) in stream order, or mtr_engine_sync for the host. */
int  mtr_engine_reduce (mtr_engine* e, mtr_comm* c, int32_t* d_hist, float* d_max, void* hip_stream);
/* What RCCL itself says about the communicator: ncclCommCount and ncclCommCuDevice (a negative status if it has been
 * aborted) — so that a job's line can prove the collective ran over N ranks on N devices (bench.py: config.rccl_nranks). */
int  mtr_comm_nranks (mtr_comm* c);
int  mtr_comm_device (mtr_comm* c);
/* ncclGetVersion of the RCCL this process runs (e.g. 22606), or a negative status. */
int  mtr_rccl_version (void);

/* Programme-level integrated loudness / range from (summed) histograms, exactly as
 * Ebu_r128_hist::calc_integ / calc_range do (ebu_r128_proc.cc:105-150). Host-side, pure C. */
void mtr_hist_loudness (const int32_t* hist_M, const int32_t* hist_S,
                        float* integrated, float* integ_thr,
                        float* range_min, float* range_max, float* range_thr);

/* ---- per-stream state: checkpoint / resume, re-sharding between engines ------------ */

/* Everything streams [first, first + count) carry from call to call — K-filter states, the open fragment and the 64-fragment
 * ring, both histograms and their counters, results and peak holds, the 47-frame interpolator history, ballistics, the
 * bank's 30 x 12 section states, levels, peak holds and dither parity, the integer meters' tables, the DR-14 histograms and
 * open window, the K-meter detector — as ONE opaque, versioned blob in host memory, together with the engine's lock-step
 * cursors (frames left in the open fragment, integration on / off, samples in the open DR-14 window, the bank's speed).
 * mtr_engine_state_import puts a blob's streams into slots [first, first + its count) of another engine of the SAME
 * configuration (meters, channels, sample rate; n_streams and the slots may differ) — in another process, on another GPU —
 * and processing continues bit for bit as if it had never stopped (tests/test_gpu_state.py).  An engine that has not
 * processed anything since it was created or reset takes the blob's cursors; any other must stand at the same ones (the
 * streams of an engine advance in lock step), else MTR_ERR_STATE.  "Takes the blob's cursors" includes integration on / off
 * and the bank's speed: a fresh engine on which mtr_engine_integr_start or mtr_engine_spectr_set_speed was called before the
 * import continues with the BLOB's setting (they are part of where the streams stand), not with the setter's.  The blob
 * carries a checksum of its payload; a blob whose cursors are out of range or whose checksum does not match is refused
 * with MTR_ERR_STATE and the engine is left as it was.  Both calls synchronise.
 * No reference counterpart: the reference persists one UI word (src/ebulv2.cc:514-553); SURVEY.md 5 (checkpoint / resume). */
size_t mtr_engine_state_bytes (const mtr_engine* e, uint32_t count);
int  mtr_engine_state_export (mtr_engine* e, uint32_t first, uint32_t count, void* blob, size_t capacity);
int  mtr_engine_state_import (mtr_engine* e, uint32_t first, const void* blob, size_t bytes);
/* Streams held by a blob (0 if it is not one). */
uint32_t mtr_state_blob_count (const void* blob, size_t bytes);

/* ---- measurement / introspection --------------------------------------------- */

/* HIP-event timing of the kernels the engine launches (off by default). */
int  mtr_engine_timing_enable (mtr_engine* e, int on);
/* Sum over the process calls since the last query: fused K-weight+true-peak kernel, gating kernel,
 * filter-bank kernel (ms) and the number of calls. Synchronises. */
int  mtr_engine_timing_query (mtr_engine* e, float* ms_fused, float* ms_gate, float* ms_bank, uint32_t* calls);
/* The same per call, without resetting anything: out [min (calls, cap)][4] = fused, gate, everything behind the gate (bank,
 * integer paths, history), and the whole call from its first to its last event (ms); *calls = timed calls since the last
 * query.  For the median beside the mean (a power-limited kernel drifts within a run).  Synchronises.
 * (On the host path every CHUNK of mtr_engine_process_host is a timed call: `calls` counts chunks there; a call whose four
 * events could not all be created is not counted.) */
int  mtr_engine_timing_calls (mtr_engine* e, float* out, uint32_t cap, uint32_t* calls);
/* With tune_prune: interpolator tile passes considered / skipped since the engine was created. */
int  mtr_engine_prune_stats (mtr_engine* e, uint64_t* considered, uint64_t* skipped);
/* With tune_prune = 2 (layout 6): 256-frame channel-blocks screened with the first of the three products / completed
 * with the other two, since the engine was created. */
int  mtr_engine_refine_stats (mtr_engine* e, uint64_t* screened, uint64_t* completed);
/* The kernel layout the engine resolved to (tune_layout = 0 picks one from the meters mask): 3, 4, 6 or 7. */
int  mtr_engine_layout (const mtr_engine* e);
/* Layout 7: process calls whose whole fragments went through the lane = time segment kernel, and the frames per stream
 * it covered, since the engine was created (the calls that did not fit it ran layout 6). */
int  mtr_engine_seg_stats (mtr_engine* e, uint64_t* calls, uint64_t* frames_per_stream);
/* How an engine of this configuration would tile and route a call of `n_frames` that starts with `frames_left_in_fragment`
 * frames of a 50 ms fragment still open (0 or sample_rate / 20: on a boundary): pure host arithmetic, no device needed —
 * the planning logic of mtr_engine_process_* (no reference counterpart: the reference walks its samples one by one,
 * ebumeter/ebu_r128_proc.cc:217-244).  n_slots = resident lane = segment waves to plan for (0: 1024, an MI355X). */
typedef struct mtr_plan_info {
	uint32_t layout;            /* as mtr_engine_layout () */
	uint32_t uses_seg;          /* 1: the call's whole fragments go through the lane = time segment kernel */
	uint32_t head_frames;       /* frames in front of them (the rest of the open fragment), through the wave-per-segment kernel */
	uint32_t body_fragments;    /* whole fragments through the lane = segment kernel */
	uint32_t segments;          /* ... cut into this many time segments per stream */
	uint32_t fragments_per_lane;/* ... of which every lane walks this many (a short segment starts one early) */
	uint32_t warm_steps;        /* K-filter warm-up in front of a segment that does not start the body: steps of 16 frames */
	uint32_t n_tiles;           /* tiles of the whole call (head + body + tail) */
	uint32_t head_tiles;        /* ... of which in front of the body */
	uint32_t n_fragments_ended; /* fragments that end inside the call */
	uint32_t kw_segments;       /* time segments per stream of a call that does not use the lane = segment kernel */
	uint32_t frames_left_after; /* of the fragment open when the call returns */
} mtr_plan_info;
int  mtr_plan_query (const mtr_config* cfg, uint32_t frames_left_in_fragment, uint64_t n_frames, uint32_t n_slots, mtr_plan_info* out);
/* K-weighting coefficients a0 a1 a2 b1 b2 c3 c4 at `sample_rate` (Ebu_r128_proc::detect_init,
 * ebumeter/ebu_r128_proc.cc:263-293) */
int  mtr_kweight_coef (float sample_rate, float* out7);
/* the 120-float polyphase table (Resampler_table ctor, fr = 1, hl = 24, np = 4) */
int  mtr_fir_table (float* out120);
/* 36 doubles [section][a0 a1 a2 b0 b1 b2] of band `band` at `rate` (bandpass_setup, src/spectr.c:89-206) */
int  mtr_band_coef (double rate, uint32_t band, double* out36);

/* Fill device memory with the repo's seeded synthetic programme signal:
 * stream s = LCG(seed + s) noise under a slow envelope plus a tone (kind 1); kind 0 = the plain LCG noise, kind 2 = that
 * noise under a level rising monotonically by 48 dB (the worst case of exact peak pruning). Used by bench.py so the
 * timed region starts with inputs resident in HBM. */
int  mtr_synth_fill_device (float* d_audio, uint32_t n_streams, uint64_t n_frames,
                            uint64_t stream_stride_frames, uint32_t seed, float sample_rate,
                            int kind, void* hip_stream);

const char* mtr_last_error (void);
const char* mtr_version (void);
int         mtr_abi_version (void);

#ifdef __cplusplus
}
#endif
#endif
