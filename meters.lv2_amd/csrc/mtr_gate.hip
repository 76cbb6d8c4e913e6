// mtr_gate.hip — fragment bookkeeping, M/S loudness, histograms, gated integration and range.
//
// Replaces the once-per-fragment part of Ebu_r128_proc::process (ebumeter/ebu_r128_proc.cc:217-244),
// addfrags (:251-260) and Ebu_r128_hist::addpoint / integrate / calc_integ / calc_range (:66-150).
//
// This TU is compiled with -ffp-contract=off and does its divisions / log10 in double rounded
// back to float, so that given the same fragment powers every float it produces and every
// histogram bin it increments is the one the reference produces on the host: floorf(10 v + 700.5f)
// and floorf(100 log10f(s) + 0.5f) are bin decisions and must not see a fused multiply-add.
//
// One workgroup (256 lanes) per stream.  The reference does this work serially per fragment; here
//   * fragment powers are summed per fragment from the tile powers of the fused kernel, in tile
//     order, starting from the 1e-30f floor (or the carried partial sum),
//   * each lane evaluates M (last 8 fragments) and S (last 60) for its own fragment with the same
//     chronological summation order as addfrags(),
//   * histogram inserts are integer atomics (order-independent, exact),
//   * calc_integ / calc_range run once, at the last fragment whose `_div2` counter wraps inside
//     this call — only that evaluation survives in the reference — one wave each, walking the
//     occupied bins in the reference's order so that the float accumulation of integrate() is the reference's.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

#define CHUNK 256

__device__ __forceinline__ float log10f_cr (float x) { return (float) log10 ((double) x); }
__device__ __forceinline__ float divf_cr (float a, float b) { return (float) ((double) a / (double) b); }

// Ebu_r128_hist::integrate / calc_integ / calc_range  (ebu_r128_proc.cc:82-150), wave-cooperative: the 64 lanes of ONE
// wave call these with the same arguments and get the same (wave-uniform) result.
//
// The reference walks all 751 bins serially and the float sum depends on that order.  One lane doing the same is a
// chain of 751 LDS round trips, five times per stream and call — it was 60 % of this kernel.  But an empty bin adds
// k * p = +0.0f to a non-negative sum and leaves it bit for bit as it was, so only the occupied bins matter: the wave
// reads 64 bins at a time, a ballot finds the occupied ones, and those are added in the reference's order (readlane
// by ascending set bit); the division by ten falls where j wraps, whether or not that century holds a point.
__device__ float hist_integrate (const int32_t* h, const float* bin_power, int i0, int lane)
{
	int   n = 0;
	float s = 0;
	for (int c = i0 / 100; c < 8; ++c) {
		const int lo = max (i0, 100 * c), hi = min (100 * c + 99, 750);
		for (int b = lo; b <= hi; b += 64) {
			const int   i = b + lane;
			const int   k = i <= hi ? h[i] : 0;
			const float p = i <= hi ? bin_power[i - 100 * c] : 0.f;
			uint64_t m = __ballot (k != 0);
			while (m) {
				const int l = __builtin_ctzll (m);
				m &= m - 1;
				const int   kk = __builtin_amdgcn_readlane (k, l);
				const float pp = __int_as_float (__builtin_amdgcn_readlane (__float_as_int (p), l));
				n += kk;
				s += (float) kk * pp;
			}
		}
		if (hi == 100 * c + 99) s = divf_cr (s, 10.0f);          // j reached 100 (:96-100)
	}
	return divf_cr (s, (float) n);
}

__device__ void hist_calc_integ (const int32_t* h, int count, const float* bp, float* vi, float* th, int lane)
{
	if (count < 50) { if (lane == 0) *vi = -200.0f; return; }
	float s = hist_integrate (h, bp, 0, lane);
	const float t = 10 * log10f_cr (s) - 10.0f;
	int k = (int) (floorf (100 * log10f_cr (s) + 0.5f)) + 600;
	if (k < 0) k = 0;
	s = hist_integrate (h, bp, k, lane);
	if (lane == 0) { *th = t; *vi = 10 * log10f_cr (s); }
}

__device__ void hist_calc_range (const int32_t* h, int count, const float* bp, float* v0, float* v1, float* th, int lane)
{
	if (count < 20) { if (lane == 0) { *v0 = -200.0f; *v1 = -200.0f; } return; }
	float s = hist_integrate (h, bp, 0, lane);
	const float t = 10 * log10f_cr (s) - 20.0f;
	// ebu_r128_proc.cc:141: the 0.5 here is a double
	int k = (int) (floorf ((float) ((double) (100 * log10f_cr (s)) + 0.5))) + 500;
	if (k < 0) k = 0;
	int n = 0;                                                   // points at or above the threshold: integers, any order
	for (int i = k + lane; i <= 750; i += 64) n += h[i];
	for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor (n, d, 64);
	const float a = 0.10f * n;
	const float b = 0.95f * n;
	// `for (i = k, s = 0; s < a; i++) s += h[i]` stops behind the first bin that takes the sum to a: an occupied one
	int i = k;
	s = 0;
	for (int base = k; base <= 750 && s < a; base += 64) {
		const int idx = base + lane;
		const int kk = idx <= 750 ? h[idx] : 0;
		uint64_t m = __ballot (kk != 0);
		while (m && s < a) {
			const int l = __builtin_ctzll (m);
			m &= m - 1;
			s += __builtin_amdgcn_readlane (kk, l);
			i = base + l + 1;
		}
	}
	// `for (j = 750, s = n; s > b; j--) s -= h[j]` likewise from the top (lane l holds bin base - l: set bits descend)
	int j = 750;
	s = n;
	for (int base = 750; base >= 0 && s > b; base -= 64) {
		const int idx = base - lane;
		const int kk = idx >= 0 ? h[idx] : 0;
		uint64_t m = __ballot (kk != 0);
		while (m && s > b) {
			const int l = __builtin_ctzll (m);
			m &= m - 1;
			s -= __builtin_amdgcn_readlane (kk, l);
			j = base - l - 1;
		}
	}
	if (lane == 0) {
		*th = t;
		*v0 = divf_cr ((float) (i - 701), 10.0f);
		*v1 = divf_cr ((float) (j - 699), 10.0f);
	}
}

// addfrags  ebu_r128_proc.cc:251-260 on a chronological array: pw[-(n-1) .. 0]
__device__ __forceinline__ float addfrags (const float* newest, int nfrag)
{
	float s = 0;
	for (int i = nfrag - 1; i >= 0; --i) s += newest[-i];
	return -0.6976f + 10 * log10f_cr (divf_cr (s, (float) nfrag));
}

__device__ __forceinline__ void hist_add (int32_t* h, int32_t* count, int32_t* error, float v)
{
	int k = (int) floorf (10 * v + 700.5f);     // ebu_r128_proc.cc:70
	if (k < 0) return;
	if (k > 750) { k = 750; atomicAdd (error, 1); }
	atomicAdd (&h[k], 1);
	atomicAdd (count, 1);
}

__global__ __launch_bounds__ (256) void k_gate (const mtr_gate_args a)
{
#pragma clang fp contract(off)
	__shared__ float   pw[64 + CHUNK];          // chronological fragment powers: 64 history + chunk
	__shared__ int32_t sh_hist[2][MTR_HIST_LEN];
	__shared__ float   sh_red[2][256];
	__shared__ int32_t sh_cnt[4];               // cnt_M cnt_S err_M err_S

	const int tid = threadIdx.x;
	// (a workgroup walks streams blockIdx.x, blockIdx.x + gridDim.x, ...: one each when the grid is the batch — the serial
	// order — sixteen each in the deferred tail's 512-workgroup launch, see mtr_launch_gate)
	for (uint32_t s = blockIdx.x; s < a.n_streams; s += gridDim.x) {
	mtr_stream_state* const st = a.state + s;
	const float* const tp = a.tile_power + (size_t) s * a.n_tiles;
	int32_t* const ghist = a.hist + (size_t) s * 2 * MTR_HIST_LEN;

	__syncthreads ();                              // (the previous stream's last readers of the shared arrays)
	for (int i = tid; i < 64; i += 256) pw[i] = st->ring[i];
	for (int i = tid; i < 2 * MTR_HIST_LEN; i += 256) (&sh_hist[0][0])[i] = ghist[i];
	if (tid < 4) sh_cnt[tid] = (tid == 0) ? st->cnt_M : (tid == 1) ? st->cnt_S : (tid == 2) ? st->err_M : st->err_S;
	const int   div1_0 = st->div1, div2_0 = st->div2;
	const float frpwr0 = st->frpwr;
	float max_M = st->max_M, max_S = st->max_S;
	float last_M = st->loud_M, last_S = st->loud_S;
	__syncthreads ();

	// Index (within this call) of the last fragment at which _div2 wraps: calc_* run there.
	// _div2 after fragment f (0-based) is (div2_0 + f + 1) mod 10.
	int f_calc = -1;
	if (a.integr && a.n_frag > 0) {
		const int r = (div2_0 + (int) a.n_frag) % 10;      // value after the last fragment
		const int f = (int) a.n_frag - 1 - r;
		if (f >= 0) f_calc = f;
	}

	for (uint32_t base = 0; base < a.n_frag; base += CHUNK) {
		const int nf = min ((int) (a.n_frag - base), CHUNK);
		// fragment mean powers of this chunk
		if (tid < nf) {
			const uint32_t f = base + tid;
			float acc = (f == 0) ? frpwr0 : 1e-30f;
			for (uint32_t j = a.frag_tile[f]; j < a.frag_tile[f + 1]; ++j) acc += tp[j];
			const float p = divf_cr (acc, a.fragm);
			pw[64 + tid] = p;
			if (a.frag_power) a.frag_power[(size_t) s * a.n_frag + f] = p;
		}
		__syncthreads ();

		float lm = -200.0f, ls = -200.0f;
		if (tid < nf) {
			lm = addfrags (&pw[64 + tid], 8);
			ls = addfrags (&pw[64 + tid], 60);
			if (!isfinite (lm) || lm < -200.f) lm = -200.0f;
			if (!isfinite (ls) || ls < -200.f) ls = -200.0f;
		}
		// histogram inserts up to and including f_calc go in now, the rest after calc_*
		const int f_abs = (int) base + tid;
		const bool addM = a.integr && tid < nf && ((div1_0 + f_abs + 1) % 2 == 0);
		const bool addS = a.integr && tid < nf && ((div2_0 + f_abs + 1) % 10 == 0);
		if (addM && f_abs <= f_calc) hist_add (sh_hist[0], &sh_cnt[0], &sh_cnt[2], lm);
		if (addS && f_abs <= f_calc) hist_add (sh_hist[1], &sh_cnt[1], &sh_cnt[3], ls);
		__syncthreads ();

		if (f_calc >= (int) base && f_calc < (int) base + nf) {
			// wave 0 integrates, wave 1 finds the range
			if (tid < 64) {
				hist_calc_integ (sh_hist[0], sh_cnt[0], a.bin_power, &st->integ, &st->integ_thr, tid);
			} else if (tid < 128) {
				hist_calc_range (sh_hist[1], sh_cnt[1], a.bin_power, &st->rmin, &st->rmax, &st->rthr, tid - 64);
			}
		}
		__syncthreads ();
		if (addM && f_abs > f_calc) hist_add (sh_hist[0], &sh_cnt[0], &sh_cnt[2], lm);
		if (addS && f_abs > f_calc) hist_add (sh_hist[1], &sh_cnt[1], &sh_cnt[3], ls);

		if (tid < nf) {                              // max-hold is order-free: per-lane, folded once at the end
			max_M = lm > max_M ? lm : max_M;
			max_S = ls > max_S ? ls : max_S;
			if (f_abs == (int) a.n_frag - 1) { sh_red[0][0] = lm; sh_red[1][0] = ls; }   // the values a getter sees
		}
		__syncthreads ();
		// slide the power window: keep the newest 64 as history for the next chunk
		float keep = 0;
		if (tid < 64) keep = pw[nf + tid];
		__syncthreads ();
		if (tid < 64) pw[tid] = keep;
		__syncthreads ();
	}

	// fold the per-lane maxima
	if (a.n_frag > 0) { last_M = sh_red[0][0]; last_S = sh_red[1][0]; }
	__syncthreads ();
	sh_red[0][tid] = max_M; sh_red[1][tid] = max_S;
	__syncthreads ();
	for (int d = 128; d >= 1; d >>= 1) {
		if (tid < d) {
			sh_red[0][tid] = sh_red[0][tid + d] > sh_red[0][tid] ? sh_red[0][tid + d] : sh_red[0][tid];
			sh_red[1][tid] = sh_red[1][tid + d] > sh_red[1][tid] ? sh_red[1][tid + d] : sh_red[1][tid];
		}
		__syncthreads ();
	}
	max_M = sh_red[0][0]; max_S = sh_red[1][0];
	// tiles of the still-open fragment: partial power carried to the next call
	if (tid == 0) {
		float acc = (a.n_frag == 0) ? frpwr0 : 1e-30f;
		for (uint32_t j = a.tail_tile; j < a.n_tiles; ++j) acc += tp[j];
		st->frpwr = acc;
		st->loud_M = last_M; st->loud_S = last_S;
		st->max_M = max_M;   st->max_S = max_S;
		st->div1 = a.integr ? (div1_0 + (int) a.n_frag) % 2 : div1_0;
		st->div2 = a.integr ? (div2_0 + (int) a.n_frag) % 10 : div2_0;
		st->cnt_M = sh_cnt[0]; st->cnt_S = sh_cnt[1]; st->err_M = sh_cnt[2]; st->err_S = sh_cnt[3];
		// true-peak hold (a deferred gate runs beside the next call's fused kernel, which is already raising tp_call:
		// then k_history has folded it on the caller's stream — mtr_fold_truepeak, mtr_internal.h)
		if (a.fold_tp) mtr_fold_truepeak (st);
	}
	for (int i = tid; i < 64; i += 256) st->ring[i] = pw[i];
	for (int i = tid; i < 2 * MTR_HIST_LEN; i += 256) ghist[i] = (&sh_hist[0][0])[i];
	}
}

// ---- long calls: the same bookkeeping spread over many workgroups per stream ----------------------------
//
// One workgroup walking 72 000 fragments (an hour of audio in one call) takes 0.7 ms — twice the K-weighting
// kernel itself.  Nothing in the reference's per-fragment work is order dependent except the single
// calc_integ / calc_range evaluation that survives (at fragment f_calc, within the last ten of the call):
// histogram inserts are integer counts, the max-hold is a max.  So:
//   k_gate_frag   grid (blocks, streams): fragments [b FPB, (b+1) FPB): powers (with 63 fragments of history
//                 recomputed), M / S, inserts for f <= f_calc into an LDS histogram flushed with global
//                 atomics, block maxima with atomicMax on sortable ints, the optional fragment-power output;
//   k_gate_final  one workgroup per stream, after it: calc_* on the merged histogram, the <= 9 inserts after
//                 f_calc, the values a getter sees, the 64-fragment ring and the counters for the next call.
// Every number is the one k_gate produces (tests/test_gpu_parity.py: one long call == many short ones).
#define GATE_FPB 1024

__device__ __forceinline__ int32_t sortable (float v) { const int32_t b = __float_as_int (v); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float unsortable (int32_t k) { return __int_as_float (k ^ ((k >> 31) & 0x7fffffff)); }

// mean power of fragment f of this call (f < 0: history of earlier calls from the ring)
__device__ __forceinline__ float frag_power_of (const mtr_gate_args& a, const mtr_stream_state* st, const float* tp, int f)
{
	if (f < 0) return f >= -64 ? st->ring[64 + f] : 0.f;
	float acc = (f == 0) ? st->frpwr : 1e-30f;
	for (uint32_t j = a.frag_tile[f]; j < a.frag_tile[f + 1]; ++j) acc += tp[j];
	return divf_cr (acc, a.fragm);
}

__device__ __forceinline__ int gate_f_calc (const mtr_gate_args& a, int div2_0)
{
	if (!a.integr || a.n_frag == 0) return -1;
	const int r = (div2_0 + (int) a.n_frag) % 10;          // _div2 after the last fragment
	const int f = (int) a.n_frag - 1 - r;
	return f >= 0 ? f : -1;
}

__global__ __launch_bounds__ (256) void k_gate_frag (const mtr_gate_args a)
{
#pragma clang fp contract(off)
	__shared__ float   pw[64 + GATE_FPB];
	__shared__ int32_t sh_hist[2][MTR_HIST_LEN];
	__shared__ int32_t sh_cnt[4];
	const uint32_t s = blockIdx.y;
	const int tid = threadIdx.x;
	const int f0 = (int) blockIdx.x * GATE_FPB;
	const int nf = min ((int) a.n_frag - f0, GATE_FPB);
	if (nf <= 0) return;
	mtr_stream_state* const st = a.state + s;
	const float* const tp = a.tile_power + (size_t) s * a.n_tiles;
	const int div1_0 = st->div1, div2_0 = st->div2;
	const int f_calc = gate_f_calc (a, div2_0);
	for (int i = tid; i < 2 * MTR_HIST_LEN; i += 256) (&sh_hist[0][0])[i] = 0;
	if (tid < 4) sh_cnt[tid] = 0;
	for (int i = tid; i < 64 + nf; i += 256) {
		const int f = f0 - 64 + i;
		const float p = frag_power_of (a, st, tp, f);
		pw[i] = p;
		if (f >= f0 && a.frag_power) a.frag_power[(size_t) s * a.n_frag + f] = p;
	}
	__syncthreads ();
	float mxM = -INFINITY, mxS = -INFINITY;
	for (int i = tid; i < nf; i += 256) {
		const int f = f0 + i;
		float lm = addfrags (&pw[64 + i], 8), ls = addfrags (&pw[64 + i], 60);
		if (!isfinite (lm) || lm < -200.f) lm = -200.0f;
		if (!isfinite (ls) || ls < -200.f) ls = -200.0f;
		mxM = lm > mxM ? lm : mxM;
		mxS = ls > mxS ? ls : mxS;
		if (a.integr && f <= f_calc) {
			if ((div1_0 + f + 1) % 2 == 0)  hist_add (sh_hist[0], &sh_cnt[0], &sh_cnt[2], lm);
			if ((div2_0 + f + 1) % 10 == 0) hist_add (sh_hist[1], &sh_cnt[1], &sh_cnt[3], ls);
		}
	}
	for (int d = 32; d >= 1; d >>= 1) {
		mxM = fmaxf (mxM, __shfl_xor (mxM, d, 64));
		mxS = fmaxf (mxS, __shfl_xor (mxS, d, 64));
	}
	if ((tid & 63) == 0 && mxM > -INFINITY) {
		atomicMax (&a.max_scratch[2 * s], sortable (mxM));
		atomicMax (&a.max_scratch[2 * s + 1], sortable (mxS));
	}
	__syncthreads ();
	int32_t* const ghist = a.hist + (size_t) s * 2 * MTR_HIST_LEN;
	for (int i = tid; i < 2 * MTR_HIST_LEN; i += 256) {
		const int32_t c = (&sh_hist[0][0])[i];
		if (c) atomicAdd (&ghist[i], c);
	}
	if (tid == 0) {
		if (sh_cnt[0]) atomicAdd (&st->cnt_M, sh_cnt[0]);
		if (sh_cnt[1]) atomicAdd (&st->cnt_S, sh_cnt[1]);
		if (sh_cnt[2]) atomicAdd (&st->err_M, sh_cnt[2]);
		if (sh_cnt[3]) atomicAdd (&st->err_S, sh_cnt[3]);
	}
}

__global__ __launch_bounds__ (256) void k_gate_final (const mtr_gate_args a)
{
#pragma clang fp contract(off)
	__shared__ float   pw[192];                 // powers of fragments n_frag - 192 .. n_frag - 1, chronological
	__shared__ int32_t sh_hist[2][MTR_HIST_LEN];
	__shared__ int32_t sh_cnt[4];
	const uint32_t s = blockIdx.x;
	const int tid = threadIdx.x;
	mtr_stream_state* const st = a.state + s;
	const float* const tp = a.tile_power + (size_t) s * a.n_tiles;
	int32_t* const ghist = a.hist + (size_t) s * 2 * MTR_HIST_LEN;
	const int n = (int) a.n_frag;
	const int div1_0 = st->div1, div2_0 = st->div2;
	const int f_calc = gate_f_calc (a, div2_0);
	if (tid < 192) pw[tid] = frag_power_of (a, st, tp, n - 192 + tid);
	for (int i = tid; i < 2 * MTR_HIST_LEN; i += 256) (&sh_hist[0][0])[i] = ghist[i];
	if (tid < 4) sh_cnt[tid] = (tid == 0) ? st->cnt_M : (tid == 1) ? st->cnt_S : (tid == 2) ? st->err_M : st->err_S;
	__syncthreads ();
	if (f_calc >= 0) {
		if (tid < 64)       hist_calc_integ (sh_hist[0], sh_cnt[0], a.bin_power, &st->integ, &st->integ_thr, tid);
		else if (tid < 128) hist_calc_range (sh_hist[1], sh_cnt[1], a.bin_power, &st->rmin, &st->rmax, &st->rthr, tid - 64);
	}
	__syncthreads ();
	if (tid == 0) {
		// the fragments after f_calc (at most nine) and the values a getter sees
		float last_M = st->loud_M, last_S = st->loud_S;
		float mxM = -INFINITY, mxS = -INFINITY;
		for (int f = max (n - 10, 0); f < n; ++f) {               // f_calc is among them; the last one always is
			const float* newest = &pw[191 - (n - 1 - f)];
			float lm = addfrags (newest, 8), ls = addfrags (newest, 60);
			if (!isfinite (lm) || lm < -200.f) lm = -200.0f;
			if (!isfinite (ls) || ls < -200.f) ls = -200.0f;
			if (a.integr && f > f_calc) {
				if ((div1_0 + f + 1) % 2 == 0)  hist_add (sh_hist[0], &sh_cnt[0], &sh_cnt[2], lm);
				if ((div2_0 + f + 1) % 10 == 0) hist_add (sh_hist[1], &sh_cnt[1], &sh_cnt[3], ls);
			}
			if (f == n - 1) { last_M = lm; last_S = ls; }
			mxM = lm > mxM ? lm : mxM;
			mxS = ls > mxS ? ls : mxS;
		}
		const float bM = unsortable (a.max_scratch[2 * s]), bS = unsortable (a.max_scratch[2 * s + 1]);
		float max_M = st->max_M, max_S = st->max_S;
		max_M = bM > max_M ? bM : max_M;
		max_S = bS > max_S ? bS : max_S;
		a.max_scratch[2 * s] = sortable (-INFINITY);
		a.max_scratch[2 * s + 1] = sortable (-INFINITY);
		float acc = (n == 0) ? st->frpwr : 1e-30f;
		for (uint32_t j = a.tail_tile; j < a.n_tiles; ++j) acc += tp[j];
		st->frpwr = acc;
		st->loud_M = last_M; st->loud_S = last_S;
		st->max_M = max_M;   st->max_S = max_S;
		st->div1 = a.integr ? (div1_0 + n) % 2 : div1_0;
		st->div2 = a.integr ? (div2_0 + n) % 10 : div2_0;
		st->cnt_M = sh_cnt[0]; st->cnt_S = sh_cnt[1]; st->err_M = sh_cnt[2]; st->err_S = sh_cnt[3];
		if (a.fold_tp) mtr_fold_truepeak (st);
	}
	__syncthreads ();
	// only the bins the late inserts touched differ from the global histogram: write all back
	for (int i = tid; i < 2 * MTR_HIST_LEN; i += 256) ghist[i] = (&sh_hist[0][0])[i];
	if (tid < 64) st->ring[tid] = pw[128 + tid];
}

// One wave that does nothing for `us` microseconds (s_memrealtime: the 100 MHz constant clock), asleep most of the time.
// The deferred tail starts with it: see mtr_engine.hip (tail_delay_us).
__global__ void k_delay (uint32_t us)
{
	const uint64_t t0 = wall_clock64 ();
	const uint64_t ticks = (uint64_t) us * 100u;
	while (wall_clock64 () - t0 < ticks) __builtin_amdgcn_s_sleep (64);
}

int mtr_launch_delay (uint32_t us, void* stream)
{
	hipLaunchKernelGGL (k_delay, dim3 (1), dim3 (64), 0, (hipStream_t) stream, us);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

int mtr_launch_gate (const mtr_gate_args& a, void* stream)
{
	// many fragments per call and few streams: spread a stream over several workgroups
	if (a.n_frag >= 4 * GATE_FPB && a.max_scratch) {
		const uint32_t nb = (a.n_frag + GATE_FPB - 1) / GATE_FPB;
		hipLaunchKernelGGL (k_gate_frag, dim3 (nb, a.n_streams), dim3 (256), 0, (hipStream_t) stream, a);
		hipLaunchKernelGGL (k_gate_final, dim3 (a.n_streams), dim3 (256), 0, (hipStream_t) stream, a);
		return hipGetLastError () == hipSuccess ? 0 : -1;
	}
	// A deferred gate (a.polite_grid) runs beside the next call's fused kernel, whose one-wave workgroups own a SIMD each for
	// the whole call: two gate workgroups per CU — two 72-VGPR waves per SIMD, 19 KB of LDS — always leave room for one of
	// those (344 VGPRs of 512, 35 KB of 160), so the gate is launched as that many workgroups, each walking its share of the
	// streams, and cannot stand in the way of k_seg's placement whenever the two are dispatched together.
	const uint32_t grid = a.polite_grid ? (a.n_streams < a.polite_grid ? a.n_streams : a.polite_grid) : a.n_streams;
	hipLaunchKernelGGL (k_gate, dim3 (grid), dim3 (256), 0, (hipStream_t) stream, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

// ---- state initialisation --------------------------------------------------------------------

__global__ void k_state_init (mtr_stream_state* st, int32_t* hist, uint32_t n_streams, int what)
{
	const uint32_t s = blockIdx.x;
	if (s >= n_streams) return;
	mtr_stream_state* p = st + s;
	const int tid = threadIdx.x;
	if (what == MTR_INIT_ALL || what == MTR_INIT_INTEGR) {
		for (int i = tid; i < 2 * MTR_HIST_LEN; i += blockDim.x) hist[(size_t) s * 2 * MTR_HIST_LEN + i] = 0;
	}
	if (tid != 0) return;
	if (what == MTR_INIT_ALL) {
		// Ebu_r128_proc::reset  ebu_r128_proc.cc:176-189
		for (int i = 0; i < 8; ++i) p->kz[i] = 0;
		p->frpwr = 1e-30f;
		for (int i = 0; i < 64; ++i) p->ring[i] = 0;
		p->loud_M = p->loud_S = -200.0f;
		for (int c = 0; c < 2; ++c) {
			p->tp_call[c] = 0; p->tp_last[c] = 0; p->tp_hold[c] = 0;
			p->tpb_z1[c] = p->tpb_z2[c] = p->tpb_m[c] = p->tpb_p[c] = 0;
		}
	}
	if (what == MTR_INIT_ALL || what == MTR_INIT_INTEGR) {
		// integr_reset  ebu_r128_proc.cc:192-204
		p->max_M = p->max_S = -200.0f;
		p->integ = p->integ_thr = -200.0f;
		p->rmin = p->rmax = p->rthr = -200.0f;
		p->div1 = p->div2 = 0;
		p->cnt_M = p->cnt_S = p->err_M = p->err_S = 0;
	}
	if (what == MTR_INIT_TP) {
		for (int c = 0; c < 2; ++c) { p->tp_call[c] = 0; p->tp_last[c] = 0; p->tp_hold[c] = 0; p->tpb_m[c] = p->tpb_p[c] = 0; }
	}
}

int mtr_launch_state_init (mtr_stream_state* st, int32_t* hist, uint32_t n_streams, int what, void* stream)
{
	hipLaunchKernelGGL (k_state_init, dim3 (n_streams), dim3 (64), 0, (hipStream_t) stream, st, hist, n_streams, what);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

// ---- cross-stream aggregate (input of the multi-GPU all-reduce) ---------------------------------

#define AGG_SPB 8    /* streams per block */

__device__ __forceinline__ void atomic_max_f32 (float* p, float v)
{
	int* const ip = reinterpret_cast<int*> (p);
	int old = *ip;
	while (v > __int_as_float (old)) {
		const int seen = atomicCAS (ip, old, __float_as_int (v));
		if (seen == old) break;
		old = seen;
	}
}

__global__ void k_aggregate_init (int32_t* d_hist, float* d_max)
{
	for (int bin = threadIdx.x; bin < 2 * MTR_HIST_LEN; bin += blockDim.x) d_hist[bin] = 0;
	if (threadIdx.x < 4) d_max[threadIdx.x] = (threadIdx.x < 2) ? 0.f : -200.f;
}

// Block b folds streams [8 b, 8 b + 8) — eight independent loads per bin — and adds what is not zero to the job's
// histograms with integer atomics (most of a stream's 1502 bins are empty); maxima by compare-and-swap.  Integer sums and
// maxima do not depend on the order: the result is deterministic.  (The first version summed 64 streams per block
// serially and folded the partials in a second pass: 0.11 ms of latency for 49 MB.)
__global__ __launch_bounds__ (256) void k_aggregate (const mtr_stream_state* st, const int32_t* hist, uint32_t n_streams,
                                                     int32_t* d_hist, float* d_max)
{
	const uint32_t s0 = blockIdx.x * AGG_SPB;
	const int ns = (int) min ((uint32_t) AGG_SPB, n_streams - s0);
	const int32_t* const h0 = hist + (size_t) s0 * 2 * MTR_HIST_LEN;
	for (int bin = threadIdx.x; bin < 2 * MTR_HIST_LEN; bin += 256) {
		int32_t acc = 0;
#pragma unroll
		for (int i = 0; i < AGG_SPB; ++i) acc += i < ns ? h0[(size_t) i * 2 * MTR_HIST_LEN + bin] : 0;
		if (acc) atomicAdd (&d_hist[bin], acc);
	}
	if (threadIdx.x < 4) {
		float m = (threadIdx.x < 2) ? 0.f : -200.f;
		for (int i = 0; i < ns; ++i) {
			const mtr_stream_state& p = st[s0 + i];
			const float v = threadIdx.x == 0 ? p.tp_hold[0] : threadIdx.x == 1 ? p.tp_hold[1] : threadIdx.x == 2 ? p.max_M : p.max_S;
			m = v > m ? v : m;
		}
		atomic_max_f32 (&d_max[threadIdx.x], m);
	}
}

int mtr_launch_aggregate (const mtr_stream_state* st, const int32_t* hist, uint32_t n_streams, int32_t* d_hist, float* d_max, void* stream)
{
	hipLaunchKernelGGL (k_aggregate_init, dim3 (1), dim3 (256), 0, (hipStream_t) stream, d_hist, d_max);
	hipLaunchKernelGGL (k_aggregate, dim3 ((n_streams + AGG_SPB - 1) / AGG_SPB), dim3 (256), 0, (hipStream_t) stream,
	                    st, hist, n_streams, d_hist, d_max);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
