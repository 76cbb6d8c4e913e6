// mtr_dr14.hip — DR-14 (the dr14 plugins' dynamic-range meter) for a whole batch of tracks (gfx950).
//
// Replaces the dr_operation_mode part of dr14_run (src/dr14.c:394-412: per sample rms_sum += v*v and
// peak_cur = max (peak_cur, v), a window closed every n_sample_cnt + 1 samples) and dr14_calc_rms_score
// (src/dr14.c:283-352: silent windows dropped, the window's RMS into an 8000-bin histogram of 0.01 dB, the
// two highest window peaks, the score = RMS of the loudest 20 % of the windows).  The true-peak and K-meter
// halves of the plugin are MTR_METER_TPBALLIST and host plumbing (lv2_dr14.c).
//
// A window is 3 s: the per-sample part is a pure streaming reduction (HBM-bound), so the call is cut at the
// window boundaries and one workgroup reduces one (stream, piece); a second, tiny kernel walks each stream's
// pieces in order, closes the windows and, at the end of the call, evaluates the score from the histogram
// exactly as the reference does after its last window.  The reference sums the squares sequentially in f32;
// here the partial sums are double (closer to the true value): a window's RMS can land in the neighbouring
// 0.01 dB bin, which moves the score by at most that — tests/test_gpu_dr14.py states +-0.02 dB.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float coeff_to_db (float c) { return c < .0001f ? -80.f : 20.f * log10f (c); }   // dr14.c:235-238
__device__ __forceinline__ float db_to_coeff (float db) { return db <= -80.f ? 0.f : powf (10.f, 0.05f * db); } // dr14.c:240-243

// piece i of a call: frames [b (i), b (i + 1)), b (0) = 0, b (i) = min (N, e0 + (i - 1) W), e0 = W - c_in
__device__ __forceinline__ uint64_t piece_start (uint32_t i, uint64_t e0, uint64_t W, uint64_t N)
{
	if (i == 0) return 0;
	const uint64_t b = e0 + (uint64_t) (i - 1) * W;
	return b < N ? b : N;
}

template <int C>
__global__ __launch_bounds__ (NT) void k_dr14_sums (const mtr_dr14_args a)
{
	const uint32_t piece = blockIdx.x, s = blockIdx.y;
	const uint64_t b0 = piece_start (piece, a.e0, a.window, a.n_frames), b1 = piece_start (piece + 1, a.e0, a.window, a.n_frames);
	const float* const src = a.audio + (size_t) s * a.stride * C;
	double sl = 0, sr = 0;
	float pl = 0.f, pr = 0.f;                                  // dr14.c:401: max (peak_cur, v), signed v, from 0
	if (C == 2) {
		// 16-byte loads: two frames per lane; a piece that starts on an odd frame of the buffer gives its first
		// frame (and one that ends on an odd frame its last) to a single lane
		const uint64_t odd = ((size_t) s * a.stride + b0 + (reinterpret_cast<size_t> (a.audio) >> 3)) & 1;
		const uint64_t h0 = b0 + (odd && b0 < b1 ? 1 : 0);
		const uint64_t np = (b1 - h0) >> 1;                    // whole pairs
		auto one = [&] (uint64_t f) {
			const float2 v = *reinterpret_cast<const float2*> (src + 2 * f);
			sl += (double) (v.x * v.x); sr += (double) (v.y * v.y);
			pl = fmaxf (pl, v.x); pr = fmaxf (pr, v.y);
		};
		if (threadIdx.x == 0 && h0 > b0) one (b0);
		if (threadIdx.x == 1 && h0 + 2 * np < b1) one (b1 - 1);
		const float4* const p4 = reinterpret_cast<const float4*> (src + 2 * h0);
		for (uint64_t i = threadIdx.x; i < np; i += NT) {
			const float4 v = p4[i];
			sl += (double) (v.x * v.x) + (double) (v.z * v.z); sr += (double) (v.y * v.y) + (double) (v.w * v.w);
			pl = fmaxf (pl, fmaxf (v.x, v.z)); pr = fmaxf (pr, fmaxf (v.y, v.w));
		}
	} else {
		for (uint64_t f = b0 + threadIdx.x; f < b1; f += NT) {
			const float v = src[f];
			sl += (double) (v * v);
			pl = fmaxf (pl, v);
		}
	}
	__shared__ double sh_s[2][NT / 64];
	__shared__ float sh_p[2][NT / 64];
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		sl += __shfl_xor (sl, d, 64); sr += __shfl_xor (sr, d, 64);
		pl = fmaxf (pl, __shfl_xor (pl, d, 64)); pr = fmaxf (pr, __shfl_xor (pr, d, 64));
	}
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (lane == 0) { sh_s[0][wid] = sl; sh_s[1][wid] = sr; sh_p[0][wid] = pl; sh_p[1][wid] = pr; }
	__syncthreads ();
	if (threadIdx.x == 0) {
		for (int w = 1; w < NT / 64; ++w) { sl += sh_s[0][w]; sr += sh_s[1][w]; pl = fmaxf (pl, sh_p[0][w]); pr = fmaxf (pr, sh_p[1][w]); }
		const size_t o = ((size_t) s * a.n_pieces + piece) * 2;
		a.piece_sum[o] = sl; a.piece_sum[o + 1] = sr;
		a.piece_peak[o] = pl; a.piece_peak[o + 1] = pr;
	}
}

// one WAVE per stream: lane 0 does the window bookkeeping of dr14_calc_rms_score over this call's pieces (a
// handful of windows), all 64 lanes look for the occupied histogram bins of the score (one thread walking
// 8000 bins per channel took 0.37 ms for 8192 streams)
template <int C>
__global__ __launch_bounds__ (64) void k_dr14_windows (const mtr_dr14_args a)
{
	const uint32_t s = blockIdx.x;
	const int lane = threadIdx.x;
	__shared__ bool st_closed;
	if (lane == 0) {
	mtr_dr14_state* const st = a.state + s;
	uint32_t* const hist = a.hist + (size_t) s * C * MTR_DR_HISTBINS;
	float rs[2] = { st->rms_sum[0], st->rms_sum[1] };
	float pk[2] = { st->peak_cur[0], st->peak_cur[1] };
	const float nsc = (float) (a.window - 1);                  // n_sample_cnt
	bool closed = false;
	for (uint32_t i = 0; i < a.n_pieces; ++i) {
		const size_t o = ((size_t) s * a.n_pieces + i) * 2;
		for (int c = 0; c < C; ++c) {
			rs[c] = (float) ((double) rs[c] + a.piece_sum[o + c]);
			pk[c] = fmaxf (pk[c], a.piece_peak[o + c]);
		}
		if (i >= a.n_windows) break;                           // the last piece leaves an open window
		// ---- a window closes (dr14.c:283-352) ----
		bool silent = true;
		for (int c = 0; c < C; ++c) if (rs[c] > 1e-9 * (double) nsc) silent = false;    // :290 (float > double product)
		if (!silent) {
			st->num_fragments++;
			closed = true;
			for (int c = 0; c < C; ++c) {
				const float rms = sqrtf (2.f * rs[c] / nsc);
				int bin = (int) rintf (100.f * (80.f + coeff_to_db (rms))) - 1;
				if (bin >= MTR_DR_HISTBINS) bin = MTR_DR_HISTBINS - 1;
				if (bin > 0) hist[c * MTR_DR_HISTBINS + bin]++;
				if (pk[c] >= st->peak_hist[c][0]) { st->peak_hist[c][1] = st->peak_hist[c][0]; st->peak_hist[c][0] = pk[c]; }
				else if (pk[c] > st->peak_hist[c][1]) st->peak_hist[c][1] = pk[c];
				pk[c] = 0.f;
			}
		}
		for (int c = 0; c < C; ++c) rs[c] = 0.f;               // silent windows keep their peak (dr14.c:296-301)
	}
	for (int c = 0; c < C; ++c) { st->rms_sum[c] = rs[c]; st->peak_cur[c] = pk[c]; }
	st_closed = closed;
	}
	// (lane 0's stores to the histogram are visible to the wave after this)
	__threadfence_block ();
	const bool closed = __builtin_amdgcn_readfirstlane ((int) st_closed) != 0;
	if (!closed) return;
	// the score after the call's last window: bins from the top, 64 at a time, in descending order
	mtr_dr14_state* const st = a.state + s;
	const uint32_t* const hist = a.hist + (size_t) s * C * MTR_DR_HISTBINS;
	const uint32_t nf = st->num_fragments;
	const float cutf = floorf (nf / 5.0f);
	const uint32_t m_cut = cutf > 1 ? (uint32_t) cutf : 1;
	for (int c = 0; c < C; ++c) {
		uint32_t n_cut = 0;
		float score = 0.f;
		if (nf > 2) {
			for (int top = MTR_DR_HISTBINS - 1; top > 0 && n_cut < m_cut; top -= 64) {
				const int b = top - lane;                          // lane 0 holds the highest bin of the chunk
				const uint32_t bc = b > 0 ? hist[c * MTR_DR_HISTBINS + b] : 0u;
				unsigned long long occupied = __ballot (bc != 0);
				while (occupied && n_cut < m_cut) {                // wave-uniform walk over the occupied bins
					const int l = __ffsll ((long long) occupied) - 1;
					occupied &= occupied - 1;
					const uint32_t cnt = (uint32_t) __builtin_amdgcn_readlane ((int) bc, l);
					const float cd = db_to_coeff ((float) ((top - l - MTR_DR_HISTBINS + 1) / 100.0));
					score += cd * cd * (float) cnt;
					n_cut += cnt;
				}
			}
		}
		if (lane == 0) {
			st->m_rms[c] = n_cut > 0 ? coeff_to_db (sqrtf (score / n_cut)) : -81.f;
			st->m_peak[c] = nf > 2 ? coeff_to_db (st->peak_hist[c][1]) : -81.f;
		}
	}
}

}  // namespace

int mtr_launch_dr14 (const mtr_dr14_args& a, void* stream)
{
	hipStream_t st = (hipStream_t) stream;
	if (a.n_channels == 2) {
		hipLaunchKernelGGL (k_dr14_sums<2>, dim3 (a.n_pieces, a.n_streams), dim3 (NT), 0, st, a);
		hipLaunchKernelGGL (k_dr14_windows<2>, dim3 (a.n_streams), dim3 (64), 0, st, a);
	} else {
		hipLaunchKernelGGL (k_dr14_sums<1>, dim3 (a.n_pieces, a.n_streams), dim3 (NT), 0, st, a);
		hipLaunchKernelGGL (k_dr14_windows<1>, dim3 (a.n_streams), dim3 (64), 0, st, a);
	}
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
