// mtr_fused2.hip — fused K-weighting + 4x true-peak kernel, wave-specialised layout (gfx950).
//
// Replaces the same reference loops as mtr_fused.hip (Ebu_r128_proc::detect_process
// ebumeter/ebu_r128_proc.cc:302-337; Resampler::process zita-resampler/resampler.cc:211-235 +
// TruePeakdsp::process_max jmeters/truepeakdsp.cc:101-124), reading each stereo frame from HBM once.
//
// What the first layout (one wave = one stream segment, mtr_fused.hip) taught on MI355X
// (profiles/r01_*): the kernel is fp32-VALU bound (v_pk_fma_f32 issues at the chip's peak rate,
// ~32.5 T instr/s), the K-weighting scan is a fixed cost per tile, and a 39-frame lane run
// (one tile = one 50 ms fragment) needs 20 KB of LDS per wave, which caps occupancy at 2 waves
// per SIMD.  Here one WORKGROUP of four waves owns a (stream, time segment) and shares the tile:
//
//   wave 3  "loader + K-filter":  streams tile j+1 from HBM straight into the other LDS buffer with
//           global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip, lane-linear destination which is
//           exactly the frame order we want), then runs the K-weighting of tile j from LDS:
//           per-lane run of K = 39 frames from zero state, 6-step wave scan with (A^K)^(2^d),
//           second pass from the true state accumulating y^2, carry to the next tile in registers.
//   waves 0..2 "FIR":  wave w computes the three non-trivial polyphase branches for the R = 13
//           outputs [l*39 + 13w, l*39 + 13w + 13) of every lane run l: 13 x 3 x 2 accumulators in
//           registers, the 13+47 window frames streamed through them once, taps in SGPRs
//           (groups of 16 taps per branch so they fit the scalar file).
//   One s_barrier per tile swaps the two LDS buffers.  LDS per workgroup = 2 x (tile + 48 frames)
//   = 39 KB at 48 kHz -> 4 workgroups = 16 waves per CU, 4 per SIMD, at <= 128 VGPRs.
//
// Lane stride in LDS is K = 39 slots (odd) -> every ds_read_b64 is bank-conflict free.
// The loader wave has ~30% less work than a FIR wave; that slack is what hides HBM latency.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

#include "mtr_wave.h"

__constant__ float c_fir2[3][48];   // 48-tap kernels of phases 1..3, index 0 = oldest window sample
__constant__ float c_firs[3][24];   // mirror-symmetric form: P = (g1[i]+g1[47-i])/2, M = (g1[i]-g1[47-i])/2, Q = g2[i]

int mtr_fused2_upload_taps (const float* g144)
{
	float pmq[3][24];
	for (int i = 0; i < 24; ++i) {
		pmq[0][i] = (float) (((double) g144[i] + (double) g144[47 - i]) * 0.5);
		pmq[1][i] = (float) (((double) g144[i] - (double) g144[47 - i]) * 0.5);
		pmq[2][i] = g144[48 + i];
	}
	if (hipMemcpyToSymbol (HIP_SYMBOL (c_firs), pmq, sizeof (pmq)) != hipSuccess) return -1;
	return hipMemcpyToSymbol (HIP_SYMBOL (c_fir2), g144, sizeof (float) * 144) == hipSuccess ? 0 : -1;
}

__device__ __forceinline__ v2f shfl_up2 (v2f v, int d)
{
	return v2f{__shfl_up (v.x, d, 64), __shfl_up (v.y, d, 64)};
}
__device__ __forceinline__ v2f bcast2 (v2f v, int l)
{
	return v2f{__shfl (v.x, l, 64), __shfl (v.y, l, 64)};
}
__device__ __forceinline__ float wave_sum (float v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor (v, d, 64);
	return v;
}
__device__ __forceinline__ float wave_max (float v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v = fmaxf (v, __shfl_xor (v, d, 64));
	return v;
}
__device__ __forceinline__ v2f scrub (v2f v)
{
	return v2f{isfinite (v.x) ? v.x : 0.f, isfinite (v.y) ? v.y : 0.f};
}

// One K-weighting step for both channels (ebu_r128_proc.cc:321-326).  Same arithmetic as the
// reference, associated so that the loop-carried chains are short: x depends on the previous x
// through one FMA, y on the previous y through add + 2 FMA.
#define KW_STEP(p, y)                                   \
	{                                                   \
		v2f t_ = (p) + 1e-15f;                          \
		t_ = t_ - b2 * z2;                              \
		const v2f x_ = t_ - b1 * z1;                    \
		v2f u_ = a1 * z1;                               \
		u_ = u_ + a2 * z2;                              \
		u_ = u_ - c4 * z4;                              \
		u_ = u_ - c3 * z3;                              \
		y = a0 * x_ + u_;                               \
		z2 = z1; z1 = x_; z4 += z3; z3 += y;            \
	}

// ---- 4x interpolator, dense form: three 48-tap branches, R outputs per register tile ----------
// xs points at the LDS slot of frame (m0 - 48), i.e. window element j of output r is xs[1 + r + j].
// Taps go through SGPRs in groups of 16 per branch: all 144 at once do not fit the scalar file
// (the compiler then parks them in VGPR lanes, one v_readlane per FMA).
template <int R>
__device__ __forceinline__ void fir_dense (const v2f* xs, int nvalid, float& pk_l, float& pk_r)
{
	constexpr int TG = 16;
	const v2f* const xw = xs + 1;
	v2f acc[R][3];
#pragma unroll
	for (int r = 0; r < R; ++r) { acc[r][0] = 0; acc[r][1] = 0; acc[r][2] = 0; }
#pragma unroll 1
	for (int g = 0; g < 48; g += TG) {
		float t0_[TG], t1_[TG], t2_[TG];
#pragma unroll
		for (int k = 0; k < TG; ++k) { t0_[k] = c_fir2[0][g + k]; t1_[k] = c_fir2[1][g + k]; t2_[k] = c_fir2[2][g + k]; }
		const v2f* const xg = xw + g;
#pragma unroll
		for (int j = 0; j < R + TG - 1; ++j) {
			const v2f x = xg[j];
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const int k = j - r;
				if (k >= 0 && k < TG) {
					acc[r][0] += t0_[k] * x;
					acc[r][1] += t1_[k] * x;
					acc[r][2] += t2_[k] * x;
				}
			}
		}
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const v2f x0 = xw[23 + r];              // phase 0 = identity: x[n - 24]
		const bool ok = r < nvalid;
		const float ml = fmaxf (fmaxf (fabsf (acc[r][0].x), fabsf (acc[r][1].x)), fmaxf (fabsf (acc[r][2].x), fabsf (x0.x)));
		const float mr = fmaxf (fmaxf (fabsf (acc[r][0].y), fabsf (acc[r][1].y)), fmaxf (fabsf (acc[r][2].y), fabsf (x0.y)));
		pk_l = fmaxf (pk_l, ok ? ml : 0.f);
		pk_r = fmaxf (pk_r, ok ? mr : 0.f);
	}
}

// ---- 4x interpolator, mirror-symmetric form -------------------------------------------------------
// The polyphase table is symmetric: branch 3 is branch 1 reversed (g3[i] = g1[47-i]) and branch 2 is
// its own mirror.  With a_i = w[i], b_i = w[47-i] (i < 24), s_i = a_i + b_i, d_i = a_i - b_i:
//     y1 = sum P_i s_i + sum M_i d_i,   y3 = sum P_i s_i - sum M_i d_i,   y2 = sum Q_i s_i
// where P = (g1[i] + g1[47-i]) / 2, M = (g1[i] - g1[47-i]) / 2, Q = g2[i]:  2 adds + 3 FMAs per
// (output, i) instead of 6 FMAs -> 120 instead of 144 packed operations per stereo frame.
// For output r and i = G*g + k:  a = xs[1 + r + i],  b = xs[48 + r - i].
// MASK = false: every one of the R outputs is a true output (the loader stages R look-ahead frames
// past the tile end, so outputs beyond `nvalid` are simply the next tile's first outputs computed
// early — harmless under max).  MASK = true only for the last tile of a call.
template <int R, bool MASK>
__device__ __forceinline__ void fir_sym (const v2f* xs, int nvalid, float& pk_l, float& pk_r)
{
	constexpr int G = 6;                         // mirror pairs per SGPR tap group
	v2f aS[R], aD[R], aQ[R];
#pragma unroll
	for (int r = 0; r < R; ++r) { aS[r] = 0; aD[r] = 0; aQ[r] = 0; }
#pragma unroll 1
	for (int g = 0; g < 24; g += G) {
		float tp[G], tm[G], tq[G];
#pragma unroll
		for (int k = 0; k < G; ++k) { tp[k] = c_firs[0][g + k]; tm[k] = c_firs[1][g + k]; tq[k] = c_firs[2][g + k]; }
		const v2f* const xl = xs + 1 + g;        // a-side: xl[r + k]
		const v2f* const xr = xs + 48 - g - (G - 1);   // b-side: xr[r + (G-1) - k]
		v2f L[R + G - 1], B[R + G - 1];
#pragma unroll
		for (int j = 0; j < R + G - 1; ++j) { L[j] = xl[j]; B[j] = xr[j]; }
#pragma unroll
		for (int r = 0; r < R; ++r) {
#pragma unroll
			for (int k = 0; k < G; ++k) {
				const v2f sv = L[r + k] + B[r + G - 1 - k];
				const v2f dv = L[r + k] - B[r + G - 1 - k];
				aS[r] += tp[k] * sv;
				aD[r] += tm[k] * dv;
				aQ[r] += tq[k] * sv;
			}
		}
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const v2f x0 = xs[24 + r];               // phase 0 = identity: x[n - 24]
		const v2f y1 = aS[r] + aD[r], y3 = aS[r] - aD[r];
		if (MASK) {
			const bool ok = r < nvalid;
			const float ml = fmaxf (fmaxf (fabsf (y1.x), fabsf (y3.x)), fmaxf (fabsf (aQ[r].x), fabsf (x0.x)));
			const float mr = fmaxf (fmaxf (fabsf (y1.y), fabsf (y3.y)), fmaxf (fabsf (aQ[r].y), fabsf (x0.y)));
			pk_l = fmaxf (pk_l, ok ? ml : 0.f);
			pk_r = fmaxf (pk_r, ok ? mr : 0.f);
		} else {
			// two v_max3_f32 per channel (|.| is a free source modifier)
			pk_l = fmaxf (fmaxf (pk_l, fabsf (y1.x)), fabsf (y3.x));
			pk_l = fmaxf (fmaxf (pk_l, fabsf (aQ[r].x)), fabsf (x0.x));
			pk_r = fmaxf (fmaxf (pk_r, fabsf (y1.y)), fabsf (y3.y));
			pk_r = fmaxf (fmaxf (pk_r, fabsf (aQ[r].y)), fabsf (x0.y));
		}
	}
}

template <int K, int R, bool EBU, bool TP, bool SYM, bool ROT>
__global__ __launch_bounds__ (TP ? 256 : 64) void k_fused2 (const mtr_fused_args a)
{
	static_assert (K == 3 * R && (K & 1) == 1, "three FIR waves x R outputs per lane run; odd lane stride");
	constexpr int LDR = TP ? 3 : 0;             // the loader / K-filter wave

	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	const int lane = threadIdx.x & 63;
	const int wid  = threadIdx.x >> 6;
	const int NB   = (int) a.buf_slots;         // 8-byte slots per LDS buffer (even)
	v2f* const buf0 = reinterpret_cast<v2f*> (smem);

	const uint32_t unit = blockIdx.x;
	const uint32_t s = unit / a.n_segs;
	const uint32_t q = unit - s * a.n_segs;

	const v2f* const src  = reinterpret_cast<const v2f*> (a.audio) + (size_t) s * a.stride;
	const v2f* const hist = reinterpret_cast<const v2f*> (a.hist) + (size_t) s * MTR_FIR_HALO;
	mtr_stream_state* const st = a.state + s;
	const bool src_even = ((((size_t) s * a.stride) & 1) == 0) && ((reinterpret_cast<size_t> (a.audio) & 15) == 0);

	const float a0 = a.a0, a1 = a.a1, a2 = a.a2, b1 = a.b1, b2 = a.b2, c3 = a.c3, c4 = a.c4;

	const uint32_t jt0 = a.seg_tile[q], jt1 = a.seg_tile[q + 1];
	const int64_t seg_start = a.tile_start[jt0];
	const int nwarm = (EBU && q > 0) ? (int) a.warm_tiles : 0;
	const int ntile = (int) (jt1 - jt0);
	constexpr int LT = 64 * K;

	// tile jj -> (first frame, length); warm-up tiles are full tiles before the segment
	auto tile_of = [&] (int jj, int64_t& t0, int& len) {
		if (jj < 0) { t0 = seg_start + (int64_t) jj * LT; len = LT; }
		else        { t0 = a.tile_start[jt0 + jj]; len = (int) (a.tile_start[jt0 + jj + 1] - (uint32_t) t0); }
	};

	// Stage frames [t0 - 48, t0 + len) into `buf`: slot i <-> frame t0 - 48 + i (frame t0 at slot 48).
	auto stage = [&] (int jj, v2f* buf) {
		int64_t t0; int len;
		tile_of (jj, t0, len);
		const int nslot = len + 48 + R;           // R look-ahead frames: the FIR epilogue needs no masks
		// 16-byte pairs need an even first frame; the one tile that ends on an odd final frame of the
		// call cannot fetch that frame as half of a pair without reading past the stream: plain path.
		const bool tail_odd = (t0 + len + R >= (int64_t) a.n_frames) && (a.n_frames & 1);
		if (t0 >= 48 && src_even && ((t0 & 1) == 0) && !tail_odd) {
			// LDS-DMA, 16 bytes (two frames) per lane per instruction, 1 KiB per wave-instruction.
			// Source addresses past the end of the call are clamped (those slots are never consumed).
			const int64_t fmax = (int64_t) a.n_frames - 2;
			for (int i = 0; i < nslot; i += 128) {
				int64_t f = t0 - 48 + i + 2 * lane;
				f = f > fmax ? (fmax & ~(int64_t) 1) : f;
				if (i + 2 * lane < nslot)       // lanes past the tile write nothing (the carry slots live up there)
				__builtin_amdgcn_global_load_lds ((const __attribute__ ((address_space (1))) void*) (src + f),
				                                  (__attribute__ ((address_space (3))) void*) (buf + i), 16, 0, 0);
			}
		} else {
			for (int i = lane; i < nslot; i += 64) {
				const int64_t f = t0 - 48 + i;
				v2f v = 0;
				if (f >= 0) v = src[f < (int64_t) a.n_frames ? f : (int64_t) a.n_frames - 1];
				else if (f >= -MTR_FIR_HALO) v = hist[MTR_FIR_HALO + f];
				buf[i] = v;
			}
		}
	};

	// Workgroup barrier that is legal in wave-uniform divergent code: every wave of the block
	// reaches exactly one of these per tile (the two role loops below have identical trip counts).
	auto block_barrier = [] () {
		__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_s_barrier ();
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
	};

	// ---- K-filter of one tile (held in `cur`) by the calling wave; k1..k4 = carried state in / out ----
	auto kfilter_tile = [&] (const v2f* cur, int jj, int len, bool warm, v2f& k1, v2f& k2, v2f& k3, v2f& k4) {
		const int run0 = lane * K;
		const int rl = min (max (len - run0, 0), K);
		const v2f* const xr = cur + 48 + run0;
		// Per-lane matrices of the scan's two row-broadcast steps: loaded per tile (consumed after pass 1)
		// so they hold no VGPRs while this wave is in a FIR role; the empty asm keeps LLVM from
		// hoisting the (loop-invariant) loads out of the tile loop.
		const float* pw = a.scan_m + 96 + 4 * K + 4;
		asm volatile ("" : "+s"(pw));
		// wave-uniform tables through the constant address space: scalar loads the stores of this
		// loop cannot clobber (see mtr_kw.hip)
		typedef const __attribute__ ((address_space (4))) float* cfloat_p;
		const cfloat_p CM = (cfloat_p) a.scan_m;
		mtrw::RowMats rm;
		rm.load (pw, lane);

		// pass 1: end state of this lane's run from a zero start state.  Only the end state is
		// needed, and it is a linear functional of the K inputs, e = sum_n A^(K-1-n) B x_n (+ the
		// constant response to the 1e-15f bias): 4 FMAs per frame with wave-uniform coefficients
		// instead of the 11-operation recurrence.  A partial run (the last active lane) is
		// skipped: nothing to its right consumes its end state.  Lane 0 adds A^K * carried state.
		v2f z1 = 0, z2 = 0, z3 = 0, z4 = 0;
		if (rl == K) {
			const cfloat_p F = CM + 96;
			z1 = F[4 * K + 0]; z2 = F[4 * K + 1]; z3 = F[4 * K + 2]; z4 = F[4 * K + 3];
#pragma unroll 1
			for (int g = 0; g < K; g += R) {          // R frames at a time: 4R coefficients in SGPRs
#pragma unroll
				for (int j = 0; j < R; ++j) {
					const v2f x = xr[g + j];
					const cfloat_p Fn = F + 4 * (g + j);
					z1 += Fn[0] * x; z2 += Fn[1] * x; z3 += Fn[2] * x; z4 += Fn[3] * x;
				}
			}
		}
		if (lane == 0) {
			const cfloat_p M = CM;
			z1 += M[0] * k1 + M[1] * k2;
			z2 += M[4] * k1 + M[5] * k2;
			z3 += M[8] * k1 + M[9] * k2 + M[10] * k3 + M[11] * k4;
			z4 += M[12] * k1 + M[13] * k2 + M[14] * k3 + M[15] * k4;
		}

		// wave scan on the DPP path (mtr_wave.h): z_l <- sum_{j<=l} (A^K)^(l-j) e_j
		mtrw::scan (z1, z2, z3, z4, CM, rm);

		if (warm) {
			k1 = mtrw::pick (z1, 63); k2 = mtrw::pick (z2, 63); k3 = mtrw::pick (z3, 63); k4 = mtrw::pick (z4, 63);
		} else {
			// pass 2: from the true start state (end state of the lane to the left), sum y^2
			z1 = mtrw::from_left (z1); z2 = mtrw::from_left (z2); z3 = mtrw::from_left (z3); z4 = mtrw::from_left (z4);
			if (lane == 0) { z1 = k1; z2 = k2; z3 = k3; z4 = k4; }
			v2f sj = 0;
			for (int n = 0; n < rl; ++n) { v2f y; KW_STEP (xr[n], y); sj += y * y; }
			const float sl = mtrw::sum63 (sj.x), sr = mtrw::sum63 (sj.y);
			if (lane == 0) a.tile_power[(size_t) s * a.n_tiles + jt0 + jj] = a.gain_l * sl + a.gain_r * sr;
			const int last = (len - 1) / K;       // the lane holding the state after the last frame
			k1 = mtrw::pick (z1, last); k2 = mtrw::pick (z2, last); k3 = mtrw::pick (z3, last); k4 = mtrw::pick (z4, last);
		}
		// ebu_r128_proc.cc:331-334: non-finite states are dropped at block ends
		k1 = scrub (k1); k2 = scrub (k2); k3 = scrub (k3); k4 = scrub (k4);
	};

	// ---- interpolator sub-run `w` (0..2) of one tile: outputs m = lane*K + R*w + r.  Window element j of
	//      output r is frame m - 47 + j, i.e. slot lane*K + R*w + 1 + r + j (frame t0 sits at slot 48) ----
	auto fir_tile = [&] (const v2f* cur, int jj, int w, float& pk_l, float& pk_r) {
		const int m0 = lane * K + R * w;
		const int len = (int) (a.tile_start[jt0 + jj + 1] - a.tile_start[jt0 + jj]);
		const int rlw = min (max (len - m0, 0), R);        // valid outputs of this lane's register tile
		if (rlw <= 0) return;
		if (SYM) {
			if (a.tile_start[jt0 + jj + 1] == (uint32_t) a.n_frames) {
				// last tile of the call: frames past its end do not exist yet
				fir_sym<7, true> (cur + m0, min (rlw, 7), pk_l, pk_r);
				if (rlw > 7) fir_sym<6, true> (cur + m0 + 7, rlw - 7, pk_l, pk_r);
			} else {
				fir_sym<7, false> (cur + m0, 7, pk_l, pk_r);
				// pin the first tile's maxima here: otherwise LLVM sinks its epilogue below the
				// second tile's loop and keeps 42 accumulator registers alive across it
				asm volatile ("" : "+v"(pk_l), "+v"(pk_r));
				fir_sym<6, false> (cur + m0 + 7, 6, pk_l, pk_r);
			}
		} else {
			fir_dense<R> (cur + m0, rlw, pk_l, pk_r);
		}
	};

	if (ROT) {
		// =================== rotating roles: wave (tile + wid) & 3 == 3 loads + K-filters, the other
		// three interpolate.  Over four tiles every wave does the same work, so the four SIMDs of a
		// CU stay evenly loaded however the dispatcher placed the workgroup's waves.  The carried
		// K-filter state travels through four LDS slots at the top of buffer 0.
		v2f* const kst = buf0 + NB - 4;
		// Exact peak pruning (a.prune): every interpolated value obeys |y_ph(n)| <= L1_ph * max|x| over its
		// 48-frame window, so a tile whose max|x| (halo and look-ahead included) times the largest L1
		// norm cannot beat the running peak of this (stream, segment) needs no interpolation at all —
		// the maximum is unchanged bit for bit.  The loader/K-filter wave scans tile j+1's max|x| in
		// its slack once the DMA has landed; the result and the shared running peak live in three LDS
		// slots at the top of buffer 1.
		v2f* const tmaxv = buf0 + 2 * NB - 4;                 // [2] by tile parity
		uint32_t* const gpk = reinterpret_cast<uint32_t*> (buf0 + 2 * NB - 2);   // running peak L, R (float bits)
		const bool prune = TP && a.prune;
		auto scan_tile = [&] (int jj, const v2f* buf, int par) {
			int64_t t0; int len;
			tile_of (jj, t0, len);
			const int nslot = len + 48 + R;
			float ml = 0.f, mr = 0.f;
			const int i0 = 1 + lane * K, i1 = min (i0 + K, nslot);
			for (int i = i0; i < i1; ++i) { const v2f x = buf[i]; ml = fmaxf (ml, fabsf (x.x)); mr = fmaxf (mr, fabsf (x.y)); }
			ml = wave_max (ml); mr = wave_max (mr);
			if (lane == 0) tmaxv[par] = v2f{ml, mr};
		};
		uint32_t n_done = 0, n_skip = 0;
		float pk_l = 0.f, pk_r = 0.f;
		if (wid == 0 && lane == 0) { gpk[0] = 0u; gpk[1] = 0u; }
		if (wid == 3) {
			v2f k1 = 0, k2 = 0, k3 = 0, k4 = 0;
			if (EBU && q == 0) {
				k1 = v2f{st->kz[0], st->kz[1]}; k2 = v2f{st->kz[2], st->kz[3]};
				k3 = v2f{st->kz[4], st->kz[5]}; k4 = v2f{st->kz[6], st->kz[7]};
			}
			if (lane == 0) { kst[0] = k1; kst[1] = k2; kst[2] = k3; kst[3] = k4; }
			stage (-nwarm, buf0);
			asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
			if (prune) scan_tile (-nwarm, buf0, 0);
		}
		block_barrier ();
		for (int jj = -nwarm; jj < ntile; ++jj) {
			const int it = jj + nwarm;
			v2f* const cur = buf0 + (it & 1) * NB;
			v2f* const nxt = buf0 + ((it & 1) ^ 1) * NB;
			const int role = (wid + it) & 3;
			if (role == 3) {
				if (jj + 1 < ntile) stage (jj + 1, nxt);
				if (EBU) {
					int64_t t0; int len;
					tile_of (jj, t0, len);
					v2f k1 = kst[0], k2 = kst[1], k3 = kst[2], k4 = kst[3];
					kfilter_tile (cur, jj, len, jj < 0, k1, k2, k3, k4);
					if (lane == 0) { kst[0] = k1; kst[1] = k2; kst[2] = k3; kst[3] = k4; }
				}
				asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
				if (prune && jj + 1 < ntile) scan_tile (jj + 1, nxt, (it & 1) ^ 1);
			} else if (TP && jj >= 0) {
				bool skip = false;
				if (prune) {
					// 2.56833 = the largest L1 norm of the three branches (phase 2); the margin covers
					// fp32 rounding of the bound and of the accumulated sums
					const float LB = 2.5684f * 1.0001f;
					const v2f tm = tmaxv[it & 1];
					const float gl = __uint_as_float (gpk[0]), gr = __uint_as_float (gpk[1]);
					skip = (LB * tm.x <= gl) && (LB * tm.y <= gr);
				}
				n_done++;
				if (skip) {
					n_skip++;
				} else {
					const float ol = pk_l, orr = pk_r;
					fir_tile (cur, jj, role, pk_l, pk_r);
					if (prune && __any ((pk_l > ol) || (pk_r > orr))) {
						const float wl = wave_max (pk_l), wr = wave_max (pk_r);
						if (lane == 0) { atomicMax (&gpk[0], __float_as_uint (wl)); atomicMax (&gpk[1], __float_as_uint (wr)); }
					}
				}
			}
			block_barrier ();
		}
		if (prune && lane == 0 && a.prune_stats) {
			atomicAdd (&a.prune_stats[0], n_done);
			atomicAdd (&a.prune_stats[1], n_skip);
		}
		if (EBU && q == a.n_segs - 1 && wid == 0 && lane == 0) {
			st->kz[0] = kst[0].x; st->kz[1] = kst[0].y; st->kz[2] = kst[1].x; st->kz[3] = kst[1].y;
			st->kz[4] = kst[2].x; st->kz[5] = kst[2].y; st->kz[6] = kst[3].x; st->kz[7] = kst[3].y;
		}
		if (TP) {
			pk_l = wave_max (pk_l);
			pk_r = wave_max (pk_r);
			if (lane == 0) {
				atomicMax (&st->tp_call[0], __float_as_uint (pk_l));
				atomicMax (&st->tp_call[1], __float_as_uint (pk_r));
			}
		}
	} else if (wid == LDR) {
		// =================== fixed roles: loader + K-filter wave ===================
		v2f k1 = 0, k2 = 0, k3 = 0, k4 = 0;        // carried K-filter state (wave-uniform)
		if (EBU && q == 0) {
			k1 = v2f{st->kz[0], st->kz[1]}; k2 = v2f{st->kz[2], st->kz[3]};
			k3 = v2f{st->kz[4], st->kz[5]}; k4 = v2f{st->kz[6], st->kz[7]};
		}
		stage (-nwarm, buf0);
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
		block_barrier ();
		for (int jj = -nwarm; jj < ntile; ++jj) {
			const int par = (jj + nwarm) & 1;
			v2f* const cur = buf0 + par * NB;
			v2f* const nxt = buf0 + (par ^ 1) * NB;
			int64_t t0; int len;
			tile_of (jj, t0, len);
			if (jj + 1 < ntile) stage (jj + 1, nxt);          // in flight while this tile is filtered
			if (EBU) kfilter_tile (cur, jj, len, jj < 0, k1, k2, k3, k4);
			asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");   // tile j+1 has landed in `nxt`
			block_barrier ();
		}
		if (EBU && q == a.n_segs - 1 && lane == 0) {
			st->kz[0] = k1.x; st->kz[1] = k1.y; st->kz[2] = k2.x; st->kz[3] = k2.y;
			st->kz[4] = k3.x; st->kz[5] = k3.y; st->kz[6] = k4.x; st->kz[7] = k4.y;
		}
	} else if (TP) {
		// =================== fixed roles: FIR wave `wid` ===================
		float pk_l = 0.f, pk_r = 0.f;
		block_barrier ();
		for (int jj = -nwarm; jj < ntile; ++jj) {
			if (jj >= 0) fir_tile (buf0 + ((jj + nwarm) & 1) * NB, jj, wid, pk_l, pk_r);
			block_barrier ();
		}
		pk_l = wave_max (pk_l);
		pk_r = wave_max (pk_r);
		if (lane == 0) {
			atomicMax (&st->tp_call[0], __float_as_uint (pk_l));
			atomicMax (&st->tp_call[1], __float_as_uint (pk_r));
		}
	}
}

template <int K, int R, bool SYM, bool ROT>
static int launch2 (bool ebu, bool tp, const mtr_fused_args& a, uint32_t n_units, hipStream_t st)
{
	const size_t lds = (size_t) 2 * a.buf_slots * sizeof (v2f);
	const dim3 grid (n_units);
	static bool raised = false;
	if (!raised) {
		const int mx = 160 * 1024;
		(void) hipFuncSetAttribute ((const void*) k_fused2<K, R, true, true, SYM, ROT>,  hipFuncAttributeMaxDynamicSharedMemorySize, mx);
		(void) hipFuncSetAttribute ((const void*) k_fused2<K, R, true, false, SYM, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
		(void) hipFuncSetAttribute ((const void*) k_fused2<K, R, false, true, SYM, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
		raised = true;
	}
	if (ebu && tp)  hipLaunchKernelGGL ((k_fused2<K, R, true, true, SYM, ROT>),  grid, dim3 (256), lds, st, a);
	else if (ebu)   hipLaunchKernelGGL ((k_fused2<K, R, true, false, SYM, ROT>), grid, dim3 (64),  lds, st, a);
	else            hipLaunchKernelGGL ((k_fused2<K, R, false, true, SYM, ROT>), grid, dim3 (256), lds, st, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

int mtr_launch_fused2 (int run, bool ebu, bool tp, const mtr_fused_args& a, uint32_t n_units, void* stream)
{
	switch (run) {
	case 39:
		if (a.fir_form == 1) return launch2<39, 13, false, false> (ebu, tp, a, n_units, (hipStream_t) stream);
		if (a.rotate && ebu && tp) return launch2<39, 13, true, true> (ebu, tp, a, n_units, (hipStream_t) stream);
		return launch2<39, 13, true, false> (ebu, tp, a, n_units, (hipStream_t) stream);
	default: return -2;
	}
}
