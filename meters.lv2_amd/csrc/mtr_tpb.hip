// mtr_tpb.hip — TruePeakdsp::process: 4x interpolation followed by the PPM-style ballistics (gfx950).
//
// Replaces jmeters/truepeakdsp.cc:41-99 (the dBTP plugins' meter, src/meters.cc:465-507, and dr14.c's
// true-peak bar) with the call-level semantics of "process(); read(m, p)": per call and channel
//     z1 *= w3, z2 *= w3 once per input frame; for each of the 4 oversampled values v = |y|:
//     if (v > z1) z1 += w1 (v - z1);  if (v > z2) z2 += w2 (v - z2);  p = max (p, v);
//     m = max (m, z1 + z2) once per input frame;  result m * g, p;  states clamped to [0, 20] on
//     entry and offset by 1e-20f on exit.
//
// The attack / release recurrence is a monotone piece-wise linear map of the state — compositions
// grow a piece per step, so it is not a cheap associative scan: time stays serial per (stream,
// channel).  What can run in parallel is everything else, so the workgroup is specialised
// (first version: one wave per stream, one lane walking 256 values per 64 frames: 543 ms per
// 31.5 GB; this one: see DESIGN.md):
//
//   * a workgroup of eight waves owns 64 streams, lane = stream, both channels packed in one v2f (32 streams
//     with two lanes per stream in the interpolators when the batch would otherwise leave CUs idle);
//   * wave 0 is the recurrence: 64 independent (z1, z2, m, p) chains, 4 steps per frame, on the
//     previous chunk's values.  It is the serial chain, so it keeps a SIMD to itself (the wave's
//     SIMD id is read from HW_ID; the wave that shares it only fetches);
//   * the six waves on the other three SIMDs interpolate: each takes 2 of the chunk's 12 frames for
//     all 64 streams (mirror-symmetric form as in k_fused2; the 72 taps stay in vector registers,
//     two per register pair) and leaves |y| of the 4 phases in LDS in time order;
//   * waves 1-7 fetch: only the 12 new frames of every row per chunk, into a ring of six chunks per
//     stream in which every frame is stored twice, 72 slots apart, so that the interpolator's
//     60-slot window is always contiguous (12 wave-wide loads per chunk instead of the 70 the first
//     version needed to re-read every row with its window — issuing those took as long as the
//     interpolation).  All loads are unconditional: one inside a divergent branch is waited for at
//     the end of its branch;
//   * the |y| array is double buffered; one barrier per chunk.
// One workgroup per CU and a serial chain: the kernel is bound by the busiest SIMD per chunk
// (tools/tpb_prof.hip prints cycles per role: two interpolators ~3300 of the chunk's ~4100 cycles,
// the recurrence ~3000).  Lane strides in LDS are odd (145 and 49 slots): conflict-free ds_read/write_b64.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

typedef float v2f __attribute__ ((ext_vector_type (2)));
typedef const __attribute__ ((address_space (4))) float* cfloat_p;

// tools/tpb_prof.hip builds this file with MTR_TPB_PROF: cycles per role and section for workgroup 0
#ifdef MTR_TPB_PROF
__device__ unsigned long long g_tpb_prof[16][4];
#define PROF_NOW(v) unsigned long long v; asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define PROF_ADD(i, d) pr[i] += (d)
#else
#define PROF_NOW(v)
#define PROF_ADD(i, d)
#endif

namespace {

constexpr int NW = 8;                    // waves: wave 0 is the recurrence, the others fetch and interpolate
constexpr int R = 2;                     // frames per interpolation item
constexpr int NGRP = 6;                  // items per chunk: one per interpolator wave when two waves share a SIMD
constexpr int F = NGRP * R;              // 12 frames per chunk
// streams per workgroup: 64 (lane = stream in every role) or, for batches that would otherwise leave CUs idle, 32
// (two lanes per stream in the interpolators, one frame of the item each; half the recurrence wave idles)
// Input rows live in a ring of six chunks per stream: the 48-frame window (four chunks) of the chunk being
// interpolated, that chunk, and the one being fetched.  Every frame is stored twice, RING slots apart, so
// any 60-slot window is contiguous in LDS and the interpolator's offsets stay compile-time constants.
constexpr int RING = 6 * F;              // 72 slots; frame f <-> slot f mod 72 (and + 72)
constexpr int IN_STRIDE = 2 * RING + 1;  // 145
constexpr int OV_STRIDE = 4 * F + 1;     // 49 slots: slot 4 f + q <-> phase q of frame c0 + f
constexpr int NTHREADS = 64 * NW;
constexpr int LPW = (64 * F / 64 + NW - 2) / (NW - 1);   // wave-wide loads per fetching wave and chunk: at most 2
static_assert (IN_STRIDE % 2 == 1 && OV_STRIDE % 2 == 1 && 48 % F == 0, "odd lane strides; whole chunks of history");

__device__ __forceinline__ v2f vabs (v2f v) { return v2f{fabsf (v.x), fabsf (v.y)}; }

// `if (v > z) z += w * (v - z)` (truepeakdsp.cc:63-64) is z <- max (z, (1 - w) z + w v): for v <= z the second argument is
// <= z, for v > z it is the reference's update.  One fused multiply-add and a maximum ON the chain (the w v products do not
// depend on the state and are formed off it) instead of subtract -> maximum -> multiply-add: the serial chain of a frame
// is 9 dependent operations instead of 13.  (A rounding change of one ulp per step in a contraction: held to the same
// 2e-6 as before, tests/test_gpu_parity.py::test_truepeak_ballistics_*.)
__device__ __forceinline__ v2f attack (v2f z, v2f wv, float c)
{
	const v2f t = __builtin_elementwise_fma (v2f{c, c}, z, wv);
	return v2f{fmaxf (z.x, t.x), fmaxf (z.y, t.y)};
}

// R outputs of the three non-trivial polyphase branches in the mirror-symmetric form of k_fused2
// (tp, tm, tq = P, M, Q: 3 x 24 taps); xs = slot of frame (first output - 48); out[r] = |x0|, |y1|, |y2|, |y3|.
// The taps live in VGPRs here (72 of the 256 a wave may use at two waves per SIMD): fetched through the
// scalar cache per group, as k_fused2 must, every group ends in an `s_waitcnt lgkmcnt(0)` that also drains
// the LDS queue, and with two waves per SIMD nobody covers that bubble.  Fully unrolled, the LDS reads of
// later groups are issued under the arithmetic of earlier ones.
// Two taps share a register pair and the multiply picks its half (op_sel), so they cost 72 registers, not 144.
struct Taps { v2f p[12], m[12], q[12]; };
// (the compiler folds that selection only for scalar-register operands, hence the two asm forms)
template <int K>
__device__ __forceinline__ void tap_fma (v2f& acc, const v2f* t, v2f x)
{
	if (K & 1) asm ("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "v"(t[K >> 1]));
	else       asm ("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "v"(t[K >> 1]));
}

constexpr int G = 6;                     // mirror pairs per tap group
// RL = frames per lane of an item (2, or 1 when two lanes share a stream); a group needs RL + G - 1 window
// slots on either side
template <int G0, int RL>
__device__ __forceinline__ void group_load (const v2f* xs, v2f* L, v2f* B)
{
	constexpr int NW_G = RL + G - 1;
	const v2f* const xl = xs + 1 + G0;
	const v2f* const xr = xs + 48 - G0 - (G - 1);
#pragma unroll
	for (int j = 0; j < NW_G; ++j) { L[j] = xl[j]; B[j] = xr[j]; }
}

template <int G0, int RL>
__device__ __forceinline__ void group_mac (const v2f* L, const v2f* B, const Taps& tp, v2f* aS, v2f* aD, v2f* aQ)
{
#define MTR_TPB_TAP(k)                                                                   \
	{                                                                                    \
		const v2f sv = L[r + k] + B[r + G - 1 - k];                                      \
		const v2f dv = L[r + k] - B[r + G - 1 - k];                                      \
		tap_fma<G0 + k> (aS[r], tp.p, sv);                                               \
		tap_fma<G0 + k> (aD[r], tp.m, dv);                                               \
		tap_fma<G0 + k> (aQ[r], tp.q, sv);                                               \
	}
#pragma unroll
	for (int r = 0; r < RL; ++r) { MTR_TPB_TAP (0) MTR_TPB_TAP (1) MTR_TPB_TAP (2) MTR_TPB_TAP (3) MTR_TPB_TAP (4) MTR_TPB_TAP (5) }
#undef MTR_TPB_TAP
}

template <int RL>
__device__ __forceinline__ void interpolate (const v2f* xs, const Taps& tp, v2f* out)
{
	constexpr int NW_G = RL + G - 1;
	v2f aS[RL], aD[RL], aQ[RL];
#pragma unroll
	for (int r = 0; r < RL; ++r) { aS[r] = 0; aD[r] = 0; aQ[r] = 0; }
	// the LDS reads of a group are issued a whole group of arithmetic ahead (left to itself the scheduler
	// sinks them next to their first use, and with two waves per SIMD that latency is not covered)
	v2f L0[NW_G], B0[NW_G], L1[NW_G], B1[NW_G];
	group_load<0, RL> (xs, L0, B0);
	group_load<6, RL> (xs, L1, B1);
	__builtin_amdgcn_sched_barrier (0);
	group_mac<0, RL> (L0, B0, tp, aS, aD, aQ);
	__builtin_amdgcn_sched_barrier (0);
	group_load<12, RL> (xs, L0, B0);
	__builtin_amdgcn_sched_barrier (0);
	group_mac<6, RL> (L1, B1, tp, aS, aD, aQ);
	__builtin_amdgcn_sched_barrier (0);
	group_load<18, RL> (xs, L1, B1);
	__builtin_amdgcn_sched_barrier (0);
	group_mac<12, RL> (L0, B0, tp, aS, aD, aQ);
	__builtin_amdgcn_sched_barrier (0);
	group_mac<18, RL> (L1, B1, tp, aS, aD, aQ);
#pragma unroll
	for (int r = 0; r < RL; ++r) {
		out[4 * r + 0] = vabs (xs[24 + r]);          // phase 0 is the identity: x[n - 24]
		out[4 * r + 1] = vabs (aS[r] + aD[r]);
		out[4 * r + 2] = vabs (aQ[r]);
		out[4 * r + 3] = vabs (aS[r] - aD[r]);
	}
}

template <int C, int NS>   // channels: 2 = interleaved stereo, 1 = mono (the right half of every v2f stays zero); streams per workgroup
__global__ __launch_bounds__ (NTHREADS) void k_tpb (const mtr_tpb_args a)
{
	static_assert (NS == 64 || NS == 32, "one or two lanes per stream");
	constexpr int RL = R * NS / 64;                                      // frames per lane of an interpolation item
	constexpr int NLOAD = NS * F / 64;                                   // wave-wide loads that bring one chunk of all rows
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	v2f* const in_buf = reinterpret_cast<v2f*> (smem);                   // [NS][IN_STRIDE]
	v2f* const ov_buf = in_buf + NS * IN_STRIDE;                         // [2][NS][OV_STRIDE]
	const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
	const uint32_t s0 = blockIdx.x * NS;
	const int srow = lane & (NS - 1), half = lane / NS;                  // interpolators and recurrence: this lane's stream; its half of an item
	const int64_t n_chunks = (int64_t) ((a.n_frames + F - 1) / F);
	const bool fir = wid != 0;

	// Role placement.  The recurrence is the serial chain of the kernel (about 40 VALU instructions per
	// frame that nothing can overlap), so its wave gets a SIMD to itself: waves that the dispatcher put on
	// wave 0's SIMD only fetch; the others share the interpolation items.  HW_ID bits 5:4 = SIMD.
	__shared__ int simd_of[NW];
	if (lane == 0) simd_of[wid] = (int) __builtin_amdgcn_s_getreg ((1 << 11) | (4 << 6) | 4);
	__syncthreads ();
	int item0 = -1, n_interp = 0;
	for (int w = 1; w < NW; ++w) {
		const bool other = simd_of[w] != simd_of[0];
		if (w == wid && other) item0 = n_interp;
		n_interp += other;
	}
	if (n_interp < 3) { item0 = wid - 1; n_interp = NW - 1; }          // unexpected placement: everyone interpolates
	item0 = __builtin_amdgcn_readfirstlane (item0);
	n_interp = __builtin_amdgcn_readfirstlane (n_interp);

	Taps taps;
	if (fir) {
		const cfloat_p pmq = (cfloat_p) a.fir_pmq;
#pragma unroll
		for (int k = 0; k < 12; ++k) {
			taps.p[k] = v2f{pmq[2 * k], pmq[2 * k + 1]};
			taps.m[k] = v2f{pmq[24 + 2 * k], pmq[25 + 2 * k]};
			taps.q[k] = v2f{pmq[48 + 2 * k], pmq[49 + 2 * k]};
			asm volatile ("" : "+v"(taps.p[k]), "+v"(taps.m[k]), "+v"(taps.q[k]));   // keep them in vector registers
		}
	}

	// ---- waves 1..NW-1 fetch: element e = 64 k + lane of a chunk is (row e / F, frame e % F); wave w owns the
	//      wave-wide loads k = w - 1 and w + 6.  Only the F new frames of a row are read per chunk (HBM and
	//      the texture path see each frame once; the first version re-read the 48-frame window every chunk
	//      and spent as long issuing loads as interpolating).  Every load is unconditional (address clamped,
	//      the mask applied when the value is stored): a load inside a divergent branch is waited for at the
	//      end of its branch.
	const float* rowp[LPW];
	int slot[LPW], ldso[LPW];
	bool live[LPW];
#pragma unroll
	for (int i = 0; i < LPW; ++i) {
		const int e = 64 * (wid - 1 + (NW - 1) * i) + lane;
		const int row = (e / F) & (NS - 1);
		slot[i] = e % F;
		ldso[i] = row * IN_STRIDE + slot[i];
		live[i] = s0 + (uint32_t) row < a.n_streams;
		rowp[i] = a.audio + (size_t) (live[i] ? s0 + (uint32_t) row : s0) * a.stride * C;
	}
	auto fetch = [&] (int64_t j, v2f (&v)[LPW]) {
#pragma unroll
		for (int i = 0; i < LPW; ++i) {
			if (wid - 1 + (NW - 1) * i >= NLOAD) break;                 // wave-uniform
			const int64_t f = j * F + slot[i];
			const float* const q = rowp[i] + (size_t) (f < (int64_t) a.n_frames ? f : 0) * C;
			v[i] = C == 2 ? *reinterpret_cast<const v2f*> (q) : v2f{q[0], 0.f};
		}
	};
	auto put = [&] (int64_t j, int ring_chunk, const v2f (&v)[LPW]) {
#pragma unroll
		for (int i = 0; i < LPW; ++i) {
			if (wid - 1 + (NW - 1) * i >= NLOAD) break;
			const bool ok = live[i] && j * F + slot[i] < (int64_t) a.n_frames;
			v2f* const dst = in_buf + ldso[i] + ring_chunk * F;
			const v2f x = ok ? v[i] : v2f{0.f, 0.f};
			dst[0] = x;
			dst[RING] = x;
		}
	};

	// ---- recurrence wave: state of stream s0 + lane ----
	v2f z1 = 0, z2 = 0, m = 0;
	v2f pkp = 0;                                                         // raw peak of the values this lane saw
	const uint32_t sl = s0 + (uint32_t) srow;
	const bool owner = lane < NS && sl < a.n_streams;                    // the lane that carries this stream's chain
	mtr_stream_state* const st = a.state + (sl < a.n_streams ? sl : 0);
	if (!fir && owner) {
		z1 = v2f{st->tpb_z1[0], st->tpb_z1[1]};
		z2 = v2f{st->tpb_z2[0], st->tpb_z2[1]};
		z1 = v2f{z1.x > 20 ? 20 : (z1.x < 0 ? 0 : z1.x), z1.y > 20 ? 20 : (z1.y < 0 ? 0 : z1.y)};   // truepeakdsp.cc:54-55
		z2 = v2f{z2.x > 20 ? 20 : (z2.x < 0 ? 0 : z2.x), z2.y > 20 ? 20 : (z2.y < 0 ? 0 : z2.y)};
	}
	// 32 streams per workgroup: the recurrence wave gives every (stream, CHANNEL) a lane of its own — lane l = stream l & 31,
	// channel l >> 5 — instead of idling half its lanes with both channels packed in a v2f: there is no packed f32 maximum,
	// so the packed form pays two v_max per attack; the scalar form is one v_fma + one v_max.  The serial chain is what
	// bounds this kernel (every stream needs its frames' instructions one after the other): 20 scalar instructions per
	// frame instead of 37 mixed ones.
	const int rch = lane >> 5;
	const uint32_t sl2 = s0 + (uint32_t) (lane & 31);
	const bool owner2 = NS == 32 && sl2 < a.n_streams && (C == 2 || rch == 0);
	mtr_stream_state* const st2 = a.state + (sl2 < a.n_streams ? sl2 : 0);
	float y1 = 0.f, y2 = 0.f, ym = 0.f;
	if (NS == 32 && !fir && owner2) {
		y1 = st2->tpb_z1[rch]; y2 = st2->tpb_z2[rch];
		y1 = y1 > 20 ? 20 : (y1 < 0 ? 0 : y1);
		y2 = y2 > 20 ? 20 : (y2 < 0 ? 0 : y2);
	}

	if (fir) {
		// prologue: the 48 frames before the call (47 of history, frame -48 is never multiplied by a non-zero
		// tap) into ring chunks 2..5, chunk 0 into ring chunk 0
		for (int e = (wid - 1) * 64 + lane; e < NS * 48; e += (NW - 1) * 64) {
			const int row = e / 48, i = e % 48;                          // frame i - 48
			v2f x = {0.f, 0.f};
			if (s0 + (uint32_t) row < a.n_streams && i >= 1) {
				// history rows are [frame][2] with a zero right channel for mono engines
				const float* const h = a.hist + ((size_t) (s0 + (uint32_t) row) * MTR_FIR_HALO + (size_t) (i - 1)) * 2;
				x = C == 2 ? v2f{h[0], h[1]} : v2f{h[0], 0.f};
			}
			v2f* const dst = in_buf + row * IN_STRIDE + 2 * F + i;
			dst[0] = x;
			dst[RING] = x;
		}
		v2f v[LPW];
		fetch (0, v);
		put (0, 0, v);
	}
	__syncthreads ();

	// iteration t: the new frames of chunk t+1 are fetched, chunk t is interpolated, chunk t-1 goes through
	// the recurrence.  rd = ring chunk where the window of chunk t starts (frame 12 t - 48), wr = where chunk
	// t+1 goes.
#ifdef MTR_TPB_PROF
	unsigned long long pr[4] = { 0, 0, 0, 0 };
#endif
	const float c1 = 1.0f - a.w1, c2 = 1.0f - a.w2;
	int rd = 2, wr = 1;
	for (int64_t t = 0; t <= n_chunks; ++t) {
		PROF_NOW (c0_);
		if (fir) {
			v2f nxt[LPW];
			const bool more = t + 1 < n_chunks;
			if (more) fetch (t + 1, nxt);
			PROF_NOW (cf_);
			PROF_ADD (2, cf_ - c0_);
			if (t < n_chunks && item0 >= 0) {
				for (int g = item0; g < NGRP; g += n_interp) {
					const v2f* const xs = in_buf + srow * IN_STRIDE + rd * F + R * g + RL * half;
					v2f o[4 * RL];
					interpolate<RL> (xs, taps, o);
					v2f* const dst = ov_buf + (t & 1) * NS * OV_STRIDE + srow * OV_STRIDE + 4 * (R * g + RL * half);
#pragma unroll
					for (int i = 0; i < 4 * RL; ++i) dst[i] = o[i];
					// the raw peak (truepeakdsp.cc:65: p = max (p, v)) does not depend on the chain: with 32 streams
					// per workgroup it is taken here, off the recurrence wave, which is that shape's bound
					const int64_t f0 = t * F + R * g + RL * half;
#pragma unroll
					for (int r = 0; r < RL; ++r) {
						if (NS == 32 && f0 + r < (int64_t) a.n_frames) {
#pragma unroll
							for (int q = 0; q < 4; ++q) {
								pkp.x = fmaxf (pkp.x, o[4 * r + q].x);
								pkp.y = fmaxf (pkp.y, o[4 * r + q].y);
							}
						}
					}
				}
			}
			PROF_NOW (c1_);
			PROF_ADD (0, c1_ - c0_);
			if (more) put (t + 1, wr, nxt);
			PROF_NOW (c2_);
			PROF_ADD (1, c2_ - c1_);
		} else if (t > 0 && NS == 32) {
			const int64_t c0 = (t - 1) * F;
			const int nf = (int) min ((int64_t) F, (int64_t) a.n_frames - c0);
			const float* const ov = reinterpret_cast<const float*> (ov_buf + ((t - 1) & 1) * NS * OV_STRIDE + (lane & 31) * OV_STRIDE) + rch;
			float v[4] = { ov[0], ov[2], ov[4], ov[6] };
#pragma unroll
			for (int f = 0; f < F; ++f) {
				float nv[4];
				if (f + 1 < F) { nv[0] = ov[8 * f + 8]; nv[1] = ov[8 * f + 10]; nv[2] = ov[8 * f + 12]; nv[3] = ov[8 * f + 14]; }
				if (f < nf) {                                            // wave-uniform: only the call's last chunk is short
					float p1[4], p2[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) { p1[q] = a.w1 * v[q]; p2[q] = a.w2 * v[q]; }
					y1 *= a.w3;
					y2 *= a.w3;
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						y1 = fmaxf (y1, __builtin_fmaf (c1, y1, p1[q]));
						y2 = fmaxf (y2, __builtin_fmaf (c2, y2, p2[q]));
					}
					ym = fmaxf (ym, y1 + y2);
				}
				if (f + 1 < F) { v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2]; v[3] = nv[3]; }
			}
		} else if (t > 0) {
			const int64_t c0 = (t - 1) * F;
			const int nf = (int) min ((int64_t) F, (int64_t) a.n_frames - c0);
			const v2f* const ov = ov_buf + ((t - 1) & 1) * NS * OV_STRIDE + srow * OV_STRIDE;
			// the values do not depend on the state: fetched a frame ahead of the chain that consumes them
			v2f v[4] = { ov[0], ov[1], ov[2], ov[3] };
#pragma unroll
			for (int f = 0; f < F; ++f) {
				v2f nv[4];
				if (f + 1 < F) { nv[0] = ov[4 * f + 4]; nv[1] = ov[4 * f + 5]; nv[2] = ov[4 * f + 6]; nv[3] = ov[4 * f + 7]; }
				if (f < nf) {                                            // wave-uniform: only the call's last chunk is short
					v2f wv1[4], wv2[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) { wv1[q] = a.w1 * v[q]; wv2[q] = a.w2 * v[q]; }
					z1 *= a.w3;
					z2 *= a.w3;
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						z1 = attack (z1, wv1[q], c1);
						z2 = attack (z2, wv2[q], c2);
						if (NS == 64) pkp = v2f{fmaxf (pkp.x, v[q].x), fmaxf (pkp.y, v[q].y)};
					}
					const v2f zz = z1 + z2;
					m = v2f{fmaxf (m.x, zz.x), fmaxf (m.y, zz.y)};
				}
				if (f + 1 < F) { v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2]; v[3] = nv[3]; }
			}
		}
		PROF_NOW (c3_);
		__syncthreads ();
		PROF_NOW (c4_);
		if (!fir) { PROF_ADD (0, c3_ - c0_); PROF_ADD (2, c4_ - c3_); }
		PROF_ADD (3, c4_ - c0_);
		rd = rd == 5 ? 0 : rd + 1;
		wr = wr == 5 ? 0 : wr + 1;
	}
#ifdef MTR_TPB_PROF
	if (blockIdx.x == 0 && lane == 0) {
		for (int i = 0; i < 4; ++i) g_tpb_prof[wid][i] = pr[i];
		g_tpb_prof[8 + wid][0] = simd_of[wid]; g_tpb_prof[8 + wid][1] = item0; g_tpb_prof[8 + wid][2] = n_interp;
	}
#endif

	// the interpolators' raw peaks per stream (non-negative floats order as unsigned ints)
	uint32_t* const pk_sh = reinterpret_cast<uint32_t*> (in_buf);        // the input ring is spent
	for (int i = threadIdx.x; i < 2 * NS; i += NTHREADS) pk_sh[i] = 0u;
	__syncthreads ();
	if (fir == (NS == 32)) {                                             // whoever tracked it (the recurrence wave at 64 streams)
		atomicMax (&pk_sh[2 * srow], __float_as_uint (pkp.x));
		atomicMax (&pk_sh[2 * srow + 1], __float_as_uint (pkp.y));
	}
	__syncthreads ();
	if (NS == 32) {
		if (!fir && owner2) {
			st2->tpb_z1[rch] = y1 + 1e-20f;                               // truepeakdsp.cc:86-87
			st2->tpb_z2[rch] = y2 + 1e-20f;
			st2->tpb_m[rch] = ym * a.g;                                   // :89, then read (m, p)
			st2->tpb_p[rch] = __uint_as_float (pk_sh[2 * (lane & 31) + rch]);
		}
		if (C == 1 && !fir && lane < 32 && sl2 < a.n_streams) {         // mono engines keep a zero right channel
			st2->tpb_z1[1] = 1e-20f; st2->tpb_z2[1] = 1e-20f; st2->tpb_m[1] = 0.f; st2->tpb_p[1] = 0.f;
		}
	} else if (!fir && owner) {
		const v2f p = v2f{__uint_as_float (pk_sh[2 * srow]), __uint_as_float (pk_sh[2 * srow + 1])};
		st->tpb_z1[0] = z1.x + 1e-20f; st->tpb_z1[1] = z1.y + 1e-20f;     // truepeakdsp.cc:86-87
		st->tpb_z2[0] = z2.x + 1e-20f; st->tpb_z2[1] = z2.y + 1e-20f;
		st->tpb_m[0] = m.x * a.g; st->tpb_m[1] = m.y * a.g;               // :89, then read (m, p)
		st->tpb_p[0] = p.x; st->tpb_p[1] = p.y;
	}
}

// the 47-frame history for mono input is stored as [f][2] with the right channel zero
__global__ void k_history_mono (const float* audio, uint64_t stride, uint64_t n_frames, const float* hist_in,
                                float* hist_out, uint32_t n_streams)
{
	const uint32_t gidx = blockIdx.x * blockDim.x + threadIdx.x;
	if (gidx >= n_streams * MTR_FIR_HALO) return;
	const uint32_t s = gidx / MTR_FIR_HALO, i = gidx % MTR_FIR_HALO;
	const int64_t f = (int64_t) n_frames - MTR_FIR_HALO + i;
	const float v = (f >= 0) ? audio[(size_t) s * stride + f] : hist_in[((size_t) s * MTR_FIR_HALO + MTR_FIR_HALO + f) * 2];
	hist_out[((size_t) s * MTR_FIR_HALO + i) * 2] = v;
	hist_out[((size_t) s * MTR_FIR_HALO + i) * 2 + 1] = 0.f;
}

}  // namespace

int mtr_launch_tpb (const mtr_tpb_args& a, void* stream)
{
	// 64 streams per workgroup (121 KiB of LDS: one workgroup per CU) when that fills the chip; 32 otherwise
	const bool narrow = (a.n_streams + 63) / 64 <= 128;
	const int ns = narrow ? 32 : 64;
	const size_t lds = (size_t) ns * (IN_STRIDE + 2 * OV_STRIDE) * sizeof (v2f);
	const size_t lds64 = (size_t) 64 * (IN_STRIDE + 2 * OV_STRIDE) * sizeof (v2f);
	static bool raised = false;
	if (!raised) {
		(void) hipFuncSetAttribute ((const void*) k_tpb<1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds64);
		(void) hipFuncSetAttribute ((const void*) k_tpb<2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds64);
		raised = true;
	}
	const dim3 grid ((a.n_streams + ns - 1) / ns);
	hipStream_t st = (hipStream_t) stream;
	if (a.n_channels == 2) {
		if (narrow) hipLaunchKernelGGL ((k_tpb<2, 32>), grid, dim3 (NTHREADS), lds, st, a);
		else        hipLaunchKernelGGL ((k_tpb<2, 64>), grid, dim3 (NTHREADS), lds, st, a);
	} else {
		if (narrow) hipLaunchKernelGGL ((k_tpb<1, 32>), grid, dim3 (NTHREADS), lds, st, a);
		else        hipLaunchKernelGGL ((k_tpb<1, 64>), grid, dim3 (NTHREADS), lds, st, a);
	}
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

int mtr_launch_history_mono (const float* audio, uint64_t stride, uint64_t n_frames, const float* hist_in,
                             float* hist_out, uint32_t n_streams, void* stream)
{
	const uint32_t n = n_streams * MTR_FIR_HALO;
	hipLaunchKernelGGL (k_history_mono, dim3 ((n + 255) / 256), dim3 (256), 0, (hipStream_t) stream,
	                    audio, stride, n_frames, hist_in, hist_out, n_streams);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
