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
//   * a workgroup of nine waves owns 64 streams, lane = stream, both channels packed in one v2f;
//   * waves 0-7 interpolate: each takes 2 of the chunk's 16 frames for all 64 streams (mirror-
//     symmetric taps as in k_fused2, held in VGPRs) and leaves |y| of the 4 phases in LDS in time
//     order; they also fetch the next chunk's input rows (one coalesced load per stream row, all
//     of them unconditional: a load inside a divergent branch is waited for on the spot);
//   * wave 8 is the recurrence: 64 independent (z1, z2, m, p) chains, 4 steps per frame, on the
//     previous chunk's values;
//   * both LDS arrays are double buffered; one barrier per chunk.  Input rows are re-fetched with
//     their 48-frame history every chunk (L2 hits; HBM sees each frame once).
// The kernel is latency-bound by construction (one workgroup per CU, the chain is serial): what
// matters is the length of the longest role per chunk, hence many narrow interpolator waves.
// Lane strides in LDS are odd (65 slots): every ds_read_b64 / ds_write_b64 is conflict free.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

typedef float v2f __attribute__ ((ext_vector_type (2)));
typedef const __attribute__ ((address_space (4))) float* cfloat_p;

namespace {

constexpr int NFIR = 8;                  // interpolator waves; wave NFIR is the recurrence
constexpr int R = 2;                     // frames per interpolator wave and chunk
constexpr int F = NFIR * R;              // 16 frames per chunk
constexpr int NS = 64;                   // streams per workgroup
constexpr int IN_SLOTS = F + 48;         // 64: slot i of a row <-> frame c0 - 48 + i, one slot per lane when a row is fetched
constexpr int IN_STRIDE = IN_SLOTS + 1;  // 65
constexpr int OV_STRIDE = 4 * F + 1;     // 65 slots: slot 4 f + q <-> phase q of frame c0 + f
constexpr int NTHREADS = 64 * (NFIR + 1);
static_assert (IN_STRIDE % 2 == 1 && OV_STRIDE % 2 == 1 && IN_SLOTS == 64 && NS % NFIR == 0, "odd lane strides; a lane per slot");

__device__ __forceinline__ v2f vabs (v2f v) { return v2f{fabsf (v.x), fabsf (v.y)}; }

// `if (v > z) z += w * (v - z)` (truepeakdsp.cc:63-64): adding w * max (v - z, 0) is the same map
__device__ __forceinline__ v2f attack (v2f z, v2f v, float w)
{
	const v2f d = v - z;
	return z + w * v2f{fmaxf (d.x, 0.f), fmaxf (d.y, 0.f)};
}

// R outputs of the three non-trivial polyphase branches in the mirror-symmetric form of k_fused2
// (pmq = P, M, Q: 3 x 24 taps); xs = slot of frame (first output - 48); out[r] = |x0|, |y1|, |y2|, |y3|
// Taps go through SGPRs in groups of 6 mirror pairs per branch (all 72 at once do not fit the scalar file;
// keeping them in VGPRs instead was tried: with the unrolled groups it spills).
__device__ __forceinline__ void interpolate (const v2f* xs, cfloat_p pmq, v2f* out)
{
	constexpr int G = 6;
	v2f aS[R], aD[R], aQ[R];
#pragma unroll
	for (int r = 0; r < R; ++r) { aS[r] = 0; aD[r] = 0; aQ[r] = 0; }
	// this kernel is latency-bound (two or three waves per SIMD): the next group's taps are fetched while
	// the current group is computed
	float tp[G], tm[G], tq[G];
#pragma unroll
	for (int k = 0; k < G; ++k) { tp[k] = pmq[k]; tm[k] = pmq[24 + k]; tq[k] = pmq[48 + k]; }
#pragma unroll 1
	for (int g = 0; g < 24; g += G) {
		float np[G], nm[G], nq[G];
		const int gn = g + G < 24 ? g + G : 0;
#pragma unroll
		for (int k = 0; k < G; ++k) { np[k] = pmq[gn + k]; nm[k] = pmq[24 + gn + k]; nq[k] = pmq[48 + gn + k]; }
		const v2f* const xl = xs + 1 + g;
		const v2f* const xr = xs + 48 - g - (G - 1);
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
#pragma unroll
		for (int k = 0; k < G; ++k) { tp[k] = np[k]; tm[k] = nm[k]; tq[k] = nq[k]; }
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		out[4 * r + 0] = vabs (xs[24 + r]);          // phase 0 is the identity: x[n - 24]
		out[4 * r + 1] = vabs (aS[r] + aD[r]);
		out[4 * r + 2] = vabs (aQ[r]);
		out[4 * r + 3] = vabs (aS[r] - aD[r]);
	}
}

__global__ __launch_bounds__ (NTHREADS) void k_tpb (const mtr_tpb_args a)
{
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	v2f* const in_buf = reinterpret_cast<v2f*> (smem);                   // [2][NS][IN_STRIDE]
	v2f* const ov_buf = in_buf + 2 * NS * IN_STRIDE;                     // [2][NS][OV_STRIDE]
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	const uint32_t s0 = blockIdx.x * NS;
	const int C = (int) a.n_channels;
	const int64_t n_chunks = (int64_t) ((a.n_frames + F - 1) / F);
	const bool fir = wid < NFIR;

	const cfloat_p pmq = (cfloat_p) a.fir_pmq;

	// ---- interpolator waves: fetch the rows of chunk j (64 rows shared out over the NFIR waves) ----
	constexpr int ROWS = NS / NFIR;                                      // 8 rows per wave
	const int row0 = wid * ROWS;
	// Every load is unconditional (addresses clamped into the stream, the mask applied when the value is
	// stored): a load inside a divergent branch is waited for at the end of its branch, and 22 global
	// latencies in series per chunk was the whole run time of the first attempt (it did not even depend
	// on the number of streams).
	auto fetch = [&] (int64_t j, v2f (&v)[ROWS]) {
		const int64_t f = j * F - 48 + lane;                             // lane <-> slot: frame of this call (f < 0: history)
		const bool in_hist = f < 0;
		const int64_t fc = f < 0 ? 0 : (f < (int64_t) a.n_frames ? f : (int64_t) a.n_frames - 1);
		const int64_t hc = f < -MTR_FIR_HALO ? 0 : (f < 0 ? MTR_FIR_HALO + f : 0);
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			uint32_t s = s0 + (uint32_t) (row0 + r);
			s = s < a.n_streams ? s : s0;
			const float* const p = in_hist ? a.hist + ((size_t) s * MTR_FIR_HALO + (size_t) hc) * 2
			                               : a.audio + ((size_t) s * a.stride + (size_t) fc) * C;
			// history rows are [frame][2] with a zero right channel for mono engines: p[0] is the sample either way
			v[r] = C == 2 ? *reinterpret_cast<const v2f*> (p) : v2f{p[0], 0.f};
		}
	};
	auto put = [&] (int64_t j, const v2f (&v)[ROWS]) {
		v2f* const dst = in_buf + (j & 1) * NS * IN_STRIDE;
		const int64_t f = j * F - 48 + lane;
		const bool ok = f >= -MTR_FIR_HALO && f < (int64_t) a.n_frames;
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			const bool live = s0 + (uint32_t) (row0 + r) < a.n_streams;
			dst[(row0 + r) * IN_STRIDE + lane] = (ok && live) ? v[r] : v2f{0.f, 0.f};
		}
	};

	// ---- recurrence wave: state of stream s0 + lane ----
	v2f z1 = 0, z2 = 0, m = 0, p = 0;
	const uint32_t sl = s0 + (uint32_t) lane;
	mtr_stream_state* const st = a.state + (sl < a.n_streams ? sl : 0);
	if (!fir && sl < a.n_streams) {
		z1 = v2f{st->tpb_z1[0], st->tpb_z1[1]};
		z2 = v2f{st->tpb_z2[0], st->tpb_z2[1]};
		z1 = v2f{z1.x > 20 ? 20 : (z1.x < 0 ? 0 : z1.x), z1.y > 20 ? 20 : (z1.y < 0 ? 0 : z1.y)};   // truepeakdsp.cc:54-55
		z2 = v2f{z2.x > 20 ? 20 : (z2.x < 0 ? 0 : z2.x), z2.y > 20 ? 20 : (z2.y < 0 ? 0 : z2.y)};
	}

	if (fir) {                                                           // prologue: chunk 0 into buffer 0
		v2f v[ROWS];
		fetch (0, v);
		put (0, v);
	}
	__syncthreads ();

	// iteration t: rows of chunk t+1 are fetched, chunk t is interpolated, chunk t-1 goes through the recurrence
	for (int64_t t = 0; t <= n_chunks; ++t) {
		if (fir) {
			v2f nxt[ROWS];
			const bool more = t + 1 < n_chunks;
			if (more) fetch (t + 1, nxt);
			if (t < n_chunks) {
				const v2f* const xs = in_buf + (t & 1) * NS * IN_STRIDE + lane * IN_STRIDE + R * wid;
				v2f o[4 * R];
				interpolate (xs, pmq, o);
				v2f* const dst = ov_buf + (t & 1) * NS * OV_STRIDE + lane * OV_STRIDE + 4 * R * wid;
#pragma unroll
				for (int i = 0; i < 4 * R; ++i) dst[i] = o[i];
			}
			if (more) put (t + 1, nxt);
		} else if (t > 0) {
			const int64_t c0 = (t - 1) * F;
			const int nf = (int) min ((int64_t) F, (int64_t) a.n_frames - c0);
			const v2f* const ov = ov_buf + ((t - 1) & 1) * NS * OV_STRIDE + lane * OV_STRIDE;
			// the values do not depend on the state: fetched a frame ahead of the chain that consumes them
			v2f v[4] = { ov[0], ov[1], ov[2], ov[3] };
#pragma unroll
			for (int f = 0; f < F; ++f) {
				v2f nv[4];
				if (f + 1 < F) { nv[0] = ov[4 * f + 4]; nv[1] = ov[4 * f + 5]; nv[2] = ov[4 * f + 6]; nv[3] = ov[4 * f + 7]; }
				if (f < nf) {                                            // wave-uniform: only the call's last chunk is short
					z1 *= a.w3;
					z2 *= a.w3;
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						z1 = attack (z1, v[q], a.w1);
						z2 = attack (z2, v[q], a.w2);
						p = v2f{fmaxf (p.x, v[q].x), fmaxf (p.y, v[q].y)};
					}
					const v2f zz = z1 + z2;
					m = v2f{fmaxf (m.x, zz.x), fmaxf (m.y, zz.y)};
				}
				if (f + 1 < F) { v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2]; v[3] = nv[3]; }
			}
		}
		__syncthreads ();
	}

	if (!fir && sl < a.n_streams) {
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
	const size_t lds = (size_t) 2 * NS * (IN_STRIDE + OV_STRIDE) * sizeof (v2f);      // 130 KiB: one workgroup per CU
	static bool raised = false;
	if (!raised) {
		(void) hipFuncSetAttribute ((const void*) k_tpb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		raised = true;
	}
	hipLaunchKernelGGL (k_tpb, dim3 ((a.n_streams + NS - 1) / NS), dim3 (NTHREADS), lds, (hipStream_t) stream, a);
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
