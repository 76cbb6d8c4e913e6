// mtr_bank.hip — 30-band 1/3-octave filter bank for gfx950 + the synthetic-signal fill kernel.
//
// Replaces the per-sample loop of spectrum_run (src/spectrumlv2.c:210-227) around
// bandpass_process / proc_one (src/spectr.c:68-87): mono down-mix, 30 bands x 6 cascaded
// TDF-II biquads in double, float EMA of v^2 with peak hold.
//
// Mapping: one lane per (stream, band), bands of consecutive streams packed back to back — lane p of the launch is
// stream p / 30, band p % 30 — so all 64 lanes of every wave work (a layout of 32 lanes per stream idles two in
// every 32: 6 % of a kernel that issues VALU instructions 100 % of the time).  The 12-state recurrence of a band
// is serial in time and, with pole radii up to 0.99991, not worth an exact time split (a 12x12 carry matrix per
// band); with >= 4096 streams there are >= 122k independent lanes, which fills the chip.  A wave of 64 lanes
// covers 2.1 streams: it stages chunks of frames of the (up to) four streams it touches through its own LDS buffers,
// forming the mono mix (L+R)/2 and adding the anti-denormal toggle once per frame instead of once per
// band; lanes then read their stream's row as an LDS broadcast.  Workgroup = one wave: no barrier anywhere.  Coefficients and states live in registers for the
// whole call.  fp64 VALU bound: 25 fp64 + 4 fp32 instructions per (frame, band), 4 cycles each, and nothing else in
// the loop — the kernel runs at that instruction floor (profiles/).
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

#define BANK_ROWS  4      /* streams a wave of 64 lanes can touch: ceil (63 / 30) + 1 */
#define BANK_CHUNK 128    /* frames staged per stream per iteration: two per lane */
#define BANK_PITCH (BANK_CHUNK + 2)   /* doubles per row: rows four banks apart, so the <= 4 rows a wave reads never collide */

// One wave per workgroup, NO barrier: the wave stages the chunks of the (up to four) streams it touches through its own
// two LDS buffers — chunk c + 1 is written, and chunk c + 2's loads are in flight, while chunk c is computed — so nothing
// in the loop ever waits for another wave (round 2's four-wave workgroups spent 17 % of the SIMD time outside the
// arithmetic: two barriers per 256 frames, on a kernel whose waves do not run in lock step).
__global__ __launch_bounds__ (64) void k_bank (const mtr_bank_args a)
{
	__shared__ double mix[2][BANK_ROWS][BANK_PITCH];    // mono mix + the +-1e-12 anti-denormal toggle, as double

	const int lane = threadIdx.x;
	const uint64_t pair0 = (uint64_t) blockIdx.x * 64;             // first (stream, band) pair of the wave
	const uint64_t pair  = pair0 + lane;
	const uint32_t s0 = (uint32_t) (pair0 / MTR_NBANDS);           // first stream the wave touches
	const uint32_t s  = (uint32_t) (pair / MTR_NBANDS);
	const int band = (int) (pair - (uint64_t) s * MTR_NBANDS);
	const int row  = (int) (s - s0);
	const bool live = s < a.n_streams;

	double W[6][5];
	double z[12];
	float  val = 0.f, mx = 0.f;
	if (live) {
#pragma unroll
		for (int i = 0; i < 6; ++i)
#pragma unroll
			for (int k = 0; k < 5; ++k) W[i][k] = a.coef[(band * 6 + i) * 5 + k];
#pragma unroll
		for (int i = 0; i < 12; ++i) z[i] = a.z[((size_t) s * MTR_NBANDS + band) * 12 + i];
		val = a.val[(size_t) s * MTR_NBANDS + band];
		mx  = a.mx[(size_t) s * MTR_NBANDS + band];
	} else {
#pragma unroll
		for (int i = 0; i < 6; ++i)
#pragma unroll
			for (int k = 0; k < 5; ++k) W[i][k] = 0;
#pragma unroll
		for (int i = 0; i < 12; ++i) z[i] = 0;
	}
	const float omega = a.omega;
	// the toggle parity each staged row starts the call with (all lanes keep all four: the staging is by frame, not by row)
	int par0[BANK_ROWS];
#pragma unroll
	for (int g = 0; g < BANK_ROWS; ++g) par0[g] = (s0 + g < a.n_streams) ? a.ac_in[s0 + g] : 0;
	const int my_par = live ? a.ac_in[s] : 0;

	// lane t stages frames 2 t, 2 t + 1 of every row
	float2 raw[BANK_ROWS][2];
	auto fetch = [&] (uint64_t base) {
#pragma unroll
		for (int g = 0; g < BANK_ROWS; ++g) {
			const uint32_t sg = s0 + g;
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const uint64_t f = base + 2 * lane + k;
				float2 v = make_float2 (0.f, 0.f);
				if (sg < a.n_streams && f < a.n_frames) {
					if (a.n_channels == 2) v = reinterpret_cast<const float2*> (a.audio)[(size_t) sg * a.stride + f];
					else                   v.x = a.audio[(size_t) sg * a.stride + f];
				}
				raw[g][k] = v;
			}
		}
	};
	auto stage = [&] (int buf, uint64_t base) {
#pragma unroll
		for (int g = 0; g < BANK_ROWS; ++g)
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				const float m = a.n_channels == 2 ? (raw[g][k].x + raw[g][k].y) / 2.0f : raw[g][k].x;   // spectrumlv2.c:216
				// bandpass_process toggles `ac` before use: sample i of the call gets +1e-12 when ac0 ^ 1 ^ (i & 1)
				const int pz = par0[g] ^ 1 ^ (int) ((base + k) & 1);                                       // (2 lane is even)
				mix[buf][g][2 * lane + k] = (double) m + (pz ? 1e-12 : -1e-12);                               // spectr.c:81-82
			}
	};

	// the staged chunk is written by all lanes and read by others: a workgroup-scope release / acquire around a wave barrier
	// makes that hand-over explicit (one wave: no cost but a wait for the LDS writes; ADVICE r3 — in-order LDS alone is not a contract)
	auto handover = [] () {
		__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier ();
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
	};
	fetch (0);
	stage (0, 0);
	handover ();
	fetch (BANK_CHUNK);
	int buf = 0;
	for (uint64_t base = 0; base < a.n_frames; base += BANK_CHUNK, buf ^= 1) {
		const int nf = (int) min ((uint64_t) BANK_CHUNK, a.n_frames - base);
		// the next chunk into the other buffer (its previous readers are behind us in this wave's own instruction stream),
		// the one after it into the registers: both land under this chunk's arithmetic
		if (base + BANK_CHUNK < a.n_frames) {
			handover ();                                  // (the other buffer's readers of the chunk before are done)
			stage (buf ^ 1, base + BANK_CHUNK);
			handover ();
			fetch (base + 2 * BANK_CHUNK);
		}
		const double* const my = &mix[buf][row][0];
		for (int n = 0; n < nf; ++n) {
			// six TDF-II sections (spectr.c:68-76).  Section 0 carries the normalisation g: numerator
			// g (1, 2, 1); sections 1-5 have (1, +-2, 1): b0 in = b2 in = in and b1 in = +-2 in are exact, so
			// sharing the product changes nothing but the operation count (25 instead of 31 fp64 ops).
			double out = my[n];
			{
				const double gi = W[0][0] * out;
				const double y = gi + z[0];
				z[0] = fma (-W[0][3], y, fma (2.0, gi, z[1]));
				z[1] = fma (-W[0][4], y, gi);
				out = y;
			}
#pragma unroll
			for (int i = 1; i < 6; ++i) {
				const double y = out + z[2 * i];
				z[2 * i]     = fma (-W[i][3], y, fma (W[i][1], out, z[2 * i + 1]));
				z[2 * i + 1] = fma (-W[i][4], y, out);
				out = y;
			}
			const float v = (float) out;
			const float q = v * v;
			val += omega * (q - val);
			// `val > mx ? val : mx` (spectrumlv2.c:222) as one v_max_f32: a NaN val loses either way, mx is never NaN
			mx = __builtin_fmaxf (mx, val);
		}
	}

	if (live) {
		// spectrum_run epilogue, state part (spectrumlv2.c:230-238)
		if (!isfinite (val)) val = 0;
		if (!isfinite (mx))  mx = 0;
#pragma unroll
		for (int i = 0; i < 12; ++i) {
			if (!isfinite (z[i])) z[i] = 0;
			a.z[((size_t) s * MTR_NBANDS + band) * 12 + i] = z[i];
		}
		a.val[(size_t) s * MTR_NBANDS + band] = val + 1e-20f;
		a.mx[(size_t) s * MTR_NBANDS + band]  = mx;
		// the parity of the next call goes to the OTHER buffer: a wave that starts late must not see this call's update
		if (band == 0) a.ac_out[s] = my_par ^ (int) (a.n_frames & 1);
	}
}

int mtr_launch_bank (const mtr_bank_args& a, void* stream)
{
	const uint64_t pairs = (uint64_t) a.n_streams * MTR_NBANDS;
	const uint32_t nb = (uint32_t) ((pairs + 63) / 64);
	hipLaunchKernelGGL (k_bank, dim3 (nb), dim3 (64), 0, (hipStream_t) stream, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

// ---- synthetic programme signal ------------------------------------------------------------------
// kind 0: LCG noise (two draws per frame, L then R), u = ((s >> 8) - 2^23) / 2^23
// kind 2: kind 0 under a monotonically rising level (2^-8 .. 1): the pruning-hostile case of bench.py
// kind 1: programme-like: env(t) * (0.5 u + 0.5 sin(2 pi f t)), f_L = 440, f_R = 3000,
//         env = 0.05 + 0.45 (0.5 + 0.5 sin(2 pi 0.2 t)); stream s uses seed + s.
// The LCG is jumped to each thread's position with the closed form s_n = A^n s_0 + C (A^n - 1)/(A - 1),
// evaluated by square-and-multiply on (A, C) pairs, so the fill is embarrassingly parallel and
// reproduces the serial generator bit-for-bit.

__device__ __forceinline__ void lcg_jump (uint32_t n, uint32_t& mul, uint32_t& add)
{
	uint32_t cm = 1664525u, ca = 1013904223u;   // one step
	mul = 1u; add = 0u;
	while (n) {
		if (n & 1u) { mul = mul * cm; add = add * cm + ca; }
		ca = ca * cm + ca;
		cm = cm * cm;
		n >>= 1;
	}
}

#define SYNTH_RUN 256   /* frames per thread */

__global__ void k_synth (float* audio, uint32_t n_streams, uint64_t n_frames, uint64_t stride,
                         uint32_t seed, float fs, int kind)
{
	const uint64_t runs = (n_frames + SYNTH_RUN - 1) / SYNTH_RUN;
	const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= runs * n_streams) return;
	const uint32_t s = (uint32_t) (g / runs);
	const uint64_t f0 = (g % runs) * SYNTH_RUN;
	uint32_t mul, add;
	lcg_jump ((uint32_t) (2 * f0), mul, add);
	uint32_t st = mul * (seed + s) + add;
	float2* dst = reinterpret_cast<float2*> (audio) + (size_t) s * stride;
	const uint64_t f1 = min (f0 + SYNTH_RUN, n_frames);
	for (uint64_t f = f0; f < f1; ++f) {
		st = 1664525u * st + 1013904223u;
		float ul = (float) (int32_t) ((st >> 8) - 8388608u) * (1.0f / 8388608.0f);
		st = 1664525u * st + 1013904223u;
		float ur = (float) (int32_t) ((st >> 8) - 8388608u) * (1.0f / 8388608.0f);
		if (kind == 1) {
			const float t = (float) f / fs;
			const float env = 0.05f + 0.45f * (0.5f + 0.5f * __sinf (6.2831853f * 0.2f * t));
			// phase reduced in integer arithmetic so long streams stay accurate
			const float pl = (float) ((f * 440ull) % (uint64_t) fs) / fs;
			const float pr = (float) ((f * 3000ull) % (uint64_t) fs) / fs;
			ul = env * (0.5f * ul + 0.5f * __sinf (6.2831853f * pl));
			ur = env * (0.5f * ur + 0.5f * __sinf (6.2831853f * pr));
		} else if (kind == 2) {
			// noise under a level that rises monotonically by 48 dB over the stream: every block's maximum beats everything
			// before it — the worst case of exact peak pruning (nothing can be skipped, every screened block is completed)
			const float env = exp2f (-8.0f + 8.0f * (float) f / (float) n_frames);
			ul *= env; ur *= env;
		}
		dst[f] = make_float2 (ul, ur);
	}
}

int mtr_launch_synth (float* d_audio, uint32_t n_streams, uint64_t n_frames, uint64_t stride,
                      uint32_t seed, float fs, int kind, void* stream)
{
	const uint64_t runs = (n_frames + SYNTH_RUN - 1) / SYNTH_RUN;
	const uint64_t n = runs * n_streams;
	hipLaunchKernelGGL (k_synth, dim3 ((uint32_t) ((n + 255) / 256)), dim3 (256), 0, (hipStream_t) stream,
	                    d_audio, n_streams, n_frames, stride, seed, fs, kind);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
