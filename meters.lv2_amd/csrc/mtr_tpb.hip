// mtr_tpb.hip — TruePeakdsp::process: 4x interpolation followed by the PPM-style ballistics.
//
// Replaces jmeters/truepeakdsp.cc:41-99 (the dBTP plugin's meter, src/meters.cc:465-507) with the
// call-level semantics of "process(); read(m, p)": per call and channel
//     z1 *= w3, z2 *= w3 once per input frame; for each of the 4 oversampled values v = |y|:
//     if (v > z1) z1 += w1 (v - z1);  if (v > z2) z2 += w2 (v - z2);  p = max (p, v);
//     m = max (m, z1 + z2) once per input frame;  result m * g, p;  states clamped to [0, 20] on
//     entry and offset by 1e-20f on exit.
// The attack/release recurrence is a monotone piece-wise linear map — not an associative linear scan —
// so time stays serial per (stream, channel).  First correct version: one wave
// per stream; the wave interpolates 64 frames at a time in parallel (lane = frame) into LDS, then
// one lane walks the 256 oversampled values of both channels (packed) in order.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

typedef float v2f __attribute__ ((ext_vector_type (2)));

__device__ __forceinline__ v2f vabs (v2f v) { return v2f{fabsf (v.x), fabsf (v.y)}; }
__device__ __forceinline__ v2f vmax (v2f a, v2f b) { return v2f{a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y}; }
// `if (v > z) z += w * (v - z)` per component (truepeakdsp.cc:63-64)
__device__ __forceinline__ v2f attack (v2f z, v2f v, float w)
{
	const v2f t = z + w * (v - z);
	return v2f{v.x > z.x ? t.x : z.x, v.y > z.y ? t.y : z.y};
}

__global__ __launch_bounds__ (64) void k_tpb (const mtr_tpb_args a)
{
	__shared__ float g[3][48];
	__shared__ v2f ov[64][4];
	const int lane = threadIdx.x;
	const uint32_t s = blockIdx.x;
	for (int i = lane; i < 144; i += 64) (&g[0][0])[i] = a.fir_g[i];
	__syncthreads ();

	const int C = (int) a.n_channels;
	const float* src = a.audio + (size_t) s * a.stride * C;
	const float* hist = a.hist + (size_t) s * MTR_FIR_HALO * 2;
	mtr_stream_state* st = a.state + s;

	auto frame = [&] (int64_t f) -> v2f {       // frame f of this call; f < 0 = history of earlier calls
		if (f < 0) return v2f{hist[(MTR_FIR_HALO + f) * 2], hist[(MTR_FIR_HALO + f) * 2 + 1]};
		return C == 2 ? v2f{src[2 * f], src[2 * f + 1]} : v2f{src[f], 0.f};
	};

	v2f z1 = 0, z2 = 0, m = 0, p = 0;
	if (lane == 0) {
		z1 = v2f{st->tpb_z1[0], st->tpb_z1[1]};
		z2 = v2f{st->tpb_z2[0], st->tpb_z2[1]};
		z1 = v2f{z1.x > 20 ? 20 : (z1.x < 0 ? 0 : z1.x), z1.y > 20 ? 20 : (z1.y < 0 ? 0 : z1.y)};   // :54-55
		z2 = v2f{z2.x > 20 ? 20 : (z2.x < 0 ? 0 : z2.x), z2.y > 20 ? 20 : (z2.y < 0 ? 0 : z2.y)};
	}

	for (uint64_t base = 0; base < a.n_frames; base += 64) {
		const int nf = (int) min ((uint64_t) 64, a.n_frames - base);
		if (lane < nf) {
			const int64_t n = (int64_t) base + lane;
			v2f y1 = 0, y2 = 0, y3 = 0;
			for (int i = 0; i < 48; ++i) {
				const v2f x = frame (n - 47 + i);
				y1 += g[0][i] * x;
				y2 += g[1][i] * x;
				y3 += g[2][i] * x;
			}
			ov[lane][0] = vabs (frame (n - 24));      // phase 0 is the identity up to 1e-17
			ov[lane][1] = vabs (y1);
			ov[lane][2] = vabs (y2);
			ov[lane][3] = vabs (y3);
		}
		__syncthreads ();
		if (lane == 0) {
			for (int i = 0; i < nf; ++i) {
				z1 *= a.w3;
				z2 *= a.w3;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const v2f v = ov[i][q];
					z1 = attack (z1, v, a.w1);
					z2 = attack (z2, v, a.w2);
					p = vmax (p, v);
				}
				m = vmax (m, z1 + z2);
			}
		}
		__syncthreads ();
	}
	if (lane == 0) {
		st->tpb_z1[0] = z1.x + 1e-20f; st->tpb_z1[1] = z1.y + 1e-20f;     // :86-87
		st->tpb_z2[0] = z2.x + 1e-20f; st->tpb_z2[1] = z2.y + 1e-20f;
		st->tpb_m[0] = m.x * a.g; st->tpb_m[1] = m.y * a.g;               // :89 then read (m, p)
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

int mtr_launch_tpb (const mtr_tpb_args& a, void* stream)
{
	hipLaunchKernelGGL (k_tpb, dim3 (a.n_streams), dim3 (64), 0, (hipStream_t) stream, a);
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
