// mtr_kmeter.hip — Kmeterdsp (the K-meters' and the TP+RMS plugins' RMS / peak detector) for a batch (gfx950).
//
// Replaces Kmeterdsp::process (jmeters/kmeterdsp.cc:56-140) with the semantics of one process () per engine
// call: per channel, on s = x^2,
//     z1 += w (s - z1) for each sample;  z2 += 4 w (z1 - z2) once per group of four samples (n mod 4 trailing
//     samples are dropped, :71);  t = max s;
//     at the end of the call: rms = sqrt (2 z2) (max-held until read), peak = sqrt (t) with the hold / fall-back
//     bookkeeping of :121-138.
//
// Only the state at the END of the call and the maximum of s enter the result, and the two-pole filter is linear
// in s: (z1, z2) after a group = A (z1, z2) + (w (1 - w)^(3 - q) s_q, q = 0..3, into z1 and 4 w times that into
// z2), A = [[a, 0], [4 w a, b]], a = (1 - w)^4, b = 1 - 4 w.  So the end state is a WEIGHTED SUM of the squares —
// the sample in slot q of the group k groups before the last one weighs A^k (1, 4 w) w (1 - w)^(3 - q) — and the
// kernel is a plain coalesced streaming reduction: lane t of a workgroup takes the groups k = k0 + t, k0 + t +
// 256, ... (its A^k advances by the constant A^256 per step, started from the closed form A^k = [[a^k, 0],
// [4 w a (a^k - b^k) / (a - b), b^k]]), sums in double, and a one-thread-per-channel kernel adds the chunks and
// A^G times the carried state and does the per-call bookkeeping.  (First version: pieces run serially from a
// zero state, four groups per lane, and combined in a tree: each lane read its own 128 contiguous bytes — 8.1 ms
// per 31.5 GB.)  The sums are re-associated and carried in double: tests/test_gpu_kmeter.py states 1e-5
// relative against the restatement.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

namespace {

constexpr int NT = 256;                  // threads per workgroup
constexpr int CH = 32 * NT;              // groups (of four samples) per workgroup: 32768 frames

struct Mat { double a, c, b; };          // [[a, 0], [c, b]]
__host__ __device__ inline Mat mat_pow (double a1, double c1, double b1, double k)
{
	// A^k for A = [[a1, 0], [c1, b1]]: c_k = c1 (a1^k - b1^k) / (a1 - b1)
	const double ak = pow (a1, k), bk = pow (b1, k);
	return Mat{ak, c1 * (ak - bk) / (a1 - b1), bk};
}

template <int C>
__global__ __launch_bounds__ (NT) void k_kmeter_pieces (const mtr_kmeter_args a)
{
	const uint32_t chunk = blockIdx.x, s = blockIdx.y;
	const float* const src = a.audio + (size_t) s * a.stride * C;
	const float w = a.omega, r = 1.f - w;
	const float u3 = w, u2 = w * r, u1 = u2 * r, u0 = u1 * r;             // weight of slot q inside its own group
	// k = groups after this one; chunk c covers k in [c CH, (c + 1) CH)
	uint64_t k = (uint64_t) chunk * CH + threadIdx.x;
	const uint64_t k_end = min ((uint64_t) (chunk + 1) * CH, a.n_groups);
	Mat m = mat_pow (a.pw1[0], a.pw1[1], a.pw1[2], (double) k);
	const Mat st = mat_pow (a.pw1[0], a.pw1[1], a.pw1[2], (double) NT);
	const double w4 = 4.0 * (double) w;
	double z1[2] = { 0, 0 }, z2[2] = { 0, 0 };
	float t[2] = { 0.f, 0.f };
	const bool wide = C == 2 ? ((((size_t) s * a.stride) & 1) == 0 && (reinterpret_cast<size_t> (a.audio) & 15) == 0)
	                         : ((((size_t) s * a.stride) & 3) == 0 && (reinterpret_cast<size_t> (a.audio) & 15) == 0);
	for (; k < k_end; k += NT) {
		const float* const p = src + (size_t) (a.n_groups - 1 - k) * 4 * C;
		float v[4 * C];
		if (wide) {
#pragma unroll
			for (int i = 0; i < C; ++i) {
				const float4 q = reinterpret_cast<const float4*> (p)[i];
				v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
			}
		} else {
#pragma unroll
			for (int i = 0; i < 4 * C; ++i) v[i] = p[i];
		}
		const double U1 = m.a, U2 = m.c + w4 * m.b;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const float s0 = v[c] * v[c], s1 = v[C + c] * v[C + c], s2 = v[2 * C + c] * v[2 * C + c], s3 = v[3 * C + c] * v[3 * C + c];
			t[c] = t[c] < s0 ? s0 : t[c];                          // kmeterdsp.cc:79: if (t < s) t = s (NaN never enters)
			t[c] = t[c] < s1 ? s1 : t[c];
			t[c] = t[c] < s2 ? s2 : t[c];
			t[c] = t[c] < s3 ? s3 : t[c];
			const double g = (double) (u0 * s0) + (double) (u1 * s1) + (double) (u2 * s2) + (double) (u3 * s3);
			z1[c] += U1 * g;
			z2[c] += U2 * g;
		}
		// A^(k + NT) = A^k A^NT
		const double na = m.a * st.a, nc = m.c * st.a + m.b * st.c, nb = m.b * st.b;
		m.a = na; m.c = nc; m.b = nb;
	}
	__shared__ double sh[NT / 64][4];
	__shared__ float sht[NT / 64][2];
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		z1[0] += __shfl_xor (z1[0], d, 64); z2[0] += __shfl_xor (z2[0], d, 64);
		z1[1] += __shfl_xor (z1[1], d, 64); z2[1] += __shfl_xor (z2[1], d, 64);
		t[0] = fmaxf (t[0], __shfl_xor (t[0], d, 64)); t[1] = fmaxf (t[1], __shfl_xor (t[1], d, 64));
	}
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (lane == 0) { sh[wid][0] = z1[0]; sh[wid][1] = z2[0]; sh[wid][2] = z1[1]; sh[wid][3] = z2[1]; sht[wid][0] = t[0]; sht[wid][1] = t[1]; }
	__syncthreads ();
	if (threadIdx.x == 0) {
		const size_t o = (size_t) s * a.n_pieces + chunk;
		double e[4] = { 0, 0, 0, 0 };
		float tt[2] = { 0.f, 0.f };
		for (int i = 0; i < NT / 64; ++i) {
			for (int j = 0; j < 4; ++j) e[j] += sh[i][j];
			tt[0] = fmaxf (tt[0], sht[i][0]); tt[1] = fmaxf (tt[1], sht[i][1]);
		}
		for (int j = 0; j < 4; ++j) a.piece_state[o * 4 + j] = e[j];
		a.piece_max[o * 2] = tt[0]; a.piece_max[o * 2 + 1] = tt[1];
	}
}

// one thread per (stream, channel): add the chunks and the carried state, then kmeterdsp.cc:108-138
__global__ void k_kmeter_final (const mtr_kmeter_args a)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n_streams * a.n_channels) return;
	const uint32_t s = i / a.n_channels, c = i % a.n_channels;
	mtr_kmeter_state* const st = a.state + (size_t) s * 2 + c;
	float zi1 = st->z1 > 50 ? 50 : (st->z1 < 0 ? 0 : st->z1);         // :66-67 (a NaN state falls through, as there)
	float zi2 = st->z2 > 50 ? 50 : (st->z2 < 0 ? 0 : st->z2);
	double e1 = 0, e2 = 0;
	float t = 0.f;
	for (uint32_t p = 0; p < a.n_pieces; ++p) {                        // the weights already carry every chunk to the end
		const size_t o = ((size_t) s * a.n_pieces + p) * 4 + 2 * c;
		e1 += a.piece_state[o]; e2 += a.piece_state[o + 1];
		t = fmaxf (t, a.piece_max[((size_t) s * a.n_pieces + p) * 2 + c]);
	}
	const Mat g = mat_pow (a.pw1[0], a.pw1[1], a.pw1[2], (double) a.n_groups);
	float z1 = (float) (g.a * (double) zi1 + e1);
	float z2 = (float) (g.c * (double) zi1 + g.b * (double) zi2 + e2);
	if (a.n_groups == 0) { z1 = zi1; z2 = zi2; }
	if (isnan (z1)) z1 = 0;                                            // :100-102
	if (isnan (z2)) z2 = 0;
	if (!isfinite (t)) t = 0;
	st->z1 = z1 + 1e-20f;
	st->z2 = z2 + 1e-20f;
	const float r = sqrtf (2.0f * z2);
	t = sqrtf (t);
	if (st->flag) { st->rms = r; st->flag = 0; }                       // :112-118
	else if (r > st->rms) st->rms = r;
	if (t >= st->peak) { st->peak = t; st->cnt = a.hold; }             // :121-138
	else if (st->cnt > 0) st->cnt -= (int32_t) a.fpp;
	else { st->peak *= a.fall; st->peak += 1e-10f; }
}

}  // namespace

void mtr_kmeter_powers (float omega, double* pw1 /* [3] */)
{
	const double w = (double) omega;
	const double a1 = pow (1.0 - w, 4.0), b1 = 1.0 - 4.0 * w, c1 = 4.0 * w * a1;
	pw1[0] = a1; pw1[1] = c1; pw1[2] = b1;
}

uint32_t mtr_kmeter_pieces (uint64_t n_groups) { return (uint32_t) ((n_groups + CH - 1) / CH); }

int mtr_launch_kmeter (const mtr_kmeter_args& a, void* stream)
{
	hipStream_t st = (hipStream_t) stream;
	if (a.n_pieces) {
		if (a.n_channels == 2) hipLaunchKernelGGL (k_kmeter_pieces<2>, dim3 (a.n_pieces, a.n_streams), dim3 (NT), 0, st, a);
		else                   hipLaunchKernelGGL (k_kmeter_pieces<1>, dim3 (a.n_pieces, a.n_streams), dim3 (NT), 0, st, a);
	}
	const uint32_t n = a.n_streams * a.n_channels;
	hipLaunchKernelGGL (k_kmeter_final, dim3 ((n + 63) / 64), dim3 (64), 0, st, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
