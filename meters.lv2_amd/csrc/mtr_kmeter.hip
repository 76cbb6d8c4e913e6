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
// in s: (z1, z2) after a group = A (z1, z2) + (its response to the group's four inputs), A = [[a, 0], [4 w a, b]],
// a = (1 - w)^4, b = 1 - 4 w.  So time is cut into pieces that END at the call's last group; a workgroup runs its
// piece from a zero state — each thread four groups serially, then a tree over the 256 threads with A^4, A^8, ...
// — and a one-thread-per-channel kernel chains the pieces (A^1024 each), adds A^G times the carried state
// (closed form: A^k = [[a^k, 0], [4 w a (a^k - b^k) / (a - b), b^k]]) and does the per-call bookkeeping.  A
// streaming reduction: HBM-bound.  The sums are re-associated (thread runs in f32 as the reference, the
// combination in double): tests/test_gpu_kmeter.py states 1e-5 relative against the restatement.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

namespace {

constexpr int NT = 256;                  // threads per workgroup
constexpr int RG = 4;                    // groups (of four samples) per thread
constexpr int PG = NT * RG;              // 1024 groups = 4096 frames per piece

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
	const uint32_t piece = blockIdx.x, s = blockIdx.y;
	// pieces end at the call's last group: piece p covers groups [G - (n_pieces - p) PG, G - (n_pieces - 1 - p) PG)
	const int64_t g_first = (int64_t) a.n_groups - (int64_t) (a.n_pieces - piece) * PG + (int64_t) threadIdx.x * RG;
	const float* const src = a.audio + (size_t) s * a.stride * C;
	const float w = a.omega;
	float z1[2] = { 0.f, 0.f }, z2[2] = { 0.f, 0.f }, t[2] = { 0.f, 0.f };
	for (int g = 0; g < RG; ++g) {
		const int64_t gi = g_first + g;
		if (gi < 0) continue;                                  // before the call: absent, not zero input
		const float* const p = src + (size_t) gi * 4 * C;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
#pragma unroll
			for (int c = 0; c < C; ++c) {
				float v = p[q * C + c];
				v *= v;
				t[c] = t[c] < v ? v : t[c];                    // kmeterdsp.cc:79: if (t < s) t = s (NaN never enters)
				z1[c] += w * (v - z1[c]);
			}
		}
#pragma unroll
		for (int c = 0; c < C; ++c) z2[c] += 4.f * w * (z1[c] - z2[c]);
	}
	// tree over the threads: e <- A^(len of the right half) e_left + e_right
	__shared__ double sh[NT][4];
	__shared__ float sht[NT][2];
	double e[4] = { z1[0], z2[0], z1[1], z2[1] };
	sh[threadIdx.x][0] = e[0]; sh[threadIdx.x][1] = e[1]; sh[threadIdx.x][2] = e[2]; sh[threadIdx.x][3] = e[3];
	sht[threadIdx.x][0] = t[0]; sht[threadIdx.x][1] = t[1];
	__syncthreads ();
	for (int lvl = 0, d = 1; d < NT; d <<= 1, ++lvl) {
		if ((threadIdx.x & (2 * d - 1)) == 2 * d - 1) {        // the right end of a span of 2 d threads
			const double pa = a.pw[3 * lvl], pc = a.pw[3 * lvl + 1], pb = a.pw[3 * lvl + 2];   // A^(RG d)
			const int l = threadIdx.x - d;
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const double l1 = sh[l][2 * c], l2 = sh[l][2 * c + 1];
				sh[threadIdx.x][2 * c]     += pa * l1;
				sh[threadIdx.x][2 * c + 1] += pc * l1 + pb * l2;
			}
			sht[threadIdx.x][0] = fmaxf (sht[threadIdx.x][0], sht[l][0]);
			sht[threadIdx.x][1] = fmaxf (sht[threadIdx.x][1], sht[l][1]);
		}
		__syncthreads ();
	}
	if (threadIdx.x == NT - 1) {
		const size_t o = ((size_t) s * a.n_pieces + piece) * 4;
		for (int i = 0; i < 4; ++i) a.piece_state[o + i] = sh[NT - 1][i];
		a.piece_max[((size_t) s * a.n_pieces + piece) * 2] = sht[NT - 1][0];
		a.piece_max[((size_t) s * a.n_pieces + piece) * 2 + 1] = sht[NT - 1][1];
	}
}

// one thread per (stream, channel): chain the pieces, add the carried state, then kmeterdsp.cc:108-138
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
	const double pa = a.pw[3 * 8], pc = a.pw[3 * 8 + 1], pb = a.pw[3 * 8 + 2];   // A^PG
	for (uint32_t p = 0; p < a.n_pieces; ++p) {
		const size_t o = ((size_t) s * a.n_pieces + p) * 4 + 2 * c;
		const double n1 = pa * e1 + a.piece_state[o], n2 = pc * e1 + pb * e2 + a.piece_state[o + 1];
		e1 = n1; e2 = n2;
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

void mtr_kmeter_powers (float omega, double* pw /* [9][3] */, double* pw1 /* [3] */)
{
	const double w = (double) omega;
	const double a1 = pow (1.0 - w, 4.0), b1 = 1.0 - 4.0 * w, c1 = 4.0 * w * a1;
	pw1[0] = a1; pw1[1] = c1; pw1[2] = b1;
	for (int lvl = 0; lvl <= 8; ++lvl) {                               // A^(RG 2^lvl); level 8 = A^PG
		const Mat m = mat_pow (a1, c1, b1, (double) (RG << lvl));
		pw[3 * lvl] = m.a; pw[3 * lvl + 1] = m.c; pw[3 * lvl + 2] = m.b;
	}
}

uint32_t mtr_kmeter_pieces (uint64_t n_groups) { return (uint32_t) ((n_groups + PG - 1) / PG); }

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
