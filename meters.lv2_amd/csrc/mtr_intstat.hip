// mtr_intstat.hip — the signal-distribution histogram, for mono streams [S][T] f32: genuinely HBM-bound
// (4 bytes per sample, a few VALU operations per sample), the integer table bit-for-bit the reference's.
//
// (The bit-usage statistics, float_stats of src/bitmeter.c, live in mtr_bitstats.hip.)
//
// k_sigdist replaces the loop of sdh_run (src/sigdistlv2.c:303-318): bin = rintf (180 + 150 x) into
// 361 bins (this TU is built with -ffp-contract=off: the bin decision must not see an FMA),
// peak bin (ties: the bin that reached the final maximum first, as the sequential `>` test gives),
// the plain sum, and the reference's running mean / variance accumulators var_m, var_s in double.
//
// The reference's recurrence (:312-315) divides by `integration_time + s + 1`, the index of the sample among ALL samples,
// also those it skipped as out of range (|x| > 1.2, NaN):
//     m <- m + (v - m) / k,     S <- S + (v - m_new) (v - m_old) = S + (1 - 1/k) (v - m_old)^2 .
//   * While every sample is binned — normalised audio — k is the count of binned samples and this is Welford's update:
//     mean and M2 of the data, order-free.  Fast path: per-lane sums about a pivot over coalesced loads, merged with
//     Chan's formula (equal to the sequential recurrence up to double rounding).
//   * Once a sample has been skipped (in this call or an earlier one) the weights 1/k no longer match the count and
//     var_m / var_s stop being moments of anything — but they are still what the reference reports, so they are
//     reproduced: the update is AFFINE in m (m' = (1 - 1/k) m + v/k) and S gains a QUADRATIC in the incoming m, so a
//     contiguous run of samples maps (m, S) -> (a m + b, S + A m^2 + B m + C); every lane builds that map for its own
//     contiguous chunk of the call (the raw sample index gives k), and the 256 maps are applied in stream order.  This
//     second pass re-reads the stream's call (L2 / HBM) and only runs for streams that have skipped a sample.
// Bins, peak bin, sample count and the plain sum `avg` are the reference's bit for bit in both regimes
// (tests/test_gpu_intstat.py checks both, var_m / var_s against the oracle at 1e-12 relative).
#include <hip/hip_runtime.h>

#include "mtr_internal.h"


__global__ __launch_bounds__ (256) void k_sigdist (const float* audio, uint64_t stride, uint64_t n_frames,
                                                   mtr_sigdist_state* out, uint32_t n_streams)
{
#pragma clang fp contract(off)
	__shared__ int32_t bins[MTR_DIST_BIN];
	__shared__ uint32_t last[MTR_DIST_BIN];              // in-call index + 1 of the last sample that fell in the bin
	__shared__ double mom[256][3];                       // n, mean, M2 per lane, then combined
	__shared__ double sums[256];
	const int tid = threadIdx.x;
	const uint32_t s = blockIdx.x;
	const float* src = audio + (size_t) s * stride;
	mtr_sigdist_state* o = out + s;
	for (int i = tid; i < MTR_DIST_BIN; i += 256) { bins[i] = 0; last[i] = 0; }
	__syncthreads ();

	// lane t takes samples t, t+256, ... (coalesced).  Its moments are kept as sums of d = v - K and d^2
	// about a pivot K (the lane's first binned sample): as accurate as a Welford update — K lies inside
	// the data — without its division per sample (~25 fp64 instructions); mean and M2 of the slice follow
	// at the end and the 256 slices are merged with Chan's pairwise formula (any partition gives the same
	// moments up to double rounding).
	uint32_t cnt = 0, skipped = 0;
	double K = 0, s1 = 0, s2 = 0, sum = 0;
	const uint64_t count0 = (uint64_t) o->count;
	const bool skipped_before = o->n_binned != o->count;       // (read before anybody writes the state back)
	auto one = [&] (float val, uint32_t i) {
		const float fb = rintf (180.f + val * 150.f);          // sigdistlv2.c:305
		// `int bin = rintf (...)` then `if (bin < 0 || bin >= 361) continue`; NaN never passes
		if (!(fb >= 0.f && fb < (float) MTR_DIST_BIN)) { skipped = 1; return; }
		const int bin = (int) fb;
		atomicAdd (&bins[bin], 1);
		atomicMax (&last[bin], i + 1u);
		const double v = (double) val;
		if (cnt == 0) K = v;
		const double d = v - K;
		sum += v;
		s1 += d;
		s2 = fma (d, d, s2);
		++cnt;
	};
	// 16-byte loads (4 samples per lane, 4 KiB per workgroup iteration) where the stream is aligned
	const bool wide = ((((size_t) s * stride) & 3) == 0) && ((reinterpret_cast<size_t> (audio) & 15) == 0);
	const uint32_t n4 = wide ? (uint32_t) (n_frames / 4) : 0;   // n_frames < 2^32 (checked by the engine)
	const float4* src4 = reinterpret_cast<const float4*> (src);
	for (uint32_t q = tid; q < n4; q += 256) {
		const float4 v = src4[q];
		one (v.x, 4 * q); one (v.y, 4 * q + 1); one (v.z, 4 * q + 2); one (v.w, 4 * q + 3);
	}
	for (uint64_t i = (uint64_t) 4 * n4 + tid; i < n_frames; i += 256) one (src[i], (uint32_t) i);
	{
		const double nd = (double) cnt;
		mom[tid][0] = nd;
		mom[tid][1] = cnt ? K + s1 / nd : 0.0;
		mom[tid][2] = cnt ? s2 - s1 * s1 / nd : 0.0;
		sums[tid] = sum;
	}
	// has this stream ever skipped a sample?  (n_binned < count: in an earlier call)
	const bool quirk = __syncthreads_or ((int) skipped) != 0 || skipped_before;
	if (quirk) {
		// the reference's recurrence with the raw sample index as divisor: lane t maps (m, S) over its contiguous chunk
		const uint64_t chunk = (n_frames + 255) / 256;
		const uint64_t i0 = (uint64_t) tid * chunk, i1 = min (i0 + chunk, n_frames);
		double a = 1.0, b = 0.0, A = 0.0, B = 0.0, Cq = 0.0;
		for (uint64_t i = i0; i < i1; ++i) {
			const float val = src[i];
			const float fb = rintf (180.f + val * 150.f);
			if (!(fb >= 0.f && fb < (float) MTR_DIST_BIN)) continue;
			const double v = (double) val, inv = 1.0 / (double) (count0 + i + 1), g = 1.0 - inv;
			const double e = v - b, ga = g * a;                    // v - m_old = e - a m_in
			A = fma (ga, a, A); B = fma (-2.0 * ga, e, B); Cq = fma (g * e, e, Cq);
			a = ga; b = fma (g, b, inv * v);
		}
		mom[tid][0] = a; mom[tid][1] = b; mom[tid][2] = A;
		sums[tid] = sum;
		__shared__ double quad[256][2];
		quad[tid][0] = B; quad[tid][1] = Cq;
		__syncthreads ();
		if (tid == 0) {
			double m = o->var_m, S = o->var_s, Sum = o->avg;
			for (int t = 0; t < 256; ++t) {
				S += fma (mom[t][2] * m, m, fma (quad[t][0], m, quad[t][1]));
				m = fma (mom[t][0], m, mom[t][1]);
			}
			for (int t = 0; t < 256; ++t) Sum += sums[t];
			int64_t nb = 0;
			for (int bq = 0; bq < MTR_DIST_BIN; ++bq) nb += bins[bq];
			o->n_binned += nb; o->var_m = m; o->var_s = S; o->avg = Sum;
		}
	} else if (tid == 0) {
		// fold the carried moments and the 256 slices in stream order (Chan et al.)
		double N = (double) o->n_binned, Mu = o->var_m, M2 = o->var_s, Sum = o->avg;
		for (int t = 0; t < 256; ++t) {
			const double nb = mom[t][0];
			if (nb == 0) continue;
			const double d = mom[t][1] - Mu, tot = N + nb;
			Mu += d * nb / tot;
			M2 += mom[t][2] + d * d * N * nb / tot;
			N = tot;
			Sum += sums[t];
		}
		o->n_binned = (int64_t) N; o->var_m = Mu; o->var_s = M2; o->avg = Sum;
	}
	// histogram + peak: merge this call's bins into the persistent ones
	for (int b = tid; b < MTR_DIST_BIN; b += 256) {
		if (bins[b]) { o->bins[b] += bins[b]; o->last[b] = count0 + last[b]; }
	}
	__syncthreads ();
	__threadfence_block ();
	if (tid == 0) {
		int pc = 0, pb = o->peak_bin;
		unsigned long long pt = ~0ull;
		for (int b = 0; b < MTR_DIST_BIN; ++b) {
			const int c = o->bins[b];
			if (c > pc || (c == pc && c > 0 && o->last[b] < pt)) { pc = c; pb = b; pt = o->last[b]; }
		}
		o->peak_cnt = pc; o->peak_bin = pb;
		o->count += (int64_t) n_frames;
	}
}

int mtr_launch_sigdist (const float* audio, uint64_t stride, uint64_t n_frames, mtr_sigdist_state* out,
                        uint32_t n_streams, void* stream)
{
	hipLaunchKernelGGL (k_sigdist, dim3 (n_streams), dim3 (256), 0, (hipStream_t) stream, audio, stride, n_frames, out, n_streams);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
