// mtr_intstat.hip — the integer paths: IEEE-754 bit-usage statistics and the signal-distribution
// histogram, for mono streams [S][T] f32.  Both are genuinely HBM-bound (4 bytes per sample, a few
// VALU operations per sample) and both reproduce the reference's integer tables bit-for-bit.
//
// k_bitstats replaces float_stats (src/bitmeter.c:63-105, table layout src/uris.h:53-60).  The
// reference walks the 23 mantissa bits of every sample and bumps hits[exp+k] (always), ones[exp+k]
// and mant[k] (bit set).  All three tables are projections of one 2-D count C[e][k] = samples with
// (effective) exponent e and mantissa bit k set, plus E[e] = samples with exponent e:
//     hits[p] = sum_{k=0..22} E[p-k] (+ normals[p-23] for the implicit one),
//     ones[p] = sum_k C[p-k][k]     (+ normals[p-23]),        mant[k] = sum_e C[e][k].
// One workgroup owns one stream and keeps C (256 x 24 int32, column 23 = E) in LDS.  A wave takes 64
// consecutive samples: 23 ballots transpose the mantissa bits so that lane k holds the 64-bit mask
// "which samples have bit k set" (lane 23: all live samples); then for every distinct exponent
// present in the wave (a handful for audio: the distribution is geometric) lane k adds
// popcount(mask_k & lanes_with_e) to C[e][k] — 24 conflict-free LDS adds per distinct exponent instead of
// ~12 same-address atomics per sample.
//
// k_sigdist replaces the loop of sdh_run (src/sigdistlv2.c:303-318): bin = rintf (180 + 150 x) into
// 361 bins (this TU is built with -ffp-contract=off: the bin decision must not see an FMA),
// peak bin (ties: the bin that reached the final maximum first, as the sequential `>` test gives),
// sum and Welford mean / M2 in double (combined across lanes with Chan's formula: equal to the
// sequential recurrence up to double rounding).
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

#define BIM_DHIT 0
#define BIM_NHIT 23
#define BIM_DONE 280
#define BIM_NONE 303
#define BIM_DSET 560

__global__ __launch_bounds__ (256) void k_bitstats (const float* audio, uint64_t stride, uint64_t n_frames,
                                                    mtr_bitstats_state* out, uint32_t n_streams)
{
	__shared__ int32_t C[256][24];
	__shared__ int32_t cnt[5];            // zero pos nan inf den
	__shared__ float   red[2][4];
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const uint32_t s = blockIdx.x;
	const uint32_t* src = reinterpret_cast<const uint32_t*> (audio) + (size_t) s * stride;
	for (int i = tid; i < 256 * 24; i += 256) (&C[0][0])[i] = 0;
	if (tid < 5) cnt[tid] = 0;
	__syncthreads ();

	int n_zero = 0, n_pos = 0, n_nan = 0, n_inf = 0, n_den = 0;
	float vmin = INFINITY, vmax = 0.f;

	// one round = 64 samples, one per lane (any assignment of samples to lanes counts the same)
	auto round64 = [&] (uint32_t bits, bool in) {
		uint32_t ex = (bits >> 23) & 0xffu;
		const uint32_t man = bits & 0x7fffffu;
		const bool special = ex == 255;
		const bool zero = ex == 0 && man == 0;
		const bool live = in && !special && !zero;
		if (in) {
			n_inf += special && man == 0;
			n_nan += special && man != 0;
			n_zero += zero;
			n_den += (ex == 0 && man != 0);
			n_pos += live && !(bits >> 31);
			if (live && ex > 0) {
				const float v = __uint_as_float (bits & 0x7fffffffu);
				vmax = v > vmax ? v : vmax;
				vmin = v < vmin ? v : vmin;
			}
		}
		if (ex == 0) ex = 1;                   // denormals sit at 2^-126 (bitmeter.c:94)
		// transpose: lane k <- mask of samples with mantissa bit k set; lane 23 <- all live samples
		// (v_writelane puts the scalar ballot straight into lane k: no compare / select per bit)
		int mine_lo = 0, mine_hi = 0;
#pragma unroll
		for (int k = 0; k < 23; ++k) {
			const unsigned long long m = __ballot (live && ((man >> k) & 1u));
			asm volatile ("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(mine_lo) : "s"((int) (uint32_t) m), "n"(k));
			asm volatile ("v_writelane_b32 %0, %1, %2" : "+v"(mine_hi) : "s"((int) (uint32_t) (m >> 32)), "n"(k));
		}
		const unsigned long long livemask = __ballot (live);
		asm volatile ("s_nop 4\n\tv_writelane_b32 %0, %1, 23" : "+v"(mine_lo) : "s"((int) (uint32_t) livemask));
		asm volatile ("v_writelane_b32 %0, %1, 23" : "+v"(mine_hi) : "s"((int) (uint32_t) (livemask >> 32)));
		const unsigned long long mine = ((unsigned long long) (uint32_t) mine_hi << 32) | (uint32_t) mine_lo;
		// one pass per distinct exponent present among the live samples of this wave
		unsigned long long todo = livemask;
		while (todo) {
			const int first = __ffsll ((long long) todo) - 1;
			const uint32_t e = (uint32_t) __builtin_amdgcn_readlane ((int) ex, first);   // scalar read, no LDS round trip
			const unsigned long long same = __ballot (live && ex == e);
			if (lane < 24) {
				const int c = __popcll (mine & same);
				if (c) atomicAdd (&C[e][lane], c);
			}
			todo &= ~same;
		}
	};

	// a wave takes chunks of 256 consecutive samples: one 16-byte load per lane (1 KiB per wave
	// instruction), the next chunk's load issued before this chunk is counted
	const bool wide = ((((size_t) s * stride) & 3) == 0) && ((reinterpret_cast<size_t> (audio) & 15) == 0);
	const uint64_t n_full = wide ? (n_frames / 256) : 0;        // whole chunks, shared out over the 4 waves
	if (n_full > (uint64_t) wid) {
		const uint4* src4 = reinterpret_cast<const uint4*> (src);
		uint64_t c = wid;
		uint4 cur = src4[c * 64 + lane];
		while (true) {
			const uint64_t cn = c + 4;
			const bool more = cn < n_full;
			uint4 nxt = cur;
			if (more) nxt = src4[cn * 64 + lane];
			round64 (cur.x, true); round64 (cur.y, true); round64 (cur.z, true); round64 (cur.w, true);
			if (!more) break;
			cur = nxt; c = cn;
		}
	}
	for (uint64_t base = n_full * 256 + (uint64_t) wid * 64; base < n_frames; base += 256) {   // tail / unaligned
		const uint64_t i = base + lane;
		const bool in = i < n_frames;
		round64 (in ? src[i] : 0u, in);
	}
	// counters and min/max
	for (int d = 32; d >= 1; d >>= 1) {
		n_zero += __shfl_xor (n_zero, d, 64); n_pos += __shfl_xor (n_pos, d, 64);
		n_nan += __shfl_xor (n_nan, d, 64);   n_inf += __shfl_xor (n_inf, d, 64);
		n_den += __shfl_xor (n_den, d, 64);
		vmin = fminf (vmin, __shfl_xor (vmin, d, 64));
		vmax = fmaxf (vmax, __shfl_xor (vmax, d, 64));
	}
	if (lane == 0) {
		atomicAdd (&cnt[0], n_zero); atomicAdd (&cnt[1], n_pos); atomicAdd (&cnt[2], n_nan);
		atomicAdd (&cnt[3], n_inf);  atomicAdd (&cnt[4], n_den);
		red[0][wid] = vmin; red[1][wid] = vmax;
	}
	__syncthreads ();

	mtr_bitstats_state* o = out + s;
	// project C onto the reference's table; positions p = e + k, p in [1, 277]
	for (int p = tid; p < 280; p += 256) {
		int hits = 0, ones = 0;
		for (int k = 0; k < 23; ++k) {
			const int e = p - k;
			if (e >= 1 && e <= 254) { hits += C[e][23]; ones += C[e][k]; }
		}
		const int e = p - 23;                       // the implicit one: normals only
		if (e >= 1 && e <= 254) {
			const int normals = C[e][23] - (e == 1 ? cnt[4] : 0);
			hits += normals; ones += normals;
		}
		o->hist[BIM_DHIT + p] += hits;
		o->hist[BIM_DONE + p] += ones;
	}
	if (tid < 23) {
		int m = 0;
		for (int e = 1; e <= 254; ++e) m += C[e][tid];
		o->hist[BIM_DSET + tid] += m;
	}
	if (tid == 0) {
		o->n_zero += cnt[0]; o->n_pos += cnt[1]; o->n_nan += cnt[2]; o->n_inf += cnt[3]; o->n_den += cnt[4];
		float mn = o->vmin, mx = o->vmax;
		for (int w = 0; w < 4; ++w) { mn = fminf (mn, red[0][w]); mx = fmaxf (mx, red[1][w]); }
		o->vmin = mn; o->vmax = mx;
	}
}

__global__ __launch_bounds__ (256) void k_sigdist (const float* audio, uint64_t stride, uint64_t n_frames,
                                                   mtr_sigdist_state* out, uint32_t n_streams)
{
#pragma clang fp contract(off)
	__shared__ int32_t bins[MTR_DIST_BIN];
	__shared__ unsigned long long last[MTR_DIST_BIN];   // index + 1 of the last sample that fell in the bin
	__shared__ double mom[256][3];                       // n, mean, M2 per lane, then combined
	__shared__ double sums[256];
	const int tid = threadIdx.x;
	const uint32_t s = blockIdx.x;
	const float* src = audio + (size_t) s * stride;
	mtr_sigdist_state* o = out + s;
	for (int i = tid; i < MTR_DIST_BIN; i += 256) { bins[i] = 0; last[i] = 0; }
	__syncthreads ();

	// lane t takes samples t, t+256, ... (coalesced); its Welford state covers that subsequence and the
	// 256 partial states are merged with Chan's pairwise formula (any partition gives the same moments)
	double n = 0, mean = 0, m2 = 0, sum = 0;
	const uint64_t count0 = (uint64_t) o->count;
	auto one = [&] (float val, uint64_t i) {
		const float fb = rintf (180.f + val * 150.f);          // sigdistlv2.c:305
		// `int bin = rintf (...)` then `if (bin < 0 || bin >= 361) continue`; NaN never passes
		if (!(fb >= 0.f && fb < (float) MTR_DIST_BIN)) return;
		const int bin = (int) fb;
		atomicAdd (&bins[bin], 1);
		atomicMax (&last[bin], (unsigned long long) (count0 + i + 1));
		sum += val;
		n += 1;
		const double d = (double) val - mean;
		mean += d / n;
		m2 += ((double) val - mean) * d;
	};
	// 16-byte loads (4 samples per lane, 4 KiB per workgroup iteration) where the stream is aligned
	const bool wide = ((((size_t) s * stride) & 3) == 0) && ((reinterpret_cast<size_t> (audio) & 15) == 0);
	const uint64_t n4 = wide ? (n_frames / 4) : 0;
	const float4* src4 = reinterpret_cast<const float4*> (src);
	for (uint64_t q = tid; q < n4; q += 256) {
		const float4 v = src4[q];
		one (v.x, 4 * q); one (v.y, 4 * q + 1); one (v.z, 4 * q + 2); one (v.w, 4 * q + 3);
	}
	for (uint64_t i = 4 * n4 + tid; i < n_frames; i += 256) one (src[i], i);
	mom[tid][0] = n; mom[tid][1] = mean; mom[tid][2] = m2; sums[tid] = sum;
	__syncthreads ();
	if (tid == 0) {
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
		if (bins[b]) { o->bins[b] += bins[b]; o->last[b] = last[b]; }
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

int mtr_launch_bitstats (const float* audio, uint64_t stride, uint64_t n_frames, mtr_bitstats_state* out,
                         uint32_t n_streams, void* stream)
{
	hipLaunchKernelGGL (k_bitstats, dim3 (n_streams), dim3 (256), 0, (hipStream_t) stream, audio, stride, n_frames, out, n_streams);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

int mtr_launch_sigdist (const float* audio, uint64_t stride, uint64_t n_frames, mtr_sigdist_state* out,
                        uint32_t n_streams, void* stream)
{
	hipLaunchKernelGGL (k_sigdist, dim3 (n_streams), dim3 (256), 0, (hipStream_t) stream, audio, stride, n_frames, out, n_streams);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
