// mtr_fused3.hip — K-weighting + true peak with the interpolator on the matrix pipe (gfx950), layout 5.
//
// OPTIONAL (tune_layout = 5), not the default: the default k_fused2 computes the 4x interpolator in exact
// f32 on the VALU and is bound by it (17 % of the HBM roofline).  This kernel is k_kw
// (mtr_kw.hip: one wave per (stream, segment), LDS-DMA, the exact time-parallel K-filter — unchanged,
// the loudness results are bit-identical to layout 4) plus the split-f16 MFMA interpolator of
// mtr_mfma_fir.h, whose peaks differ from the f32 result by at most 0.0055 dB (taps rounded to f16;
// samples carried as hi + lo with 22 bits) — inside the +-0.01 dB of the parity clause, but not
// bit-identical, hence opt-in.
//
// Per tile (one 50 ms fragment), one wave:
//   1. the tile has landed in the f32 buffer (LDS-DMA); each lane reads its K-frame run into registers,
//      the buffer is free again and the DMA of the next tile is issued;
//   2. the run is written back as {hi, lo} f16 words, one array per channel, behind a 47-frame halo:
//      array position i <-> frame t0 - 47 + i, so a window start is a plain index and every MFMA operand
//      is one aligned 16-byte LDS read;
//   3. K-filter: pass 1 -> DPP scan -> pass 2 (as k_kw);
//   4. ceil(len / 256) x 2 channels MFMA tiles of 256 frames x 4 phases (7 MFMAs each), max |y| per lane;
//   5. the last 47 words of each array move to the front: the next tile's halo.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"
#include "mtr_mfma_fir.h"
#include "mtr_wave.h"

// -DMTR_F3_PROF: cycles per phase, summed over the tiles of workgroup 0 (read back with mtr_debug_f3_prof)
#ifdef MTR_F3_PROF
__device__ unsigned long long g_f3_prof[8];
#define PROF_NOW(v) unsigned long long v; asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define PROF_ADD(i, d) pr[i] += (d)
#else
#define PROF_NOW(v)
#define PROF_ADD(i, d)
#endif

namespace {

__device__ __forceinline__ v2f scrub (v2f v) { return v2f{isfinite (v.x) ? v.x : 0.f, isfinite (v.y) ? v.y : 0.f}; }

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

constexpr int HALO = MTR_FIR_HALO;       // 47

// INPLACE: the word arrays take the place of the f32 tile once it sits in registers, so a wave needs ONE
// region of LDS (20 KB at K = 39) and two waves fit a SIMD — at the price of fetching the next tile only
// after this one's products (the other wave of the SIMD covers that wait).  The halo words cross the
// fetch in registers.
template <int K, bool EBU, bool INPLACE>
__global__ __launch_bounds__ (64) void k_kwtp (const mtr_fused_args a)
{
	static_assert ((K & 1) == 1, "odd lane stride: conflict-free LDS accesses");
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	// [W left][W right][f32 tile buffer] (INPLACE: the tile lies over the two word arrays).  Operand reads of
	// columns past the end of a short tile — masked outputs — are clamped into their array (mfir::fetch_b)
	const int wn = (int) a.mfma_words;                                   // words per channel, a multiple of 4
	uint32_t* const WL = reinterpret_cast<uint32_t*> (smem);
	uint32_t* const WR = WL + wn;
	v2f* const buf = INPLACE ? reinterpret_cast<v2f*> (smem) : reinterpret_cast<v2f*> (WR + wn);
	const int lane = threadIdx.x;

	const uint32_t unit = blockIdx.x;
	const uint32_t s = unit / a.n_segs;
	const uint32_t q = unit - s * a.n_segs;
	const v2f* const src = reinterpret_cast<const v2f*> (a.audio) + (size_t) s * a.stride;
	mtr_stream_state* const st = a.state + s;
	const bool src_even = ((((size_t) s * a.stride) & 1) == 0) && ((reinterpret_cast<size_t> (a.audio) & 15) == 0);
	// The K-filter coefficients live in vector registers here: pass 1 wants 4 K scalars at once, the scalar file
	// overflows, and spilled scalars cost a v_readlane per use — eight per frame in pass 2 when they were these.
	v2f a0 = a.a0, a1 = a.a1, a2 = a.a2, b1 = a.b1, b2 = a.b2, c3 = a.c3, c4 = a.c4;
	asm volatile ("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b1), "+v"(b2), "+v"(c3), "+v"(c4));

	const uint32_t jt0 = a.seg_tile[q], jt1 = a.seg_tile[q + 1];
	const int64_t seg_start = a.tile_start[jt0];
	const int nwarm = (q > 0) ? (int) a.warm_tiles : 0;
	const int ntile = (int) (jt1 - jt0);
	constexpr int LT = 64 * K;

	auto tile_of = [&] (int jj, int64_t& t0, int& len) {
		if (jj < 0) { t0 = seg_start + (int64_t) jj * LT; len = LT; }
		else        { t0 = a.tile_start[jt0 + jj]; len = (int) (a.tile_start[jt0 + jj + 1] - (uint32_t) t0); }
	};
	// Frames [t0 - off, t0 + len) -> slots [0, len + off), off = t0 & 1 (mtr_kw.hip)
	auto stage = [&] (int jj) {
		int64_t t0; int len;
		tile_of (jj, t0, len);
		const int off = (int) (t0 & 1);
		const int nslot = len + off;
		const bool tail_odd = (t0 + len == (int64_t) a.n_frames) && (a.n_frames & 1);
		if (src_even && !tail_odd) {
			const v2f* const p = src + (t0 - off) + 2 * lane;
			for (int i = 0; i < nslot; i += 128) {
				if (i + 2 * lane < nslot)
				__builtin_amdgcn_global_load_lds ((const __attribute__ ((address_space (1))) void*) (p + i),
				                                  (__attribute__ ((address_space (3))) void*) (buf + i), 16, 0, 0);
			}
		} else {
			for (int i = lane; i < nslot; i += 64) buf[i] = src[t0 - off + i];
		}
	};

	v2f k1 = 0, k2 = 0, k3 = 0, k4 = 0;            // carried K-filter state, wave-uniform
	if (EBU && q == 0) {
		k1 = v2f{st->kz[0], st->kz[1]}; k2 = v2f{st->kz[2], st->kz[3]};
		k3 = v2f{st->kz[4], st->kz[5]}; k4 = v2f{st->kz[6], st->kz[7]};
	}
	typedef const __attribute__ ((address_space (4))) float* cfloat_p;
	const cfloat_p CM = (cfloat_p) a.scan_m;
	const cfloat_p F = CM + 96;
	const v2f e1 = F[4 * K + 0], e2 = F[4 * K + 1], e3 = F[4 * K + 2], e4 = F[4 * K + 3];
	mtrw::RowMats rm;
	if (EBU) rm.load (a.scan_m + 96 + 4 * K + 4, lane);

	mfir::AFrag A;
	A.load (a.mfma_a, lane);
	float pk_l = 0.f, pk_r = 0.f;
	float run_l = 0.f, run_r = 0.f, tmax_prev = 3.0e38f;                 // pruning: wave-wide peaks so far; no bound on the first halo
	uint32_t n_done = 0, n_skip = 0;

	// Both word arrays start as zeros: a matrix column reads up to 9 words past its last output's window (K is
	// padded from 55 to 56 samples) and whole columns past the end of a short tile; those products carry
	// zero taps or are masked, but 0 x NaN is NaN, so what they read must at least be finite.
	for (int i = lane; i < 2 * wn; i += 64) WL[i] = 0u;
	asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
	// halo of the first tile: the 47 frames before the call (segment 0) — later segments get theirs from the
	// warm-up tiles, which are converted like any other
	uint32_t halo_l = 0u, halo_r = 0u;                                   // INPLACE: lanes 0..46 carry the halo words
	if (lane < HALO) {
		if (q == 0) {
			const float* const h = a.hist + ((size_t) s * HALO + (size_t) lane) * 2;
			halo_l = mfir::split_word (h[0]);
			halo_r = mfir::split_word (h[1]);
		}
		WL[lane] = halo_l;
		WR[lane] = halo_r;
	}
	asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");

#ifdef MTR_F3_PROF
	unsigned long long pr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
	stage (-nwarm);
	for (int jj = -nwarm; jj < ntile; ++jj) {
		int64_t t0; int len;
		tile_of (jj, t0, len);
		PROF_NOW (c0_);
		const int run0 = lane * K;
		const int rl = min (max (len - run0, 0), K);
		const v2f* const xr = buf + (int) (t0 & 1) + run0;

		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");      // this tile has landed
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
		PROF_NOW (c1_); PROF_ADD (0, c1_ - c0_);

		v2f x[K];
#pragma unroll
		for (int n = 0; n < K; ++n) x[n] = xr[n];
		asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");    // every read has returned: the buffer is free
		if (!INPLACE && jj + 1 < ntile) stage (jj + 1);
		// slots past the tile hold stale data: zero them once, here, so that nothing below needs a per-frame test
		// (a per-frame `if (n < rl)` is a lane-mask update and a branch per frame; zeroing the tail in LDS
		// before the reads instead was measured: no gain)
#pragma unroll
		for (int n = 0; n < K; ++n) x[n] = n < rl ? x[n] : v2f{0.f, 0.f};
		PROF_NOW (c2_); PROF_ADD (1, c2_ - c1_);

		// Exact peak pruning (a.prune, as in k_fused2): |y| <= L1 * max|x| over the 48-frame window, so a tile
		// whose max|x| (its own and the previous tile's, which holds the halo) times the largest L1 norm
		// cannot beat what both channels already hold needs no products — the result is unchanged bit for bit.
		// 2.5684 = the largest L1 norm of the f32 taps; the margin covers their rounding to f16 and the f32 sums.
		bool skip = false;
		if (a.prune && jj >= -1) {
			float tm = 0.f;
#pragma unroll
			for (int n = 0; n < K; ++n) tm = fmaxf (fmaxf (tm, fabsf (x[n].x)), fabsf (x[n].y));
			tm = mtrw::max63 (tm);
			skip = jj >= 0 && 2.5684f * 1.002f * (float) (1 << MTR_MFMA_TAP_SHIFT) * fmaxf (tm, tmax_prev) <= fminf (run_l, run_r);
			tmax_prev = tm;
			n_done += jj >= 0; n_skip += skip;
		}

		// A pruned tile's words are never multiplied: all that is needed of them is the next tile's halo, and its 47
		// frames are still there as f32 (in place, nothing has been written over the tile yet).
		const bool no_words = INPLACE && skip && len >= HALO;
		if (no_words) {
			if (lane < HALO) {
				const v2f h = buf[(int) (t0 & 1) + len - HALO + lane];
				mfir::split_words (h.x, h.y, halo_l, halo_r);
			}
			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		// the run as {hi, lo} words behind the halo
		if (!no_words) {
			if (INPLACE) {
				// the tile was lying over both arrays: put the halo back, and zeros into the pad behind the left array
				if (lane < HALO) { WL[lane] = halo_l; WR[lane] = halo_r; }
				if (lane < 12) WL[HALO + 64 * K + lane] = 0u;
			}
			uint32_t* const wl = WL + HALO + run0;
			uint32_t* const wr = WR + HALO + run0;
#pragma unroll
			for (int n = 0; n < K; ++n) {
				uint32_t l_, r_;
				mfir::split_words (x[n].x, x[n].y, l_, r_);
				wl[n] = l_; wr[n] = r_;              // all 64 K positions: past the tile they are zeros
			}
		}

		PROF_NOW (c3_); PROF_ADD (2, c3_ - c2_);
		if (EBU) {
			v2f z1 = e1, z2 = e2, z3 = e3, z4 = e4;
#pragma unroll
			for (int n = 0; n < K; ++n) {
				z1 += F[4 * n + 0] * x[n]; z2 += F[4 * n + 1] * x[n]; z3 += F[4 * n + 2] * x[n]; z4 += F[4 * n + 3] * x[n];
			}
			if (rl != K) { z1 = 0; z2 = 0; z3 = 0; z4 = 0; }
			if (lane == 0) {
				const cfloat_p M = CM;
				z1 += M[0] * k1 + M[1] * k2;
				z2 += M[4] * k1 + M[5] * k2;
				z3 += M[8] * k1 + M[9] * k2 + M[10] * k3 + M[11] * k4;
				z4 += M[12] * k1 + M[13] * k2 + M[14] * k3 + M[15] * k4;
			}
			PROF_NOW (p1_); PROF_ADD (5, p1_ - c3_);
			mtrw::scan (z1, z2, z3, z4, CM, rm);
			PROF_NOW (p2_); PROF_ADD (0, p2_ - p1_);
			if (jj < 0) {
				k1 = mtrw::pick (z1, 63); k2 = mtrw::pick (z2, 63); k3 = mtrw::pick (z3, 63); k4 = mtrw::pick (z4, 63);
			} else {
				z1 = mtrw::from_left (z1); z2 = mtrw::from_left (z2); z3 = mtrw::from_left (z3); z4 = mtrw::from_left (z4);
				if (lane == 0) { z1 = k1; z2 = k2; z3 = k3; z4 = k4; }
				// Only one lane has a partial run (the tile's last active one): the set of lanes that take step n is
				// "up to and including it" while n < its run length and "all before it" afterwards — two
				// loop-invariant lane masks and a scalar test per step, not a vector compare per step.
				const int last_l = (len - 1) / K, rl_last = len - last_l * K;
				const bool upto = lane <= last_l, before = lane < last_l;
				v2f sj = 0;
#pragma unroll
				for (int n = 0; n < K; ++n) {
					if (n < rl_last) { if (upto) { v2f y; KW_STEP (x[n], y); sj += y * y; } }
					else             { if (before) { v2f y; KW_STEP (x[n], y); sj += y * y; } }
				}
				const float sl = mtrw::sum63 (sj.x), sr = mtrw::sum63 (sj.y);
				if (lane == 0) a.tile_power[(size_t) s * a.n_tiles + jt0 + jj] = a.gain_l * sl + a.gain_r * sr;
				const int last = (len - 1) / K;
				k1 = mtrw::pick (z1, last); k2 = mtrw::pick (z2, last); k3 = mtrw::pick (z3, last); k4 = mtrw::pick (z4, last);
			}
			k1 = scrub (k1); k2 = scrub (k2); k3 = scrub (k3); k4 = scrub (k4);
		}

		PROF_NOW (c4_); PROF_ADD (3, c4_ - c3_);
		// the interpolator: 256 output frames x 4 phases per MFMA tile and channel
		if (jj >= 0 && !skip) {
			__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");            // the words of every lane are in LDS
			const int fo = 8 * (lane & 31) + 4 * (lane >> 5);              // + (r & 3): output frame of register r in its tile
			mfir::BFrag bl, br;
			const int last = (wn - 52) & ~3;                                 // 6 steps of 8 words + 4 behind a fragment's start
			mfir::fetch_b (bl, WL, 0, lane, last);
			mfir::fetch_b (br, WR, 0, lane, last);
			int b0 = 0;
			for (; b0 + 256 <= len; b0 += 256) {
				// the next block's operands are fetched under this block's products (past the last block the
				// read is harmless: finite words inside the allocation)
				mfir::BFrag nl, nr;
				mfir::fetch_b (nl, WL, b0 + 256, lane, last);
				mfir::fetch_b (nr, WR, b0 + 256, lane, last);
				mfir::f16x yl, yr;
				mfir::tile2 (A, bl, br, yl, yr);
#pragma unroll
				for (int r = 0; r < 16; r += 2) {
					pk_l = fmaxf (fmaxf (pk_l, fabsf (yl[r])), fabsf (yl[r + 1]));
					pk_r = fmaxf (fmaxf (pk_r, fabsf (yr[r])), fabsf (yr[r + 1]));
				}
				bl = nl; br = nr;
			}
			if (b0 < len) {
				// the tile's last block: frames past its end belong to the next tile (or do not exist yet).
				// Register r holds output frame b0 + fo + (r & 3): four lane masks.
				mfir::f16x yl, yr;
				mfir::tile2 (A, bl, br, yl, yr);
				const int lim = len - b0 - fo;
				float ml = 0.f, mr = 0.f;
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const bool ok = (r & 3) < lim;
					ml = fmaxf (ml, ok ? fabsf (yl[r]) : 0.f);
					mr = fmaxf (mr, ok ? fabsf (yr[r]) : 0.f);
				}
				pk_l = fmaxf (pk_l, ml);
				pk_r = fmaxf (pk_r, mr);
			}
			if (a.prune) { run_l = mtrw::max63 (pk_l); run_r = mtrw::max63 (pk_r); }
		}

		PROF_NOW (c5_); PROF_ADD (4, c5_ - c4_);
		// the next tile's halo: the last 47 frames before t0 + len sit at positions len .. len + 46
		{
			if (!no_words && lane < HALO) { halo_l = WL[len + lane]; halo_r = WR[len + lane]; }
			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
			if (INPLACE) {
				if (jj + 1 < ntile) stage (jj + 1);                // over the words: they are spent
			} else if (lane < HALO) {
				WL[lane] = halo_l; WR[lane] = halo_r;
			}
		}
		PROF_NOW (c6_); PROF_ADD (5, c6_ - c5_); PROF_ADD (6, c6_ - c0_); PROF_ADD (7, 1);
	}
#ifdef MTR_F3_PROF
	if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 8; ++i) g_f3_prof[i] = pr[i];
#endif
	if (EBU && q == a.n_segs - 1 && lane == 0) {
		st->kz[0] = k1.x; st->kz[1] = k1.y; st->kz[2] = k2.x; st->kz[3] = k2.y;
		st->kz[4] = k3.x; st->kz[5] = k3.y; st->kz[6] = k4.x; st->kz[7] = k4.y;
	}
	if (a.prune && lane == 0 && a.prune_stats) {
		atomicAdd (&a.prune_stats[0], n_done);
		atomicAdd (&a.prune_stats[1], n_skip);
	}
	const float unscale = 1.f / (float) (1 << MTR_MFMA_TAP_SHIFT);       // exact
	pk_l = mtrw::max63 (pk_l) * unscale;
	pk_r = mtrw::max63 (pk_r) * unscale;
	if (lane == 0) {
		atomicMax (&st->tp_call[0], __float_as_uint (pk_l));
		atomicMax (&st->tp_call[1], __float_as_uint (pk_r));
	}
}

template <int K>
int launch_kwtp (bool ebu, bool inplace, const mtr_fused_args& a, uint32_t n_units, hipStream_t st)
{
	const size_t words = (size_t) 2 * a.mfma_words * sizeof (uint32_t), tile = (size_t) a.buf_slots * sizeof (v2f);
	if (inplace) {
		const size_t lds = words > tile ? words : tile;
		if (ebu) hipLaunchKernelGGL ((k_kwtp<K, true, true>), dim3 (n_units), dim3 (64), lds, st, a);
		else     hipLaunchKernelGGL ((k_kwtp<K, false, true>), dim3 (n_units), dim3 (64), lds, st, a);
	} else {
		const size_t lds = words + tile;
		if (ebu) hipLaunchKernelGGL ((k_kwtp<K, true, false>), dim3 (n_units), dim3 (64), lds, st, a);
		else     hipLaunchKernelGGL ((k_kwtp<K, false, false>), dim3 (n_units), dim3 (64), lds, st, a);
	}
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

}  // namespace

int mtr_launch_kwtp (int run, bool ebu, const mtr_fused_args& a, uint32_t n_units, void* stream)
{
	const bool inplace = a.fir_form != 3;                                // tune_fir = 3: separate tile buffer (first form, kept for comparison)
	switch (run) {
	case 39: return launch_kwtp<39> (ebu, inplace, a, n_units, (hipStream_t) stream);
	case 19: return launch_kwtp<19> (ebu, inplace, a, n_units, (hipStream_t) stream);
	default: return -2;
	}
}

#ifdef MTR_F3_PROF
extern "C" int mtr_debug_f3_prof (unsigned long long* out)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_f3_prof), 8 * sizeof (unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
