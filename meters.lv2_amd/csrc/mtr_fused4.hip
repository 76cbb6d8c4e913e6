// mtr_fused4.hip — layout 6: layout 5's arithmetic (mtr_fused3.hip: k_kw's K-filter + the split-f16 MFMA
// interpolator of mtr_mfma_fir.h, OPT-IN, peaks within 0.0056 dB of the f32 interpolator) laid out for
// occupancy: a workgroup of TWO waves owns a (stream, segment) and walks it in 19-frame lane runs
// (tiles of 1216 frames, half a 50 ms fragment at 48 kHz):
//
//   wave 0  fetches the tile (LDS-DMA), runs the K-filter — reading its run from LDS in both passes instead of
//           holding it in registers — and, inside pass 2, writes the run back as {hi, lo} f16 words;
//   wave 1  turns that tile's words into products once wave 0 has written them, while wave 0 already fetches
//           and filters the next tile (pass 1 and the scan do not touch the words).
//
// Layout 5 needs ~240 registers and 20 KB of LDS per wave (two waves per SIMD, VALU busy 55 %, neither pipe
// the bound: latency is).  Here each role fits 128 registers and a workgroup 20 KB, so a CU holds 8 workgroups
// = 16 waves = FOUR per SIMD.  Two barriers per tile: "words free" (wave 1 is done with the previous tile's
// words; wave 0 may overwrite them) and "words ready".
#include <hip/hip_runtime.h>

#include "mtr_internal.h"
#include "mtr_mfma_fir.h"
#include "mtr_wave.h"

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

__device__ __forceinline__ void wg_barrier ()
{
	__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_s_barrier ();
	__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
}

template <int K, bool EBU>
__global__ __launch_bounds__ (128) __attribute__ ((amdgpu_waves_per_eu (4, 4))) void k_kwtp2 (const mtr_fused_args a)
{
	static_assert ((K & 1) == 1, "odd lane stride: conflict-free LDS accesses");
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	const int wn = (int) a.mfma_words;                                   // words per channel, a multiple of 4
	uint32_t* const WL = reinterpret_cast<uint32_t*> (smem);
	uint32_t* const WR = WL + wn;
	v2f* const buf = reinterpret_cast<v2f*> (WR + wn);                    // the f32 tile
	const int lane = threadIdx.x & 63;
	const int wid = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);

	const uint32_t unit = blockIdx.x;
	const uint32_t s = unit / a.n_segs;
	const uint32_t q = unit - s * a.n_segs;
	mtr_stream_state* const st = a.state + s;

	const uint32_t jt0 = a.seg_tile[q], jt1 = a.seg_tile[q + 1];
	const int64_t seg_start = a.tile_start[jt0];
	const int nwarm = (q > 0) ? (int) a.warm_tiles : 0;
	const int ntile = (int) (jt1 - jt0);
	constexpr int LT = 64 * K;

	auto tile_of = [&] (int jj, int64_t& t0, int& len) {
		if (jj < 0) { t0 = seg_start + (int64_t) jj * LT; len = LT; }
		else        { t0 = a.tile_start[jt0 + jj]; len = (int) (a.tile_start[jt0 + jj + 1] - (uint32_t) t0); }
	};

	// zeros everywhere first (see mtr_fused3.hip: what a product may read must be finite)
	for (int i = threadIdx.x; i < 2 * wn; i += 128) WL[i] = 0u;
	wg_barrier ();

	if (wid == 0) {
		// ============================ wave 0: fetch + K-filter + split ============================
		const v2f* const src = reinterpret_cast<const v2f*> (a.audio) + (size_t) s * a.stride;
		const bool src_even = ((((size_t) s * a.stride) & 1) == 0) && ((reinterpret_cast<size_t> (a.audio) & 15) == 0);
		const float a0 = a.a0, a1 = a.a1, a2 = a.a2, b1 = a.b1, b2 = a.b2, c3 = a.c3, c4 = a.c4;
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

		stage (-nwarm);
		for (int jj = -nwarm; jj < ntile; ++jj) {
			int64_t t0; int len;
			tile_of (jj, t0, len);
			const int run0 = lane * K;
			const int rl = min (max (len - run0, 0), K);
			const v2f* const xr = buf + (int) (t0 & 1) + run0;

			asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");      // this tile has landed
			__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");

			v2f z1 = 0, z2 = 0, z3 = 0, z4 = 0;
			if (EBU) {
				// pass 1 + scan: everything that does not need the word arrays (wave 1 may still be reading them)
				mtrw::RowMats rm;
				rm.load (a.scan_m + 96 + 4 * K + 4, lane);
				if (rl == K) {
					z1 = F[4 * K + 0]; z2 = F[4 * K + 1]; z3 = F[4 * K + 2]; z4 = F[4 * K + 3];
#pragma unroll
					for (int n = 0; n < K; ++n) {
						const v2f x = xr[n];
						z1 += F[4 * n + 0] * x; z2 += F[4 * n + 1] * x; z3 += F[4 * n + 2] * x; z4 += F[4 * n + 3] * x;
					}
				}
				if (lane == 0) {
					const cfloat_p M = CM;
					z1 += M[0] * k1 + M[1] * k2;
					z2 += M[4] * k1 + M[5] * k2;
					z3 += M[8] * k1 + M[9] * k2 + M[10] * k3 + M[11] * k4;
					z4 += M[12] * k1 + M[13] * k2 + M[14] * k3 + M[15] * k4;
				}
				mtrw::scan (z1, z2, z3, z4, CM, rm);
			}

			wg_barrier ();                                         // words free: wave 1 is done with the previous tile

			uint32_t* const wl = WL + HALO + run0;
			uint32_t* const wr = WR + HALO + run0;
			if (EBU && jj >= 0) {
				z1 = mtrw::from_left (z1); z2 = mtrw::from_left (z2); z3 = mtrw::from_left (z3); z4 = mtrw::from_left (z4);
				if (lane == 0) { z1 = k1; z2 = k2; z3 = k3; z4 = k4; }
				// pass 2 (see mtr_fused3.hip for the two lane masks) with the split of every frame riding along
				const int last_l = (len - 1) / K, rl_last = len - last_l * K;
				const bool upto = lane <= last_l, before = lane < last_l;
				v2f sj = 0;
#pragma unroll
				for (int n = 0; n < K; ++n) {
					const v2f x = n < rl ? xr[n] : v2f{0.f, 0.f};
					uint32_t l_, r_;
					mfir::split_words (x.x, x.y, l_, r_);
					wl[n] = l_; wr[n] = r_;
					if (n < rl_last) { if (upto) { v2f y; KW_STEP (x, y); sj += y * y; } }
					else             { if (before) { v2f y; KW_STEP (x, y); sj += y * y; } }
				}
				const float sl = mtrw::sum63 (sj.x), sr = mtrw::sum63 (sj.y);
				if (lane == 0) a.tile_power[(size_t) s * a.n_tiles + jt0 + jj] = a.gain_l * sl + a.gain_r * sr;
				k1 = mtrw::pick (z1, last_l); k2 = mtrw::pick (z2, last_l); k3 = mtrw::pick (z3, last_l); k4 = mtrw::pick (z4, last_l);
			} else {
				if (EBU) { k1 = mtrw::pick (z1, 63); k2 = mtrw::pick (z2, 63); k3 = mtrw::pick (z3, 63); k4 = mtrw::pick (z4, 63); }
#pragma unroll
				for (int n = 0; n < K; ++n) {
					const v2f x = n < rl ? xr[n] : v2f{0.f, 0.f};
					uint32_t l_, r_;
					mfir::split_words (x.x, x.y, l_, r_);
					wl[n] = l_; wr[n] = r_;
				}
			}
			if (EBU) { k1 = scrub (k1); k2 = scrub (k2); k3 = scrub (k3); k4 = scrub (k4); }

			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");    // every read of the tile has returned
			wg_barrier ();                                         // words ready
			if (jj + 1 < ntile) stage (jj + 1);                    // lands while wave 1 multiplies and this wave waits
		}
		if (EBU && q == a.n_segs - 1 && lane == 0) {
			st->kz[0] = k1.x; st->kz[1] = k1.y; st->kz[2] = k2.x; st->kz[3] = k2.y;
			st->kz[4] = k3.x; st->kz[5] = k3.y; st->kz[6] = k4.x; st->kz[7] = k4.y;
		}
	} else {
		// ============================ wave 1: the products ============================
		mfir::AFrag A;
		A.load (a.mfma_a, lane);
		float pk_l = 0.f, pk_r = 0.f;
		uint32_t halo_l = 0u, halo_r = 0u;                           // lanes 0..46: the 47 words before the next tile
		if (lane < HALO && q == 0) {
			const float* const h = a.hist + ((size_t) s * HALO + (size_t) lane) * 2;
			halo_l = mfir::split_word (h[0]);
			halo_r = mfir::split_word (h[1]);
		}
		const int fo = 8 * (lane & 31) + 4 * (lane >> 5);            // + (r & 3): output frame of register r in its block
		for (int jj = -nwarm; jj < ntile; ++jj) {
			int64_t t0; int len;
			tile_of (jj, t0, len);
			wg_barrier ();                                           // words free (this wave's own reads are done)
			if (lane < HALO) { WL[lane] = halo_l; WR[lane] = halo_r; }
			wg_barrier ();                                           // words ready
			if (jj >= 0) {
				int b0 = 0;
				for (; b0 + 256 <= len; b0 += 256) {
					mfir::f16x yl, yr;
					mfir::tile2_stream (A, WL, WR, b0, lane, yl, yr);
#pragma unroll
					for (int r = 0; r < 16; r += 2) {
						pk_l = fmaxf (fmaxf (pk_l, fabsf (yl[r])), fabsf (yl[r + 1]));
						pk_r = fmaxf (fmaxf (pk_r, fabsf (yr[r])), fabsf (yr[r + 1]));
					}
				}
				if (b0 < len) {
					// the tile's last block: frames past its end belong to the next tile (or do not exist yet)
					mfir::f16x yl, yr;
					mfir::tile2_stream (A, WL, WR, b0, lane, yl, yr);
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
			}
			if (lane < HALO) { halo_l = WL[len + lane]; halo_r = WR[len + lane]; }
			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		const float unscale = 1.f / (float) (1 << MTR_MFMA_TAP_SHIFT);   // exact
		pk_l = mtrw::max63 (pk_l) * unscale;
		pk_r = mtrw::max63 (pk_r) * unscale;
		if (lane == 0) {
			atomicMax (&st->tp_call[0], __float_as_uint (pk_l));
			atomicMax (&st->tp_call[1], __float_as_uint (pk_r));
		}
	}
}

template <int K>
int launch_kwtp2 (bool ebu, const mtr_fused_args& a, uint32_t n_units, hipStream_t st)
{
	const size_t lds = (size_t) 2 * a.mfma_words * sizeof (uint32_t) + (size_t) a.buf_slots * sizeof (v2f);
	if (ebu) hipLaunchKernelGGL ((k_kwtp2<K, true>), dim3 (n_units), dim3 (128), lds, st, a);
	else     hipLaunchKernelGGL ((k_kwtp2<K, false>), dim3 (n_units), dim3 (128), lds, st, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

}  // namespace

int mtr_launch_kwtp2 (int run, bool ebu, const mtr_fused_args& a, uint32_t n_units, void* stream)
{
	switch (run) {
	case 19: return launch_kwtp2<19> (ebu, a, n_units, (hipStream_t) stream);
	default: return -2;
	}
}
