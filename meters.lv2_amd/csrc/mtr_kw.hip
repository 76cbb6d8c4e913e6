// mtr_kw.hip — K-weighting only (EBU R128 without true peak), the HBM-bound layout (gfx950).
//
// Replaces Ebu_r128_proc::detect_process (ebumeter/ebu_r128_proc.cc:302-337) for a whole batch.
// Same exact time-parallel scheme as k_fused2's K-filter role (per-lane K-frame run from a zero
// state -> wave scan with powers of A^K -> second pass from the true state), laid out for
// bandwidth instead of for sharing a tile with three interpolator waves:
//
//   * one WAVE owns a (stream, time segment); workgroup = one wave, so there is no barrier at all;
//   * ONE LDS buffer per wave (a tile is only transposed through it: DMA writes frame order, lane l
//     reads its run [lK, lK+K) with odd stride K -> conflict-free 64-bit LDS reads);
//   * pass 1 reads the run into registers and keeps it there: as soon as the reads have returned the
//     buffer is free, the LDS-DMA of the NEXT tile is issued and lands while the scan and pass 2
//     (two thirds of the arithmetic) run from registers;
//   * 20 KB (K = 39, one 50 ms fragment per tile) or 10 KB (K = 19, half a fragment) of LDS per
//     wave -> 8 or 16 waves per CU; the other waves of the SIMD cover the DMA wait;
//   * all cross-lane traffic on the DPP path (mtr_wave.h), none through ds_bpermute.
//
// No halo, no look-ahead: the K-filter needs only the tile itself.  Tiles may start on an odd frame
// (44.1 kHz fragments are 2205 frames): the DMA then starts one frame early and the runs read from
// slot offset 1, so the 16-byte source alignment of global_load_lds_dwordx4 always holds.
#include <hip/hip_runtime.h>

#include <utility>

#include "mtr_internal.h"
#include "mtr_wave.h"
#include "mtr_kw_steps.h"

namespace {

__device__ __forceinline__ v2f scrub (v2f v) { return v2f{isfinite (v.x) ? v.x : 0.f, isfinite (v.y) ? v.y : 0.f}; }

template <int K>
__global__ __launch_bounds__ (64) void k_kw (const mtr_fused_args a)
{
	static_assert ((K & 1) == 1, "odd lane stride: conflict-free 64-bit LDS reads");
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	v2f* const buf = reinterpret_cast<v2f*> (smem);
	const int lane = threadIdx.x;

	const uint32_t unit = blockIdx.x;
	const uint32_t s = unit / a.n_segs;
	const uint32_t q = unit - s * a.n_segs;
	const v2f* const src = reinterpret_cast<const v2f*> (a.audio) + (size_t) s * a.stride;
	mtr_stream_state* const st = a.state + s;
	const bool src_even = ((((size_t) s * a.stride) & 1) == 0) && ((reinterpret_cast<size_t> (a.audio) & 15) == 0);
	const float a0 = a.a0, a1 = a.a1, a2 = a.a2, b1 = a.b1, b2 = a.b2, c3 = a.c3, c4 = a.c4;

	const uint32_t jt0 = a.seg_tile[q], jt1 = a.seg_tile[q + 1];
	const int64_t seg_start = a.tile_start[jt0];
	const int nwarm = (q > 0) ? (int) a.warm_tiles : 0;
	const int ntile = (int) (jt1 - jt0);
	constexpr int LT = 64 * K;

	auto tile_of = [&] (int jj, int64_t& t0, int& len) {
		if (jj < 0) { t0 = seg_start + (int64_t) jj * LT; len = LT; }
		else        { t0 = a.tile_start[jt0 + jj]; len = (int) (a.tile_start[jt0 + jj + 1] - (uint32_t) t0); }
	};

	// Frames [t0 - off, t0 + len) -> slots [0, len + off), off = t0 & 1.
	auto stage = [&] (int jj) {
		int64_t t0; int len;
		tile_of (jj, t0, len);
		const int off = (int) (t0 & 1);
		const int nslot = len + off;
		// the last pair of a call that ends on an odd frame count would read one frame past the stream
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

	v2f k1 = 0, k2 = 0, k3 = 0, k4 = 0;            // carried state, wave-uniform
	if (q == 0) {
		k1 = v2f{st->kz[0], st->kz[1]}; k2 = v2f{st->kz[2], st->kz[3]};
		k3 = v2f{st->kz[4], st->kz[5]}; k4 = v2f{st->kz[6], st->kz[7]};
	}
	// The constant tables are read through the constant address space: they are never written while a
	// kernel runs, and only this tells the compiler that the tile_power stores in the loop cannot
	// clobber them, so wave-uniform entries stay scalar loads instead of 160 VGPRs of hoisted copies.
	typedef const __attribute__ ((address_space (4))) float* cfloat_p;
	const cfloat_p CM = (cfloat_p) a.scan_m;       // (A^K)^(2^d), d = 0..5
	const cfloat_p F = CM + 96;                    // end-state functionals, 4 per frame of a run, + the bias response
	const v2f e1 = F[4 * K + 0], e2 = F[4 * K + 1], e3 = F[4 * K + 2], e4 = F[4 * K + 3];
	mtrw::RowMats rm;
	rm.load (a.scan_m + 96 + 4 * K + 4, lane);

	stage (-nwarm);
	for (int jj = -nwarm; jj < ntile; ++jj) {
		int64_t t0; int len;
		tile_of (jj, t0, len);
		const int run0 = lane * K;
		const int rl = min (max (len - run0, 0), K);
		const v2f* const xr = buf + (int) (t0 & 1) + run0;

		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");      // this tile has landed
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");

		// the lane's run into registers (slots past the tile hold stale data: masked below)
		v2f x[K];
#pragma unroll
		for (int n = 0; n < K; ++n) x[n] = xr[n];
		asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");    // every read has returned: the buffer is free
		if (jj + 1 < ntile) stage (jj + 1);                    // lands while the scan and pass 2 run from registers

		// pass 1: end state of the run from a zero start state = a linear functional of its K inputs
		v2f z1 = e1, z2 = e2, z3 = e3, z4 = e4;
#pragma unroll
		for (int n = 0; n < K; ++n) {
			z1 += F[4 * n + 0] * x[n]; z2 += F[4 * n + 1] * x[n]; z3 += F[4 * n + 2] * x[n]; z4 += F[4 * n + 3] * x[n];
		}
		if (rl != K) { z1 = 0; z2 = 0; z3 = 0; z4 = 0; }       // partial run: nothing to its right consumes it
		if (lane == 0) {
			const cfloat_p M = CM;
			z1 += M[0] * k1 + M[1] * k2;
			z2 += M[4] * k1 + M[5] * k2;
			z3 += M[8] * k1 + M[9] * k2 + M[10] * k3 + M[11] * k4;
			z4 += M[12] * k1 + M[13] * k2 + M[14] * k3 + M[15] * k4;
		}
		// wave scan on the DPP path: z_l <- sum_{j<=l} (A^K)^(l-j) e_j
		mtrw::scan (z1, z2, z3, z4, CM, rm);

		if (jj < 0) {                                          // warm-up tile: only the state matters
			k1 = mtrw::pick (z1, 63); k2 = mtrw::pick (z2, 63); k3 = mtrw::pick (z3, 63); k4 = mtrw::pick (z4, 63);
		} else {
			// pass 2: from the true start state (end state of the lane to the left), sum y^2
			z1 = mtrw::from_left (z1); z2 = mtrw::from_left (z2); z3 = mtrw::from_left (z3); z4 = mtrw::from_left (z4);
			if (lane == 0) { z1 = k1; z2 = k2; z3 = k3; z4 = k4; }
			// frames 0 .. K - 2 in hand-scheduled pairs (mtr_kw_steps.h): one lane per tile has a partial run — steps
			// n < rl_last run in the lanes up to and including it, the others in the lanes before it; frame K - 1 (K is odd)
			// in the lanes whose run is whole
			const int last_l = (len - 1) / K, rl_last = len - last_l * K;
			const uint64_t upto = __ballot (lane <= last_l), before = __ballot (lane < last_l);
			const v2f A0 = v2f{a0, a0}, A1 = v2f{a1, a1}, A2 = v2f{a2, a2}, B1 = v2f{b1, b1}, B2 = v2f{b2, b2}, C3 = v2f{c3, c3}, C4 = v2f{c4, c4};
			const v2f eps2 = v2f{1e-15f, 1e-15f};
			v2f sj = 0;
			[&]<int... P> (std::integer_sequence<int, P...>) {
				(kw_pair<2 * P> (x[2 * P], x[2 * P + 1], z1, z2, z3, z4, sj, A0, A1, A2, B1, B2, C3, C4, eps2, upto, before, rl_last), ...);
			} (std::make_integer_sequence<int, (K - 1) / 2> {});
			if ((rl_last & 1) && rl_last < K) {
				// the partial lane stopped between the two steps of a pair: its shelving states sit swapped
				const v2f w1 = mtrw::pick (z1, last_l), w2 = mtrw::pick (z2, last_l);
				if (lane == last_l) { z1 = w2; z2 = w1; }
			}
			if (rl == K) { v2f y; KW_STEP (x[K - 1], y); sj += y * y; }
			const float pw = mtrw::sum63 (a.gain_l * sj.x + a.gain_r * sj.y);
			if (lane == 0) a.tile_power[(size_t) s * a.n_tiles + jt0 + jj] = pw;
			const int last = (len - 1) / K;                    // the lane holding the state after the last frame
			k1 = mtrw::pick (z1, last); k2 = mtrw::pick (z2, last); k3 = mtrw::pick (z3, last); k4 = mtrw::pick (z4, last);
		}
		// ebu_r128_proc.cc:331-334: non-finite states are dropped at block ends
		k1 = scrub (k1); k2 = scrub (k2); k3 = scrub (k3); k4 = scrub (k4);
	}
	if (q == a.n_segs - 1 && lane == 0) {
		st->kz[0] = k1.x; st->kz[1] = k1.y; st->kz[2] = k2.x; st->kz[3] = k2.y;
		st->kz[4] = k3.x; st->kz[5] = k3.y; st->kz[6] = k4.x; st->kz[7] = k4.y;
	}
}

template <int K>
int launch_kw (const mtr_fused_args& a, uint32_t n_units, hipStream_t st)
{
	const size_t lds = (size_t) a.buf_slots * sizeof (v2f);
	hipLaunchKernelGGL ((k_kw<K>), dim3 (n_units), dim3 (64), lds, st, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

}  // namespace

int mtr_launch_kw (int run, const mtr_fused_args& a, uint32_t n_units, void* stream)
{
	switch (run) {
	case 39: return launch_kw<39> (a, n_units, (hipStream_t) stream);
	case 19: return launch_kw<19> (a, n_units, (hipStream_t) stream);
	default: return -2;
	}
}
