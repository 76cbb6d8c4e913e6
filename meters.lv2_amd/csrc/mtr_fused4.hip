// mtr_fused4.hip — K-weighting + true peak, the interpolator on the matrix pipe at f32 grade (gfx950), layout 6.
//
// Replaces, for a whole batch, Ebu_r128_proc::detect_process (ebumeter/ebu_r128_proc.cc:302-337) and
// Resampler::process + TruePeakdsp::process_max (zita-resampler/resampler.cc:211-235,
// jmeters/truepeakdsp.cc:101-124).  The 4x interpolator is 144 MACs per channel-sample against 4 bytes: a dense
// contraction, and what bounds the exact-f32 VALU kernel (mtr_fused2.hip) at 17 % of the HBM roofline.  Here it runs
// on v_mfma_f32_16x16x32_f16 with samples AND taps split into two f16 halves and three of the four partial
// products kept (mtr_mfma16_fir.h): 2^-23-relative factors, f32 accumulation — as accurate as the f32 fmaf chain
// (measured against a float64 interpolator: 0.8e-7 of the peak vs the chain's 3e-7), held by the parity tests to
// the same 2e-6 relative bound as the VALU interpolator.
//
// One wave per (stream, time segment), workgroup = one wave, no barrier.  Per tile (one 50 ms fragment, <= 64 K frames):
//   1. the tile is in the lanes' run registers (K frames per lane; K even: runs start on even frames) and lanes 0..23
//      hold the 48 frames in front of it — loaded by the previous tile's step 6;
//   2. per channel: max |x| over halo and tile -> a power-of-two scale that puts it in [2^14, 2^15): the f16 halves
//      then carry 22+ bits at any level, from denormal streams to +/-3e38; an Inf / NaN sample poisons only the
//      outputs it reaches, and the VALU keeps max |x| (phase 0) exact;
//   3. the run goes to LDS as f16 hi / lo pair words (four arrays, 20 KB: all the LDS the kernel uses);
//   4. K-filter on the unscaled registers: pass 1 -> DPP scan -> pass 2 (as k_kw: loudness is the same arithmetic);
//   5. the next tile's global loads are issued, straight into the run registers (dead from here on);
//   6. ceil(len / 256) blocks x 2 channels x 18 MFMAs under which those loads land, |max| of the twelve
//      accumulators, last block masked.
// Two waves per SIMD.  (The first form — the f32 tile staged through LDS by LDS-DMA, read back transposed — was 2-3 %
// slower on every shape measured and left the library in round 3; so did layouts 1, 2 and 5.)
// Big batches that start on a fragment boundary take the lane = time segment kernel instead (mtr_seg.hip, layout 7);
// this kernel serves every other call, and finishes those (the part of the call behind the last whole fragment).
#include <hip/hip_runtime.h>

#include <utility>

#include "mtr_internal.h"
#include "mtr_mfma16_fir.h"
#include "mtr_wave.h"
#include "mtr_kw_steps.h"

// -DMTR_F4_PROF: shader cycles per phase, summed over the tiles of one workgroup (read back with mtr_debug_f4_prof)
#ifdef MTR_F4_PROF
__device__ unsigned long long g_f4_prof[12];
#define PROF_NOW(v) unsigned long long v; asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define PROF_ADD(i, d) pr_[i] += (d)
#else
#define PROF_NOW(v)
#define PROF_ADD(i, d)
#endif

namespace {

__device__ __forceinline__ v2f scrub (v2f v) { return v2f{isfinite (v.x) ? v.x : 0.f, isfinite (v.y) ? v.y : 0.f}; }

constexpr int HALO = MTR_M16_HALO;       // 48

// max (m, |a|, |b|) — one v_max3_f32 with source modifiers; NaN operands lose
__device__ __forceinline__ float max3abs (float m, float a, float b) { return fmaxf (fmaxf (m, fabsf (a)), fabsf (b)); }

// power-of-two scale that puts a value with exponent field `e` (0 = zero / denormal .. 255 = Inf) into [2^14, 2^15),
// and 2^-15 / scale
__device__ __forceinline__ void pow2_scale (uint32_t e_, float& scale, float& unscale)
{
	const int e = (int) e_;
	const int se = min (238, 268 - e);                   // exponent field of the scale: 127 + 14 - (e - 127), clamped for tm < 2^-97
	scale = __uint_as_float ((uint32_t) se << 23);
	unscale = __uint_as_float ((uint32_t) (239 - se) << 23);      // 2^-(se - 127) * 2^-15
}

template <int K, bool EBU>
__global__ __launch_bounds__ (64, 2) void k_kwtp16 (const mtr_fused_args a)
{
	static_assert ((K & 1) == 0, "even runs: sample pairs never straddle two lanes");
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	constexpr int WN = (HALO + 64 * K) / 2;                            // words per array: positions 0 .. HALO + 64 K - 1
	static_assert (WN % 4 == 0, "arrays start on 16 bytes");
	constexpr int CMAX = (HALO + 64 * K) / 16 - 4;                      // last column whose 64-sample window lies inside the arrays
	uint32_t* const HL = reinterpret_cast<uint32_t*> (smem);
	uint32_t* const HR = HL + WN;
	uint32_t* const LL = HL + 2 * WN;
	uint32_t* const LR = HL + 3 * WN;
	const int lane = threadIdx.x;

	const uint32_t unit = blockIdx.x;
	const uint32_t s = unit / a.n_segs;
	const uint32_t q = unit - s * a.n_segs;
	const v2f* const src = reinterpret_cast<const v2f*> (a.audio) + (size_t) s * a.stride;
	mtr_stream_state* const st = a.state + s;
	const bool src_even = ((((size_t) s * a.stride) & 1) == 0) && ((reinterpret_cast<size_t> (a.audio) & 15) == 0);
	v2f a0 = a.a0, a1 = a.a1, a2 = a.a2, b1 = a.b1, b2 = a.b2, c3 = a.c3, c4 = a.c4;
	asm volatile ("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b1), "+v"(b2), "+v"(c3), "+v"(c4));
	const v2f eps2 = v2f{1e-15f, 1e-15f};

	const uint32_t jt0 = a.seg_tile[q], jt1 = a.seg_tile[q + 1];
	const int64_t seg_start = a.tile_start[jt0];
	const int nwarm = (q > 0) ? (int) a.warm_tiles : 0;
	const int ntile = (int) (jt1 - jt0);
	constexpr int LT = 64 * K;
	const int64_t id_end = (int64_t) a.n_frames - 24;                  // phase 0 of this call ends with frame n_frames - 25

	auto tile_of = [&] (int jj, int64_t& t0, int& len) {
		if (jj < 0) { t0 = seg_start + (int64_t) jj * LT; len = LT; }
		else        { t0 = a.tile_start[jt0 + jj]; len = (int) (a.tile_start[jt0 + jj + 1] - (uint32_t) t0); }
	};
	// The next tile goes straight into the run registers — which are dead during the products — with per-lane
	// global loads issued BEFORE the MFMA phase: lane l reads its own 304 contiguous bytes (19 x 16 bytes; the 64 lanes
	// of one instruction touch 64 cache lines, each of which the following seven instructions hit again in the L1),
	// lanes 0..23 the two halo frames in front of the tile.  No f32 image in LDS, no DMA issue behind the products, no
	// transposing read: the stream arrives under the matrix work.  Frames past the tile in the last lanes' runs are the
	// next tile's (real, finite or not): every use below masks them (K-filter: lane masks; products: outputs past the
	// tile are masked and no valid output's window reaches them; scale: lanes past the tile are left out).
	const v2f* const hst = reinterpret_cast<const v2f*> (a.hist) + (size_t) s * MTR_FIR_HALO;
	auto frame_at = [&] (int64_t f) -> v2f {                             // call-relative frame, history in front of frame 0
		if (f >= 0) return src[f];
		if (f >= -MTR_FIR_HALO && q == 0) return hst[f + MTR_FIR_HALO];
		return v2f{0.f, 0.f};
	};
	auto fetch = [&] (int jj, v2f (&xn)[K], v2f& g0, v2f& g1) {
		int64_t t0; int len;
		tile_of (jj, t0, len);
		const int64_t base = t0 + (int64_t) K * lane;
		if (src_even && !(t0 & 1) && t0 + (int64_t) LT <= (int64_t) a.n_frames) {
			const float4* const p4 = reinterpret_cast<const float4*> (src + base);
#pragma unroll
			for (int i = 0; i < K / 2; ++i) { const float4 v = p4[i]; xn[2 * i] = v2f{v.x, v.y}; xn[2 * i + 1] = v2f{v.z, v.w}; }
		} else {
#pragma unroll
			for (int n = 0; n < K; ++n) xn[n] = base + n < (int64_t) a.n_frames ? src[base + n] : v2f{0.f, 0.f};
		}
		g0 = v2f{0.f, 0.f}; g1 = v2f{0.f, 0.f};
		if (lane < HALO / 2) { g0 = frame_at (t0 - HALO + 2 * lane); g1 = frame_at (t0 - HALO + 2 * lane + 1); }
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

	m16::AFrag A;
	A.load (a.mfma_a, lane);
	float pk_l = 0.f, pk_r = 0.f;                                      // the segment's peaks so far, per lane
	uint32_t n_done = 0, n_skip = 0, n_blk = 0, n_fin = 0;

	// halo of the first tile: the 47 frames before the call behind one zero (segment 0); later segments start with
	// warm-up tiles, whose own halo is never multiplied.  Lane i < 24 holds positions 2 i and 2 i + 1.
	v2f h0 = v2f{0.f, 0.f}, h1 = v2f{0.f, 0.f};
	v2f x[K];
	fetch (-nwarm, x, h0, h1);
	const int wrun = HALO / 2 + (K / 2) * lane;                        // first word of this lane's run in each array
	const int col8 = 8 * (lane & 15), kg4 = 4 * (lane >> 4);

#ifdef MTR_F4_PROF
	unsigned long long pr_[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
	for (int jj = -nwarm; jj < ntile; ++jj) {
		int64_t t0; int len;
		tile_of (jj, t0, len);

		PROF_NOW (c0_);
		asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");      // this tile has landed in the run registers
		PROF_NOW (c1_); PROF_ADD (0, c1_ - c0_);
		// the per-lane matrices of the scan's row-broadcast steps: 24 registers, fetched per tile (L1 / L2 hits that
		// land under the split) rather than held across the products
		mtrw::RowMats rm;
		if (EBU) rm.load (a.scan_m + 96 + 4 * K + 4, lane);

		// this tile's halo (positions 0..47) is in h0 / h1
		const v2f ph0 = h0, ph1 = h1;
		PROF_NOW (c2_); PROF_ADD (1, c2_ - c1_);

		// per-lane max |x| per channel
		float ml = 0.f, mr = 0.f;
#pragma unroll
		for (int n = 0; n < K; n += 2) { ml = max3abs (ml, x[n].x, x[n + 1].x); mr = max3abs (mr, x[n].y, x[n + 1].y); }
		float hl = 0.f, hr = 0.f;
		if (lane < HALO / 2) { hl = fmaxf (fabsf (ph0.x), fabsf (ph1.x)); hr = fmaxf (fabsf (ph0.y), fabsf (ph1.y)); }
		// The scales need only the exponents of the two window maxima: both ride through ONE wave reduction as a pair
		// of 16-bit fields (v_pk_max_u16); NaNs have lost every fmaxf above, an Inf gives exponent 255.
		const bool inside = K * lane < len;                           // lanes past the tile hold the next tile's frames
		const uint32_t epair = (__float_as_uint (fmaxf (inside ? ml : 0.f, hl)) >> 23) | ((__float_as_uint (fmaxf (inside ? mr : 0.f, hr)) >> 23) << 16);
		const uint32_t emax = mtrw::max63_u16x2 (epair);

		// Phase 0 (|x[n - 24]|) of this call covers frames [-24, n_frames - 24): every frame of the tile unless the call
		// ends within 24 frames of it; the 24 frames before the call's first tile come from the history.  Peaks are
		// kept per lane and reduced once, at the end of the segment.
		if (jj >= 0) {
			float il = ml, ir = mr;
			if (t0 + LT > id_end) {
				il = 0.f; ir = 0.f;
				const int64_t lim = id_end - t0 - (int64_t) K * lane;          // frames of this lane's run inside the range
#pragma unroll
				for (int n = 0; n < K; ++n) if (n < lim) { il = fmaxf (il, fabsf (x[n].x)); ir = fmaxf (ir, fabsf (x[n].y)); }
			}
			if (q == 0 && jj == 0 && lane >= HALO / 4 && lane < HALO / 2) {
				// positions 24..47 <-> frames t0 - 24 .. t0 - 1; position p counts while frame t0 + p - 48 < n_frames - 24 (t0 = 0
				// unless this launch finishes a call behind the lane = segment kernel: a tail shorter than 24 frames leaves
				// the last frames in front of it to the next call, as k_seg's p0_end does — ADVICE r3)
				const int64_t plim = (int64_t) a.n_frames + 24 - t0;
				if (2 * lane < plim)     { il = fmaxf (il, fabsf (ph0.x)); ir = fmaxf (ir, fabsf (ph0.y)); }
				if (2 * lane + 1 < plim) { il = fmaxf (il, fabsf (ph1.x)); ir = fmaxf (ir, fabsf (ph1.y)); }
			}
			pk_l = fmaxf (pk_l, il);
			pk_r = fmaxf (pk_r, ir);
		}

		// Exact peak pruning (a.prune): |y| <= L1 * max |x| over the window, 2.5684 = the largest L1 norm of the taps;
		// a channel whose bound cannot beat its peak so far needs no products (the result is unchanged bit for bit).
		// Only this option needs the maxima and the peaks so far wave-wide, per tile.
		bool need_l = jj >= 0, need_r = jj >= 0;
		float tml = 0.f, tmr = 0.f, run_l = 0.f, run_r = 0.f;
		if (a.prune && jj >= 0) {
			tml = mtrw::max63 (fmaxf (inside ? ml : 0.f, hl)); tmr = mtrw::max63 (fmaxf (inside ? mr : 0.f, hr));
			run_l = mtrw::max63 (pk_l); run_r = mtrw::max63 (pk_r);
			need_l = 2.5684f * 1.002f * tml > run_l;
			need_r = 2.5684f * 1.002f * tmr > run_r;
			n_done += 1; n_skip += !(need_l || need_r);
		}

		PROF_NOW (c3_); PROF_ADD (2, c3_ - c2_);
		float sc_l, sc_r, un_l, un_r;
		pow2_scale (emax & 0xffffu, sc_l, un_l);
		pow2_scale (emax >> 16, sc_r, un_r);
		if (need_l || need_r) {
			// the run as f16 hi / lo pair words behind the halo, scaled
			const v2f sc = v2f{sc_l, sc_r};
			uint32_t* const w = HL + wrun;
#pragma unroll
			for (int i = 0; i < K / 2; ++i) {
				const v2f u = x[2 * i] * sc, v = x[2 * i + 1] * sc;
				uint32_t hi, lo;
				m16::split_pair (u.x, v.x, hi, lo);
				w[i] = hi; w[2 * WN + i] = lo;
				m16::split_pair (u.y, v.y, hi, lo);
				w[WN + i] = hi; w[3 * WN + i] = lo;
			}
			if (lane < HALO / 2) {
				const v2f u = ph0 * sc, v = ph1 * sc;
				uint32_t hi, lo;
				m16::split_pair (u.x, v.x, hi, lo);
				HL[lane] = hi; LL[lane] = lo;
				m16::split_pair (u.y, v.y, hi, lo);
				HR[lane] = hi; LR[lane] = lo;
			}
		}

		PROF_NOW (c4_); PROF_ADD (3, c4_ - c3_);
		if (EBU) {
			v2f z1 = e1, z2 = e2, z3 = e3, z4 = e4;
#pragma unroll
			for (int n = 0; n < K; ++n) {
				z1 += F[4 * n + 0] * x[n]; z2 += F[4 * n + 1] * x[n]; z3 += F[4 * n + 2] * x[n]; z4 += F[4 * n + 3] * x[n];
			}
			const int rl = min (max (len - K * lane, 0), K);
			if (rl != K) { z1 = 0; z2 = 0; z3 = 0; z4 = 0; }
			if (lane == 0) {
				const cfloat_p M = CM;
				z1 += M[0] * k1 + M[1] * k2;
				z2 += M[4] * k1 + M[5] * k2;
				z3 += M[8] * k1 + M[9] * k2 + M[10] * k3 + M[11] * k4;
				z4 += M[12] * k1 + M[13] * k2 + M[14] * k3 + M[15] * k4;
			}
			PROF_NOW (k1_); PROF_ADD (9, k1_ - c4_);
			mtrw::scan (z1, z2, z3, z4, CM, rm);
			PROF_NOW (k2_); PROF_ADD (10, k2_ - k1_);
			if (jj < 0) {
				k1 = mtrw::pick (z1, 63); k2 = mtrw::pick (z2, 63); k3 = mtrw::pick (z3, 63); k4 = mtrw::pick (z4, 63);
			} else {
				z1 = mtrw::from_left (z1); z2 = mtrw::from_left (z2); z3 = mtrw::from_left (z3); z4 = mtrw::from_left (z4);
				if (lane == 0) { z1 = k1; z2 = k2; z3 = k3; z4 = k4; }
				// one lane has a partial run (the tile's last active one): steps n < rl_last run in the lanes up to and
				// including it, the others in the lanes before it (kw_pair)
				const int last_l = (len - 1) / K, rl_last = len - last_l * K;
				const uint64_t upto = __ballot (lane <= last_l), before = __ballot (lane < last_l);
				v2f sj = 0;
				[&]<int... P> (std::integer_sequence<int, P...>) {
					(kw_pair<2 * P> (x[2 * P], x[2 * P + 1], z1, z2, z3, z4, sj, a0, a1, a2, b1, b2, c3, c4, eps2, upto, before, rl_last), ...);
				} (std::make_integer_sequence<int, K / 2> {});
				if (rl_last & 1) {
					// the partial lane stopped between the two steps of a pair: its shelving states sit swapped
					const v2f w1 = mtrw::pick (z1, last_l), w2 = mtrw::pick (z2, last_l);
					if (lane == last_l) { z1 = w2; z2 = w1; }
				}
				const float pw = mtrw::sum63 (a.gain_l * sj.x + a.gain_r * sj.y);
				if (lane == 0) a.tile_power[(size_t) s * a.n_tiles + jt0 + jj] = pw;
				k1 = mtrw::pick (z1, last_l); k2 = mtrw::pick (z2, last_l); k3 = mtrw::pick (z3, last_l); k4 = mtrw::pick (z4, last_l);
			}
			k1 = scrub (k1); k2 = scrub (k2); k3 = scrub (k3); k4 = scrub (k4);
		}

		PROF_NOW (c5_); PROF_ADD (4, c5_ - c4_);
		// the run registers are dead from here on: the next tile's loads go out now and land under the products
		v2f xn[K], g0 = v2f{0.f, 0.f}, g1 = v2f{0.f, 0.f};
		if (jj + 1 < ntile) fetch (jj + 1, xn, g0, g1);
		// The interpolator: 256 output frames x 3 phases per block and channel.  Operands ping-pong between the channels:
		// the right channel's fragments land under the left channel's 18 products and the next block's left fragments
		// under the right channel's; each channel's |max| rides between the other channel's MFMAs (two VALU
		// instructions per 16x16x32 MFMA are free, tools/coissue.hip).
		if (need_l || need_r) {
			__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");            // the words of every lane are in LDS
			float pl = 0.f, pr = 0.f;
			const int nfull = len >> 8;                                      // blocks whose 256 outputs all belong to the tile
			const int fo = 2 * col8 + kg4;                                   // + r: output frame of register r inside its block
			m16::BFrag bl, br;
			m16::f4 yl[3], yr[3];
			// (the scale of a tile below 2^-97 is clamped and its scaled maximum can sit in f16's subnormals, where the
			// bound below does not hold: such a tile takes the dense loop)
			if (a.prune >= 2 && (!need_l || tml * sc_l >= 16384.f) && (!need_r || tmr * sc_r >= 16384.f)) {
				// Refinement (tune_prune = 2, exact like the tile-level pruning): the first product, Ghi Xhi, for every block.
				// The other two move an output by at most eps = 2^-10 L1 max |x|: |Xlo| <= 8 = 2^-11 of a scaled maximum
				// that is >= 2^14, |Glo| <= 2^-11 |G|, one L1 max |x| each.  A block none of whose first-product outputs
				// comes within eps of the peak this (stream, segment, channel) had reached before the tile cannot hold
				// the final maximum and is dropped; a block that might gets the remaining twelve MFMAs on the same
				// accumulators — bit for bit the dense values.  (5 % on eps covers the f32 accumulation order.)
				const float k15 = (float) (1 << MTR_M16_TAP_SHIFT);
				const float thr_l = run_l * sc_l * k15 - (2.5684f * 1.05f / 1024.f) * k15 * (tml * sc_l);
				const float thr_r = run_r * sc_r * k15 - (2.5684f * 1.05f / 1024.f) * k15 * (tmr * sc_r);
				const int nb = (len + 255) >> 8;
				for (int b = 0; b < nb; ++b) {
					const int w = min (128 * b + col8, 8 * CMAX) + kg4;
					const int lim = len - 256 * b - fo;                          // registers r < lim are outputs of this tile
					bl.h0 = *reinterpret_cast<const uint4*> (HL + w); bl.h1 = *reinterpret_cast<const uint4*> (HL + w + 16);
					br.h0 = *reinterpret_cast<const uint4*> (HR + w); br.h1 = *reinterpret_cast<const uint4*> (HR + w + 16);
					m16::block_first (A, bl.h0, bl.h1, yl);
					m16::block_first (A, br.h0, br.h1, yr);
					bool hit_l = false, hit_r = false;
#pragma unroll
					for (int p = 0; p < 3; ++p)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							hit_l |= r < lim && fabsf (yl[p][r]) >= thr_l;
							hit_r |= r < lim && fabsf (yr[p][r]) >= thr_r;
						}
					const bool go_l = need_l && __ballot (hit_l) != 0, go_r = need_r && __ballot (hit_r) != 0;
					n_blk += 2; n_fin += (go_l ? 1 : 0) + (go_r ? 1 : 0);
					if (go_l) {
						bl.l0 = *reinterpret_cast<const uint4*> (LL + w); bl.l1 = *reinterpret_cast<const uint4*> (LL + w + 16);
						m16::block_rest (A, bl.h0, bl.h1, bl.l0, bl.l1, yl);
#pragma unroll
						for (int p = 0; p < 3; ++p)
#pragma unroll
							for (int r = 0; r < 4; ++r) pl = fmaxf (pl, r < lim ? fabsf (yl[p][r]) : 0.f);
					}
					if (go_r) {
						br.l0 = *reinterpret_cast<const uint4*> (LR + w); br.l1 = *reinterpret_cast<const uint4*> (LR + w + 16);
						m16::block_rest (A, br.h0, br.h1, br.l0, br.l1, yr);
#pragma unroll
						for (int p = 0; p < 3; ++p)
#pragma unroll
							for (int r = 0; r < 4; ++r) pr = fmaxf (pr, r < lim ? fabsf (yr[p][r]) : 0.f);
					}
				}
			} else {
#pragma unroll
			for (int p = 0; p < 3; ++p) yr[p] = m16::f4{0.f, 0.f, 0.f, 0.f};
			// word index of this lane's window in block b: 128 b + col8 + kg4.  Full blocks lie inside the arrays; only the
			// partial block's trailing columns can start past the last whole window and are clamped (their outputs are masked).
			int w = col8 + kg4;
			const int wlast = min (128 * nfull + col8, 8 * CMAX) + kg4;
			m16::fetch_b (bl, HL, LL, nfull ? w : wlast);
			for (int b = 0; b < nfull; ++b, w += 128) {
				m16::fetch_b (br, HR, LR, w);
				__builtin_amdgcn_sched_barrier (0);
				m16::block (A, bl, yl);
#pragma unroll
				for (int p = 0; p < 3; ++p) { pr = max3abs (pr, yr[p][0], yr[p][1]); pr = max3abs (pr, yr[p][2], yr[p][3]); }   // block b - 1
#pragma unroll
				for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier (0x8, 3, 0); __builtin_amdgcn_sched_group_barrier (0x2, 1, 0); }
				__builtin_amdgcn_sched_barrier (0);
				m16::fetch_b (bl, HL, LL, b + 1 < nfull ? w + 128 : wlast);   // past the last block: a harmless clamped read
				__builtin_amdgcn_sched_barrier (0);
				m16::block (A, br, yr);
#pragma unroll
				for (int p = 0; p < 3; ++p) { pl = max3abs (pl, yl[p][0], yl[p][1]); pl = max3abs (pl, yl[p][2], yl[p][3]); }
#pragma unroll
				for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier (0x8, 3, 0); __builtin_amdgcn_sched_group_barrier (0x2, 1, 0); }
				__builtin_amdgcn_sched_barrier (0);
			}
#pragma unroll
			for (int p = 0; p < 3; ++p) { pr = max3abs (pr, yr[p][0], yr[p][1]); pr = max3abs (pr, yr[p][2], yr[p][3]); }
			if ((len & 255) && (len & 255) <= 128) {
				// the tile's last block holds at most eight columns per channel (48 kHz: 2400 = 9 x 256 + 96): both channels
				// share ONE block — columns 0..7 read the left arrays, columns 8..15 the right ones, at the same positions
				const bool rside = (lane & 8) != 0;
				const int cc8 = 8 * (lane & 7);
				const int wm = min (128 * nfull + cc8, 8 * CMAX) + kg4;
				m16::fetch_b (br, rside ? HR : HL, rside ? LR : LL, wm);
				m16::block (A, br, yr);
				const int lim = len - 256 * nfull - (2 * cc8 + kg4);         // registers r < lim are outputs of this tile
				float pm = 0.f;
#pragma unroll
				for (int p = 0; p < 3; ++p)
#pragma unroll
					for (int r = 0; r < 4; ++r) pm = fmaxf (pm, r < lim ? fabsf (yr[p][r]) : 0.f);
				pl = fmaxf (pl, rside ? 0.f : pm);
				pr = fmaxf (pr, rside ? pm : 0.f);
			} else if (len & 255) {
				// the tile's last block: frames past its end belong to the next tile (or do not exist yet)
				m16::fetch_b (br, HR, LR, wlast);
				m16::block (A, bl, yl);
				m16::block (A, br, yr);
				const int lim = len - 256 * nfull - fo;                      // registers r < lim are outputs of this tile
				if ((len & 3) == 0) {
					// a lane's four registers are all in or all out: one branch on the lane mask, no per-register tests
					if (lim > 0) {
#pragma unroll
						for (int p = 0; p < 3; ++p) {
							pl = max3abs (pl, yl[p][0], yl[p][1]); pl = max3abs (pl, yl[p][2], yl[p][3]);
							pr = max3abs (pr, yr[p][0], yr[p][1]); pr = max3abs (pr, yr[p][2], yr[p][3]);
						}
					}
				} else {
#pragma unroll
					for (int p = 0; p < 3; ++p)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							pl = fmaxf (pl, r < lim ? fabsf (yl[p][r]) : 0.f);
							pr = fmaxf (pr, r < lim ? fabsf (yr[p][r]) : 0.f);
						}
				}
			}
			}
			pk_l = fmaxf (pk_l, pl * un_l);                                  // back to the samples' own scale (exact)
			pk_r = fmaxf (pk_r, pr * un_r);
		}
		PROF_NOW (c6_); PROF_ADD (5, c6_ - c5_);

#pragma unroll
		for (int n = 0; n < K; ++n) x[n] = xn[n];
		h0 = g0; h1 = g1;
		PROF_NOW (c7_); PROF_ADD (6, c7_ - c6_); PROF_ADD (7, c7_ - c0_); PROF_ADD (8, 1);
	}
#ifdef MTR_F4_PROF
	if (blockIdx.x == gridDim.x / 2 && lane == 0) for (int i = 0; i < 12; ++i) g_f4_prof[i] = pr_[i];
#endif
	if (EBU && q == a.n_segs - 1 && lane == 0) {
		st->kz[0] = k1.x; st->kz[1] = k1.y; st->kz[2] = k2.x; st->kz[3] = k2.y;
		st->kz[4] = k3.x; st->kz[5] = k3.y; st->kz[6] = k4.x; st->kz[7] = k4.y;
	}
	if (a.prune && lane == 0 && a.prune_stats) {
		atomicAdd (&a.prune_stats[0], n_done);
		atomicAdd (&a.prune_stats[1], n_skip);
		atomicAdd (&a.prune_stats[2], n_blk);
		atomicAdd (&a.prune_stats[3], n_fin);
	}
	pk_l = mtrw::max63 (pk_l);
	pk_r = mtrw::max63 (pk_r);
	if (lane == 0) {
		atomicMax (&st->tp_call[0], __float_as_uint (pk_l));
		atomicMax (&st->tp_call[1], __float_as_uint (pk_r));
	}
}

template <int K>
int launch_kwtp16 (bool ebu, const mtr_fused_args& a, uint32_t n_units, hipStream_t st)
{
	const size_t lds = ((size_t) 4 * ((HALO + 64 * K) / 2) * sizeof (uint32_t) + 15) & ~(size_t) 15;
	if (ebu) hipLaunchKernelGGL ((k_kwtp16<K, true>), dim3 (n_units), dim3 (64), lds, st, a);
	else     hipLaunchKernelGGL ((k_kwtp16<K, false>), dim3 (n_units), dim3 (64), lds, st, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

}  // namespace

#ifdef MTR_F4_PROF
extern "C" int mtr_debug_f4_prof (unsigned long long* out)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_f4_prof), 12 * sizeof (unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// New 47-frame history = the last 47 frames of (old history ++ this call's audio): what Resampler::process keeps in its
// window between calls (zita-resampler/resampler.cc:229-262).
// With a deferred gate (mtr_engine.hip) it also folds the call's true peak, in the gate's place: one lane per stream.
__global__ void k_history (const float* audio, uint64_t stride, uint64_t n_frames, const float* hist_in,
                           float* hist_out, uint32_t n_streams, mtr_stream_state* fold_state)
{
	const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n_streams * MTR_FIR_HALO) return;
	const uint32_t s = g / MTR_FIR_HALO, i = g % MTR_FIR_HALO;
	const int64_t f = (int64_t) n_frames - MTR_FIR_HALO + i;   // frame index in this call, may be < 0
	const v2f* src = reinterpret_cast<const v2f*> (audio) + (size_t) s * stride;
	const v2f* hin = reinterpret_cast<const v2f*> (hist_in) + (size_t) s * MTR_FIR_HALO;
	v2f* hout = reinterpret_cast<v2f*> (hist_out) + (size_t) s * MTR_FIR_HALO;
	hout[i] = (f >= 0) ? src[f] : hin[MTR_FIR_HALO + f];
	if (fold_state && i == 0) mtr_fold_truepeak (fold_state + s);
}

int mtr_launch_history (const float* audio, uint64_t stride, uint64_t n_frames, const float* hist_in,
                        float* hist_out, uint32_t n_streams, mtr_stream_state* fold_state, void* stream)
{
	const uint32_t n = n_streams * MTR_FIR_HALO;
	hipLaunchKernelGGL (k_history, dim3 ((n + 255) / 256), dim3 (256), 0, (hipStream_t) stream,
	                    audio, stride, n_frames, hist_in, hist_out, n_streams, fold_state);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}

int mtr_launch_kwtp16 (int run, bool ebu, const mtr_fused_args& a, uint32_t n_units, void* stream)
{
	switch (run) {
	case 38: return launch_kwtp16<38> (ebu, a, n_units, (hipStream_t) stream);
	default: return -2;
	}
}
