// mtr_bitstats.hip — IEEE-754 bit-usage statistics within the HBM budget (gfx950).
//
// Replaces float_stats (src/bitmeter.c:63-105, table layout src/uris.h:53-60) for a batch of mono
// streams [S][T] f32, bit for bit.  The reference walks the 23 mantissa bits of every sample and bumps
// hits[exp+k] (always), ones[exp+k] and mant[k] (bit set), plus hits/ones[exp+23] for the implicit
// one of a normal.  Everything is a projection of two counts: E[e] = samples with (effective)
// exponent e, and ones[p] = samples whose 24-bit significand, shifted left by e, has bit p set.
//
// The first version (23 ballots per 64 samples + one pass per distinct exponent, ~200 wave
// instructions per 64 samples) was VALU-bound at 9 % of the HBM peak.  This one counts POSITIONALLY
// with bit-sliced ("vertical") counters, lane-locally:
//
//   * exponents are grouped in eight classes of 32 (class = e >> 5; class 3 = [2^-31, 2) holds all
//     of normalised audio); within the wave's current "hot" class a sample becomes four words:
//        W  = significand << (e & 31)        (56 bits -> two words)   -> ones[32 c + bit]
//        1 << (e & 31)                                                -> E[32 c + bit]
//        mantissa                                                     -> mant[bit]
//     and the sign bit rides in the spare top bit of W's high word;
//   * each lane adds its words into four bit-sliced counters (12 planes = 4095 per bit position)
//     through a Harley-Seal carry-save tree: one full adder (two v_bitop3_b32: 3-input xor and
//     majority) per word, amortised — about 2.5 VALU ops per word, ~20 per sample all told, no LDS
//     and no cross-lane traffic in the loop;
//   * four samples (one 16-byte load) are vetted together with two v_max3: if every lane's four are
//     normals of the hot class (the case for audio) the words are built without any masking;
//   * the planes are unloaded once per <= 255 blocks (ballot + popcount per plane and bit);
//   * samples outside the hot class (other magnitudes, denormals) take the first version's
//     transposition path under a wave-uniform branch, so every input is still counted exactly;
//     a wave that keeps hitting that branch re-picks its hot class.
//
// One workgroup per stream, four waves, a wave takes blocks of 1024 samples (16 per lane, four
// 16-byte loads), the next block's loads in flight while this one is counted.
#include <hip/hip_runtime.h>

#include "mtr_internal.h"

#define BIM_DHIT 0
#define BIM_NHIT 23
#define BIM_DONE 280
#define BIM_NONE 303
#define BIM_DSET 560

namespace {

// full adder on 32 independent bit columns: acc <- acc ^ a ^ b, returns the carries (majority)
__device__ __forceinline__ uint32_t csa (uint32_t& acc, uint32_t a, uint32_t b)
{
	// v_bitop3_b32 (new on gfx950): any 3-input boolean function by truth table; 0xE8 = majority, 0x96 = parity
	const uint32_t c = __builtin_amdgcn_bitop3_b32 (acc, a, b, 0xE8);
	acc = __builtin_amdgcn_bitop3_b32 (acc, a, b, 0x96);
	return c;
}

constexpr int NPLANE = 12;                         // counts up to 4095 per lane and bit position
constexpr int NUP = 3;                             // carry-save levels above a block (weights 16, 32, 64)
constexpr int BLK_PER_FLUSH = 248;                 // multiple of 2^NUP; 248 x 16 = 3968 <= 4095
constexpr int RING = 3;                            // LDS blocks per wave: one being counted, two in flight

// 32 counters, one per bit position of the words added, stored as bit planes
struct VCount {
	uint32_t p[NPLANE];
	uint32_t b0, b1, b2, b3;                       // carries waiting for their partner inside a block of 16
	uint32_t u0, u1, u2;                           // ... and across blocks (weights 16, 32, 64), zero when empty
	__device__ __forceinline__ void clear ()
	{
#pragma unroll
		for (int j = 0; j < NPLANE; ++j) p[j] = 0;
		u0 = u1 = u2 = 0;
		b0 = b1 = b2 = b3 = 0;
	}
	// I = position of the word inside a block of 16 (compile-time): 8 + 4 + 2 + 1 full adders per block.
	// nblk = blocks already added since the last clear (wave-uniform): the carry of weight 16 meets its
	// partner every second block, the one of weight 32 every fourth, ... ; above NUP levels it ripples.
	template <int I>
	__device__ __forceinline__ void add (uint32_t w, int nblk)
	{
		if (!(I & 1)) { b0 = w; return; }
		uint32_t c = csa (p[0], b0, w);
		if (!(I & 2)) { b1 = c; return; }
		c = csa (p[1], b1, c);
		if (!(I & 4)) { b2 = c; return; }
		c = csa (p[2], b2, c);
		if (!(I & 8)) { b3 = c; return; }
		c = csa (p[3], b3, c);
		static_assert (NUP == 3, "three explicit levels");
		if (!(nblk & 1)) { u0 = c; return; }
		c = csa (p[4], u0, c); u0 = 0;
		if (!(nblk & 2)) { u1 = c; return; }
		c = csa (p[5], u1, c); u1 = 0;
		if (!(nblk & 4)) { u2 = c; return; }
		c = csa (p[6], u2, c); u2 = 0;
#pragma unroll
		for (int j = 4 + NUP; j < NPLANE; ++j) { const uint32_t t = p[j] & c; p[j] ^= c; c = t; }
	}
	// count of bit position b summed over the wave (wave-uniform); only valid at block boundaries
	__device__ __forceinline__ int total (int b) const
	{
		int n = 0;
#pragma unroll
		for (int j = 0; j < NPLANE; ++j) n += __popcll (__ballot ((p[j] >> b) & 1u)) << j;
		n += __popcll (__ballot ((u0 >> b) & 1u)) << 4;
		n += __popcll (__ballot ((u1 >> b) & 1u)) << 5;
		n += __popcll (__ballot ((u2 >> b) & 1u)) << 6;
		return n;
	}
};

__global__ __launch_bounds__ (256) void k_bitstats (const float* audio, uint64_t stride, uint64_t n_frames,
                                                    mtr_bitstats_state* out, uint32_t n_streams)
{
	__shared__ uint4 ring[4][RING][256];   // per wave: RING blocks of 1024 samples, filled by LDS-DMA
	__shared__ int32_t Oh[288];            // ones by position p = e + k (implicit one of a normal at k = 23)
	__shared__ int32_t Eh[256];            // samples by (effective) exponent
	__shared__ int32_t Mh[24];             // mantissa bit k set
	__shared__ int32_t cnt[6];             // zero nan inf den sign-bits-of-positional-quads positive-candidates
	// min / max over the normals of t = |x| bits - 0x00800000 (normal <=> t < 0x7f000000); the max is
	// meaningful once the min says a normal was seen
	__shared__ uint32_t tmin, tmax;
	const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
	const uint32_t s = blockIdx.x;
	const uint32_t* src = reinterpret_cast<const uint32_t*> (audio) + (size_t) s * stride;
	for (int i = tid; i < 288; i += 256) Oh[i] = 0;
	Eh[tid] = 0;
	if (tid < 24) Mh[tid] = 0;
	if (tid < 6) cnt[tid] = 0;
	if (tid == 0) { tmin = 0xffffffffu; tmax = 0u; }
	__syncthreads ();

	// The general path keeps nothing in registers (its counters go straight to LDS): registers decide
	// the occupancy of this kernel and the positional path needs them all.
	int n_hotquad = 0;                     // quads counted positionally (wave-uniform): 256 live samples each
	float vmin_f = INFINITY, vmax_f = 0.f; // |x| range over the positionally counted quads (all normals)

	// ---- per-exponent path (the first version's transposition), for the lanes in `sel` only: 23 ballots
	// turn the mantissa bits so that lane k holds the mask "which selected samples have bit k set"
	// (lane 23: all selected, lane 24: the selected normals, whose implicit one counts at k = 23); then
	// one pass per distinct exponent e among them adds popcounts straight into the positional tables.
	auto slow_count = [&] (uint32_t bits, bool sel) {
		uint32_t ex = (bits >> 23) & 0xffu;
		const uint32_t man = bits & 0x7fffffu;
		const bool seln = sel && ex != 0;
		if (ex == 0) ex = 1;                   // denormals sit at 2^-126 (bitmeter.c:94)
		// (v_writelane after a VALU-written VCC needs wait states hipcc does not insert around asm)
		int mine_lo = 0, mine_hi = 0;
#pragma unroll
		for (int k = 0; k < 23; ++k) {
			const unsigned long long m = __ballot (sel && ((man >> k) & 1u));
			asm volatile ("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(mine_lo) : "s"((int) (uint32_t) m), "n"(k));
			asm volatile ("v_writelane_b32 %0, %1, %2" : "+v"(mine_hi) : "s"((int) (uint32_t) (m >> 32)), "n"(k));
		}
		const unsigned long long selmask = __ballot (sel), nrmmask = __ballot (seln);
		asm volatile ("s_nop 4\n\tv_writelane_b32 %0, %1, 23" : "+v"(mine_lo) : "s"((int) (uint32_t) selmask));
		asm volatile ("v_writelane_b32 %0, %1, 23" : "+v"(mine_hi) : "s"((int) (uint32_t) (selmask >> 32)));
		asm volatile ("s_nop 4\n\tv_writelane_b32 %0, %1, 24" : "+v"(mine_lo) : "s"((int) (uint32_t) nrmmask));
		asm volatile ("v_writelane_b32 %0, %1, 24" : "+v"(mine_hi) : "s"((int) (uint32_t) (nrmmask >> 32)));
		const unsigned long long mine = ((unsigned long long) (uint32_t) mine_hi << 32) | (uint32_t) mine_lo;
		if (lane < 23) {
			const int c = __popcll (mine);
			if (c) atomicAdd (&Mh[lane], c);
		}
		unsigned long long todo = selmask;
		while (todo) {                           // one pass per distinct exponent among the selected samples
			const int first = __ffsll ((long long) todo) - 1;
			const uint32_t e = (uint32_t) __builtin_amdgcn_readlane ((int) ex, first);
			const unsigned long long same = __ballot (sel && ex == e);
			if (lane < 25) {
				const int c = __popcll (mine & same);
				if (c) atomicAdd (lane == 23 ? &Eh[e] : &Oh[e + (lane < 23 ? lane : 23)], c);
			}
			todo &= ~same;
		}
	};

	VCount wl, wh, we, wm;
	wl.clear (); wh.clear (); we.clear (); wm.clear ();
	uint32_t cstar = 3;                    // the hot class, 1..6 (classes 0 and 7 hold denormals / inf / nan)
	int odd_ct = 0;                        // other-class branches taken in the current block (wave-uniform)
	int since = 0;                         // blocks added since the last unload (wave-uniform)

	auto pick_class = [&] (uint32_t bits) {
		const uint32_t cls = (bits & 0x7fffffffu) >> 28;
		const unsigned long long ok = __ballot (cls >= 1 && cls <= 6);
		cstar = ok ? (uint32_t) __builtin_amdgcn_readlane ((int) cls, __ffsll ((long long) ok) - 1) : 3u;
	};

	auto flush = [&] () {
		const int pos0 = 32 * (int) cstar;
		for (int b = 0; b < 32; ++b) {
			const int nl = wl.total (b), ne = we.total (b);
			if (lane == 0) {
				if (nl) atomicAdd (&Oh[pos0 + b], nl);
				if (ne) atomicAdd (&Eh[pos0 + b], ne);
			}
		}
		for (int b = 0; b < 24; ++b) {
			const int nh = wh.total (b), nm = wm.total (b);
			if (lane == 0) {
				if (nh) atomicAdd (&Oh[pos0 + 32 + b], nh);
				if (nm) atomicAdd (&Mh[b], nm);
			}
		}
		const int ns = wh.total (31);
		if (lane == 0 && ns) atomicAdd (&cnt[4], ns);
		wl.clear (); wh.clear (); we.clear (); wm.clear ();
	};

	// Four samples per lane of which some, in some lane, are not normals of the hot class: the whole
	// quad goes the reference's way, sample by sample (float_stats' own tests and counters) with the
	// per-exponent path doing the tables.  Rolled: this is the cold path and must stay small.
	auto general_quad = [&] (const uint4& v) {
		int n_other = 0;
#pragma unroll 1
		for (int j = 0; j < 4; ++j) {
			const uint32_t bits = j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
			const uint32_t av = bits & 0x7fffffffu;
			const uint32_t t = av - 0x00800000u;
			const bool normal = t < 0x7f000000u;
			const bool live = av != 0u && av < 0x7f800000u;
			const int c_zero = __popcll (__ballot (av == 0u)), c_nan = __popcll (__ballot (av > 0x7f800000u));
			const int c_inf = __popcll (__ballot (av == 0x7f800000u)), c_den = __popcll (__ballot (live && !normal));
			const int c_pos = __popcll (__ballot (live && !(bits >> 31)));
			if (lane == 0) {
				if (c_zero) atomicAdd (&cnt[0], c_zero);
				if (c_nan) atomicAdd (&cnt[1], c_nan);
				if (c_inf) atomicAdd (&cnt[2], c_inf);
				if (c_den) atomicAdd (&cnt[3], c_den);
				if (c_pos) atomicAdd (&cnt[5], c_pos);
			}
			if (normal) { atomicMin (&tmin, t); atomicMax (&tmax, t); }
			if (__ballot (live)) slow_count (bits, live);
			n_other += __ballot (normal && (av >> 28) != cstar) != 0;
		}
		odd_ct += n_other;
	};
	// a position of the block that went the general way adds nothing to the vertical counters
#define MTR_BIT_NONE(I)                                                                       \
	{ wl.add<I> (0u, since); wh.add<I> (0u, since); we.add<I> (0u, since); wm.add<I> (0u, since); }
	// one sample known to be a normal of the hot class in every lane: no masks, no tests
#define MTR_BIT_HOT(I, bits_)                                                                 \
	{                                                                                         \
		const uint32_t bits = (bits_);                                                        \
		const uint32_t m24 = (bits & 0x7fffffu) | 0x800000u;                                  \
		const uint32_t sh = (bits >> 23) & 31u;                                               \
		const unsigned long long W = (unsigned long long) m24 << sh;                          \
		wl.add<I> ((uint32_t) W, since);                                                      \
		wh.add<I> ((uint32_t) (W >> 32) | (bits & 0x80000000u), since);                       \
		we.add<I> (1u << sh, since);                                                          \
		wm.add<I> (bits & 0x7fffffu, since);                                                  \
	}
	// four samples of one 16-byte load: vetted together
#define MTR_BIT_QUAD(Q, v)                                                                    \
	{                                                                                         \
		/* opaque until here: otherwise LLVM hoists the vetting of all four quads to the top of the block */ \
		asm volatile ("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w));               \
		/* all four in the hot class <=> the class bits 30..28 of every (bits ^ class) are clear; classes  \
		   1..6 hold normals only, so no other test is needed */                              \
		const uint32_t cs = cstar << 28;                                                      \
		const uint32_t xo = (((v).x ^ cs) | ((v).y ^ cs)) | (((v).z ^ cs) | ((v).w ^ cs));    \
		if (__ballot ((xo & 0x70000000u) != 0u) == 0) {                                       \
			/* |x| rides on the source modifiers of v_min3_f32 / v_max3_f32 */                \
			const float f0 = __uint_as_float ((v).x), f1 = __uint_as_float ((v).y);           \
			const float f2 = __uint_as_float ((v).z), f3 = __uint_as_float ((v).w);           \
			vmin_f = fminf (fminf (vmin_f, fabsf (f0)), fabsf (f1));                          \
			vmin_f = fminf (fminf (vmin_f, fabsf (f2)), fabsf (f3));                          \
			vmax_f = fmaxf (fmaxf (vmax_f, fabsf (f0)), fabsf (f1));                          \
			vmax_f = fmaxf (fmaxf (vmax_f, fabsf (f2)), fabsf (f3));                          \
			MTR_BIT_HOT (4 * (Q) + 0, (v).x) MTR_BIT_HOT (4 * (Q) + 1, (v).y)                 \
			MTR_BIT_HOT (4 * (Q) + 2, (v).z) MTR_BIT_HOT (4 * (Q) + 3, (v).w)                 \
			++n_hotquad;                                                                      \
		} else {                                                                              \
			general_quad (v);                                                                 \
			MTR_BIT_NONE (4 * (Q) + 0) MTR_BIT_NONE (4 * (Q) + 1)                             \
			MTR_BIT_NONE (4 * (Q) + 2) MTR_BIT_NONE (4 * (Q) + 3)                             \
		}                                                                                     \
	}

	// ---- blocks of 1024 samples per wave: lane l takes samples 4 (64 q + l) .. + 3 of quarter q ------------
	const bool wide = ((((size_t) s * stride) & 3) == 0) && ((reinterpret_cast<size_t> (audio) & 15) == 0);
	const uint64_t n_blk = (n_frames + 1023) / 1024;
	const uint4* const src4 = reinterpret_cast<const uint4*> (src);
	// The wave's blocks travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction,
	// no VGPR holds data in flight), RING - 1 blocks ahead of the one being counted; the lane-linear
	// destination is exactly the [quarter][lane] order the lanes read back with one ds_read_b128 per quad.
	uint4 (* const myring)[256] = ring[wid];
	auto stage = [&] (uint64_t blk, int slot) {
		const uint64_t base = blk * 1024;
		if (wide && base + 1024 <= n_frames) {
#pragma unroll
			for (int q = 0; q < 4; ++q)
				__builtin_amdgcn_global_load_lds ((const __attribute__ ((address_space (1))) void*) (src4 + (blk * 4 + q) * 64 + lane),
				                                  (__attribute__ ((address_space (3))) void*) (&myring[slot][q * 64]), 16, 0, 0);
		} else {                               // the last block, or an unaligned stream: zero padded, guarded, synchronous
#pragma unroll 1
			for (int q = 0; q < 4; ++q) {
				const uint64_t i = base + (uint64_t) (q * 64 + lane) * 4;
				uint4 v;
				v.x = i + 0 < n_frames ? src[i + 0] : 0u;
				v.y = i + 1 < n_frames ? src[i + 1] : 0u;
				v.z = i + 2 < n_frames ? src[i + 2] : 0u;
				v.w = i + 3 < n_frames ? src[i + 3] : 0u;
				myring[slot][q * 64 + lane] = v;
			}
			// everything issued so far, DMA included, has landed once this returns
			asm volatile ("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		}
	};

	const int64_t nk = n_blk > (uint64_t) wid ? (int64_t) ((n_blk - wid + 3) / 4) : 0;   // this wave's blocks: wid + 4 k
	if (nk > 0) {
		stage (wid, 0);
		if (nk > 1) stage (wid + 4, 1);
		for (int64_t k = 0; k < nk; ++k) {
			const int slot = (int) (k % RING);
			// slot (k + 2) % RING held block k - 1: its reads have all returned (their data has been counted)
			asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
			if (k + 2 < nk) stage (wid + 4 * (uint64_t) (k + 2), (int) ((k + 2) % RING));
			// block k has landed when at most the DMA instructions of the newer blocks are outstanding
			if (k + 2 < nk)      asm volatile ("s_waitcnt vmcnt(8)" ::: "memory");
			else if (k + 1 < nk) asm volatile ("s_waitcnt vmcnt(4)" ::: "memory");
			else                 asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
			const uint4* const blkp = myring[slot] + lane;
			uint4 qa = blkp[0], qb = blkp[64];
			if (k == 0) pick_class (qa.x);
			odd_ct = 0;
			MTR_BIT_QUAD (0, qa)
			qa = blkp[128];
			MTR_BIT_QUAD (1, qb)
			qb = blkp[192];
			MTR_BIT_QUAD (2, qa)
			MTR_BIT_QUAD (3, qb)
			++since;
			const bool last = k + 1 == nk;
			const bool repick = odd_ct >= 8;       // most of this block lives in another class
			if (since == BLK_PER_FLUSH || last || repick) { flush (); since = 0; }
			if (repick && !last) {
				// the next block is RING - 1 ahead in flight: look at it once it is needed — re-pick from this one's tail
				pick_class (qb.w);
			}
		}
	}
#undef MTR_BIT_QUAD
#undef MTR_BIT_HOT
#undef MTR_BIT_NONE

	// ---- workgroup totals -----------------------------------------------------------------------------
	for (int d = 32; d >= 1; d >>= 1) {
		vmin_f = fminf (vmin_f, __shfl_xor (vmin_f, d, 64));
		vmax_f = fmaxf (vmax_f, __shfl_xor (vmax_f, d, 64));
	}
	if (lane == 0) {
		if (vmax_f > 0.f) {                // this wave counted positional quads: fold their |x| range in
			atomicMin (&tmin, __float_as_uint (vmin_f) - 0x00800000u);
			atomicMax (&tmax, __float_as_uint (vmax_f) - 0x00800000u);
		}
		// every sample of a positionally counted quad is live; their negatives are counted by the sign plane
		atomicAdd (&cnt[5], 256 * n_hotquad);
	}
	__syncthreads ();

	mtr_bitstats_state* o = out + s;
	const int dens = cnt[3];
	// project onto the reference's table; positions p = e + k, p in [1, 277]
	for (int p = tid; p < 280; p += 256) {
		int hits = 0;
		for (int k = 0; k < 23; ++k) {
			const int e = p - k;
			if (e >= 1 && e <= 254) hits += Eh[e];
		}
		const int e = p - 23;                       // the implicit one: normals only
		if (e >= 1 && e <= 254) hits += Eh[e] - (e == 1 ? dens : 0);
		o->hist[BIM_DHIT + p] += hits;
		o->hist[BIM_DONE + p] += Oh[p];
	}
	if (tid < 23) o->hist[BIM_DSET + tid] += Mh[tid];
	if (tid == 0) {
		const int pads = (int) (n_blk * 1024 - n_frames);            // zero padding of the last block
		o->n_zero += cnt[0] - pads; o->n_nan += cnt[1]; o->n_inf += cnt[2]; o->n_den += dens;
		o->n_pos += cnt[5] - cnt[4];                                 // minus the set sign bits of the positional quads
		if (tmin < 0x7f000000u) {                                    // a normal was seen
			o->vmin = fminf (o->vmin, __uint_as_float (tmin + 0x00800000u));
			o->vmax = fmaxf (o->vmax, __uint_as_float (tmax + 0x00800000u));
		}
	}
}

}  // namespace

int mtr_launch_bitstats (const float* audio, uint64_t stride, uint64_t n_frames, mtr_bitstats_state* out,
                         uint32_t n_streams, void* stream)
{
	hipLaunchKernelGGL (k_bitstats, dim3 (n_streams), dim3 (256), 0, (hipStream_t) stream, audio, stride, n_frames, out, n_streams);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
