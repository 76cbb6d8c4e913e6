// mtr_tpb.hip — TruePeakdsp::process: 4x interpolation followed by the PPM-style ballistics (gfx950).
//
// Replaces jmeters/truepeakdsp.cc:41-99 (the dBTP plugins' meter, src/meters.cc:465-507, and dr14.c's
// true-peak bar) with the call-level semantics of "process(); read(m, p)": per call and channel
//     z1 *= w3, z2 *= w3 once per input frame; for each of the 4 oversampled values v = |y|:
//     if (v > z1) z1 += w1 (v - z1);  if (v > z2) z2 += w2 (v - z2);  p = max (p, v);
//     m = max (m, z1 + z2) once per input frame;  result m * g, p;  states clamped to [0, 20] on
//     entry and offset by 1e-20f on exit.
//
// The attack / release recurrence is a monotone piece-wise linear map of the state; compositions grow a piece per
// value, so it is not a cheap associative scan and time stays serial per (stream, channel, filter).  What this kernel does
// (round 5's form; round 4's — one wave walking all 64 chains as packed pairs, one wave fetching and splitting, operands
// read in the iteration that used them — is in git and in profiles/r04_tpb.md):
//
//   * `if (v > z) z += w (v - z)` is z <- max (z, a z + w v) with a = 1 - w: a monotone max-affine map, and maps of
//     that kind compose.  Two attacks in a row are z <- max (z, a z + c, a^2 z + c'), c = w max (v1, v2), c' = (w a) v1 + w v2:
//     the intercepts depend on the values only — any lane can form them, for any frame — so what is left ON the chain per
//     frame and filter is the release (folded into the first pair's slopes) — a multiply, two PACKED fused multiply-adds (a
//     pair map's two intercepts sit side by side) and two v_max3 — plus a DPP add and half a v_max3 for m = max (z1 + z2).  Exact in real arithmetic; in f32 a few ulps from the
//     reference's sequence (tests/test_gpu_parity.py::test_truepeak_ballistics_*: 4e-6 of the value; tools/fuzz_tpb.py).
//   * the interpolator is the matrix-pipe one of mtr_mfma16_fir.h (samples and taps as two f16 halves, three partial
//     products, f32 accumulation: within 4e-7 of the exact-f32 chain): a workgroup owns 64 (stream, channel)
//     columns = 4 blocks of 16, a chunk is 16 frames = the rows of one block.
//   * EVERY SAMPLE IS SPLIT ONCE into a ring of f16 hi / lo pair words (six slots of 16 samples per column), under a
//     power-of-two scale per column that follows the maximum of five slots with hysteresis: it stands while that maximum
//     stays in [2^7, 2^15) scaled, and when it has to move — a sample too large for it, or a stretch that has become 2^5
//     quieter than the scale was made for — the new chunk is written under the new scale and the three chunks in front of it
//     are split AGAIN from the exact f32 ring, in a cold path with its own barrier (`rescale`).  The products READ their
//     operands (four ds_read_b128 per unit, from places chosen so that no two lanes of a read group share a bank: f16_rot);
//     phase 0 (x[n - 24]) comes from the f32 ring, exactly.
//   * a chunk travels HBM -> LDS by LDS-DMA issued as inline assembly (the compiler's own vmcnt bookkeeping would wait for
//     whatever is in flight at every barrier), five chunks ahead of its split, which is two chunks ahead of its products.
//   * TWELVE WAVES, three per SIMD, each with its own copy of the loop, one barrier per chunk:
//       waves 0, 1   the chains of columns 0 .. 31 | 32 .. 63: lane = (column, filter), plain f32 (round 4: one wave, both
//                    filters of 64 columns as packed pairs — eleven instructions per frame on ONE wave; here 6.5 on each of two);
//       waves 2, 3   the split of the same halves: lane = (half of a chunk, column); wave 2 also sends the LDS-DMA;
//       the rest     a block's products as TWO units — unit A: phase 1 (6 MFMAs) and the frame's first pair map (x[n - 24], y1);
//                    unit B: phases 2 and 3 (12 MFMAs) and the second — whose operands are read one iteration AHEAD, behind
//                    the MFMAs of the chunk before and under its maps.
//     What bounds it is what a SIMD can issue per chunk (VALU + matrix pipe busy ~85 % of the time): round 5 cut the
//     workgroup's VALU instructions per chunk from 963 to ~725 while adding a second chain wave, removed the LDS bank
//     conflicts (SQ_LDS_BANK_CONFLICT 1.1e9 -> 1.2e8 per launch) and the LDS latency on every role's critical path.
//     22.9 -> 17.9 - 19.3 ms per 8192 streams x 10 s across boxes (profiles/r05_tpb.md).
#include <hip/hip_runtime.h>

#include "mtr_internal.h"
#include "mtr_mfma16_fir.h"

typedef float v2f __attribute__ ((ext_vector_type (2)));

// tools/tpb_prof.hip builds this file with MTR_TPB_PROF: cycles per wave and section for workgroup 0
#ifdef MTR_TPB_PROF
__device__ unsigned long long g_tpb_prof[16][4];
#define PROF_NOW(v) unsigned long long v; asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define PROF_ADD(i, d) pr[i] += (d)
#else
#define PROF_NOW(v)
#define PROF_ADD(i, d)
#endif

// tools/tpb_prof.hip elimination builds — a role switched off: WRONG RESULTS, timing only.  They exist only in a library built
// with MTR_TIMING_ONLY_BUILD, whose mtr_version () says so and which meters.lv2_amd/engine.py refuses to load outside tools/.
#ifdef MTR_TIMING_ONLY_BUILD
#ifndef MTR_TPB_DBG_NOCHAIN
#define MTR_TPB_DBG_NOCHAIN 0
#endif
#ifndef MTR_TPB_DBG_NOPROD
#define MTR_TPB_DBG_NOPROD 0
#endif
#ifndef MTR_TPB_DBG_NOFETCH
#define MTR_TPB_DBG_NOFETCH 0
#endif
#ifndef MTR_TPB_DBG_NOSPLIT
#define MTR_TPB_DBG_NOSPLIT 0
#endif
#ifndef MTR_TPB_DBG_NODMA
#define MTR_TPB_DBG_NODMA 0
#endif
#else
#define MTR_TPB_DBG_NOCHAIN 0
#define MTR_TPB_DBG_NOPROD 0
#define MTR_TPB_DBG_NOFETCH 0
#define MTR_TPB_DBG_NOSPLIT 0
#define MTR_TPB_DBG_NODMA 0
#endif

namespace {

constexpr int NW = 12;                         // waves 0, 1: the chains of columns 0 .. 31 | 32 .. 63; 2, 3: fetch + split of the same halves;
                                               // unit A of blocks 0 .. 3 on waves 4, 5, 8, 6, unit B on 9, 10, 7, 11.  Waves of equal w mod 4
                                               // share a SIMD, and what a SIMD issues per chunk is what bounds the kernel: SIMD 0 carries a chain
                                               // (~105 VALU instructions per chunk) and two A units (~45 + 6 MFMAs each), SIMD 1 a chain, an A and
                                               // a B unit (~48 + 12 MFMAs), SIMD 2 a split wave (~60), an A and a B, SIMD 3 a split wave and two
                                               // B units.  (A chain, an A and a B on every SIMD, or both chains' SIMDs with two A units, are
                                               // 0.5 - 1 % slower; sixteen waves with the B units off the chains' SIMDs balance better and
                                               // lose it at the barrier: profiles/r05_tpb.md)
constexpr int F = 16;                          // frames per chunk = rows of one MFMA block
constexpr int NCOL = 64;                       // (stream, channel) columns per workgroup: 32 stereo or 64 mono streams
constexpr int NSLOT = 6;                       // ring slots of 16 samples per column: the 64-sample window of the chunk in the products and
                                               // the two chunks the split is ahead of it
constexpr int RING = NSLOT * F;
constexpr int RSTRIDE = RING + 4;              // floats per column of the f32 ring: 25 x 16 bytes — odd, so that the 16-byte stores of eight
                                               // neighbouring columns (the split's lanes) sit in eight bank groups
constexpr int RING_B = NCOL * RSTRIDE * 4;
constexpr int HSTRIDE = 256;                   // bytes per column and array of the f16 ring: sixteen 16-byte places for its twelve pieces (8 samples
                                               // as four pair words each), piece q at place (q + rot (column)) mod 16 — see f16_rot
constexpr int HRING_B = NCOL * HSTRIDE;        // one array: hi | lo
constexpr int AUX_B = 512;                     // float un [NCOL]; int flag [2]
constexpr int CROW = 1024;                     // a chunk of maps: [half][frame] rows of [group of 16 columns][filter][column] x (c, c') = 8 bytes
constexpr int CBUF_B = 2 * F * CROW;
constexpr int NSTG = 4;                        // staging buffers, as the LDS-DMA leaves a chunk: [chunk % 4][piece 64 i + lane] x 16 bytes — the chunk the
                                               // split reads and the three behind it that are landing or on their way from HBM
constexpr int AHEAD = 5;                       // iteration t sends chunk t + AHEAD and waits for chunk t + 3: two chunks' time of flight behind
                                               // the one that is landing (AHEAD = 7 with eight buffers, four chunks: measured 1 % SLOWER)
constexpr int STG_B = NSTG * 4 * 1024;
constexpr int LDS_BYTES = RING_B + 2 * HRING_B + AUX_B + 2 * CBUF_B + STG_B;
constexpr int NTHREADS = 64 * NW;
static_assert (RING % 8 == 0 && (RSTRIDE * 4) % 16 == 0, "operand slices never wrap inside the ring");
static_assert (LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// Where piece q of a column sits in its 256 bytes of the f16 ring.  ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} (MI355X_MICROARCH.md, LDS), one cycle each
// while the sixteen 16-byte places differ mod 16.  A products lane (cc, kg) reads piece q0 + kg of column cc: a group holds the
// columns {0-3, 12-15} of one kg and {4-11} of the next, i.e. two consecutive pieces — so every column's rotation must be EVEN
// (the two pieces then differ in parity) and distinct within each of the two column sets: rot = (c & 6) + 8 (c & 1).  (Round 4's
// 176-byte columns put 11 c + kg mod 16 there: three places of every group twice — SQ_LDS_BANK_CONFLICT 1.1e9 per launch, 17.8 % of
// the LDS-active cycles.)  The split's ds_write_b128 goes in groups of eight NEIGHBOURING lanes whose places must differ mod 8:
// rot mod 8 = c & 6 takes four values on eight columns, so odd columns write the OTHER piece of a chunk's two in the same
// instruction (a lane of the split owns half a chunk of its column: which half alternates with the column's parity).
__device__ __forceinline__ int f16_rot (int col) { return (col & 6) | ((col & 1) << 3); }
__device__ __forceinline__ int f16_place (int q, int rot) { return ((q + rot) & 15) << 4; }

__device__ __forceinline__ float max3f (float a, float b, float c) { return __builtin_fmaxf (__builtin_fmaxf (a, b), c); }
// max (a, b) / max (|a|, b) as ONE instruction (as a C expression every operand that is not provably quiet costs a canonicalising
// v_max (x, x) first, and an |x| that is used twice a v_and); a NaN loses, as in fmaxf

__device__ __forceinline__ float max3_plain (float a, float b, float c) { float r; asm ("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max_abs_plain (float a, float b) { float r; asm ("v_max_f32 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the maximum of a value over lanes l and l ^ 32, in both (v_permlane32_swap: the upper half of one operand against the lower of the other)
__device__ __forceinline__ float max_across_halves (float m)
{
	const auto r = __builtin_amdgcn_permlane32_swap (__float_as_uint (m), __float_as_uint (m), false, false);
	return __builtin_fmaxf (__uint_as_float (r[0]), __uint_as_float (r[1]));
}

// The scale of a column: a power of two, 2^(se - 127).  Made for a window maximum W it puts W into [2^12, 2^13); it stands
// until a sample reaches 2^15 under it (cap: the bit pattern of that sample — non-negative floats order as uints, an Inf
// lies above every cap that matters) or the window's maximum falls below 2^7 under it (low).
struct ColScale {
	int se;
	float sc, un;                              // scale; 2^-15 / scale (the taps carry 2^15)
	uint32_t cap, low;
	__device__ __forceinline__ static int se_for (float w) { return min (238, 266 - (int) (__float_as_uint (w) >> 23)); }
	__device__ __forceinline__ void set (int se_)
	{
		se = se_;
		sc = __uint_as_float ((uint32_t) se << 23);
		un = __uint_as_float ((uint32_t) (239 - se) << 23);
		cap = (uint32_t) (269 - se) << 23;
		low = (uint32_t) (261 - se) << 23;
	}
};

template <int C>   // channels: 2 = interleaved stereo (column = stream + 32 channel), 1 = mono (column = stream)
__global__ __launch_bounds__ (NTHREADS) void k_tpb (const mtr_tpb_args a)
{
	constexpr int NSTR = NCOL / C;                                       // streams per workgroup
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem[];
	float* const ring = reinterpret_cast<float*> (smem);                 // [NCOL][RSTRIDE] f32: exact samples, chunk j at position 16 ((j + 3) mod 6)
	unsigned char* const ringh = smem + RING_B;                          // [NCOL][HSTRIDE]: f16 hi pair words, the two pieces of chunk j = 2 ((j + 3) mod 6), + 1
	unsigned char* const ringl = ringh + HRING_B;                        // ... lo
	float* const un_sh = reinterpret_cast<float*> (ringl + HRING_B);     // [NCOL]: 2^-15 / scale of every column, as the ring holds it
	int* const flag_sh = reinterpret_cast<int*> (un_sh + NCOL);          // [2]: a rescale is pending for the iteration of this parity
	unsigned char* const cbuf = smem + RING_B + 2 * HRING_B + AUX_B;     // [2][half][frame][CROW]
	unsigned char* const stg = cbuf + 2 * CBUF_B;
	const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
	const uint32_t s0 = blockIdx.x * NSTR;
	// (32-bit: gfx950 has no scalar 64-bit signed compare, and every `t < n_chunks` of every wave and iteration was two or three
	// VALU instructions — ~80 of a workgroup's ~850 per chunk.  mtr_launch_tpb refuses calls of 2^31 - 4096 frames or more.)
	const int n_frames = (int) a.n_frames;
	const int n_chunks = (n_frames + F - 1) / F;

	// ---- the chains (waves 0, 1): lane = (column ci of the wave's 32, filter phi) ------------------------------------------
	// One filter of one column per lane, in plain f32 — round 4 walked both filters of a column as a packed pair in ONE wave of 64
	// columns: eleven instructions per frame of which five packed; here a wave issues nine unpacked ones, and there are two waves.
	const int hw_ = wid & 1;                                             // which half of the columns (chains and split alike)
	const int ci = lane >> 1, phi = lane & 1;
	const int ccol = 32 * hw_ + ci;
	const uint32_t csl = s0 + (uint32_t) (C == 2 ? ci : ccol);
	const int cch = C == 2 ? hw_ : 0;
	const bool cowner = csl < a.n_streams;
	mtr_stream_state* const cst = a.state + (cowner ? csl : 0);
	float z = 0.f, zm = 0.f;
	if (wid < 2 && cowner) {
		z = phi ? cst->tpb_z2[cch] : cst->tpb_z1[cch];
		z = z > 20 ? 20 : (z < 0 ? 0 : z);                                 // truepeakdsp.cc:54-55
	}
	// slopes of the frame's two pair maps: w3, a w3, a^2 w3 (the release folded into the first one) | a, a^2
	const float a1 = 1.0f - a.w1, a2 = 1.0f - a.w2;
	const float ap = phi ? a2 : a1;
	const float sl0 = a.w3, sl1 = (float) ((double) a.w3 * (double) ap), sl2_ = (float) ((double) a.w3 * (double) ap * (double) ap);
	const float ap2 = (float) ((double) ap * (double) ap);

	// ---- the split (waves 2, 3): lane = (half hh of the lanes, column cl of the wave's 32); it owns samples 8 hp .. 8 hp + 7 of every chunk ----
	const int hh = lane >> 5, cl = lane & 31;
	const int scol = 32 * hw_ + cl;
	const int hp = hh ^ (cl & 1);                                        // (odd columns: the halves swap lanes — f16_rot)
	const uint32_t ssl = s0 + (uint32_t) (C == 2 ? cl : scol);
	const int sch = C == 2 ? hw_ : 0;
	const bool sowner = ssl < a.n_streams;
	const float* const srow = a.audio + (size_t) (sowner ? ssl : s0) * a.stride * C;
	const int srot = f16_rot (scol);

	// A WHOLE chunk travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no register holds data in flight, so nothing
	// makes the wave wait for it but the one s_waitcnt below), issued by wave 2, AHEAD (= 5) iterations ahead of the barrier behind
	// which the split reads it: HBM's latency is ~2500 cycles under this kernel's access pattern — one 128-byte line per stream
	// and chunk, 1 MB in flight chip-wide per chunk of prefetch distance — more than a whole chunk's time.  (Rounds 2 and 3 loaded
	// a chunk into registers and used it in the same iteration; round 4's first forms kept it in registers across the barrier:
	// the compiler's wait counts at a loop's back edge are conservative, vmcnt (0), and every form waited for the youngest
	// loads.  That wait was what bound this kernel.)  The call's ragged last chunk, and every chunk of a batch whose streams do
	// not start on 16 bytes, is loaded with plain loads by the lanes that split it.
	const bool dma_ok = (reinterpret_cast<size_t> (a.audio) & 15) == 0 && (a.n_streams == 1 || ((a.stride * C) & 3) == 0);
	// What lane l of DMA instruction i brings: stereo — frames 4 i + 2 (l >> 5), + 1 of stream l & 31 (both channels); mono —
	// frames 4 i .. + 3 of stream l.  (A stream past the batch re-reads stream s0; its column is zeroed when it is split.)
	constexpr int NP = 4;
	const float* dsrc[NP];
	{
		const uint32_t row = C == 2 ? (uint32_t) (lane & 31) : (uint32_t) lane;
		const float* const base = a.audio + (size_t) (s0 + row < a.n_streams ? s0 + row : s0) * a.stride * C;
#pragma unroll
		for (int i = 0; i < NP; ++i) dsrc[i] = base + (C == 2 ? (4 * i + 2 * (lane >> 5)) * 2 : 4 * i);
	}
	const uint32_t stg_lds = (uint32_t) (size_t) (__attribute__ ((address_space (3))) unsigned char*) stg;
	// (as inline assembly: the compiler must not know that these write LDS — it would wait for them, vmcnt (0), in front of
	// every LDS access and every barrier that follows, and the point is that they stay in flight across three barriers)
	auto dma = [&] (int j) __attribute__ ((always_inline)) {        // whole chunks only: (j + 1) F <= n_frames; into staging buffer j mod 4
		const float* const g0 = dsrc[0] + (size_t) (uint32_t) j * (F * C);
		const float* const g1 = dsrc[1] + (size_t) (uint32_t) j * (F * C);
		const float* const g2 = dsrc[2] + (size_t) (uint32_t) j * (F * C);
		const float* const g3 = dsrc[3] + (size_t) (uint32_t) j * (F * C);
		const uint32_t l = stg_lds + (uint32_t) ((int) (j & (NSTG - 1)) * NP) * 1024u;
		// (m0 — the LDS base of an LDS-DMA — is saved and restored inside the statement: the compiler may keep a value of its own
		// there, and does not accept m0 on a clobber list; an s_nop between a write of m0 and the instruction that reads it)
		uint32_t m0_was;
		asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
		              "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
		              "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
		              "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
		              : "=&s"(m0_was) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(l) : "memory", "scc");
	};
	// a chunk has landed when at most the DMA instructions (four per chunk) of the `younger` chunks behind it are outstanding
	auto dma_wait = [] (int younger) __attribute__ ((always_inline)) {
		if (younger >= 4)      asm volatile ("s_waitcnt vmcnt(16)" ::: "memory");
		else if (younger == 3) asm volatile ("s_waitcnt vmcnt(12)" ::: "memory");
		else if (younger == 2) asm volatile ("s_waitcnt vmcnt(8)" ::: "memory");
		else if (younger == 1) asm volatile ("s_waitcnt vmcnt(4)" ::: "memory");
		else                   asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
	};
	// this lane's eight samples of chunk j: from the staging buffer the LDS-DMA filled ...
	auto take_staged = [&]<int CH> (int j, float (&x)[8]) __attribute__ ((always_inline)) {
		const unsigned char* const b = stg + (int) (j & (NSTG - 1)) * (NP * 1024) + hp * 2048;
		if (C == 2) {
#pragma unroll
			for (int k = 0; k < 4; ++k) {                                    // frames 8 hp + 2 k, + 1: instruction 2 hp + (k >> 1), lanes 32 (k & 1) + stream
				typedef float f4v __attribute__ ((ext_vector_type (4)));
				f4v v = *reinterpret_cast<const f4v*> (b + (k >> 1) * 1024 + (32 * (k & 1) + cl) * 16);
				// (the whole 16 bytes, as ONE ds_read_b128: left to itself the compiler reads the channel's two words with a
				// ds_read2_b32, whose lanes — 16 bytes apart — sit four to a bank)
				asm volatile ("" : "+v"(v));
				x[2 * k] = CH ? v.y : v.x; x[2 * k + 1] = CH ? v.w : v.z;
			}
		} else {
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const float4 v = *reinterpret_cast<const float4*> (b + i * 1024 + scol * 16);
				x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
			}
		}
		// (a column past the batch holds a copy of stream s0's: nothing of it is ever stored — eight selects per chunk saved)
	};
	// ... or straight from memory (any chunk, zeros behind the call's last frame)
	auto take_ragged = [&] (int j, float (&x)[8]) __attribute__ ((always_inline)) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int f = j * F + 8 * hp + i;
			x[i] = (sowner && f < n_frames) ? srow[(size_t) f * C + sch] : 0.f;
		}
	};

	// ---- the split: every sample goes ONCE into the f16 ring, under its column's scale (both lanes of a column keep the same copy of it) ----
	ColScale cs;
	cs.set (238);
	float hm1 = 0.f, hm2 = 0.f, hm3 = 0.f, hm4 = 0.f;                    // max |x| of the four slots in front of the newest one
	bool pend = false;                                                   // this column's older slots still carry the previous scale
	auto half_max = [] (const float (&x)[8]) {
		float m = 0.f;
#pragma unroll
		for (int i = 0; i < 8; i += 2) m = max3f (m, fabsf (x[i]), fabsf (x[i + 1]));      // (a NaN loses every maximum; an Inf is one)
		return m;
	};
	auto put_f32 = [&] (int pos, const float (&x)[8]) {                  // exact, for phase 0 and for a later rescale
		float* const d = ring + scol * RSTRIDE + pos + 8 * hp;
		*reinterpret_cast<float4*> (d) = float4{x[0], x[1], x[2], x[3]};
		*reinterpret_cast<float4*> (d + 4) = float4{x[4], x[5], x[6], x[7]};
	};
	auto get_f32 = [&] (int pos, float (&x)[8]) {
		const float* const d = ring + scol * RSTRIDE + pos + 8 * hp;
		const float4 u = *reinterpret_cast<const float4*> (d), v = *reinterpret_cast<const float4*> (d + 4);
		x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
	};
	auto put_f16 = [&] (int pos, const float (&x)[8]) {                  // split under cs.sc into piece 2 (pos / 16) + hp of both arrays
		uint32_t hw[4], lw[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) m16::split_pair (x[2 * i] * cs.sc, x[2 * i + 1] * cs.sc, hw[i], lw[i]);
		const int off = scol * HSTRIDE + f16_place ((pos >> 3) + hp, srot);
		*reinterpret_cast<uint4*> (ringh + off) = uint4{hw[0], hw[1], hw[2], hw[3]};
		*reinterpret_cast<uint4*> (ringl + off) = uint4{lw[0], lw[1], lw[2], lw[3]};
	};
	// The chunk at ring position pos: its maximum, the scale check, both rings.  The scale is made for the maximum of FIVE slots — the
	// new chunk and the four in front of it: the products read a chunk two iterations after it was split, so the window they read
	// then reaches four slots back from the chunk split one iteration before.  If the scale has to move, the NEW chunk is written
	// under the new scale (nobody reads its slot before the next barrier); the older slots and un_sh are left to `rescale` at the top
	// of the next iteration — the products of this one are reading them.
	auto split = [&] (const float (&x)[8], int pos, int next_par) __attribute__ ((always_inline)) {
		const float m0 = max_across_halves (half_max (x));
		const float w = max3f (max3f (m0, hm1, hm2), hm3, hm4);
		const int se_w = ColScale::se_for (w);
		const bool move = __float_as_uint (m0) >= cs.cap || (__float_as_uint (w) < cs.low && se_w != cs.se);
		if (__builtin_expect (__ballot (move) != 0, 0)) {
			if (move) { cs.set (se_w); pend = true; }
			if (lane == 0) flag_sh[next_par] = 1;
		}
		put_f32 (pos, x);
		put_f16 (pos, x);
		hm4 = hm3; hm3 = hm2; hm2 = hm1; hm1 = m0;
	};
	// A pending move, found when chunk t + 1 was split (last iteration): the three chunks in front of it — t - 2, t - 1, t, the slots
	// behind the first of the window that starts at ring position w0 — are split AGAIN, from the exact samples of the f32 ring, under
	// the new scale (round 4 rescaled the quantised words in place: after a drop of ~100 dB inside one window the older samples kept
	// 16 bits instead of 22 — ADVICE r4), and the column's un follows.  This iteration's products are not touched by it: their
	// operands and their un were read an iteration ago, under the old scale, and the window of chunk t held nothing newer; the first
	// to read the new words are the operands of chunk t + 1, which the products fetch behind the cold barrier.
	auto rescale = [&] (int w0) {
		if (pend) {
#pragma unroll 1
			for (int k = 1; k < 4; ++k) {
				int p = w0 + 16 * k; p -= p >= RING ? RING : 0;
				float x[8];
				get_f32 (p, x);
				put_f16 (p, x);
			}
			un_sh[scol] = cs.un;
			pend = false;
		}
	};

	// ---- the products (waves 4 .. 11) and the per-frame maps: lane (cc, kg) = column cc of the block, frames 4 kg .. + 3 of the chunk ----
	// A block's products run as TWO units on two waves — unit A: phase 1 (6 MFMAs) and the frame's first pair map (x[n - 24], y1);
	// unit B: phases 2 and 3 (12 MFMAs) and the second.
	const int cc = lane & 15, kg = lane >> 4;
	// (unit A of blocks 0 .. 3 on waves 4, 5, 8, 6; unit B on 9, 10, 7, 11 — see NW)
	const bool unit_a = wid == 4 || wid == 5 || wid == 8 || wid == 6;
	const bool unit_b = wid == 9 || wid == 10 || wid == 7 || wid == 11;
	const int blk = wid == 4 ? 0 : wid == 5 ? 1 : wid == 8 ? 2 : wid == 6 ? 3 : wid == 9 ? 0 : wid == 10 ? 1 : wid == 7 ? 2 : 3;
	const int ucol = 16 * blk + cc;
	const int urot = f16_rot (ucol);
	const bool prod_wave = unit_a || unit_b;
	float pk = 0.f;                                                      // raw peak of the values this lane produced (column cc of its block)
	// the operands of a chunk: read from the rings one iteration AHEAD of the products, behind the MFMAs of the chunk before and
	// under its maps (round 4 read them at the top of the iteration they were used in: ~200 cycles of LDS latency per chunk on
	// every unit's critical path)
	struct Ops { m16::BFrag B; float4 x0; float un; };
	Ops ops[2];
	// where this lane's operands sit for a window that starts at slot sw = chunk mod 6: window positions 32 st + 8 kg .. + 7 are piece
	// (2 sw + 4 st + kg) mod 12 of the column — the piece of step 1 at slot sw is the piece of step 0 at slot sw + 2 — and x[n - 24]
	// of the lane's four frames is at ring position 16 sw + 24 + 4 kg.  Six addresses each, computed once: the loop is unrolled over
	// the ring's period and names them (round 5's first form computed them per chunk: 25 of a unit's 125 instructions).
	int pa[NSLOT], ox[NSLOT];
#pragma unroll
	for (int k = 0; k < NSLOT; ++k) {
		int q0 = 2 * k + kg; q0 -= q0 >= 2 * NSLOT ? 2 * NSLOT : 0;
		pa[k] = ucol * HSTRIDE + f16_place (q0, urot);
		int o0 = 16 * k + 24 + 4 * kg; o0 -= o0 >= RING ? RING : 0;
		ox[k] = (ucol * RSTRIDE + o0) * 4;
	}
	auto fetch_ops = [&]<bool UB, int SW> (Ops& o) __attribute__ ((always_inline)) {
		constexpr int S1 = (SW + 2) % NSLOT;
		o.B.h0 = *reinterpret_cast<const uint4*> (ringh + pa[SW]);
		o.B.h1 = *reinterpret_cast<const uint4*> (ringh + pa[S1]);
		o.B.l0 = *reinterpret_cast<const uint4*> (ringl + pa[SW]);
		o.B.l1 = *reinterpret_cast<const uint4*> (ringl + pa[S1]);
		o.x0 = float4{0.f, 0.f, 0.f, 0.f};
		if (!UB) o.x0 = *reinterpret_cast<const float4*> (reinterpret_cast<const unsigned char*> (ring) + ox[SW]);
		o.un = un_sh[ucol];
	};
	// one of the two units of a block (UB = false: phase 1 + the first pair map; true: phases 2, 3 + the second): the MFMAs ...
	// (af = the unit's own tap fragments: [phase of the unit][window step][hi | lo], four of the table's twelve for unit A, eight for B)
	auto unit_mfma = [&]<bool UB> (const m16::h8* af, const Ops& o, m16::f4 (&y)[2]) __attribute__ ((always_inline)) {
		constexpr int NPH = UB ? 2 : 1;
#pragma unroll
		for (int p = 0; p < NPH; ++p) y[p] = m16::f4{0.f, 0.f, 0.f, 0.f};
		// (the order of m16::block: consecutive MFMAs write different accumulators where there are two)
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (af[(p * 2 + 0) * 2 + 0], o.B.h0, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (af[(p * 2 + 1) * 2 + 0], o.B.h1, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (af[(p * 2 + 0) * 2 + 0], o.B.l0, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (af[(p * 2 + 1) * 2 + 0], o.B.l1, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (af[(p * 2 + 0) * 2 + 1], o.B.h0, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (af[(p * 2 + 1) * 2 + 1], o.B.h1, y[p]);
	};
	// ... and the maps: two attacks in a row are z <- max (z, a z + max (b1, b2), a^2 z + (a b1 + b2)), b = w v — two intercepts per
	// pair and filter, written straight to where that filter's chain lane reads them as ONE 8-byte word: row [half][frame], then
	// [group of 16 columns][filter][column] — the sixteen lanes of a unit that share a frame store 128 contiguous bytes per filter (a
	// ds_write2_b64 is served as two accesses of sixteen-lane groups on 32 banks: measured — with the second filter 64 bytes behind
	// the first, in groups of eight columns, every one of them cost a second cycle), and a chain wave's thirty-two-lane groups
	// (sixteen columns x two filters) read 256 contiguous bytes with ds_read_b64, 64 banks — as long as the compiler does not fuse two
	// of them into a ds_read2st64_b64, which is served in sixteen-lane groups on 32 banks: see `chain`
	auto unit_maps = [&]<bool UB, bool FULL> (int par, const float4& x0, float un, const m16::f4 (&y)[2], int nfl) __attribute__ ((always_inline)) {
		constexpr int NPH = UB ? 2 : 1;
		const float xr[4] = { x0.x, x0.y, x0.z, x0.w };
		const v2f W = v2f{a.w1, a.w2}, WA = v2f{a.w1 * a1, a.w2 * a2};
		unsigned char* const row = cbuf + par * CBUF_B + ((UB ? F : 0) + 4 * kg) * CROW + blk * 256 + cc * 8;
		float mprev = 0.f;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			unsigned char* const cd = row + r * CROW;
			// the frame's two values of this unit, as the ballistics see them: |x [n - 24]| (exact) and |y1|, or |y2| and |y3| (un-scaled:
			// un is a power of two)
			const float va = UB ? fabsf (y[0][r]) * un : xr[r] /* the modifiers below take its magnitude */, vb = fabsf (y[NPH - 1][r]) * un;
			// c = max (w va, w vb) = w max (va, vb) (w > 0 and rounding is monotone: the same bits), c' = a (w va) + w vb = (w a) va + w vb
			const float mx = max_abs_plain (va, vb);
			const v2f bb = W * vb;
			// (c and c' of a filter as plain VALU instructions whose results the compiler can put side by side for the 8-byte store: a
			// packed multiply leaves the two filters' c side by side instead, its own v_fmac_f32 overwrites the addend — two moves per
			// frame and filter either way)
			float g1x, g1y, g2x, g2y;
			asm ("v_mul_f32 %0, %1, %2" : "=v"(g1x) : "v"(W.x), "v"(mx));
			asm ("v_mul_f32 %0, %1, %2" : "=v"(g1y) : "v"(W.y), "v"(mx));
			asm ("v_fma_f32 %0, %1, |%2|, %3" : "=v"(g2x) : "v"(WA.x), "v"(va), "v"(bb.x));
			asm ("v_fma_f32 %0, %1, |%2|, %3" : "=v"(g2y) : "v"(WA.y), "v"(va), "v"(bb.y));
			*reinterpret_cast<v2f*> (cd) = v2f{g1x, g2x};                     // filter 1: (c, c')
			*reinterpret_cast<v2f*> (cd + 128) = v2f{g1y, g2y};               // filter 2
			// the raw peak (truepeakdsp.cc:65); only the call's ragged last chunk has frames that do not count
			const float mk = FULL ? mx : mx * (r < nfl ? 1.f : 0.f);
			if (r & 1) pk = max3_plain (pk, mprev, mk);                      // (one v_max3 per two frames)
			mprev = mk;
		}
	};

	// ---- prologue: the 48 frames before the call (47 of history; frame -48 is never multiplied by a non-zero tap), chunks 0 and 1 ----
	for (int e = threadIdx.x; e < NCOL * 48; e += NTHREADS) {
		const int col = e / 48, i = e % 48;                                  // frame i - 48 -> ring position i
		const uint32_t s = s0 + (uint32_t) (C == 2 ? (col & 31) : col);
		float x = 0.f;
		if (s < a.n_streams && i >= 1)                                       // history rows are [frame][2], right channel zero for mono engines
			x = a.hist[((size_t) s * MTR_FIR_HALO + (size_t) (i - 1)) * 2 + (C == 2 ? (col >> 5) : 0)];
		ring[col * RSTRIDE + i] = x;
	}
	if (threadIdx.x < 2) flag_sh[threadIdx.x] = 0;
	float xc0[8], xc1[8];
	if (wid == 2 || wid == 3) { take_ragged (0, xc0); take_ragged (1, xc1); }
	__syncthreads ();
	if (wid == 2 || wid == 3) {
		// the first window and the chunk behind it: one scale for the five slots, from their common maximum
		float xh[3][8];
		get_f32 (0, xh[0]); get_f32 (16, xh[1]); get_f32 (32, xh[2]);
		hm4 = max_across_halves (half_max (xh[0])); hm3 = max_across_halves (half_max (xh[1])); hm2 = max_across_halves (half_max (xh[2]));
		hm1 = max_across_halves (half_max (xc0));
		const float m0 = max_across_halves (half_max (xc1));
		cs.set (ColScale::se_for (max3f (max3f (m0, hm1, hm2), hm3, hm4)));
		put_f16 (0, xh[0]); put_f16 (16, xh[1]); put_f16 (32, xh[2]);
		put_f32 (48, xc0); put_f16 (48, xc0);
		put_f32 (64, xc1); put_f16 (64, xc1);
		un_sh[scol] = cs.un;
		hm4 = hm3; hm3 = hm2; hm2 = hm1; hm1 = m0;
		if (wid == 2 && dma_ok && !MTR_TPB_DBG_NOFETCH && !MTR_TPB_DBG_NODMA) {   // chunks 2 .. 4: on their way before the loop starts,
			int sent = 0;                                                  // and chunk 2 has landed behind the barrier below
#pragma unroll
			for (int j = 2; j < AHEAD; ++j)
				if ((j + 1) * F <= n_frames) { dma (j); ++sent; }
			dma_wait (sent - 1);
		}
	}
	__syncthreads ();

	// iteration t: chunk t + 5 leaves HBM, chunk t + 2 is split, chunk t goes through the products (whose operands were read an
	// iteration ago) and the maps, chunk t - 1 through the chains
#ifdef MTR_TPB_PROF
	unsigned long long pr[4] = { 0, 0, 0, 0 };
#endif
	// one frame of a chain: z <- max (w3 z, a w3 z + c1, a^2 w3 z + c2), then z <- max (z, a z + c3, a^2 z + c4); m = max (m, z1 + z2)
	// (the other filter's state sits in the neighbouring lane: one DPP add)
	// The sixteen frames' maps are read with ds_read_b64 written out as inline assembly, in two batches of eight frames, each waited
	// for by hand: left to the compiler the 32 reads become 16 ds_read2st64_b64 — half the instructions, but served at half the
	// width and, in this layout, two lanes to a bank: 512 LDS cycles per chunk and workgroup where these take 128 (the LDS was busy
	// 74 % of the time with them: the kernel's bound after the instruction diet).  The waits name the registers they protect —
	// the compiler knows nothing of what an asm statement has in flight — and the second one the state as well, so that it stays
	// behind the first eight frames.
	const uint32_t csrc = (uint32_t) (size_t) (__attribute__ ((address_space (3))) unsigned char*) cbuf
	                      + (uint32_t) ((2 * hw_ + (ci >> 4)) * 256 + phi * 128 + (ci & 15) * 8);
#define TPB_RD(dst, off) asm volatile ("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(csrc), "n"(off))
#define TPB_WAIT8(q, b) asm volatile ("s_waitcnt lgkmcnt(0)" : "+v"(q##1[b]), "+v"(q##1[b + 1]), "+v"(q##1[b + 2]), "+v"(q##1[b + 3]), "+v"(q##1[b + 4]), "+v"(q##1[b + 5]), "+v"(q##1[b + 6]), "+v"(q##1[b + 7]), \
	                                              "+v"(q##2[b]), "+v"(q##2[b + 1]), "+v"(q##2[b + 2]), "+v"(q##2[b + 3]), "+v"(q##2[b + 4]), "+v"(q##2[b + 5]), "+v"(q##2[b + 6]), "+v"(q##2[b + 7]), "+v"(z))
	const v2f SL12 = v2f{sl1, sl2_}, AP12 = v2f{ap, ap2};
	auto chain = [&]<bool FULL, int PARITY> (int nf) {
		constexpr int P = PARITY * CBUF_B;
		v2f q1[F], q2[F];
		TPB_RD (q1[0], P + 0 * CROW); TPB_RD (q2[0], P + (F + 0) * CROW); TPB_RD (q1[1], P + 1 * CROW); TPB_RD (q2[1], P + (F + 1) * CROW);
		TPB_RD (q1[2], P + 2 * CROW); TPB_RD (q2[2], P + (F + 2) * CROW); TPB_RD (q1[3], P + 3 * CROW); TPB_RD (q2[3], P + (F + 3) * CROW);
		TPB_RD (q1[4], P + 4 * CROW); TPB_RD (q2[4], P + (F + 4) * CROW); TPB_RD (q1[5], P + 5 * CROW); TPB_RD (q2[5], P + (F + 5) * CROW);
		TPB_RD (q1[6], P + 6 * CROW); TPB_RD (q2[6], P + (F + 6) * CROW); TPB_RD (q1[7], P + 7 * CROW); TPB_RD (q2[7], P + (F + 7) * CROW);
		TPB_WAIT8 (q, 0);
		TPB_RD (q1[8], P + 8 * CROW); TPB_RD (q2[8], P + (F + 8) * CROW); TPB_RD (q1[9], P + 9 * CROW); TPB_RD (q2[9], P + (F + 9) * CROW);
		TPB_RD (q1[10], P + 10 * CROW); TPB_RD (q2[10], P + (F + 10) * CROW); TPB_RD (q1[11], P + 11 * CROW); TPB_RD (q2[11], P + (F + 11) * CROW);
		TPB_RD (q1[12], P + 12 * CROW); TPB_RD (q2[12], P + (F + 12) * CROW); TPB_RD (q1[13], P + 13 * CROW); TPB_RD (q2[13], P + (F + 13) * CROW);
		TPB_RD (q1[14], P + 14 * CROW); TPB_RD (q2[14], P + (F + 14) * CROW); TPB_RD (q1[15], P + 15 * CROW); TPB_RD (q2[15], P + (F + 15) * CROW);
#pragma unroll
		for (int f = 0; f < F; ++f) {
			if (f == 8) TPB_WAIT8 (q, 8);
			if (FULL || f < nf) {                                            // wave-uniform: only the call's last chunk is short
				// (the two intercepts of a pair map sit side by side, as they were read: ONE packed fused multiply-add per pair map — the
				// same bits as two plain ones, 6.5 instructions per frame instead of 8.5)
				const float u0 = sl0 * z;
				const v2f u12 = __builtin_elementwise_fma (SL12, v2f{z, z}, q1[f]);
				const float zh = max3f (u0, u12.x, u12.y);
				const v2f v12 = __builtin_elementwise_fma (AP12, v2f{zh, zh}, q2[f]);
				z = max3f (zh, v12.x, v12.y);
				const float zo = __uint_as_float (__builtin_amdgcn_update_dpp (0, __float_as_uint (z), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
				zm = __builtin_fmaxf (zm, z + zo);                            // z1 + z2 (the same bits in both lanes)
			}
		}
	};
#undef TPB_RD
#undef TPB_WAIT8
	// THE LOOP, once per role: every wave runs the same iterations and the same barriers, but each role's copy of the loop has
	// its own registers (in ONE loop with the roles as branches the compiler re-fetched the products' twelve tap fragments from
	// global memory in every iteration).
	int sw = 0;                                                          // chunk t mod 6: the first slot of its window, and its pieces' base
	const int n_it = n_chunks + 1;
	// iterations [t0, t1) of the loop (the parity of an iteration is a compile-time constant in `work`: the buffers it selects)
	// One iteration.  A column's scale moved when chunk t + 1 was split (last iteration): the three chunks in front of it are split
	// again and its un follows (`rescale`), in a cold path with its own barrier.  The flag is uniform; its read goes out HERE and is
	// waited for where the role needs the answer — behind the chains' frames, behind the products' MFMAs and in front of the operands
	// they read ahead, with the split's own data — instead of an LDS round trip at the top of every iteration of every wave.  Every
	// role asks once and, if the answer is yes, passes the cold barrier once; the split clears the flag behind it.  (PAR = t & 1 and,
	// for the products, SW = t mod 6 are compile-time constants in `work`: the buffers and ring addresses they select.)
	auto iteration = [&]<int PAR, int SW> (int t, auto&& work) __attribute__ ((always_inline)) {
		const int flag = flag_sh[PAR];
		auto moved = [&] () __attribute__ ((always_inline)) { return __builtin_expect (__builtin_amdgcn_readfirstlane (flag) != 0, 0); };
		PROF_NOW (c0_);
		work.template operator()<PAR, SW> (t, moved);
		PROF_NOW (c1_);
		__syncthreads ();
		PROF_NOW (c2_);
		PROF_ADD (0, c1_ - c0_); PROF_ADD (2, c2_ - c1_); PROF_ADD (3, c2_ - c0_);
		sw = sw + 1 >= NSLOT ? 0 : sw + 1;
	};
	// iterations [t0, t1) in pairs: what an even iteration reads ahead for the odd one behind it stays in the registers it was loaded
	// into (with one iteration per trip and the parity as a branch the compiler rotated ~40 registers per trip)
	auto run_range = [&] (int t0, int t1, auto&& work) __attribute__ ((always_inline)) {
		int t = t0;
		if (t < t1 && (t & 1)) { iteration.template operator()<1, -1> (t, work); ++t; }
		for (; t + 1 < t1; t += 2) {
			iteration.template operator()<0, -1> (t, work);
			iteration.template operator()<1, -1> (t + 1, work);
		}
		if (t < t1) iteration.template operator()<0, -1> (t, work);
	};
	// ... and all of them in sixes, the ring's period (the products)
	auto run_six = [&] (auto&& work) __attribute__ ((always_inline)) {
		int t = 0;
		for (; t + 5 < n_it; t += 6) {
			iteration.template operator()<0, 0> (t, work);     iteration.template operator()<1, 1> (t + 1, work);
			iteration.template operator()<0, 2> (t + 2, work); iteration.template operator()<1, 3> (t + 3, work);
			iteration.template operator()<0, 4> (t + 4, work); iteration.template operator()<1, 5> (t + 5, work);
		}
		if (t < n_it) iteration.template operator()<0, 0> (t++, work);
		if (t < n_it) iteration.template operator()<1, 1> (t++, work);
		if (t < n_it) iteration.template operator()<0, 2> (t++, work);
		if (t < n_it) iteration.template operator()<1, 3> (t++, work);
		if (t < n_it) iteration.template operator()<0, 4> (t++, work);
	};
	auto run = [&] (auto&& work) __attribute__ ((always_inline)) { run_range (0, n_it, work); };
	if (wid < 2) {
		run ([&]<int PAR, int> (int t, auto&& moved) __attribute__ ((always_inline)) {
			if (t >= 1 && !MTR_TPB_DBG_NOCHAIN) {
				const int left = n_frames - (t - 1) * F;
				if (left >= F) chain.template operator()<true, PAR ^ 1> (F);
				else chain.template operator()<false, PAR ^ 1> ((int) left);
			}
			if (moved ()) __syncthreads ();
		});
	} else if (wid < 4) {
		// Two loops.  The first serves the whole chunks that came by LDS-DMA, and holds NO vector-memory instruction the compiler
		// knows of: any such load makes it count vmcnt, and its conservative waits (vmcnt (0) where paths join) would wait for the
		// DMA in flight as well.  The second takes over where chunk t + 2 is the call's ragged last one (or for the whole call,
		// when the streams do not start on 16 bytes) and drains the pipeline.
		const int n_whole = n_frames / F;
		const int t_dma = dma_ok && !MTR_TPB_DBG_NOFETCH ? (n_whole > 2 ? n_whole - 2 : 0) : 0;     // chunks 2 .. n_whole - 1 are staged
		auto staged = [&]<int CH, bool SENDER> () __attribute__ ((always_inline)) {
			run_range (0, t_dma, [&]<int PAR, int> (int t, auto&& moved) __attribute__ ((always_inline)) {
				// chunk t + 5 leaves HBM (into the staging buffer chunk t + 1 was read from, an iteration ago)
				if (SENDER && (t + AHEAD + 1) * F <= n_frames && !MTR_TPB_DBG_NODMA) dma (t + AHEAD);
				float x[8];
				take_staged.template operator()<CH> (t + 2, x);
				if (moved ()) { rescale (16 * sw); __syncthreads (); if (lane == 0) flag_sh[PAR] = 0; }
				if (!MTR_TPB_DBG_NOSPLIT) {
					int sp = sw + 5; sp -= sp >= NSLOT ? NSLOT : 0;              // chunk t + 2's slot
					split (x, 16 * sp, PAR ^ 1);
				}
				// chunk t + 3 — the next iteration's — has landed when at most the DMA instructions of the younger chunks are outstanding;
				// the wait stands in front of the barrier that lets BOTH split waves read it
				if (SENDER && !MTR_TPB_DBG_NODMA) {
					int younger = 0;
#pragma unroll
					for (int k = 4; k <= AHEAD; ++k) younger += (t + k + 1) * F <= n_frames ? 1 : 0;
					dma_wait (younger);
				}
			});
		};
		if (wid == 2) staged.template operator()<0, true> ();
		else          staged.template operator()<C == 2 ? 1 : 0, false> ();
		run_range (t_dma, n_it, [&]<int PAR, int> (int t, auto&& moved) __attribute__ ((always_inline)) {
			if (moved ()) { rescale (16 * sw); __syncthreads (); if (lane == 0) flag_sh[PAR] = 0; }
			if (t + 2 < n_chunks && !MTR_TPB_DBG_NOFETCH) {
				float x[8];
				take_ragged (t + 2, x);
				int sp = sw + 5; sp -= sp >= NSLOT ? NSLOT : 0;
				split (x, 16 * sp, PAR ^ 1);
			}
		});
	} else {
		auto run_unit = [&]<bool UB> () __attribute__ ((always_inline)) {
			constexpr int NF = UB ? 8 : 4, F0 = UB ? 4 : 0;
			m16::h8 af[NF];
#pragma unroll
			for (int f = 0; f < NF; ++f) af[f] = *reinterpret_cast<const m16::h8*> (a.mfma_a + (F0 + f) * 512 + lane * 8);
			// (used — waited for — right here: a load still pending in front of the loop makes the compiler guard every register it
			// might land in with a vmcnt wait inside it)
#pragma unroll
			for (int f = 0; f < NF; ++f) asm volatile ("" : "+v"(af[f]));
			fetch_ops.template operator()<UB, 0> (ops[0]);                  // the first chunk's operands
			run_six ([&]<int PAR, int SW> (int t, auto&& moved) __attribute__ ((always_inline)) {
				if (t < n_chunks && !MTR_TPB_DBG_NOPROD) {
					m16::f4 y[2];
					unit_mfma.template operator()<UB> (af, ops[PAR], y);         // on the operands (and the un) read an iteration ago
					// the next chunk's operands — behind the cold barrier if the rest of its window is being split again under a new scale;
					// unconditionally (behind the call's last chunk they are whatever the ring holds: nobody uses them, and a branch here
					// would put a wait for them in front of the maps)
					__builtin_amdgcn_sched_barrier (0);                            // (the wait for the flag stays BEHIND the MFMAs)
					if (moved ()) __syncthreads ();
					fetch_ops.template operator()<UB, (SW + 1) % NSLOT> (ops[PAR ^ 1]);
					if ((t + 1) * F <= n_frames) unit_maps.template operator()<UB, true> (PAR, ops[PAR].x0, ops[PAR].un, y, 4);
					else {
						const int left = n_frames - t * F - 4 * kg;           // this lane's frames are 4 kg .. 4 kg + 3 of the chunk
						unit_maps.template operator()<UB, false> (PAR, ops[PAR].x0, ops[PAR].un, y, left >= 4 ? 4 : (left > 0 ? (int) left : 0));
					}
				} else if (moved ()) __syncthreads ();
			});
		};
		if (unit_a)      run_unit.template operator()<false> ();
		else if (unit_b) run_unit.template operator()<true> ();
		else run ([&]<int, int> (int, auto&& moved) __attribute__ ((always_inline)) { if (moved ()) __syncthreads (); });   // (idle: keeps the barriers' count)
	}
#ifdef MTR_TPB_PROF
	if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 4; ++i) g_tpb_prof[wid][i] = pr[i];
#endif

	// the raw peaks per column (non-negative floats order as unsigned ints)
	uint32_t* const pk_sh = reinterpret_cast<uint32_t*> (ring);          // the ring is spent
	if (wid == 0) pk_sh[lane] = 0u;
	__syncthreads ();
	if (prod_wave) atomicMax (&pk_sh[ucol], __float_as_uint (pk));
	__syncthreads ();
	if (wid < 2 && cowner) {
		if (phi) cst->tpb_z2[cch] = z + 1e-20f;                            // truepeakdsp.cc:86-87
		else {
			cst->tpb_z1[cch] = z + 1e-20f;
			cst->tpb_m[cch] = zm * a.g;                                      // :89, then read (m, p)
			cst->tpb_p[cch] = __uint_as_float (pk_sh[ccol]);
			if (C == 1) { cst->tpb_z1[1] = 1e-20f; cst->tpb_z2[1] = 1e-20f; cst->tpb_m[1] = 0.f; cst->tpb_p[1] = 0.f; }   // mono engines keep a zero right channel
		}
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

}  // namespace

int mtr_launch_tpb (const mtr_tpb_args& a, void* stream)
{
	if (a.n_frames >= 0x7ffff000ull) return -1;                        // (the kernel counts frames and chunks in 32 bits, a few chunks ahead)
	static bool raised = false;
	if (!raised) {
		(void) hipFuncSetAttribute ((const void*) k_tpb<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		(void) hipFuncSetAttribute ((const void*) k_tpb<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		raised = true;
	}
	const uint32_t nstr = NCOL / (a.n_channels == 2 ? 2 : 1);
	const dim3 grid ((a.n_streams + nstr - 1) / nstr);
	hipStream_t st = (hipStream_t) stream;
	if (a.n_channels == 2) hipLaunchKernelGGL ((k_tpb<2>), grid, dim3 (NTHREADS), LDS_BYTES, st, a);
	else                   hipLaunchKernelGGL ((k_tpb<1>), grid, dim3 (NTHREADS), LDS_BYTES, st, a);
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
