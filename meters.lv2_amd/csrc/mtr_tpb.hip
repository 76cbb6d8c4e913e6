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
// value, so it is not a cheap associative scan and time stays serial per (stream, channel).  What this kernel does
// (round 4's form; round 3's split every 64-sample window again for every 16-frame chunk — each sample four times, in
// the lanes of the products — which made a block of products a 1600-cycle dependent sequence):
//
//   * `if (v > z) z += w (v - z)` is z <- max (z, a z + w v) with a = 1 - w: a monotone max-affine map, and maps of
//     that kind compose.  The intercepts depend on the values only — any lane can form them, for any frame — so what is
//     left ON the chain per frame and filter is the release, two fused multiply-adds and a v_max3, twice (the frame's
//     four values as two pair maps, see below; round 3 applied them as one five-piece map).  Exact
//     in real arithmetic; in f32 a few ulps from the reference's sequence (held to the 2e-6 of
//     tests/test_gpu_parity.py::test_truepeak_ballistics_*; 600 fuzzed shapes: <= 7.4e-7 up to 48 kHz, 1.6e-6 at 192 kHz).
//   * the interpolator is the matrix-pipe one of mtr_mfma16_fir.h (samples and taps as two f16 halves, three partial
//     products, f32 accumulation: within 4e-7 of the exact-f32 chain): a workgroup owns 64 (stream, channel)
//     columns = 4 blocks of 16, a chunk is 16 frames = the rows of one block.
//   * EVERY SAMPLE IS SPLIT ONCE, by the wave that fetched it (lane = column), into a ring of f16 hi / lo pair words
//     (five slots of 16 samples per column, 176 bytes per column and array: conflict-free 16-byte accesses), under a
//     power-of-two scale per column that follows the window's maximum with hysteresis: it stands while the maximum of
//     the 64-sample window stays in [2^7, 2^15) scaled, and when it has to move — a sample too large for it, or a window
//     that has become 2^5 quieter than the scale was made for — the column's three older slots are rescaled in place (a
//     power of two: exact but for what falls below f16's range, 2^-27 of the new maximum) in a cold path with its own
//     barrier.  22 bits of every sample within 2^-11 of the window's maximum, as with a scale per window.  The products
//     then READ their operands (four ds_read_b128 per block) instead of forming them; phase 0 (x[n - 24]) comes from an
//     f32 ring the same wave fills, exactly.
//   * a chunk travels HBM -> LDS by LDS-DMA, three iterations ahead of its products, issued as inline assembly: the
//     compiler's own vmcnt bookkeeping (conservative at loop back edges and where paths join, and aware that an LDS-DMA
//     writes LDS) waited for whatever was in flight at every barrier — in rounds 2 and 3, and in this round's first four
//     forms, the wave that fetched sat out HBM's latency in every chunk, and the whole workgroup with it at the barrier.
//   * the lanes that produce a frame's four values form its maps as well: two attacks in a row are
//     z <- max (z, a z + max (b1, b2), a^2 z + (a b1 + b2)), b = w v, so a frame is two such maps applied one after the
//     other on the chain — two intercepts per pair and filter, ten instructions per frame and column instead of
//     twenty-two for the frame's single five-piece map — written straight to where the chains read them: no values in
//     LDS, no map waves, one chunk less between products and chains.
//   * twelve waves, three per SIMD (156 registers), each with ITS OWN copy of the loop (one loop with the roles as branches
//     made the compiler fetch the products' tap fragments from global memory in every iteration): wave 0 walks the 64 chains
//     (chunk t - 1); a block's products run as TWO units on two waves — unit A: phase 1 (6 MFMAs) and the frame's first
//     pair map (x[n - 24], y1), unit B: phases 2 and 3 (12 MFMAs) and the second — the B units on waves 1, 5, 2, 6, the A
//     units on waves 4, 9, 10, 7 (two B and an A on SIMDs 1 and 2, an A beside the chains, an A beside the split); wave 3
//     sends chunk t + 3 on its way and splits chunk t + 1; waves 8 and 11 only keep the barriers' count.  One barrier per
//     chunk; the chains (~1400 cycles per chunk) are what bounds it now.
//     (The forms this one was measured against — one or two blocks per products wave, values and five-piece maps through
//     LDS, map waves — live in git and in profiles/r04_tpb.md, r04_tpb_experiments.md; ONE form ships.)
#include <hip/hip_runtime.h>

#include "mtr_internal.h"
#include "mtr_mfma16_fir.h"

typedef float v2f __attribute__ ((ext_vector_type (2)));

// tools/tpb_prof.hip builds this file with MTR_TPB_PROF: cycles per wave and section for workgroup 0
#ifdef MTR_TPB_PROF
__device__ unsigned long long g_tpb_prof[12][4];
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

constexpr int NW = 12;                         // wave 0: the chains; wave 3: fetch + split; eight product units; two idle
constexpr int F = 16;                          // frames per chunk = rows of one MFMA block
constexpr int NCOL = 64;                       // (stream, channel) columns per workgroup: 32 stereo or 64 mono streams
constexpr int RING = 5 * F;                    // samples per column: the 64-sample window of a chunk + the chunk being fetched
constexpr int RSTRIDE = RING + 4;              // floats per column of the f32 ring (16-byte rows; 84 = 20 mod 64: sixteen columns hit sixteen bank groups)
constexpr int RING_B = NCOL * RSTRIDE * 4;
constexpr int HSTRIDE = 176;                   // bytes per column and array of the f16 ring: ten 16-byte pieces (8 samples as four pair words) + 16:
                                               // 11 c mod 16 is a permutation — sixteen columns' pieces sit in sixteen bank groups
constexpr int HRING_B = NCOL * HSTRIDE;        // one array: hi | lo
constexpr int AUX_B = 512;                     // float un [NCOL]; int flag [2]
constexpr int CBUF_B = F * 2 * NCOL * 16;      // a chunk of maps: [frame][half][column] x (c_k of filter 1, of filter 2) for k = 1, 2 | 3, 4
constexpr int STG_B = 3 * 4 * 64 * 16;         // three chunks in flight from HBM, as the LDS-DMA leaves them: [chunk][piece 64 i + lane] x 16 bytes
constexpr int LDS_BYTES = RING_B + 2 * HRING_B + AUX_B + 2 * CBUF_B + STG_B;
constexpr int NTHREADS = 64 * NW;
static_assert (RING % 8 == 0 && (RSTRIDE * 4) % 16 == 0, "operand slices never wrap inside the ring");
static_assert (LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// A block's products run as two units on two waves — unit A: phase 1 (6 MFMAs) and the frame's first pair map (x[n - 24], y1);
// unit B: phases 2 and 3 (12 MFMAs) and the second — three waves per SIMD (w % 4): the 30 MFMAs of two B units and an A unit
// on SIMDs 1 and 2, an A unit beside the chains and one beside the split
constexpr int ASET[4] = { 4, 9, 10, 7 }, BSET[4] = { 1, 5, 2, 6 };

__device__ __forceinline__ float max3f (float a, float b, float c) { return __builtin_fmaxf (__builtin_fmaxf (a, b), c); }
__device__ __forceinline__ v2f fma2 (v2f a, v2f b, v2f c) { return __builtin_elementwise_fma (a, b, c); }
__device__ __forceinline__ v2f max2 (v2f a, v2f b) { return v2f{__builtin_fmaxf (a.x, b.x), __builtin_fmaxf (a.y, b.y)}; }

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
	float* const ring = reinterpret_cast<float*> (smem);                 // [NCOL][RSTRIDE] f32: exact samples
	unsigned char* const ringh = smem + RING_B;                          // [NCOL][HSTRIDE]: f16 hi pair words, slot k at 32 k
	unsigned char* const ringl = ringh + HRING_B;                        // ... lo
	float* const un_sh = reinterpret_cast<float*> (ringl + HRING_B);     // [NCOL]: 2^-15 / scale of every column, as the ring holds it
	int* const flag_sh = reinterpret_cast<int*> (un_sh + NCOL);          // [2]: a rescale is pending for the iteration of this parity
	unsigned char* const cbuf = smem + RING_B + 2 * HRING_B + AUX_B;     // [2][F][2][NCOL] float4
	const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane (threadIdx.x >> 6);
	const uint32_t s0 = blockIdx.x * NSTR;
	const int64_t n_frames = (int64_t) a.n_frames;
	const int64_t n_chunks = (n_frames + F - 1) / F;

	// ---- the chains: lane = column ----------------------------------------------------------------------------------------
	const int ch = C == 2 ? lane >> 5 : 0;
	const uint32_t sl = s0 + (uint32_t) (C == 2 ? (lane & 31) : lane);
	const bool owner = sl < a.n_streams;
	mtr_stream_state* const st = a.state + (owner ? sl : 0);
	float z1 = 0.f, z2 = 0.f, zm = 0.f;                                  // (the two filters are walked as one packed pair)
	if (wid == 0 && owner) {
		z1 = st->tpb_z1[ch]; z2 = st->tpb_z2[ch];
		z1 = z1 > 20 ? 20 : (z1 < 0 ? 0 : z1);                             // truepeakdsp.cc:54-55
		z2 = z2 > 20 ? 20 : (z2 < 0 ? 0 : z2);
	}
	// slopes of the frame's map: a^k w3 (the release first, then up to four attacks)
	const float a1 = 1.0f - a.w1, a2 = 1.0f - a.w2;
	float s1[5], s2[5];
	{
		double p1 = (double) a.w3, p2 = (double) a.w3;
		for (int k = 0; k < 5; ++k) { s1[k] = (float) p1; s2[k] = (float) p2; p1 *= (double) a1; p2 *= (double) a2; }
	}

	// ---- wave 3: what it fetches ------------------------------------------------------------------------------------------
	// A chunk is 256 pieces of 16 bytes: stereo piece p = (stream p / 8, frames 2 (p % 8), + 1), mono piece p = (stream p / 4,
	// frames 4 (p % 4) .. + 3).  Wave 3 fetches them, four per lane.
	constexpr int NP = 4;
	const float* prow[NP];
	int pfr[NP], pdst[NP];
	bool plive[NP];
#pragma unroll
	for (int i = 0; i < NP; ++i) {
		const int p = 64 * i + lane;
		const int row = C == 2 ? p >> 3 : p >> 2;
		pfr[i] = C == 2 ? 2 * (p & 7) : 4 * (p & 3);
		plive[i] = s0 + (uint32_t) row < a.n_streams;
		prow[i] = a.audio + (size_t) (plive[i] ? s0 + (uint32_t) row : s0) * a.stride * C;
		pdst[i] = row * RSTRIDE + pfr[i];                                // (stereo: the right channel's column is 32 further)
	}
	// A WHOLE chunk travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no register holds data in flight, so nothing
	// makes the wave wait for it but the one s_waitcnt below), THREE iterations ahead of its products: HBM's latency is
	// ~2500 cycles under this kernel's access pattern — one 128-byte line per stream and chunk, 1 MB in flight chip-wide per
	// chunk of prefetch distance — more than a whole chunk's time.  (Rounds 2 and 3 loaded a chunk into registers and used
	// it in the same iteration; round 4's first forms kept it in registers across the barrier, one and two chunks ahead: the
	// compiler's wait counts at a loop's back edge are conservative, vmcnt (0), and every form waited for the youngest
	// loads.  That wait was what bound this kernel: tools/tpb_prof.hip, MTR_TPB_DBG_NOFETCH.)  The lane-linear destination
	// is the [piece] order the lanes read back.  The call's ragged last chunk, and every chunk of a batch whose streams do
	// not start on 16 bytes, is loaded and stored in one go.
	unsigned char* const stg = cbuf + 2 * CBUF_B;
	const bool dma_ok = (reinterpret_cast<size_t> (a.audio) & 15) == 0 && (a.n_streams == 1 || ((a.stride * C) & 3) == 0);
	// What lane l of DMA instruction i brings: stereo — frames 4 i + 2 (l >> 5), + 1 of stream l & 31 (both channels); mono —
	// frames 4 i .. + 3 of stream l.  The sixteen lanes that later read a piece index k of sixteen neighbouring streams find
	// them 16 bytes apart: conflict-free.  (A stream past the batch re-reads stream s0; its column is zeroed when it is split.)
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
	auto dma = [&] (int64_t j, int buf) __attribute__ ((always_inline)) {   // whole chunks only: (j + 1) F <= n_frames
#pragma unroll
		for (int i = 0; i < NP; ++i) {
			const float* const g = dsrc[i] + (size_t) j * (F * C);
			const uint32_t l = stg_lds + (uint32_t) (buf * NP + i) * 1024u;
			// (m0 is saved and restored inside the statement: the compiler may keep a value of its own there — it does not accept m0 on
			// a clobber list — and the LDS-DMA reads its LDS base from it)
			uint32_t m0_was;
			asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_was) : "v"(g), "s"(l) : "memory");
		}
	};
	auto store = [&] (int i, int slot, float4 v) __attribute__ ((always_inline)) {
		float* const d = ring + pdst[i] + slot;
		if (!plive[i]) v = float4{0.f, 0.f, 0.f, 0.f};                   // a row of a stream past the batch (it re-read stream s0)
		if (C == 2) {                                                    // v = L0 R0 L1 R1
			*reinterpret_cast<v2f*> (d) = v2f{v.x, v.z};
			*reinterpret_cast<v2f*> (d + 32 * RSTRIDE) = v2f{v.y, v.w};
		} else *reinterpret_cast<float4*> (d) = v;
	};
	auto fetch_put_ragged = [&] (int64_t j, int slot) __attribute__ ((always_inline)) {   // any chunk, zeros behind the call's last frame
#pragma unroll 1
		for (int i = 0; i < NP; ++i) {
			const int64_t f = j * F + pfr[i];
			float x[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int64_t fk = C == 2 ? f + (k >> 1) : f + k;
				x[k] = fk < n_frames ? prow[i][(size_t) fk * C + (C == 2 ? (k & 1) : 0)] : 0.f;
			}
			store (i, slot, float4{x[0], x[1], x[2], x[3]});
		}
	};

	// ---- wave 3, lane = column: every sample is split ONCE into the f16 ring, under the column's scale -----------------------
	ColScale cs;
	cs.set (238);
	float hm1 = 0.f, hm2 = 0.f, hm3 = 0.f;                               // max |x| of the three slots in front of the newest one
	int pend_k = 0;                                                      // this column's older slots still carry a scale 2^-pend_k off
	auto wave_sync = [] () {                                             // LDS written by other lanes of THIS wave is about to be read
		__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier ();
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
	};
	auto read_slot = [&] (int pos, float (&x)[F]) {                      // the column's 16 samples at ring position pos (a multiple of 16)
		const float* const col = ring + lane * RSTRIDE + pos;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const float4 v = *reinterpret_cast<const float4*> (col + 4 * q);
			x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
		}
	};
	auto slot_max = [] (const float (&x)[F]) {
		float m = 0.f;
#pragma unroll
		for (int i = 0; i < F; i += 2) m = max3f (m, fabsf (x[i]), fabsf (x[i + 1]));      // (a NaN loses every maximum; an Inf is one)
		return m;
	};
	auto write_slot = [&] (int pos, const float (&x)[F]) {               // ... split under cs.sc into slot pos / 16 of both arrays
		uint32_t hw[8], lw[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) m16::split_pair (x[2 * i] * cs.sc, x[2 * i + 1] * cs.sc, hw[i], lw[i]);
		unsigned char* const h = ringh + lane * HSTRIDE + 2 * pos;       // 16 samples = 32 bytes
		unsigned char* const l = ringl + lane * HSTRIDE + 2 * pos;
		*reinterpret_cast<uint4*> (h)      = uint4{hw[0], hw[1], hw[2], hw[3]};
		*reinterpret_cast<uint4*> (h + 16) = uint4{hw[4], hw[5], hw[6], hw[7]};
		*reinterpret_cast<uint4*> (l)      = uint4{lw[0], lw[1], lw[2], lw[3]};
		*reinterpret_cast<uint4*> (l + 16) = uint4{lw[4], lw[5], lw[6], lw[7]};
	};
	// the chunk at ring position pos has landed in the f32 ring: its maximum, the scale check, the split.  If the scale has to
	// move, the NEW chunk is written under the new scale (nobody reads its slot before the next barrier), the older slots and
	// un_sh are left to `rescale` at the top of the next iteration — the products of this one are reading them.
	auto split_regs = [&] (const float (&x)[F], int pos, int next_par) __attribute__ ((always_inline)) {
		const float m0 = slot_max (x);
		const float w = max3f (max3f (m0, hm1, hm2), hm3, 0.f);
		const int se_w = ColScale::se_for (w);
		const bool move = __float_as_uint (m0) >= cs.cap || (__float_as_uint (w) < cs.low && se_w != cs.se);
		if (__builtin_expect (__ballot (move) != 0, 0)) {
			if (move) { pend_k += cs.se - se_w; cs.set (se_w); }           // (every move is served at the next iteration's top: pend_k never adds up)
			if (lane == 0) flag_sh[next_par] = 1;
		}
		write_slot (pos, x);
		hm3 = hm2; hm2 = hm1; hm1 = m0;
	};
	auto split = [&] (int pos, int next_par) __attribute__ ((always_inline)) {      // from the f32 ring (the chunk was stored there piece by piece)
		float x[F];
		read_slot (pos, x);
		split_regs (x, pos, next_par);
	};
	// from the staging area the LDS-DMA filled: this column's sixteen samples go to the f32 ring (exact: phase 0) and,
	// split, to the f16 ring — one trip through the LDS each way
	auto split_staged = [&] (int buf, int pos, int next_par) __attribute__ ((always_inline)) {
		float x[F];
		const unsigned char* const b = stg + buf * (NP * 1024);
		if (C == 2) {
			const int r = lane & 31;
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const float4 v = *reinterpret_cast<const float4*> (b + (k >> 1) * 1024 + (r + 32 * (k & 1)) * 16);
				x[2 * k] = ch ? v.y : v.x; x[2 * k + 1] = ch ? v.w : v.z;
			}
		} else {
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const float4 v = *reinterpret_cast<const float4*> (b + i * 1024 + lane * 16);
				x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
			}
		}
		if (!owner) {
#pragma unroll
			for (int i = 0; i < F; ++i) x[i] = 0.f;
		}
		float* const col = ring + lane * RSTRIDE + pos;
#pragma unroll
		for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*> (col + 4 * q) = float4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
		split_regs (x, pos, next_par);
	};
	// a pending move: the three slots in front of the newest (at ring position pos) times 2^-pend_k, exactly (a power of two;
	// what falls below f16's range is 2^-27 of the window's new maximum), and the column's un for the products
	auto rescale = [&] (int pos) {
		if (pend_k != 0) {
			const int k = max (-60, min (60, -pend_k));
			const float r = __uint_as_float ((uint32_t) (127 + k) << 23);
			typedef _Float16 h2v __attribute__ ((ext_vector_type (2)));
#pragma unroll 1
			for (int back = 1; back <= 3; ++back) {
				int p = pos - 16 * back; p += p < 0 ? RING : 0;
#pragma unroll 1
				for (int arr = 0; arr < 2; ++arr) {
					unsigned char* const b = (arr ? ringl : ringh) + lane * HSTRIDE + 2 * p;
#pragma unroll
					for (int i = 0; i < 2; ++i) {
						uint4 v = *reinterpret_cast<uint4*> (b + 16 * i);
						uint32_t* const wv = reinterpret_cast<uint32_t*> (&v);
#pragma unroll
						for (int j = 0; j < 4; ++j) {
							const h2v hv = __builtin_bit_cast (h2v, wv[j]);
							wv[j] = m16::hi_pair ((float) hv.x * r, (float) hv.y * r);
						}
						*reinterpret_cast<uint4*> (b + 16 * i) = v;
					}
				}
			}
			un_sh[lane] = cs.un;
			pend_k = 0;
		}
	};

	const v2f AA = v2f{a1, a2};
	// ---- the products and the per-frame maps: lane (cc, kg) = column cc of the block, frames 4 kg .. + 3 of the chunk ---------
	const int cc = lane & 15, kg = lane >> 4;
	m16::AFrag A;
	const int unit_a = wid == ASET[0] ? 0 : wid == ASET[1] ? 1 : wid == ASET[2] ? 2 : wid == ASET[3] ? 3 : -1;
	const int unit_b = wid == BSET[0] ? 0 : wid == BSET[1] ? 1 : wid == BSET[2] ? 2 : wid == BSET[3] ? 3 : -1;
	const int my_block = unit_a >= 0 ? unit_a : unit_b;
	const bool prod_wave = my_block >= 0;
	if (prod_wave) {
		A.load (a.mfma_a, lane);
		// (used — waited for — right here: a load still pending where the roles part makes the compiler guard every register it
		// might land in with a vmcnt wait, in EVERY role's loop, and in wave 3's that wait would be for the LDS-DMA in flight)
#pragma unroll
		for (int f = 0; f < MTR_M16_FRAGS; ++f) asm volatile ("" : "+v"(A.a[f]));
	}
	float pk[2] = { 0.f, 0.f };                                          // raw peak of the values this lane produced (column cc of its block)

	// one of the two units of a block (UB = false: phase 1 + the first pair map; true: phases 2, 3 + the second)
	auto unit = [&]<bool UB> (int par, int w0, int b, int nfl) __attribute__ ((always_inline)) {
		int q0 = (w0 >> 3) + kg; q0 -= q0 >= 10 ? 10 : 0;
		int q1 = q0 + 4; q1 -= q1 >= 10 ? 10 : 0;
		const int col = 16 * b + cc;
		const unsigned char* const h = ringh + col * HSTRIDE;
		const unsigned char* const l = ringl + col * HSTRIDE;
		m16::BFrag B;
		B.h0 = *reinterpret_cast<const uint4*> (h + 16 * q0);
		B.h1 = *reinterpret_cast<const uint4*> (h + 16 * q1);
		B.l0 = *reinterpret_cast<const uint4*> (l + 16 * q0);
		B.l1 = *reinterpret_cast<const uint4*> (l + 16 * q1);
		float4 x0 = float4{0.f, 0.f, 0.f, 0.f};
		if (!UB) { int o0 = w0 + 24 + 4 * kg; o0 -= o0 >= RING ? RING : 0; x0 = *reinterpret_cast<const float4*> (ring + col * RSTRIDE + o0); }
		const float un = un_sh[col];
		constexpr int P0 = UB ? 1 : 0, NPH = UB ? 2 : 1;
		m16::f4 y[NPH];
#pragma unroll
		for (int p = 0; p < NPH; ++p) y[p] = m16::f4{0.f, 0.f, 0.f, 0.f};
		// (the order of m16::block: consecutive MFMAs write different accumulators where there are two)
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (A.a[((P0 + p) * 2 + 0) * 2 + 0], B.h0, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (A.a[((P0 + p) * 2 + 1) * 2 + 0], B.h1, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (A.a[((P0 + p) * 2 + 0) * 2 + 0], B.l0, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (A.a[((P0 + p) * 2 + 1) * 2 + 0], B.l1, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (A.a[((P0 + p) * 2 + 0) * 2 + 1], B.h0, y[p]);
#pragma unroll
		for (int p = 0; p < NPH; ++p) M16_MFMA (A.a[((P0 + p) * 2 + 1) * 2 + 1], B.h1, y[p]);
		const float xr[4] = { x0.x, x0.y, x0.z, x0.w };
		const v2f W = v2f{a.w1, a.w2};
		float pm = 0.f, px = 0.f;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const float keep = r < nfl ? 1.f : 0.f;
			unsigned char* const cd = cbuf + par * CBUF_B + (((4 * kg + r) * 2 + (UB ? 1 : 0)) * NCOL + col) * 16;
			if (!UB) {
				px = __builtin_fmaxf (px, fabsf (xr[r]) * keep);
				pm = __builtin_fmaxf (pm, fabsf (y[0][r]) * keep);
				const v2f b1 = W * fabsf (xr[r]), b2 = W * (fabsf (y[0][r]) * un);
				const v2f d1 = max2 (b1, b2), d2 = fma2 (AA, b1, b2);
				*reinterpret_cast<float4*> (cd) = float4{d1.x, d1.y, d2.x, d2.y};
			} else {
				pm = __builtin_fmaxf (pm, __builtin_fmaxf (fabsf (y[0][r]), fabsf (y[NPH - 1][r])) * keep);
				const v2f b3 = W * (fabsf (y[0][r]) * un), b4 = W * (fabsf (y[NPH - 1][r]) * un);
				const v2f e1 = max2 (b3, b4), e2 = fma2 (AA, b3, b4);
				*reinterpret_cast<float4*> (cd) = float4{e1.x, e1.y, e2.x, e2.y};
			}
		}
		pk[0] = max3f (pk[0], px, pm * un);
	};

	// ---- prologue: the 48 frames before the call (47 of history; frame -48 is never multiplied by a non-zero tap) and chunk 0 ----
	if (wid != 0) {
		for (int e = (wid - 1) * 64 + lane; e < NCOL * 48; e += (NW - 1) * 64) {
			const int col = e / 48, i = e % 48;                              // frame i - 48 -> ring position 32 + i
			const uint32_t s = s0 + (uint32_t) (C == 2 ? (col & 31) : col);
			float x = 0.f;
			if (s < a.n_streams && i >= 1)                                   // history rows are [frame][2], right channel zero for mono engines
				x = a.hist[((size_t) s * MTR_FIR_HALO + (size_t) (i - 1)) * 2 + (C == 2 ? (col >> 5) : 0)];
			ring[col * RSTRIDE + 32 + i] = x;
		}
		if (wid == 3) fetch_put_ragged (0, 0);
	}
	if (threadIdx.x < 2) flag_sh[threadIdx.x] = 0;
	__syncthreads ();
	if (wid == 3) {
		// the first window: one scale for its four slots, from their common maximum
		float x[4][F];
		read_slot (32, x[0]); read_slot (48, x[1]); read_slot (64, x[2]); read_slot (0, x[3]);
		hm3 = slot_max (x[0]); hm2 = slot_max (x[1]); hm1 = slot_max (x[2]);
		const float m0 = slot_max (x[3]);
		cs.set (ColScale::se_for (max3f (max3f (m0, hm1, hm2), hm3, 0.f)));
		write_slot (32, x[0]); write_slot (48, x[1]); write_slot (64, x[2]); write_slot (0, x[3]);
		un_sh[lane] = cs.un;
		hm3 = hm2; hm2 = hm1; hm1 = m0;
		if (dma_ok && !MTR_TPB_DBG_NOFETCH) {                           // chunks 1 and 2: on their way before the loop starts
			if (2 * F <= n_frames) dma (1, 1);
			if (3 * F <= n_frames) dma (2, 2);
		}
	}
	__syncthreads ();

	// iteration t: chunk t + 1 is fetched and split, chunk t goes through the products, chunk t - 1 through the maps, chunk t - 2 through the chains
#ifdef MTR_TPB_PROF
	unsigned long long pr[4] = { 0, 0, 0, 0 };
#endif
	// one frame of the chains: z <- max (w3 z, a w3 z + c1, ..., a^4 w3 z + c4) for both filters (packed), then m
	v2f zz = v2f{z1, z2};
	v2f sl2[5];
#pragma unroll
	for (int k = 0; k < 5; ++k) sl2[k] = v2f{s1[k], s2[k]};
	const v2f AA2 = v2f{(float) ((double) a1 * (double) a1), (float) ((double) a2 * (double) a2)};   // (the second pair's slopes are a, a^2)
	auto chain = [&]<bool FULL> (int par, int nf) {
		const unsigned char* const src = cbuf + par * CBUF_B + lane * 16;
		float4 q1[F], q2[F];                                             // the maps do not depend on the state: all sixteen frames' reads go out first
#pragma unroll
		for (int f = 0; f < F; ++f) {
			q1[f] = *reinterpret_cast<const float4*> (src + (f * 2 + 0) * NCOL * 16);
			q2[f] = *reinterpret_cast<const float4*> (src + (f * 2 + 1) * NCOL * 16);
		}
#pragma unroll
		for (int f = 0; f < F; ++f) {
			if (FULL || f < nf) {                                            // wave-uniform: only the call's last chunk is short
				const v2f u0 = sl2[0] * zz, u1 = fma2 (sl2[1], zz, v2f{q1[f].x, q1[f].y}), u2 = fma2 (sl2[2], zz, v2f{q1[f].z, q1[f].w});
				const v2f zh = v2f{max3f (u0.x, u1.x, u2.x), max3f (u0.y, u1.y, u2.y)};
				const v2f w1_ = fma2 (AA, zh, v2f{q2[f].x, q2[f].y}), w2_ = fma2 (AA2, zh, v2f{q2[f].z, q2[f].w});
				zz = v2f{max3f (zh.x, w1_.x, w2_.x), max3f (zh.y, w1_.y, w2_.y)};
				zm = __builtin_fmaxf (zm, zz.x + zz.y);
			}
		}
	};
	constexpr int LAG = 1;                                               // chunks between the products and the chains
	// THE LOOP, once per role: every wave runs the same iterations and the same barriers, but each role's copy of the loop has
	// its own registers (in ONE loop with the roles as branches the compiler re-fetched the products' twelve tap fragments from
	// global memory in every iteration — the chains' thirty-two map registers were live across the same loop — and a block of
	// products took 800 cycles, more than half of them waiting for those loads).
	int slot_w = 32, slot_p = F % RING;                                  // window start of chunk t; where chunk t + 1 goes
	const int64_t n_it = n_chunks + LAG;
	// iterations [t0, t1) of the loop (the parity of an iteration is a compile-time constant in `work`: the buffers it selects)
	auto run_range = [&]<bool SPLITTER> (int64_t t0, int64_t t1, auto&& work) __attribute__ ((always_inline)) {
		auto iteration = [&]<int PAR> (int64_t t) __attribute__ ((always_inline)) {
			// a column's scale moved when chunk t was split (last iteration): its older slots and its un follow now, before this
			// iteration's products read them — the cold path, with its own barrier (the flag is uniform: every wave reads it here,
			// and wave 3 clears it only behind that barrier)
			const int moved = __builtin_amdgcn_readfirstlane (flag_sh[PAR]);
			if (__builtin_expect (moved != 0, 0)) {
				if constexpr (SPLITTER) rescale (slot_w + 48 >= RING ? slot_w + 48 - RING : slot_w + 48);   // chunk t's own slot = the window's last
				__syncthreads ();
				if (SPLITTER && lane == 0) flag_sh[PAR] = 0;
			}
			PROF_NOW (c0_);
			work.template operator()<PAR> (t, slot_w, slot_p);
			PROF_NOW (c1_);
			__syncthreads ();
			PROF_NOW (c2_);
			PROF_ADD (0, c1_ - c0_); PROF_ADD (2, c2_ - c1_); PROF_ADD (3, c2_ - c0_);
			slot_w = slot_w + F >= RING ? slot_w + F - RING : slot_w + F;
			slot_p = slot_p + F >= RING ? slot_p + F - RING : slot_p + F;
		};
		for (int64_t t = t0; t < t1; ++t) {
			if (t & 1) iteration.template operator()<1> (t);
			else       iteration.template operator()<0> (t);
		}
	};
	auto run = [&]<bool SPLITTER> (auto&& work) __attribute__ ((always_inline)) { run_range.template operator()<SPLITTER> (0, n_it, work); };
	if (wid == 0) {
		run.template operator()<false> ([&]<int PAR> (int64_t t, int, int) __attribute__ ((always_inline)) {
			if (t >= LAG && !MTR_TPB_DBG_NOCHAIN) {
				const int64_t left = n_frames - (t - LAG) * F;
				if (left >= F) chain.template operator()<true> (PAR ^ (LAG & 1), F);
				else chain.template operator()<false> (PAR ^ (LAG & 1), (int) left);
			}
		});
	} else if (wid == 3) {
		// Two loops.  The first serves the whole chunks that came by LDS-DMA, and holds NO vector-memory instruction the compiler
		// knows of: any such load makes it count vmcnt, and its conservative waits (vmcnt (0) where paths join) would wait for the
		// DMA in flight as well.  The second takes over where chunk t + 1 is the call's ragged last one (or for the whole call,
		// when the streams do not start on 16 bytes) and drains the pipeline.
		const int64_t n_whole = n_frames / F;
		const int64_t t_dma = dma_ok && !MTR_TPB_DBG_NOFETCH ? (n_whole > 1 ? n_whole - 1 : 0) : 0;     // chunks 1 .. n_whole - 1 are staged
		run_range.template operator()<true> (0, t_dma, [&]<int PAR> (int64_t t, int, int slot_p) __attribute__ ((always_inline)) {
			// chunk t + 3 leaves HBM; chunk t + 1, sent three iterations ago, is split from the staging area into both rings
			const int sb1 = (int) ((t + 1) % 3);
			if ((t + 4) * F <= n_frames && !MTR_TPB_DBG_NODMA) dma (t + 3, sb1 == 0 ? 2 : sb1 - 1);   // (t + 3) % 3: the slot chunk t was read from, an iteration ago
			// chunk t + 1 has landed when at most the DMA instructions of the younger chunks are outstanding
			const int younger = ((t + 3) * F <= n_frames ? 1 : 0) + ((t + 4) * F <= n_frames ? 1 : 0);
			if (MTR_TPB_DBG_NODMA) { }
			else if (younger == 2) asm volatile ("s_waitcnt vmcnt(8)" ::: "memory");
			else if (younger == 1) asm volatile ("s_waitcnt vmcnt(4)" ::: "memory");
			else                   asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
			if (!MTR_TPB_DBG_NOSPLIT) split_staged (sb1, slot_p, PAR ^ 1);
		});
		run_range.template operator()<true> (t_dma, n_it, [&]<int PAR> (int64_t t, int, int slot_p) __attribute__ ((always_inline)) {
			if (t + 1 < n_chunks && !MTR_TPB_DBG_NOFETCH) {
				fetch_put_ragged (t + 1, slot_p);
				wave_sync ();
				split (slot_p, PAR ^ 1);
			}
		});
	} else {
		auto run_unit = [&]<bool UB> (int b) __attribute__ ((always_inline)) {
			run.template operator()<false> ([&]<int PAR> (int64_t t, int slot_w, int) __attribute__ ((always_inline)) {
				if (t < n_chunks && !MTR_TPB_DBG_NOPROD) {
					const int64_t left = n_frames - t * F - 4 * kg;             // this lane's frames are 4 kg .. 4 kg + 3 of the chunk
					unit.template operator()<UB> (PAR, slot_w, b, left >= 4 ? 4 : (left > 0 ? (int) left : 0));
				}
			});
		};
		if (unit_a >= 0) run_unit.template operator()<false> (unit_a);
		else if (unit_b >= 0) run_unit.template operator()<true> (unit_b);
		else run.template operator()<false> ([&]<int PAR> (int64_t, int, int) __attribute__ ((always_inline)) { });   // (idle: keeps the barriers' count)
	}
	z1 = zz.x; z2 = zz.y;
#ifdef MTR_TPB_PROF
	if (blockIdx.x == 0 && lane == 0) for (int i = 0; i < 4; ++i) g_tpb_prof[wid][i] = pr[i];
#endif

	// the raw peaks per column (non-negative floats order as unsigned ints)
	uint32_t* const pk_sh = reinterpret_cast<uint32_t*> (ring);          // the ring is spent
	if (wid == 0) pk_sh[lane] = 0u;
	__syncthreads ();
	if (prod_wave) atomicMax (&pk_sh[16 * my_block + cc], __float_as_uint (pk[0]));
	__syncthreads ();
	if (wid == 0 && owner) {
		st->tpb_z1[ch] = z1 + 1e-20f;                                    // truepeakdsp.cc:86-87
		st->tpb_z2[ch] = z2 + 1e-20f;
		st->tpb_m[ch] = zm * a.g;                                        // :89, then read (m, p)
		st->tpb_p[ch] = __uint_as_float (pk_sh[lane]);
		if (C == 1) { st->tpb_z1[1] = 1e-20f; st->tpb_z2[1] = 1e-20f; st->tpb_m[1] = 0.f; st->tpb_p[1] = 0.f; }   // mono engines keep a zero right channel
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
