// mtr_seg.hip — K-weighting + true peak with LANE = TIME SEGMENT (gfx950), layout 7: the batch path.
//
// Replaces, for a whole batch, Ebu_r128_proc::detect_process (ebumeter/ebu_r128_proc.cc:302-337) and
// Resampler::process + TruePeakdsp::process_max (zita-resampler/resampler.cc:211-235, jmeters/truepeakdsp.cc:101-124),
// like k_kwtp16 (mtr_fused4.hip), whose arithmetic it shares: the K-weighting step is the reference's recurrence in f32,
// the interpolator runs on v_mfma_f32_16x16x32_f16 with samples and taps split into two f16 halves and three of the
// four partial products kept (mtr_mfma16_fir.h).
//
// What is different is the decomposition.  k_kwtp16 gives a wave ONE stream and its 64 lanes consecutive 38-frame runs
// of one 50 ms tile: the serial K-filter then needs an end-state pass, a DPP scan and a masked second pass (22 packed
// instructions per frame instead of the recurrence's 11), every tile has serial phases (scale reduction, split, pass 1,
// scan, pass 2, products), and its two waves per SIMD take turns (tools/f4_census.py: 1353 VALU + 342 MFMA + 361 SALU per
// tile).  Here a lane owns a whole TIME SEGMENT of a stream (the 64 lanes of a wave = 64 (stream, segment) units) and
// walks it 16 frames per step:
//   * the K-filter is the plain recurrence, 11 packed instructions per frame, state in registers; a segment that does
//     not start the call is warmed up over the 0.075 s in front of it (slowest pole 0.99502 per sample: 1.6e-8;
//     k_kwtp16 warms its own time segments the same way), segment 0 starts from the carried state;
//   * the 16 new frames of a lane are ONE column of the block-Toeplitz product: rows = the 16 outputs, window = the
//     lane's last 64 samples, which sit in a four-slot ring in LDS (f16 hi / lo words, 128 bytes per column and
//     array, 16-byte chunks XOR-swizzled by the column: conflict-free ds_read_b128 / ds_write_b128).  One step = 64 columns = 4 blocks x 2 channels x 18 MFMAs;
//   * the scale of a column is its lane's own: a power of two that puts the segment's running maximum into
//     [2^3, 2^15) — it only ever shrinks, and when it must (a sample 2^12 above what the scale was made for) the lane's
//     ring words are rescaled in place (exact: a power of two) and the peaks so far leave the scaled domain.  An Inf
//     or NaN sample poisons only the columns it reaches;
//   * no tile phases: every step is the same code, and the products of step j - 1 run UNDER the scalar and packed work
//     of step j in one instruction stream (one wave per SIMD, all the registers, 38 KB of LDS) — the MFMA shadows
//     carry the split, the maxima and a quarter of the recurrence (tools/issue_model.hip: two independent full-rate or
//     one half-rate VALU instruction per 16x16x32 MFMA cost 2 cycles; a packed-f32 one waits for the matrix pipe), the
//     rest of the recurrence is one packed block per step, software-pipelined over frames;
//   * loads: lane l reads its own 128 bytes per step (8 x global_load_dwordx4, one cache line), three steps ahead.
//
// The launch covers the whole 50 ms tiles of a call, n_main per lane; the rest of a fragment the call started in (`head`
// frames in front of the first tile) and what is left behind the last whole tile go to k_kwtp16 in the same stream order
// (mtr_engine.hip).  A tile need not be a whole number of steps (44.1 / 88.2 kHz): see ALIGNED below.  Σ y² per tile leaves through tile_power exactly as from k_kwtp16, so k_gate does
// not know which kernel ran.
#include <hip/hip_runtime.h>

#include <utility>

#include "mtr_internal.h"
#include "mtr_mfma16_fir.h"
#include "mtr_wave.h"

// -DMTR_SEG_PROF: shader cycles per part of a step, summed over one wave's main loop (tools/seg_prof.py)
#ifdef MTR_SEG_PROF
__device__ unsigned long long g_seg_prof[8];
#define SPROF_NOW(v) unsigned long long v; asm volatile ("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
#define SPROF_ADD(i, d) sprof_[i] += (d)
#else
#define SPROF_NOW(v)
#define SPROF_ADD(i, d)
#endif

namespace {

constexpr int R     = MTR_SEG_STEP;          // frames per lane and step = one MFMA column
constexpr int COLB  = 128;                   // bytes per column and array: 4 ring slots x 32; the eight 16-byte chunks of a column sit at
                                             // chunk ^ (column & 7): conflict-free in ds_read_b128's 16-lane groups and ds_write_b128's 8-lane groups
constexpr int ARRB  = 64 * COLB;             // one array: HL | HR | LL | LR
constexpr int BLKB  = 16 * COLB;             // 16 columns = one MFMA block
constexpr int XCHG  = 4 * ARRB;              // exchange area: float [2 ch][4 blocks][4 kg][16 c]
constexpr int LDS_BYTES = XCHG + 2048;

typedef unsigned char lds_u8;          // (generic pointers into the dynamic LDS block: the compiler infers the address space)

__device__ __forceinline__ float max3abs (float m, float a, float b) { return fmaxf (fmaxf (m, fabsf (a)), fabsf (b)); }
__device__ __forceinline__ v2f fma2 (v2f a, v2f b, v2f c) { return __builtin_elementwise_fma (a, b, c); }
__device__ __forceinline__ v2f scrub (v2f v) { return v2f{isfinite (v.x) ? v.x : 0.f, isfinite (v.y) ? v.y : 0.f}; }

// The scale of one channel of one lane.  A running maximum with exponent field e goes to [2^3, 2^4); the scale stands
// until a sample reaches 2^15 under it (cap = the bit pattern of that sample: non-negative floats order as uints, an
// Inf sets cap above every pattern).
struct Scale {
	float sc, un;          // scale, 2^-15 / scale
	uint32_t cap;
	__device__ __forceinline__ void set (float mx)
	{
		const int e = (int) (__float_as_uint (mx) >> 23);
		const int se = min (238, 257 - e);
		sc = __uint_as_float ((uint32_t) se << 23);
		un = __uint_as_float ((uint32_t) (239 - se) << 23);
		cap = (uint32_t) (269 - se) << 23;
	}
};

struct KCoef { v2f a0, a1, a2, b1, b2, c3, c4, eps; };
struct KState { v2f z1, z2, z3, z4, sj; };

// ebu_r128_proc.cc:321-327 for both channels of a frame; the association of mtr_kw_steps.h's hand-scheduled pair
__device__ __forceinline__ void kstep (const KCoef& k, KState& s, v2f p)
{
	v2f t = p + k.eps;
	v2f u = k.a1 * s.z1;
	t = fma2 (-k.b2, s.z2, t);
	u = fma2 (k.a2, s.z2, u);
	const v2f x = fma2 (-k.b1, s.z1, t);
	u = fma2 (-k.c4, s.z4, u);
	s.z4 = s.z4 + s.z3;
	u = fma2 (-k.c3, s.z3, u);
	const v2f y = fma2 (k.a0, x, u);
	s.z3 = s.z3 + y;
	s.sj = fma2 (y, y, s.sj);
	s.z2 = s.z1; s.z1 = x;
}

// The recurrence of one step — 16 frames x 11 operations — as ONE sequence for the hand-placed schedule of the main loop,
// software-pipelined over frames: the x-chain of frame n + 1 (operations 0..4: it only needs x (n), x (n - 1)) runs between the
// y-chain of frame n (5..10), so that no operation reads the result of one of its two predecessors (a dependent packed
// operation issues after 8 cycles, an independent one after 5: tools/issue_model.hip).  Same operations, same operands,
// same association as kstep — only their order in the instruction stream differs.
//     0: t = p + eps         1: u = a1 z1          2: t -= b2 z2        3: u += a2 z2       4: x = t - b1 z1
//     5: u -= c4 z4          6: z4 += z3           7: u -= c3 z3        8: y = a0 x + u     9: z3 += y      10: sj += y y
constexpr int KOPS = 11 * R;
struct KSeq {
	int frame[KOPS], op[KOPS];
	constexpr KSeq () : frame{}, op{}
	{
		int n = 0;
		for (int o = 0; o < 5; ++o) { frame[n] = 0; op[n] = o; ++n; }
		constexpr int ord[11][2] = { {5, 0}, {0, 1}, {6, 0}, {7, 0}, {1, 1}, {2, 1}, {8, 0}, {3, 1}, {4, 1}, {9, 0}, {10, 0} };
		for (int f = 0; f < R; ++f)
			for (int i = 0; i < 11; ++i) {
				if (ord[i][1] && f + 1 == R) continue;
				frame[n] = f + ord[i][1]; op[n] = ord[i][0]; ++n;
			}
	}
};
constexpr KSeq kseq_tab{};
struct KWork { v2f t[R], u[R], y[R], x[R + 2]; };         // x[f + 2] = x of frame f; x[0], x[1] = z2, z1 carried into the step
template <int OP>
__device__ __forceinline__ void kopx (const KCoef& k, KState& s, KWork& w, const v2f (&p)[R], int f)
{
	if constexpr (OP == 0) w.t[f] = p[f] + k.eps;
	else if constexpr (OP == 1) w.u[f] = k.a1 * w.x[f + 1];
	else if constexpr (OP == 2) w.t[f] = fma2 (-k.b2, w.x[f], w.t[f]);
	else if constexpr (OP == 3) w.u[f] = fma2 (k.a2, w.x[f], w.u[f]);
	else if constexpr (OP == 4) w.x[f + 2] = fma2 (-k.b1, w.x[f + 1], w.t[f]);
	else if constexpr (OP == 5) w.u[f] = fma2 (-k.c4, s.z4, w.u[f]);
	else if constexpr (OP == 6) s.z4 = s.z4 + s.z3;
	else if constexpr (OP == 7) w.u[f] = fma2 (-k.c3, s.z3, w.u[f]);
	else if constexpr (OP == 8) w.y[f] = fma2 (k.a0, w.x[f + 2], w.u[f]);
	else if constexpr (OP == 9) s.z3 = s.z3 + w.y[f];
	else s.sj = fma2 (w.y[f], w.y[f], s.sj);
}

// The lo word of the f16 split (mtr_mfma16_fir.h: lo_pair) as two separate instructions, so that the two that share an MFMA's
// shadow are independent (the halves of one word are not: v_fma_mixhi keeps the word's low half).
__device__ __forceinline__ void lo_first (uint32_t& lw, uint32_t hw, float x0)
{
	asm ("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lw) : "v"(hw), "v"(x0));
}
__device__ __forceinline__ void lo_second (uint32_t& lw, uint32_t hw, float x1)
{
	asm ("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lw) : "v"(hw), "v"(x1));
}

// ALIGNED: the tile (a 50 ms fragment) is a whole number of steps — 48, 96, 192, 32 kHz; otherwise (44.1, 88.2 kHz: 2205, 4410 frames)
// the step in which a tile ends is followed by a second run of its recurrence, frame by frame from the state the step started with,
// with the reference's end-of-fragment actions at the exact frame (every other step is the aligned kernel's code).
template <bool EBU, bool ALIGNED>
__global__ __launch_bounds__ (64, 1) void k_seg (const mtr_seg_args a)
{
	extern __shared__ __attribute__ ((aligned (16))) unsigned char smem_[];
	lds_u8* const smem = smem_;
	const int lane = threadIdx.x;

	// ---- which (stream, segment) this lane owns --------------------------------------------------------------------------
	const uint32_t n_units = a.n_streams * a.n_segs;
	uint32_t unit = blockIdx.x * 64u + (uint32_t) lane;
	const bool live = unit < n_units;                               // lanes past the batch shadow the last unit, silently
	if (!live) unit = n_units - 1;
	const uint32_t s = unit / a.n_segs, q = unit - s * a.n_segs;
	const uint32_t fq = q * a.seg_base + min (q, a.seg_rem);          // first tile this lane answers for
	const uint32_t cq = a.seg_base + (q < a.seg_rem ? 1u : 0u);
	const uint32_t p0 = fq + cq - a.n_main;                           // first tile it processes (one early where its segment is short)
	const int64_t F0 = (int64_t) p0 * a.tile_frames;
	const v2f* const src = reinterpret_cast<const v2f*> (a.audio) + (size_t) s * a.stride + a.head;   // frame 0 = the launch's first tile
	mtr_stream_state* const st = a.state + s;
	const int n_steps = (int) (((uint64_t) a.n_main * a.tile_frames + R - 1) / R);   // (ALIGNED: n_main tiles of spt steps)
	const int spt = (int) (a.tile_frames / R);
	// Unaligned tiles: a lane's last tile ends last_rows frames into the launch's last step.  What follows is the next segment's
	// (read again there: harmless for a maximum, and the recurrence is rerun to the exact frame, below) — but behind a stream's
	// LAST segment it is the call's tail, another stream, or nobody's memory: those lanes never read it.  The stream pointer
	// (of every lane: the main loop stays uniform) stops one step early, the last step's frames are fetched one by one in front
	// of it — zeros behind the tile's end in a last segment — and the rows of that step's products behind the end do not count.
	const int last_rows = ALIGNED ? R : (int) ((uint64_t) a.n_main * a.tile_frames - (uint64_t) (n_steps - 1) * R);   // 1 .. 16, wave-uniform
	const bool cut = last_rows < R;
	const bool lastq = !ALIGNED && q == a.n_segs - 1;
	const int n_loads = n_steps - (cut ? 1 : 0);                      // steps the stream pointers walk

	KCoef kc;
	kc.a0 = a.a0; kc.a1 = a.a1; kc.a2 = a.a2; kc.b1 = a.b1; kc.b2 = a.b2; kc.c3 = a.c3; kc.c4 = a.c4; kc.eps = 1e-15f;
	KState ks;
	ks.z1 = 0; ks.z2 = 0; ks.z3 = 0; ks.z4 = 0; ks.sj = 0;
	if (EBU && q == 0) {
		ks.z1 = v2f{st->kz[0], st->kz[1]}; ks.z2 = v2f{st->kz[2], st->kz[3]};
		ks.z3 = v2f{st->kz[4], st->kz[5]}; ks.z4 = v2f{st->kz[6], st->kz[7]};
	}

	// ---- the stream: four register buffers of 16 frames, loaded three steps ahead ------------------------------------------
	const bool warm = EBU && q > 0;
	typedef float f4v_ __attribute__ ((ext_vector_type (4)));
	typedef f4v_ float4_a8 __attribute__ ((aligned (8)));              // (a segment may start on any frame: 8-byte alignment is all there is)
	const float4_a8* lp = reinterpret_cast<const float4_a8*> (src + F0 - (warm ? (int64_t) a.warm_steps * R : 0));
	v2f xq[4][R];
	auto load = [&]<int B> () __attribute__ ((always_inline)) {
#pragma unroll
		for (int i = 0; i < R / 2; ++i) {
			const f4v_ v = lp[i];        // (plain loads: the eight 16-byte reads of a lane's line merge in the L1 — as nt loads they go to the L2 one by one: 17.4 ms)
			xq[B][2 * i] = v2f{v.x, v.y}; xq[B][2 * i + 1] = v2f{v.z, v.w};
		}
	};
	load.template operator()<0> (); lp += R / 2;
	load.template operator()<1> (); lp += R / 2;
	load.template operator()<2> (); lp += R / 2;

	m16::AFrag A;
	A.load (a.mfma_a, lane);
	// the twelve tap fragments live in accumulation registers for the whole kernel: the matrix pipe reads them there, and the
	// 48 architectural registers they would take are what the step's working set needs (otherwise the register allocator
	// parks some fragments in AGPRs anyway and copies one back per chunk: 32 v_accvgpr_read per step; measured -1.5 %)
#pragma unroll
	for (int f = 0; f < MTR_M16_FRAGS; ++f) asm volatile ("" : "+a"(A.a[f]));

	// ---- K-filter warm-up of the segments that do not start the call (warm_steps is a multiple of 4) -------------------------
	if (warm) {
		for (uint32_t w = 0; w < a.warm_steps; w += 4) {
			/* (the recurrence in the main loop's software-pipelined order — the x-chain of frame n + 1 between the y-chain of frame n: \
			 * the same operations on the same operands, 5 cycles apiece instead of the 8 a dependent packed pair takes) */ \
#define MTR_WARM(B)                                                              \
			{                                                                        \
				KWork kw_;                                                           \
				kw_.x[0] = ks.z2; kw_.x[1] = ks.z1;                                  \
				[&]<int... Is> (std::integer_sequence<int, Is...>) __attribute__ ((always_inline)) { \
					(kopx<kseq_tab.op[Is]> (kc, ks, kw_, xq[B], kseq_tab.frame[Is]), ...); \
				} (std::make_integer_sequence<int, KOPS>{});                         \
				ks.z2 = kw_.x[R]; ks.z1 = kw_.x[R + 1];                              \
				load.template operator()<(B + 3) & 3> (); lp += R / 2;                \
			}
			MTR_WARM (0) MTR_WARM (1) MTR_WARM (2) MTR_WARM (3)
#undef MTR_WARM
		}
		ks.z1 = scrub (ks.z1); ks.z2 = scrub (ks.z2); ks.z3 = scrub (ks.z3); ks.z4 = scrub (ks.z4);
		ks.sj = 0;
	}

	// ---- the ring: 48 frames in front of the segment (history of the previous call for segment 0) ---------------------------
	const int cc = lane & 15, kg = lane >> 4;
	int RA[4];                                                        // operand read address of window quarter v, this lane
#pragma unroll
	for (int v = 0; v < 4; ++v) RA[v] = cc * COLB + ((((kg & 1) + 2 * ((v + (kg >> 1)) & 3)) ^ (cc & 7)) * 16);
	const int WA = lane * COLB;                                       // this lane's column
	int WS[4];                                                        // ... and where the first half of ring slot s sits in it (the second: ^ 16)
#pragma unroll
	for (int sl = 0; sl < 4; ++sl) WS[sl] = WA + (((2 * sl) ^ (lane & 7)) * 16);

	Scale scl, scr;
	v2f pk0 = v2f{0.f, 0.f};                                           // phase 0: max |x[n - 24]|, exact
	v2f pkf = v2f{0.f, 0.f};                                           // interpolated peaks that have left the scaled domain
	float pm[4][2][2];                                                // running |max| of the accumulators, scaled, per block and channel: two
	                                                                  // independent chains each (two dependent v_max3 in one MFMA's shadow cost 3 cycles)
#pragma unroll
	for (int b = 0; b < 4; ++b) { pm[b][0][0] = 0.f; pm[b][0][1] = 0.f; pm[b][1][0] = 0.f; pm[b][1][1] = 0.f; }

	auto split_store = [&] (const v2f (&x)[R], int slot) __attribute__ ((always_inline)) {
		const v2f sc = v2f{scl.sc, scr.sc};
		uint32_t hl[R / 2], hr[R / 2], ll[R / 2], lr[R / 2];
#pragma unroll
		for (int i = 0; i < R / 2; ++i) {
			const v2f u = x[2 * i] * sc, v = x[2 * i + 1] * sc;
			m16::split_pair (u.x, v.x, hl[i], ll[i]);
			m16::split_pair (u.y, v.y, hr[i], lr[i]);
		}
		lds_u8* const w0 = smem + WS[slot];
		lds_u8* const w1 = smem + (WS[slot] ^ 16);
		*reinterpret_cast<uint4*> (w0)            = uint4{hl[0], hl[1], hl[2], hl[3]};
		*reinterpret_cast<uint4*> (w1)            = uint4{hl[4], hl[5], hl[6], hl[7]};
		*reinterpret_cast<uint4*> (w0 + ARRB)     = uint4{hr[0], hr[1], hr[2], hr[3]};
		*reinterpret_cast<uint4*> (w1 + ARRB)     = uint4{hr[4], hr[5], hr[6], hr[7]};
		*reinterpret_cast<uint4*> (w0 + 2 * ARRB) = uint4{ll[0], ll[1], ll[2], ll[3]};
		*reinterpret_cast<uint4*> (w1 + 2 * ARRB) = uint4{ll[4], ll[5], ll[6], ll[7]};
		*reinterpret_cast<uint4*> (w0 + 3 * ARRB) = uint4{lr[0], lr[1], lr[2], lr[3]};
		*reinterpret_cast<uint4*> (w1 + 3 * ARRB) = uint4{lr[4], lr[5], lr[6], lr[7]};
	};

	{
		const v2f* const hst = reinterpret_cast<const v2f*> (a.hist) + (size_t) s * MTR_FIR_HALO;
		v2f px[3][R];
#pragma unroll
		for (int k = 0; k < 3; ++k)
#pragma unroll
			for (int n = 0; n < R; ++n) {
				const int64_t f = F0 - 48 + R * k + n;
				const int64_t fc = f + (int64_t) a.head;                     // the call's own frame; in front of it: the history
				px[k][n] = fc >= 0 ? src[f] : (fc >= -MTR_FIR_HALO ? hst[fc + MTR_FIR_HALO] : v2f{0.f, 0.f});
			}
		float ml = 0.f, mr = 0.f;
#pragma unroll
		for (int k = 0; k < 3; ++k)
#pragma unroll
			for (int n = 0; n < R; n += 2) { ml = max3abs (ml, px[k][n].x, px[k][n + 1].x); mr = max3abs (mr, px[k][n].y, px[k][n + 1].y); }
		scl.set (ml); scr.set (mr);
		split_store (px[0], 1); split_store (px[1], 2); split_store (px[2], 3);
		if (F0 == 0) {
			// phase 0 of the launch's first outputs is frames -24 .. -1: the history of a launch that starts the call, else the
			// call's own frames in front of it (whose owner, the kernel of the call's head, stops 24 frames short of them)
#pragma unroll
			for (int n = 8; n < R; ++n) pk0 = v2f{fmaxf (pk0.x, fabsf (px[1][n].x)), fmaxf (pk0.y, fabsf (px[1][n].y))};
#pragma unroll
			for (int n = 0; n < R; ++n) pk0 = v2f{fmaxf (pk0.x, fabsf (px[2][n].x)), fmaxf (pk0.y, fabsf (px[2][n].y))};
		}
	}

	// ---- pieces of a step ---------------------------------------------------------------------------------------------------
	// the accumulators' running maxima leave the scaled domain: pm (accumulator layout: lane (c, kg), block b = column
	// 16 b + c) -> pkf of the lane that owns the column, through the exchange area
	auto flush_pm = [&] () __attribute__ ((always_inline)) {
		float* const X = reinterpret_cast<float*> (smem_ + XCHG);
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			X[((0 * 4 + b) * 4 + kg) * 16 + cc] = fmaxf (pm[b][0][0], pm[b][0][1]);
			X[((1 * 4 + b) * 4 + kg) * 16 + cc] = fmaxf (pm[b][1][0], pm[b][1][1]);
			pm[b][0][0] = 0.f; pm[b][0][1] = 0.f; pm[b][1][0] = 0.f; pm[b][1][1] = 0.f;
		}
		__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier ();
		__builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
		float fl = 0.f, fr = 0.f;
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			fl = fmaxf (fl, X[((0 * 4 + kg) * 4 + g) * 16 + cc]);         // column `lane` = block lane >> 4 (= kg), c = lane & 15
			fr = fmaxf (fr, X[((1 * 4 + kg) * 4 + g) * 16 + cc]);
		}
		pkf = v2f{fmaxf (pkf.x, fl * scl.un), fmaxf (pkf.y, fr * scr.un)};
		__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier ();
	};

	// a sample has outgrown a lane's scale: peaks out of the scaled domain, new scales, the lane's ring words rescaled
	// in place by the (power-of-two) ratio — every word of the four slots; the slot about to be overwritten included
	auto rescale = [&] (float ml, float mr) __attribute__ ((always_inline)) {
		flush_pm ();
		const bool el = __float_as_uint (ml) >= scl.cap, er = __float_as_uint (mr) >= scr.cap;
		const float ol = scl.sc, orr = scr.sc;
		if (el) scl.set (ml);
		if (er) scr.set (mr);
		if (el || er) {
			typedef _Float16 h2v __attribute__ ((ext_vector_type (2)));
			const _Float16 rl = (_Float16) (scl.sc / ol), rr = (_Float16) (scr.sc / orr);      // 1 or 2^-12 and below (0 under 2^-24)
#pragma unroll 1
			for (int arr = 0; arr < 4; ++arr) {
				const _Float16 r = (arr & 1) ? rr : rl;
				const h2v r2 = h2v{r, r};
#pragma unroll 1
				for (int i = 0; i < 8; ++i) {
					uint4* const p = reinterpret_cast<uint4*> (smem + arr * ARRB + WA + 16 * i);
					uint4 v = *p;
					v.x = __builtin_bit_cast (uint32_t, (h2v) (__builtin_bit_cast (h2v, v.x) * r2));
					v.y = __builtin_bit_cast (uint32_t, (h2v) (__builtin_bit_cast (h2v, v.y) * r2));
					v.z = __builtin_bit_cast (uint32_t, (h2v) (__builtin_bit_cast (h2v, v.z) * r2));
					v.w = __builtin_bit_cast (uint32_t, (h2v) (__builtin_bit_cast (h2v, v.w) * r2));
					*p = v;
				}
			}
		}
		__builtin_amdgcn_fence (__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier ();
	};

	// Operands of (block, channel) bc for the products of the step whose window ends in ring slot (U + 3) & 3: window
	// quarter v sits in slot (U + v) & 3.
	m16::BFrag B0, B1;
	m16::f4 y0[3], y1[3];
#pragma unroll
	for (int p = 0; p < 3; ++p) { y0[p] = m16::f4{0.f, 0.f, 0.f, 0.f}; y1[p] = m16::f4{0.f, 0.f, 0.f, 0.f}; }
	auto fetch = [&]<int U> (m16::BFrag& B, int bc) __attribute__ ((always_inline)) {
		const int b = bc >> 1, ch = bc & 1;
		const lds_u8* const h = smem + ch * ARRB + b * BLKB;
		const lds_u8* const l = h + 2 * ARRB;
		B.h0 = *reinterpret_cast<const uint4*> (h + RA[U]);
		B.l0 = *reinterpret_cast<const uint4*> (l + RA[U]);
		B.h1 = *reinterpret_cast<const uint4*> (h + RA[(U + 2) & 3]);
		B.l1 = *reinterpret_cast<const uint4*> (l + RA[(U + 2) & 3]);
	};
	// |max| of the accumulators of (block, channel) bc into pm
	auto fold = [&] (const m16::f4 (&y)[3], int bc) __attribute__ ((always_inline)) {
		float ma = pm[bc >> 1][bc & 1][0], mb = pm[bc >> 1][bc & 1][1];
#pragma unroll
		for (int p = 0; p < 3; ++p) { ma = max3abs (ma, y[p][0], y[p][1]); mb = max3abs (mb, y[p][2], y[p][3]); }
		pm[bc >> 1][bc & 1][0] = ma; pm[bc >> 1][bc & 1][1] = mb;
	};
	// ... of the launch's LAST step: lane (c, kg) holds rows (= frames of the step) 4 kg .. 4 kg + 3 of column 16 b + c; where that
	// column is the last segment of its stream, only the rows in front of its tile's end count
	auto fold_last = [&] (const m16::f4 (&y)[3], int bc) __attribute__ ((always_inline)) {
		if constexpr (ALIGNED) fold (y, bc);
		else {
			const uint32_t ucol = min (blockIdx.x * 64u + 16u * (uint32_t) (bc >> 1) + (uint32_t) cc, n_units - 1);
			const int rows = (ucol % a.n_segs == a.n_segs - 1) ? last_rows : R;
			float ma = pm[bc >> 1][bc & 1][0], mb = pm[bc >> 1][bc & 1][1];
#pragma unroll
			for (int p = 0; p < 3; ++p) {
				ma = max3abs (ma, 4 * kg + 0 < rows ? y[p][0] : 0.f, 4 * kg + 1 < rows ? y[p][1] : 0.f);
				mb = max3abs (mb, 4 * kg + 2 < rows ? y[p][2] : 0.f, 4 * kg + 3 < rows ? y[p][3] : 0.f);
			}
			pm[bc >> 1][bc & 1][0] = ma; pm[bc >> 1][bc & 1][1] = mb;
		}
	};
	// the products of the call's last step (nothing left to run under them); every chunk folds its predecessor's accumulators
	// (the first fold is still the step before's)
	auto products = [&]<int U> () __attribute__ ((always_inline)) {
		fetch.template operator()<U> (B0, 0);
#pragma unroll
		for (int bc = 0; bc < 8; bc += 2) {
			fetch.template operator()<U> (B1, bc + 1);
			m16::block (A, B0, y0);
			if (bc == 0) fold (y1, 7); else fold_last (y1, bc - 1);
			if (bc + 2 < 8) fetch.template operator()<U> (B0, bc + 2);
			m16::block (A, B1, y1);
			fold_last (y0, bc);
		}
		fold_last (y1, 7);
	};

	int tile_left = spt;
	uint32_t tile = p0;
	int j = 0;

	// max |x| per channel of buffer B: always computed one step early, under the products of the step before
	float ml = 0.f, mr = 0.f;
	auto maxabs = [&]<int B> () __attribute__ ((always_inline)) {
		ml = 0.f; mr = 0.f;
#pragma unroll
		for (int n = 0; n < R; n += 2) { ml = max3abs (ml, xq[B][n].x, xq[B][n + 1].x); mr = max3abs (mr, xq[B][n].y, xq[B][n + 1].y); }
	};

	// one step: the scalar / packed work of step j on buffer U (ring slot U), and — PROD — the products of step j - 1,
	// interleaved by hand (the source order is the schedule: see the chunks below)
#ifdef MTR_SEG_PROF
	unsigned long long sprof_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
	// KW: the recurrence rides in this step's schedule (EBU; not in the step of an unaligned tile's end)
	auto step = [&]<int U, bool PROD, bool KW> () __attribute__ ((always_inline)) {
		v2f (&x)[R] = xq[U];
		SPROF_NOW (c0_);
		// phase 0 = |x[n - 24]| for the frames of this call: everything but its last 24 frames
		if (F0 + (int64_t) R * (j + 1) <= a.p0_end) pk0 = v2f{fmaxf (pk0.x, ml), fmaxf (pk0.y, mr)};
		else {
			const int64_t lim = a.p0_end - F0 - (int64_t) R * j;
#pragma unroll
			for (int n = 0; n < R; ++n) if (n < lim) pk0 = v2f{fmaxf (pk0.x, fabsf (x[n].x)), fmaxf (pk0.y, fabsf (x[n].y))};
		}
		if (__builtin_expect (__ballot (__float_as_uint (ml) >= scl.cap || __float_as_uint (mr) >= scr.cap) != 0, 0)) {
			// (the last chunk's accumulators of the step before are still waiting for their fold, which chunk 0 does: they
			// belong to the old scale, so they are folded here, in front of the flush, and cleared)
			fold (y1, 7);
#pragma unroll
			for (int p = 0; p < 3; ++p) y1[p] = m16::f4{0.f, 0.f, 0.f, 0.f};
			rescale (ml, mr);
			if (PROD) fetch.template operator()<U> (B0, 0);               // (fetched before the ring was rescaled: again)
		}

		// the stream, three steps ahead (the pointer stops with the segment: the last loads re-read its last line)
#if !(defined (MTR_TIMING_ONLY_BUILD) && defined (MTR_SEG_DBG_NOADV))   /* (elimination runs of a timing-only library, wrong results: the same line over and over) */
		lp += (j + 3 < (ALIGNED ? n_steps : n_loads)) ? R / 2 : 0;
#endif
		load.template operator()<(U + 3) & 3> ();
		SPROF_NOW (c1_); SPROF_ADD (0, c1_ - c0_);

		// Eight chunks, one per (block, channel) of the products of step j - 1.  THE SOURCE ORDER IS THE SCHEDULE (this TU
		// is compiled without the machine schedulers, csrc/Makefile).  The issue model of one wave (tools/issue_model.hip,
		// profiles/r03_issue_model.txt), in shader cycles per MFMA:
		//     16x16x32 MFMA alone 17.2;  + 2 independent full-rate VALU (v_fma / v_mul / v_max3) 19.2;  + 3: 23.8;  + 2 DEPENDENT: 22.2
		//     + 1 half-rate VALU (v_fma_mix, v_cvt_pk, 8 cycles alone) 18.5;  + 2 of them 26.5;  mixlo + mixhi of ONE word 31.3
		//     + 1 v_pk_fma_f32: 34.3 (a packed-f32 instruction waits for the matrix pipe);  packed among themselves 5.1, 8.2 if
		//     the next one reads the last one's result;  + 1 ds_read_b128: + 6
		// So: behind every MFMA either two independent full-rate instructions (hipcc unpacks a packed one it finds there into its
		// two halves) or ONE half-rate instruction, and everything that does not fit — most of the recurrence — in ONE packed
		// block per step behind the last chunk (one wait for the matrix pipe per step instead of one per chunk):
		//   MFMA 0-1: the scale of two frames    2-3: f16 hi words (v_cvt_pk)    4-7: lo words (v_fma_mix, one each)
		//   8: the next step's maxima    9-11: |max| of the previous chunk's accumulators, two chains
		//   12-17: six operations of the recurrence's sequence
		const v2f sc2 = v2f{scl.sc, scr.sc};
		uint32_t hl[R / 2], hr[R / 2], ll[R / 2], lr[R / 2];
		float nl = 0.f, nr = 0.f;
		const v2f (&xn)[R] = xq[(U + 1) & 3];
		KWork kw;
		kw.x[0] = ks.z2; kw.x[1] = ks.z1;
		constexpr int KGAP = 6;                                     // operations of the recurrence behind the MFMAs of one chunk
		auto kseq = [&]<int I> () __attribute__ ((always_inline)) {
			if constexpr (KW && I < KOPS) kopx<kseq_tab.op[I]> (kc, ks, kw, x, kseq_tab.frame[I]);
		};
		auto chunk = [&]<int BC> (m16::BFrag& Bc, m16::BFrag& Bn, m16::f4 (&yc)[3], m16::f4 (&yp)[3]) __attribute__ ((always_inline)) {
			constexpr int PB = (BC + 7) & 7;
			const v2f xa = x[2 * BC], xb = x[2 * BC + 1];
			if (PROD && BC < 7) fetch.template operator()<U> (Bn, BC + 1);
#if defined (MTR_TIMING_ONLY_BUILD) && defined (MTR_SEG_DBG_NOPROD)    /* (elimination runs of a timing-only library, wrong results: no products) */
#define MTR_M(I)
#else
#define MTR_M(I) if (PROD) m16::block_mfma<I> (A, Bc, yc)
#endif
			MTR_M (0);  const v2f um = xa * sc2;
			MTR_M (1);  const v2f vm = xb * sc2;
			MTR_M (2);  hl[BC] = m16::hi_pair (um.x, vm.x);
			MTR_M (3);  hr[BC] = m16::hi_pair (um.y, vm.y);
			MTR_M (4);  lo_first (ll[BC], hl[BC], um.x);
			MTR_M (5);  lo_first (lr[BC], hr[BC], um.y);
			MTR_M (6);  lo_second (ll[BC], hl[BC], vm.x);
			MTR_M (7);  lo_second (lr[BC], hr[BC], vm.y);
			MTR_M (8);  nl = max3abs (nl, xn[2 * BC].x, xn[2 * BC + 1].x); nr = max3abs (nr, xn[2 * BC].y, xn[2 * BC + 1].y);
			float ma = pm[PB >> 1][PB & 1][0], mb = pm[PB >> 1][PB & 1][1];
			MTR_M (9);  if (PROD) { ma = max3abs (ma, yp[0][0], yp[0][1]); mb = max3abs (mb, yp[0][2], yp[0][3]); }
			MTR_M (10); if (PROD) { ma = max3abs (ma, yp[1][0], yp[1][1]); mb = max3abs (mb, yp[1][2], yp[1][3]); }
			MTR_M (11); if (PROD) { ma = max3abs (ma, yp[2][0], yp[2][1]); mb = max3abs (mb, yp[2][2], yp[2][3]); }
			if (PROD) asm volatile ("" : "+v"(ma), "+v"(mb));   // consumed here: the maxima must not sink behind the accumulators' next writers
			pm[PB >> 1][PB & 1][0] = ma; pm[PB >> 1][PB & 1][1] = mb;
			// The ring stores of this step ride in the last two chunks — behind the last operand fetch that still reads the
			// slot they overwrite (chunk 6 fetches chunk 7's) — and the next step's first operands are fetched behind them:
			// neither the stores' VGPR transfer nor the fetch's latency is left for the step's head and tail.
#define MTR_ST(ARR, W_) \
			if constexpr (BC == 6) *reinterpret_cast<uint4*> (smem + WS[U] + (ARR) * ARRB) = uint4{W_[0], W_[1], W_[2], W_[3]}; \
			if constexpr (BC == 7) *reinterpret_cast<uint4*> (smem + (WS[U] ^ 16) + (ARR) * ARRB) = uint4{W_[4], W_[5], W_[6], W_[7]}
			MTR_M (12); kseq.template operator()<KGAP * BC + 0> ();
			MTR_ST (0, hl);
			MTR_M (13); kseq.template operator()<KGAP * BC + 1> ();
			MTR_ST (1, hr);
			MTR_M (14); kseq.template operator()<KGAP * BC + 2> ();
			MTR_ST (2, ll);
			MTR_M (15); kseq.template operator()<KGAP * BC + 3> ();
			MTR_ST (3, lr);
			MTR_M (16); kseq.template operator()<KGAP * BC + 4> ();
			if constexpr (BC == 7) fetch.template operator()<(U + 1) & 3> (Bn, 0);      // (Bn of the last chunk = B0 of the next step)
			MTR_M (17); kseq.template operator()<KGAP * BC + 5> ();
#undef MTR_ST
#undef MTR_M
		};
		chunk.template operator()<0> (B0, B1, y0, y1);
		SPROF_NOW (c2_); SPROF_ADD (1, c2_ - c1_);
		chunk.template operator()<1> (B1, B0, y1, y0);
		chunk.template operator()<2> (B0, B1, y0, y1); chunk.template operator()<3> (B1, B0, y1, y0);
		chunk.template operator()<4> (B0, B1, y0, y1); chunk.template operator()<5> (B1, B0, y1, y0);
		chunk.template operator()<6> (B0, B1, y0, y1);
		SPROF_NOW (c3_); SPROF_ADD (2, c3_ - c2_);
		chunk.template operator()<7> (B1, B0, y1, y0);
		SPROF_NOW (c4_); SPROF_ADD (3, c4_ - c3_);
		// the rest of the recurrence's sequence: one packed block
		if (KW) {
			[&]<int... Is> (std::integer_sequence<int, Is...>) __attribute__ ((always_inline)) {
				(kseq.template operator()<8 * KGAP + Is> (), ...);
			} (std::make_integer_sequence<int, KOPS - 8 * KGAP>{});
			ks.z2 = kw.x[R]; ks.z1 = kw.x[R + 1];
		}
		asm volatile ("" : "+v"(nl), "+v"(nr));
		ml = nl; mr = nr;
		++j;
		if (KW && ALIGNED && --tile_left == 0) {
			if (live && tile >= fq) a.tile_power[(size_t) s * a.n_tiles + a.tile0 + tile] = a.gain_l * ks.sj.x + a.gain_r * ks.sj.y;
			ks.sj = 0;
			ks.z1 = scrub (ks.z1); ks.z2 = scrub (ks.z2); ks.z3 = scrub (ks.z3); ks.z4 = scrub (ks.z4);   // ebu_r128_proc.cc:331-334
			tile_left = spt; ++tile;
		}
		SPROF_NOW (c5_); SPROF_ADD (4, c5_ - c4_); SPROF_ADD (5, c5_ - c0_); SPROF_ADD (6, 1);
	};

	// An unaligned tile ends inside a step (the same step and frame for all lanes: every lane starts on a tile boundary).  That
	// step runs like every other one — the very code of the aligned kernel, products, split and recurrence — and then its
	// recurrence is run AGAIN from the state the step started with, frame by frame: Σ y² closes at the exact frame, the states
	// are scrubbed there (ebu_r128_proc.cc:331-334), and the lane's last tile end is where its filter state is final — the frames
	// of the step behind it belong to the next segment (or to k_kwtp16's tail of the call).  One step in ~138 pays for sixteen
	// dependent recurrence steps (+ 0.3 % of the launch); the other 137 carry no trace of the boundary (round 3 compiled a second
	// form of the step without the recurrence and sixteen copies of the tile end into the loop: 450 registers, 1700 accumulator
	// copies, + 1.9 % cycles on EVERY step).
	int frames_left = (int) a.tile_frames;                            // of the open tile, at the start of the next step
	uint32_t tiles_done = 0;
	KState kfin = ks;
	auto kslow = [&]<int U> () __attribute__ ((always_inline)) {
		const v2f (&x)[R] = xq[U];
		const int k = frames_left;                                    // 1 .. 16: frames of this step that belong to the open tile (wave-uniform)
#pragma unroll
		for (int n = 0; n < R; ++n) if (n < k) kstep (kc, ks, x[n]);
		if (live && tile >= fq) a.tile_power[(size_t) s * a.n_tiles + a.tile0 + tile] = a.gain_l * ks.sj.x + a.gain_r * ks.sj.y;
		ks.sj = 0;
		ks.z1 = scrub (ks.z1); ks.z2 = scrub (ks.z2); ks.z3 = scrub (ks.z3); ks.z4 = scrub (ks.z4);
		++tile;
		if (++tiles_done == a.n_main) kfin = ks;
#pragma unroll
		for (int n = 0; n < R; ++n) if (n >= k) kstep (kc, ks, x[n]);
		frames_left = (int) a.tile_frames - (R - k);
	};
	// the launch's last step where tiles are unaligned: the lanes of a last segment fetch their last_rows frames one by one
	auto last_frames = [&]<int U> () __attribute__ ((always_inline)) {
		const v2f* f = src + F0 + (int64_t) (n_steps - 1) * R;
		int rows = lastq ? last_rows : R;                             // frames of the step that are this lane's to read
		asm volatile ("" : "+v"(f), "+v"(rows));                      // (computed HERE, once: not sixteen addresses and masks kept across the main loop)
		v2f t[R];
#pragma unroll
		for (int n = 0; n < R; ++n) t[n] = f[min (n, rows - 1)];
#pragma unroll
		for (int n = 0; n < R; ++n) xq[U][n] = n < rows ? t[n] : v2f{0.f, 0.f};
		maxabs.template operator()<U> ();                             // (the maxima computed a step early saw a stale line)
	};
	auto do_step = [&]<int U, bool PROD> () __attribute__ ((always_inline)) {
		if constexpr (!ALIGNED) if (cut && j == n_steps - 1) last_frames.template operator()<U> ();
		if constexpr (!EBU || ALIGNED) step.template operator()<U, PROD, EBU> ();
		else {
			const bool boundary = frames_left <= R;                   // wave-uniform
			KState at_start = ks;
			step.template operator()<U, PROD, true> ();
			// (the step's recurrence is needed HERE, whichever way the branch below goes: without this the compiler sinks all 176
			// operations out of the MFMA shadows they were placed in, into the branch that uses them — one lump behind the step)
			asm volatile ("" : "+v"(ks.z1), "+v"(ks.z2), "+v"(ks.z3), "+v"(ks.z4), "+v"(ks.sj));
			if (boundary) { ks = at_start; kslow.template operator()<U> (); }
			else frames_left -= R;
		}
	};

	// (the first three loads above were steps 0..2; step 0 has no products in front of it)
	lp -= R / 2;                                                      // `step` advances before it loads
	maxabs.template operator()<0> ();
	do_step.template operator()<0, false> ();
	while (j + 4 <= n_steps) {
		do_step.template operator()<1, true> ();
		do_step.template operator()<2, true> ();
		do_step.template operator()<3, true> ();
		do_step.template operator()<0, true> ();
	}
	if (j < n_steps) do_step.template operator()<1, true> ();
	if (j < n_steps) do_step.template operator()<2, true> ();
	if (j < n_steps) do_step.template operator()<3, true> ();
	switch (n_steps & 3) {                                           // the products of the last step
	case 0:  products.template operator()<0> (); break;
	case 1:  products.template operator()<1> (); break;
	case 2:  products.template operator()<2> (); break;
	default: products.template operator()<3> (); break;
	}
	flush_pm ();
#ifdef MTR_SEG_PROF
	if (blockIdx.x == gridDim.x / 2 && lane == 0) for (int i = 0; i < 8; ++i) g_seg_prof[i] = sprof_[i];
#endif

	if (live) {
		atomicMax (&st->tp_call[0], __float_as_uint (fmaxf (pk0.x, pkf.x)));
		atomicMax (&st->tp_call[1], __float_as_uint (fmaxf (pk0.y, pkf.y)));
		if (EBU && q == a.n_segs - 1) {
			const KState& kf = ALIGNED ? ks : kfin;                       // (unaligned: the state at the lane's last tile end)
			st->kz[0] = kf.z1.x; st->kz[1] = kf.z1.y; st->kz[2] = kf.z2.x; st->kz[3] = kf.z2.y;
			st->kz[4] = kf.z3.x; st->kz[5] = kf.z3.y; st->kz[6] = kf.z4.x; st->kz[7] = kf.z4.y;
		}
	}
}

}  // namespace

size_t mtr_seg_lds_bytes (void) { return LDS_BYTES; }

#ifdef MTR_SEG_PROF
extern "C" int mtr_debug_seg_prof (unsigned long long* out)
{
	return hipMemcpyFromSymbol (out, HIP_SYMBOL (g_seg_prof), 8 * sizeof (unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int mtr_launch_seg (bool ebu, const mtr_seg_args& a, uint32_t n_waves, void* stream)
{
	hipStream_t st = (hipStream_t) stream;
	const bool aligned = a.tile_frames % R == 0;
	if (!ebu && aligned) hipLaunchKernelGGL ((k_seg<false, true>), dim3 (n_waves), dim3 (64), LDS_BYTES, st, a);   // (no tiles without Σ y²; the
	else if (!ebu)    hipLaunchKernelGGL ((k_seg<false, false>), dim3 (n_waves), dim3 (64), LDS_BYTES, st, a);   // same code but for the launch's last step)
	else if (aligned) hipLaunchKernelGGL ((k_seg<true, true>), dim3 (n_waves), dim3 (64), LDS_BYTES, st, a);
	else              hipLaunchKernelGGL ((k_seg<true, false>), dim3 (n_waves), dim3 (64), LDS_BYTES, st, a);
	return hipGetLastError () == hipSuccess ? 0 : -1;
}
