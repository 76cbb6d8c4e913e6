// tools/tpb_prof.hip — where does a chunk of k_tpb go?  Builds the kernel with cycle counters per wave
// (work / barrier wait / total) and prints them for workgroup 0.
// Build: tools/build_tpb_prof.sh (the default build and the elimination builds: a role switched off, -DMTR_TIMING_ONLY_BUILD — wrong
// results by construction, timing only)
#define MTR_TPB_PROF 1
#include "../meters.lv2_amd/csrc/mtr_tpb.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main (int argc, char** argv)
{
	const uint32_t S = argc > 1 ? atoi (argv[1]) : 8192;
	const uint64_t T = argc > 2 ? atoll (argv[2]) : 48000;
	const uint64_t ST = T + (argc > 3 ? atoll (argv[3]) : 0);          // stride in frames: does the streams' distance in memory matter?
	float *audio, *hist; uint16_t* afr; mtr_stream_state* st;
	hipMalloc (&audio, (size_t) S * ST * 8);
	hipMalloc (&hist, (size_t) S * MTR_FIR_HALO * 8);
	hipMalloc (&afr, MTR_M16_A_HALVES * 2);
	hipMalloc (&st, (size_t) S * sizeof (mtr_stream_state));
	hipMemset (hist, 0, (size_t) S * MTR_FIR_HALO * 8);
	hipMemset (st, 0, (size_t) S * sizeof (mtr_stream_state));
	std::vector<float> h ((size_t) 1 << 22);
	uint32_t r = 12345;
	for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((int) (r >> 8) - (1 << 23)) / 8388608.f; }
	for (size_t o = 0; o < (size_t) S * ST * 2; o += h.size ())
		hipMemcpy (audio + o, h.data (), std::min (h.size (), (size_t) S * ST * 2 - o) * 4, hipMemcpyHostToDevice);
	std::vector<float> taps (144, 0.01f);
	std::vector<uint16_t> a16 (MTR_M16_A_HALVES);
	mtr_m16_build_a (taps.data (), a16.data ());
	hipMemcpy (afr, a16.data (), a16.size () * 2, hipMemcpyHostToDevice);
	mtr_tpb_args a{};
	a.audio = audio; a.stride = ST; a.n_frames = T; a.hist = hist; a.mfma_a = afr; a.state = st;
	a.n_streams = S; a.n_channels = 2; a.w1 = 0.0208f; a.w2 = 0.0896f; a.w3 = 0.99996f; a.g = 0.502f;
	hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
	mtr_launch_tpb (a, nullptr);
	hipDeviceSynchronize ();
	hipEventRecord (e0);
	mtr_launch_tpb (a, nullptr);
	hipEventRecord (e1);
	hipDeviceSynchronize ();
	float ms; hipEventElapsedTime (&ms, e0, e1);
	unsigned long long pr[16][4] = {};
	hipMemcpyFromSymbol (pr, HIP_SYMBOL (g_tpb_prof), sizeof pr);
	const double nchunk = (double) ((T + F - 1) / F + 1);
	printf ("stride %llu frames (= 128 B x %.3f): ", (unsigned long long) ST, ST * 8 / 128.0);
	printf ("S=%u T=%llu: %.3f ms, %.0f ns per chunk of %d frames\n", S, (unsigned long long) T, ms, ms * 1e6 / nchunk, F);
	printf ("shader cycles per chunk:\n wave  work  barrier wait  total   (waves 0, 1: the chains of columns 0 .. 31 | 32 .. 63; 2: LDS-DMA + split of columns 0 .. 31; 3: split of 32 .. 63; 4 .. 7: unit A of blocks 0 .. 3 (phase 1 + first pair map); 10, 11, 14, 15: unit B (phases 2, 3 + second pair map; 8 .. 11 with MTR_TPB_NW=12); the rest idle; waves of equal w mod 4 share a SIMD)\n");
	for (int w = 0; w < NW && w < 16; ++w) printf ("  %2d  %7.1f  %7.1f  %7.1f\n", w, pr[w][0] / nchunk, pr[w][2] / nchunk, pr[w][3] / nchunk);
	return 0;
}
