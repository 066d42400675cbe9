// dev/gpu_ubench_scan.hip - what bounds the referee's sequential scan (kernels.h: ref_exact_window_dev)?  One wavefront runs the
// reference's recursion  y = r0 + (B1*y1 + B2*y2)  (src/demod.c:74-79, every product and sum rounded to float) over N steps, r0
// handed over with v_readlane, in several shapes; the shader clocks per step say whether the chain of three dependent operations
// is cheaper as packed (I, Q) arithmetic (round 5's form), as separate fp32 operations, or split over lanes / wavefronts.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dev/gpu_ubench_scan.hip -o /tmp/ubench_scan && /tmp/ubench_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lane_of(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template<int V>
__global__ __launch_bounds__(512) void k_scan(const float *in, float *out, int nblk, unsigned long long *clk) {
	#pragma clang fp contract(off)
	const int lane = threadIdx.x & 63; const bool w0_ = threadIdx.x < 64;
	float fa = in[lane], fb = in[64 + lane];
	const float b1 = in[128], b2 = in[129];
	const v2f B1 = v2f{b1, b1}, B2 = v2f{b2, b2};
	v2f y1 = v2f{0.f, 0.f}, y2 = v2f{0.f, 0.f};
	float yi1 = 0.f, yi2 = 0.f, yq1 = 0.f, yq2 = 0.f;
	const bool odd = lane & 1;
	const unsigned long long t0 = clock64();
	const unsigned long long w0 = wall_clock64();
	for(int b = 0; b < nblk; b++) {
		asm volatile("" : "+v"(fa), "+v"(fb));
		if(V == 0) {
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const v2f r0 = v2f{lane_of(fa, j), lane_of(fb, j)};
				const v2f yv = r0 + (B1 * y1 + B2 * y2);
				y2 = y1; y1 = yv;
			}
		} else if(V == 1) {
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const float ri = lane_of(fa, j), rq = lane_of(fb, j);
				const float yi = ri + (b1 * yi1 + b2 * yi2);
				const float yq = rq + (b1 * yq1 + b2 * yq2);
				yi2 = yi1; yi1 = yi; yq2 = yq1; yq1 = yq;
			}
		} else if(V == 2) {
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const float ri = lane_of(fa, j);
				const float yi = ri + (b1 * yi1 + b2 * yi2);
				yi2 = yi1; yi1 = yi;
			}
		} else if(V == 3) {
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const float ri = lane_of(fa, j), rq = lane_of(fb, j);
				const float r = odd ? rq : ri;
				const float yi = r + (b1 * yi1 + b2 * yi2);
				yi2 = yi1; yi1 = yi;
			}
		} else if(V == 4) {
			// the product B2*y2 a step early, by hand (what the compiler should do anyway)
			float t2 = b2 * yi2;
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const float ri = lane_of(fa, j);
				const float m = b1 * yi1;
				const float s = m + t2;
				t2 = b2 * yi1;
				const float yi = ri + s;
				yi1 = yi;
			}
			yi2 = t2;
		} else if(V == 5) {
			// r0 from LDS-free DPP broadcast instead of v_readlane: row_bcast is not general; use v_readfirstlane after a rotate (wave_ror:1)
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const float ri = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fa)));
				fa = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, fa), 0x134 /* wave_rol:1 */, 0xf, 0xf, false));
				const float yi = ri + (b1 * yi1 + b2 * yi2);
				yi2 = yi1; yi1 = yi;
			}
		} else if(V == 6 || V == 7) {
			// V6: the two products of a step in ONE instruction - a component lives in a pair of lanes, the even one multiplies by B1, the
			// odd one by B2 (per-lane coefficient), y comes to both from the even lane (quad_perm [0,0,2,2] on the multiply), the sum
			// m + t2 takes the odd lane's product of the step before (quad_perm [1,0,3,2] on the add): 3 VALU per step.  V7: the same with
			// the feed-forward value from a register (as the kernel has it) instead of v_readlane
			const float bb = odd ? b2 : b1;
			float pprev = 0.f;
			#pragma unroll
			for(int j = 0; j < 64; j++) {
				const float ri = V == 6 ? lane_of(fa, j) : (j & 1 ? fb : fa);
				const float ysrc = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, yi1), 0xA0, 0xf, 0xf, false));
				const float p = bb * ysrc;
				const float tp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, pprev), 0xB1, 0xf, 0xf, false));
				const float s = p + tp;
				pprev = p;
				yi1 = ri + s;
			}
			yi2 = pprev;
		} else if(V == 8 || V == 9 || V == 10) {
			// V4 with the feed-forward value from a register; V9 / V10: only lanes 0-31 / 0-15 active (does the SIMD skip the idle quarters?)
			if(V == 8 || (V == 9 && lane < 32) || (V == 10 && lane < 16)) {
				float t2 = b2 * yi2;
				#pragma unroll
				for(int j = 0; j < 64; j++) {
					const float ri = j & 1 ? fb : fa;
					const float m = b1 * yi1;
					const float s = m + t2;
					t2 = b2 * yi1;
					yi1 = ri + s;
				}
				yi2 = t2;
			}
		}
	}
	const unsigned long long t1 = clock64();
	const unsigned long long w1 = wall_clock64();
	if(lane == 0 && w0_) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
	out[lane] = y1.x + y1.y + y2.x + y2.y + yi1 + yi2 + yq1 + yq2;
}

// the same with W wavefronts per SIMD busy with the same thing (does a second scan on the SIMD slow the first?)
template<int V>
static void run(const char *what, int waves_per_simd) {
	float *d_in, *d_out; unsigned long long *d_clk;
	std::vector<float> h(130);
	for(int i = 0; i < 128; i++) h[i] = 1e-3f * (float)((i * 37) % 101 - 50);
	h[128] = 1.9692f; h[129] = -0.97f;
	hipMalloc(&d_in, 130 * 4); hipMalloc(&d_out, 64 * 4 * 4096); hipMalloc(&d_clk, 16);
	hipMemcpy(d_in, h.data(), 130 * 4, hipMemcpyHostToDevice);
	const int nblk = 1 << 13;    // 2^19 steps
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for(int rep = 0; rep < 3; rep++) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k_scan<V>, dim3(1), dim3(64 * waves_per_simd * 4), 0, 0, d_in, d_out, nblk, d_clk);
		hipEventRecord(e1); hipEventSynchronize(e1); { hipError_t e = hipGetLastError(); if(e != hipSuccess) printf("launch: %s\n", hipGetErrorString(e)); }
		float ms = 0; hipEventElapsedTime(&ms, e0, e1);
		unsigned long long c[2]; hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
		if(rep == 2) printf("%-58s waves/SIMD %d: %7.3f ms  %6.2f ns/step  %6.1f shader clocks/step (s_memtime)  %6.2f ns/step (s_memrealtime @100 MHz)\n", what, waves_per_simd, ms, ms * 1e6 / (nblk * 64.0), (double)c[0] / (nblk * 64.0), (double)c[1] * 10.0 / (nblk * 64.0));
	}
	hipFree(d_in); hipFree(d_out); hipFree(d_clk);
}

int main() {
	// warm the clock
	{ float *d; hipMalloc(&d, 1 << 26); for(int i = 0; i < 200; i++) hipMemsetAsync(d, i, 1 << 26, 0); hipDeviceSynchronize(); hipFree(d); }
	for(int w = 1; w <= 2; w++) {
		run<0>("V0 packed (I,Q): 2 readlane + 4 v_pk (round 5)", w);
		run<1>("V1 I and Q as separate fp32 chains: 2 readlane + 8 VALU", w);
		run<2>("V2 one component only: 1 readlane + 4 VALU", w);
		run<3>("V3 I in even lanes, Q in odd: 2 readlane + select + 4", w);
		run<4>("V4 one component, B2*y2 a step early by hand", w);
		run<5>("V5 one component, r0 by readfirstlane + wave_rol", w);
		run<6>("V6 lane pair (B1 | B2 per lane), DPP: 1 readlane + 3 VALU", w);
		run<7>("V7 lane pair, r0 from a register: 3 VALU", w);
		run<8>("V8 one component, r0 from a register: 4 VALU", w);
		run<9>("V9 = V8, lanes 0-31 only", w);
		run<10>("V10 = V8, lanes 0-15 only", w);
	}
	return 0;
}
