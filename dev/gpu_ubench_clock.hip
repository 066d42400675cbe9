// Development micro-benchmark (not part of the product): TRUE issue costs in shader clocks, measured with the chip at its working
// clock.  Each kernel is launched back to back for >= 0.3 s before it is timed (the shader clock needs a few hundred ms to leave its
// idle state: profiles/r03_clocks_under_load.txt), and lane 0 of workgroup 0 brackets its loop with s_memtime (shader clocks) and
// s_memrealtime (100 MHz), so that cycles per instruction need no assumption about the clock.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/ubc dev/gpu_ubench_clock.hip && /tmp/ubc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

struct Clk { unsigned long long core, real; };

template<int PK>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float s, Clk *clk) {
	v2f a[8];
	for(int i = 0; i < 8; i++) a[i] = v2f{(float)threadIdx.x + i, (float)i};
	const v2f m = v2f{s, s}, c = v2f{0.5f, 0.25f};
	const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	for(int it = 0; it < iters; it++) {
		#pragma unroll
		for(int i = 0; i < 8; i++) {
			if(PK) a[i] = __builtin_elementwise_fma(a[i], m, c);
			else {
				float x = a[i].x, y = a[i].y;
				asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(s), "v"(c.x));
				asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(s), "v"(c.y));
				a[i].x = x; a[i].y = y;
			}
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
	float r = 0.f;
	for(int i = 0; i < 8; i++) r += a[i].x + a[i].y;
	out[blockIdx.x * 256 + threadIdx.x] = r;
	if(blockIdx.x == 0 && threadIdx.x == 0) { clk->core = t1 - t0; clk->real = r1 - r0; }
}

// K1's inner loop in miniature (see gpu_ubench_valu.hip).  GATHER 0: the LUT entry is a loop-invariant register (VALU work only);
// 1: one ds_read_b128 gather per channel-sample as in K1
template<int GATHER>
__global__ __launch_bounds__(256, 4) void k_mix(float *out, int iters, unsigned step, const float *taps, Clk *clk) {
	__shared__ float4 lut[256];
	lut[threadIdx.x] = make_float4(0.001f * (float)threadIdx.x, 1.f, 1e-6f, 2e-6f);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	constexpr int CR = 4;
	unsigned ph[CR], dph[CR];
	for(int c = 0; c < CR; c++) { ph[c] = (unsigned)lane * 2654435761u + c * 977u; dph[c] = step * (2 * c + 1) + 12345u; }
	v2f A0[CR], A1[CR];
	for(int c = 0; c < CR; c++) { A0[c] = v2f{0.f, 0.f}; A1[c] = v2f{0.f, 0.f}; }
	float xr = 0.25f + lane * 1e-3f, xi = -0.125f;
	const float4 e0 = lut[lane];
	const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	for(int it = 0; it < iters; it++) {
		#pragma unroll 5
		for(int j = 0; j < 20; j++) {
			const float g0 = taps[j], g1 = taps[20 + j];
			const v2f X = v2f{xr, xi}, Xr = v2f{-xi, xr};
			#pragma unroll
			for(int c = 0; c < CR; c++) {
				const unsigned p = ph[c];
				const float F = (float)(p & 0xffffu);
				float4 e = e0;
				if(GATHER) e = lut[(p >> 16) & 0xffu];
				else asm volatile("" : "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w));     // opaque: keeps the FMA per channel-sample
				const v2f sc = __builtin_elementwise_fma(v2f{e.z, e.w}, v2f{F, F}, v2f{e.x, e.y});
				const v2f m = __builtin_elementwise_fma(v2f{sc.y, sc.y}, X, v2f{sc.x, sc.x} * Xr);
				A0[c] = __builtin_elementwise_fma(v2f{g0, g0}, m, A0[c]);
				A1[c] = __builtin_elementwise_fma(v2f{g1, g1}, m, A1[c]);
				ph[c] = p + dph[c];
			}
			xr += 1e-6f; xi -= 1e-6f;
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
	float r = 0.f;
	for(int c = 0; c < CR; c++) r += A0[c].x + A0[c].y + A1[c].x + A1[c].y;
	out[blockIdx.x * 256 + threadIdx.x] = r;
	if(blockIdx.x == 0 && threadIdx.x == 0) { clk->core = t1 - t0; clk->real = r1 - r0; }
}

// The "one lane = one channel's sequential run" layout in miniature: no scan, no transposed tile - every lane walks its own run of
// blocks, x comes from LDS (same address for the lanes of a run: a broadcast), outputs are kept for 8 blocks and written as 64 bytes.
// MINB = workgroups of 256 per CU the launch bound asks for (4 -> 4 waves per SIMD / 128 registers, 8 -> 8 waves / 64 registers)
struct RunArgs { float g0[20], g1[20]; float P[4]; float c[3]; };
template<int MINB>
__global__ __launch_bounds__(256, MINB) void k_run(float4 *out, int nblocks, const unsigned *dphv, RunArgs a, Clk *clk) {
	__shared__ __align__(16) float4 lut[256];
	__shared__ float2 xt[4][80];                           // per wave: 4 blocks of samples (refilled every 4 blocks)
	lut[threadIdx.x] = make_float4(0.001f * (float)threadIdx.x, 1.f - 1e-4f * (float)threadIdx.x, 1e-6f, 2e-6f);
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int gid = blockIdx.x * 256 + threadIdx.x;
	unsigned ph = (unsigned)gid * 2654435761u;
	const unsigned dph = dphv[gid & 255];
	float t0r = 0.f, t0i = 0.f, t1r = 0.f, t1i = 0.f;
	float4 keep[4];
	const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	for(int b = 0; b < nblocks; b++) {
		if((b & 3) == 0) {                                  // stage 80 samples for this wave (conversion amortised over the wave)
			for(int t = lane; t < 80; t += 64) xt[wave][t] = make_float2((float)((b * 80 + t) & 1023) * (1.f / 32768.f), (float)((b * 81 + t) & 511) * (1.f / 32768.f));
			__builtin_amdgcn_wave_barrier();
		}
		const float2 *xr = &xt[wave][(b & 3) * 20];
		v2f A0 = v2f{0.f, 0.f}, A1 = v2f{0.f, 0.f}, M = v2f{0.f, 0.f};
		#pragma unroll
		for(int j = 0; j < 20; j++) {
			const float2 x = xr[j];
			const v2f X = v2f{x.x, x.y}, Xr = v2f{-x.y, x.x};
			const float F = (float)(ph & 0xffffu);
			const float4 e = lut[(ph >> 16) & 0xffu];
			const v2f sc = __builtin_elementwise_fma(v2f{e.z, e.w}, v2f{F, F}, v2f{e.x, e.y});
			const v2f m = __builtin_elementwise_fma(v2f{sc.y, sc.y}, X, v2f{sc.x, sc.x} * Xr);
			A0 = __builtin_elementwise_fma(v2f{a.g0[j], a.g0[j]}, m, A0);
			A1 = __builtin_elementwise_fma(v2f{a.g1[j], a.g1[j]}, m, A1);
			M = m;
			ph += dph;
		}
		const float n0r = __builtin_fmaf(a.P[0], t0r, __builtin_fmaf(a.P[1], t1r, A0.x)), n0i = __builtin_fmaf(a.P[0], t0i, __builtin_fmaf(a.P[1], t1i, A0.y));
		const float n1r = __builtin_fmaf(a.P[2], t0r, __builtin_fmaf(a.P[3], t1r, A1.x)), n1i = __builtin_fmaf(a.P[2], t0i, __builtin_fmaf(a.P[3], t1i, A1.y));
		t0r = n0r; t0i = n0i; t1r = n1r; t1i = n1i;
		const float yr = __builtin_fmaf(a.c[0], n0r, __builtin_fmaf(a.c[1], n1r, a.c[2] * M.x)), yi = __builtin_fmaf(a.c[0], n0i, __builtin_fmaf(a.c[1], n1i, a.c[2] * M.y));
		if(b & 1) { keep[(b >> 1) & 3].z = yr; keep[(b >> 1) & 3].w = yi; } else { keep[(b >> 1) & 3].x = yr; keep[(b >> 1) & 3].y = yi; }
		if((b & 7) == 7) {
			float4 *dst = out + ((size_t)gid * (size_t)(nblocks / 2) + (size_t)(b >> 1) - 3);
			dst[0] = keep[0]; dst[1] = keep[1]; dst[2] = keep[2]; dst[3] = keep[3];
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
	if(blockIdx.x == 0 && threadIdx.x == 0) { clk->core = t1 - t0; clk->real = r1 - r0; }
}

// LUT gathers: 16 bytes per lane from a 4 KiB table at pseudo-random entries - from LDS (ds_read_b128), from global memory through
// the vector L1 (global_load_dwordx4), or alternating (SPLIT: 2 of 3 from LDS, 1 of 3 from L1) - is the second path worth using?
template<int MODE>   // 0 LDS, 1 L1, 2 split 2:1
__global__ __launch_bounds__(256, 4) void k_lutgather(float *out, int iters, const float4 *glut, Clk *clk) {
	__shared__ __align__(16) float4 lut[256];
	lut[threadIdx.x] = glut[threadIdx.x];
	__syncthreads();
	unsigned p = (unsigned)(blockIdx.x * 256 + threadIdx.x) * 2654435761u;
	const unsigned dp = 63913u * (2 * (threadIdx.x & 63) + 1) * 40u + 12345u;
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
	for(int it = 0; it < iters; it++) {
		#pragma unroll
		for(int i = 0; i < 12; i++) {
			const unsigned idx = (p >> 16) & 0xffu;
			float4 e;
			if(MODE == 0 || (MODE == 2 && i % 3 != 2)) e = lut[idx]; else e = glut[idx];
			acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
			p += dp;
		}
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
	out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
	if(blockIdx.x == 0 && threadIdx.x == 0) { clk->core = t1 - t0; clk->real = r1 - r0; }
}

template<typename F> static void run(const char *name, F launch, double insts_per_wave, double flops, hipEvent_t e0, hipEvent_t e1, Clk *dclk) {
	// warm: >= 0.4 s of the same kernel
	hipEventRecord(e0); float ms = 0.f; int n = 0;
	do { launch(); n++; hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } while(ms < 400.f);
	hipEventRecord(e0);
	for(int i = 0; i < 20; i++) launch();
	hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
	Clk h; hipMemcpy(&h, dclk, sizeof h, hipMemcpyDeviceToHost);
	const double mhz = (double)h.core / ((double)h.real / 100.0);             // s_memrealtime counts at 100 MHz
	printf("%-46s %8.3f ms/launch  %7.1f TFLOP/s  s_memtime/s_memrealtime -> %6.0f MHz  wave 0: %7.2f shader clocks per loop item of its own\n",
	       name, ms / 20, flops / (ms / 20) / 1e9, mhz, (double)h.core / insts_per_wave);
}

int main() {
	float *out; hipMalloc(&out, 256 * 8192 * sizeof(float));
	Clk *dclk; hipMalloc(&dclk, sizeof(Clk));
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float h_taps[40]; for(int i = 0; i < 40; i++) h_taps[i] = 0.01f * (float)(i + 1);
	float *taps; hipMalloc(&taps, sizeof h_taps); hipMemcpy(taps, h_taps, sizeof h_taps, hipMemcpyHostToDevice);
	const int iters = 16384;
	for(int wg_per_cu : {4, 8}) {
		const int grid = 256 * wg_per_cu;
		const double flops = (double)grid * 256 * iters * 16 * 2;
		char nm[96];
		snprintf(nm, sizeof nm, "v_fma_f32    x16 per iteration, %d waves/SIMD", wg_per_cu);
		run(nm, [&] { hipLaunchKernelGGL(k_fma<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, dclk); }, (double)iters * 16, flops, e0, e1, dclk);
		snprintf(nm, sizeof nm, "v_pk_fma_f32 x8  per iteration, %d waves/SIMD", wg_per_cu);
		run(nm, [&] { hipLaunchKernelGGL(k_fma<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, dclk); }, (double)iters * 8, flops, e0, e1, dclk);
	}
	{
		const int grid = 256 * 4, it2 = 1024;               // 4 workgroups per CU = 4 waves per SIMD, one round
		const double cs = (double)it2 * 20 * 4;             // channel-samples per wave
		const double flops = (double)grid * 256 * cs * 30;  // SURVEY's 30 flop per channel-sample
		run("K1-like loop, LUT entry in registers", [&] { hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(256), 0, 0, out, it2, 40u * 63913u, taps, dclk); }, cs, flops, e0, e1, dclk);
		run("K1-like loop, ds_read_b128 gather", [&] { hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(256), 0, 0, out, it2, 40u * 63913u, taps, dclk); }, cs, flops, e0, e1, dclk);
	}
	{
		float4 *glut; hipMalloc(&glut, 4096); hipMemset(glut, 0, 4096);
		const int grid = 256 * 4, it3 = 2048;
		const double gathers = (double)it3 * 12;
		const char *nm[3] = {"LUT gather 16 B/lane: LDS", "LUT gather 16 B/lane: vector L1", "LUT gather 16 B/lane: 2 LDS : 1 L1"};
		for(int mode = 0; mode < 3; mode++) {
			auto l = [&] { if(mode == 0) hipLaunchKernelGGL(k_lutgather<0>, dim3(grid), dim3(256), 0, 0, out, it3, (const float4 *)glut, dclk);
			               else if(mode == 1) hipLaunchKernelGGL(k_lutgather<1>, dim3(grid), dim3(256), 0, 0, out, it3, (const float4 *)glut, dclk);
			               else hipLaunchKernelGGL(k_lutgather<2>, dim3(grid), dim3(256), 0, 0, out, it3, (const float4 *)glut, dclk); };
			run(nm[mode], l, gathers, 0.0, e0, e1, dclk);
			// per CU: 16 waves x gathers wave-gathers per launch
		}
		printf("  (per wave-gather and CU: ms/launch * clock / (16 waves * %d gathers))\n", it3 * 12);
	}
	{
		RunArgs ra;
		for(int j = 0; j < 20; j++) { ra.g0[j] = 0.01f * (j + 1); ra.g1[j] = 0.02f * (20 - j); }
		ra.P[0] = 0.5f; ra.P[1] = -0.2f; ra.P[2] = 0.3f; ra.P[3] = 0.4f; ra.c[0] = 0.1f; ra.c[1] = 0.2f; ra.c[2] = 0.3f;
		std::vector<unsigned> hd(256);
		for(int i = 0; i < 256; i++) hd[i] = 63913u * (2 * i + 1) + 12345u;
		unsigned *dphv; hipMalloc(&dphv, 1024); hipMemcpy(dphv, hd.data(), 1024, hipMemcpyHostToDevice);
		const int nblocks = 512;
		float4 *out4; hipMalloc(&out4, (size_t)256 * 8 * 256 * 256 * (nblocks / 2) * sizeof(float4) / 256 + 65536);
		for(int minb : {4, 8}) {
			const int grid = 256 * minb;                   // one round: every workgroup resident
			const double steps = (double)nblocks * 20;     // wave-steps (= channel-samples per lane)
			const double flops = (double)grid * 256 * steps * 30;
			char nm[96]; snprintf(nm, sizeof nm, "lane = one channel's run, %d waves/SIMD", minb);
			if(minb == 4) run(nm, [&] { hipLaunchKernelGGL(k_run<4>, dim3(grid), dim3(256), 0, 0, out4, nblocks, dphv, ra, dclk); }, steps, flops, e0, e1, dclk);
			else run(nm, [&] { hipLaunchKernelGGL(k_run<8>, dim3(grid), dim3(256), 0, 0, out4, nblocks, dphv, ra, dclk); }, steps, flops, e0, e1, dclk);
		}
	}
	return 0;
}
