// Development micro-benchmark (not part of the product): issue rates that K1's design rests on.
//   v_fma_f32 vs v_pk_fma_f32 throughput per wave, 4 and 8 waves per SIMD
//   ds_read_b128 gather: same address / conflict-free / random 256-entry LUT (16 B entries)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub dev/gpu_ubench_valu.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template<int PK>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float s) {
	v2f a[8];
	for(int i = 0; i < 8; i++) a[i] = v2f{(float)threadIdx.x + i, (float)i};
	const v2f m = v2f{s, s}, c = v2f{0.5f, 0.25f};
	for(int it = 0; it < iters; it++) {
		#pragma unroll
		for(int i = 0; i < 8; i++) {
			if(PK) a[i] = __builtin_elementwise_fma(a[i], m, c);
			else {   // inline asm: the compiler would SLP-pack two adjacent scalar FMAs into v_pk_fma_f32
				float x = a[i].x, y = a[i].y;
				asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(s), "v"(c.x));
				asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(s), "v"(c.y));
				a[i].x = x; a[i].y = y;
			}
		}
	}
	float r = 0.f;
	for(int i = 0; i < 8; i++) r += a[i].x + a[i].y;
	out[blockIdx.x * 256 + threadIdx.x] = r;
}

// mode 0: every lane the same entry; 1: lane-distinct slots (conflict-free); 2: pseudo-random entries; 3: stride-16 entries (worst case)
__global__ __launch_bounds__(256) void k_gather(float *out, int iters, int mode, unsigned step) {
	__shared__ float4 lut[256];
	lut[threadIdx.x] = make_float4((float)threadIdx.x, 1.f, 2.f, 3.f);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	unsigned p = mode == 0 ? 7u : mode == 1 ? (unsigned)lane : mode == 2 ? (unsigned)lane * 2654435761u >> 8 : (unsigned)lane * 16u;
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
	for(int it = 0; it < iters; it++) {
		#pragma unroll
		for(int i = 0; i < 8; i++) {
			const float4 e = lut[p & 255u];
			acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
			p += step;
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// K1's inner loop in miniature: per "channel-sample" one LUT gather (ds_read_b128), 2 SDWA-style integer ops, the packed-FP32 chain
//   sc = fma(e.zw, F, e.xy); sx = sc * (i x); m = fma(sc.y, x, sx); A0 = fma(g0, m, A0); A1 = fma(g1, m, A1); ph += dph
// MODE 0: the two tap sums as v_pk_fma_f32 (what K1 does).  MODE 1: the two tap sums on the matrix pipe - one
// v_mfma_f32_4x4x1_16b_f32 per component (re, im): D[r] += A[r] * B with B = the lane's mixed sample and A = (g0, g1, 0, 0) taken from
// the four lanes of the block, i.e. rows 0/1 are the two sums, rows 2/3 idle; same fp32 FMA per element (bit-identical), different
// issue port.  The A operand is rebuilt per sample with two v_cndmask (shared by the 4 "channels" of the wave, as in K1).
typedef float v4f __attribute__((ext_vector_type(4)));
template<int MODE>
__global__ __launch_bounds__(256, 4) void k_mix(float *out, int iters, unsigned step, const float *taps) {
	__shared__ float4 lut[256];
	lut[threadIdx.x] = make_float4(0.001f * (float)threadIdx.x, 1.f, 1e-6f, 2e-6f);
	__syncthreads();
	const int lane = threadIdx.x & 63;
	constexpr int CR = 4;
	unsigned ph[CR], dph[CR];
	for(int c = 0; c < CR; c++) { ph[c] = (unsigned)lane * 2654435761u + c * 977u; dph[c] = step * (2 * c + 1) + 12345u; }
	v2f A0[CR], A1[CR];
	v4f Dr[CR], Di[CR];
	for(int c = 0; c < CR; c++) { A0[c] = v2f{0.f, 0.f}; A1[c] = v2f{0.f, 0.f}; Dr[c] = v4f{0.f, 0.f, 0.f, 0.f}; Di[c] = v4f{0.f, 0.f, 0.f, 0.f}; }
	const bool l0 = (lane & 3) == 0, l1 = (lane & 3) == 1;
	float xr = 0.25f + lane * 1e-3f, xi = -0.125f;
	for(int it = 0; it < iters; it++) {
		#pragma unroll 5
		for(int j = 0; j < 20; j++) {
			const float g0 = taps[j], g1 = taps[20 + j];            // uniform: scalar loads, as K1's kernarg taps
			const v2f X = v2f{xr, xi}, Xr = v2f{-xi, xr};
			float ga = 0.f;
			if(MODE == 1) ga = l0 ? g0 : l1 ? g1 : 0.f;             // the A operand: (g0, g1, 0, 0) across the four lanes of a block
			#pragma unroll
			for(int c = 0; c < CR; c++) {
				const unsigned p = ph[c];
				const float F = (float)(p & 0xffffu);
				const float4 e = lut[(p >> 16) & 0xffu];
				const v2f sc = __builtin_elementwise_fma(v2f{e.z, e.w}, v2f{F, F}, v2f{e.x, e.y});
				const v2f m = __builtin_elementwise_fma(v2f{sc.y, sc.y}, X, v2f{sc.x, sc.x} * Xr);
				if(MODE == 0) {
					A0[c] = __builtin_elementwise_fma(v2f{g0, g0}, m, A0[c]);
					A1[c] = __builtin_elementwise_fma(v2f{g1, g1}, m, A1[c]);
				} else {
					Dr[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ga, m.x, Dr[c], 0, 0, 0);
					Di[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(ga, m.y, Di[c], 0, 0, 0);
				}
				ph[c] = p + dph[c];
			}
			xr += 1e-6f; xi -= 1e-6f;
		}
	}
	float r = 0.f;
	for(int c = 0; c < CR; c++) r += A0[c].x + A0[c].y + A1[c].x + A1[c].y + Dr[c].x + Dr[c].y + Di[c].x + Di[c].y;
	out[blockIdx.x * 256 + threadIdx.x] = r;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
	float *out; hipMalloc(&out, 256 * 8192 * sizeof(float));
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4096;
	for(int wg_per_cu : {4, 8}) {
		const int grid = 256 * wg_per_cu;
		for(int pk = 0; pk < 2; pk++) {
			for(int rep = 0; rep < 2; rep++) {
				hipEventRecord(e0);
				if(pk) hipLaunchKernelGGL(k_fma<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
				else hipLaunchKernelGGL(k_fma<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f);
				hipEventRecord(e1); hipEventSynchronize(e1);
			}
			const double ms = time_ms(e0, e1);
			const double flops = (double)grid * 256 * iters * 16 * 2;      // 16 scalar FMAs (8 pairs) per iteration per lane
			printf("fma pk=%d wg/cu=%d: %.3f ms  %.1f TFLOP/s\n", pk, wg_per_cu, ms, flops / ms / 1e9);
		}
	}
	{
		float h_taps[40]; for(int i = 0; i < 40; i++) h_taps[i] = 0.01f * (float)(i + 1);
		float *taps; hipMalloc(&taps, sizeof h_taps); hipMemcpy(taps, h_taps, sizeof h_taps, hipMemcpyHostToDevice);
		const int grid = 256 * 4 * 4, it2 = 256;               // 4 workgroups per CU resident (launch bound), 4 rounds
		for(int mode = 0; mode < 2; mode++) {
			for(int rep = 0; rep < 3; rep++) {
				hipEventRecord(e0);
				if(mode) hipLaunchKernelGGL(k_mix<1>, dim3(grid), dim3(256), 0, 0, out, it2, 40u * 63913u, taps);
				else hipLaunchKernelGGL(k_mix<0>, dim3(grid), dim3(256), 0, 0, out, it2, 40u * 63913u, taps);
				hipEventRecord(e1); hipEventSynchronize(e1);
			}
			const double ms = time_ms(e0, e1);
			const double cs = (double)grid * 256 * it2 * 20 * 4;       // lane-level channel-samples
			printf("K1-like loop, tap sums on %s: %.3f ms  %.3e channel-samples/s  (%.1f cycles per wave-level channel-sample per SIMD at 2.4 GHz)\n",
			       mode ? "the matrix pipe (2 x v_mfma_f32_4x4x1 per channel-sample)" : "the VALU (2 x v_pk_fma_f32)", ms, cs / ms * 1e3,
			       ms * 1e-3 * 2.4e9 * 1024 / (cs / 64));
		}
	}
	for(int mode = 0; mode < 4; mode++) {
		for(unsigned step : {0u, 1u, 3u, 16u, 39u, 40u}) {
			const int grid = 256 * 4;
			for(int rep = 0; rep < 2; rep++) {
				hipEventRecord(e0);
				hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, out, iters, mode, step);
				hipEventRecord(e1); hipEventSynchronize(e1);
			}
			const double ms = time_ms(e0, e1);
			const double gathers = (double)grid * 4 * iters * 8;               // wave-level ds_read_b128 instructions
			// LDS cycles per wave-instruction if the LDS pipe were the only limit: 256 CUs at ~2.4 GHz
			printf("gather mode=%d step=%u: %.3f ms  %.2f ns per wave-gather per CU  (~%.1f LDS clk)\n", mode, step, ms,
			       ms * 1e6 / (gathers / 256), ms * 1e6 / (gathers / 256) * 2.4);
		}
	}
	return 0;
}
