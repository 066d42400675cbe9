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
