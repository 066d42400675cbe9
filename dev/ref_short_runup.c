// dev/ref_short_runup.c - how far from the reference's own fp32 trajectory (src/demod.c:302-329) is a scan that started from a zero
// state W input samples earlier, BEFORE the two have become bit-identical?  (They do after ~1.5e4 samples on average, 1.2e5 at
// worst: dev/iir_state_coalescence.c.)  The zero-start transient is gone after a few hundred samples; what is left is a difference of
// a few units in the last place that lingers until the two happen to round alike.  If that residue is bounded - by a small multiple
// of an ulp of the signal around the sample - a referee scan with a SHORT run-up gives the reference's samples to within that bound:
// good enough to decide all but a sliver of the marginal decisions, at a sixteenth of the cost.
//   built and driven by dev/ref_short_runup.py
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef struct { float xr[3], xi[3], yr[3], yi[3]; uint32_t phi; } st_t;
static inline void step(st_t *v, float re, float im, uint32_t dphi, int mix, const float *sin_t, const float *cos_t, const float *A, const float *B) {
	v->xr[2] = v->xr[1]; v->xr[1] = v->xr[0]; v->xi[2] = v->xi[1]; v->xi[1] = v->xi[0];
	v->yr[2] = v->yr[1]; v->yr[1] = v->yr[0]; v->yi[2] = v->yi[1]; v->yi[1] = v->yi[0];
	if(mix) {
		uint32_t idx = v->phi >> 16; float fr = (float)(v->phi & 0xffff) / 65536.0f;
		float s1 = sin_t[idx], s2 = sin_t[idx + 1], sn = s1 + (s2 - s1) * fr;
		float c1 = cos_t[idx], c2 = cos_t[idx + 1], cs = c1 + (c2 - c1) * fr;
		float mr = re * cs - im * sn, mi = im * cs + re * sn;
		re = mr; im = mi;
		v->phi = (v->phi + dphi) & 0xffffff;
	}
	v->xr[0] = re; v->xi[0] = im;
	float r = A[0] * v->xr[0]; r += A[1] * v->xr[1] + A[2] * v->xr[2]; r += B[1] * v->yr[1] + B[2] * v->yr[2]; v->yr[0] = r;
	r = A[0] * v->xi[0]; r += A[1] * v->xi[1] + A[2] * v->xi[2]; r += B[1] * v->yi[1] + B[2] * v->yi[2]; v->yi[0] = r;
}
// the whole trajectory from the stream's start, decimated: y[2k], y[2k+1] = output after input sample os (k + 1) - 1
void full(const int16_t *raw, long n, int os, uint32_t dphi, int mix, const float *sin_t, const float *cos_t, const float *A, const float *B, float *y) {
	st_t v; memset(&v, 0, sizeof v);
	for(long s = 0; s < n; s++) {
		step(&v, (float)raw[2 * s] / 32768.0f, (float)raw[2 * s + 1] / 32768.0f, dphi, mix, sin_t, cos_t, A, B);
		if((s + 1) % os == 0) { long k = (s + 1) / os - 1; y[2 * k] = v.yr[0]; y[2 * k + 1] = v.yi[0]; }
	}
}
// zero state at input sample s0 (a multiple of os); for every decimated sample k after it, up to kmax of them: the difference from
// the full trajectory y, relative to the largest |y| among k .. k-3, is folded into env[(k - k0) / bin] (maximum); returns the
// number of decimated samples until the two are bit-identical for good (within the stretch looked at), -1: never
long from_zero(const int16_t *raw, long n, int os, uint32_t dphi, int mix, const float *sin_t, const float *cos_t, const float *A, const float *B,
		const float *y, long s0, long kmax, int bin, double *env, double *env_abs) {
	st_t v; memset(&v, 0, sizeof v);
	v.phi = (uint32_t)(((uint64_t)s0 * dphi) & 0xffffff);
	long k0 = s0 / os, last_diff = -1;
	for(long s = s0; s < n && (s + 1) / os - 1 - k0 < kmax; s++) {
		step(&v, (float)raw[2 * s] / 32768.0f, (float)raw[2 * s + 1] / 32768.0f, dphi, mix, sin_t, cos_t, A, B);
		if((s + 1) % os == 0) {
			long k = (s + 1) / os - 1;
			double dr = (double)v.yr[0] - y[2 * k], di = (double)v.yi[0] - y[2 * k + 1], d = sqrt(dr * dr + di * di);
			if(d != 0.0) last_diff = k - k0;
			double m = 0;
			for(long j = k; j >= 0 && j > k - 4; j--) { double a = hypot(y[2 * j], y[2 * j + 1]); if(a > m) m = a; }
			double rel = m > 0 ? d / m : (d > 0 ? 1e9 : 0);
			long b = (k - k0) / bin;
			if(rel > env[b]) env[b] = rel;
			if(d > env_abs[b]) env_abs[b] = d;
		}
	}
	return last_diff + 1;
}
