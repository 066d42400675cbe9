// design.h - host-side, one-time derivation of every constant the kernels use.
//
// Mirrors the reference's init-time code so the device works from the same
// numbers the CPU path would:
//   chebyshev_lpf_init()/chebyshev_lpf_calc_pole()   src/chebyshev.c:32-119
//   input_lpf_init()                                 src/demod.c:367-370
//   sincosf_lut_init()                               src/demod.c:372-377
//   vdl2_channel_init() (NCO step)                   src/demod.c:385
//   demod_sync_init()                                src/demod.c:84-96
// and then rewrites the 2-pole IIR in block form for the channeliser kernel
// (see DESIGN.md "K1"): per decimated output the recurrence runs once on a 2x2
// state, fed by an `oversample`-tap dot product of the mixed input.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace vdl2 {

constexpr int kSymbolRate = 10500;   // dumpvdl2.h:46
constexpr int kSps = 10;             // dumpvdl2.h:44
constexpr int kMaxOversample = 32;
constexpr int kFixW = 128;           // blocks after which a start state has decayed below fp32 resolution
constexpr int kRunMax = 8;           // max decimated outputs per thread-run in K1

struct LpfCoeffs { float A[3]; float B[3]; };

// 2-pole, 0.5 % ripple Chebyshev low-pass, same float operation order as chebyshev.c
inline LpfCoeffs design_lpf(float fc, float ripple) {
	const int np = 2;
	float rp, ip;
	sincosf((float)(M_PI / (2 * np)), &ip, &rp);
	rp = -rp;
	if(ripple != 0.f) {
		float es = sqrtf(powf(100.f / (100.f - ripple), 2.f) - 1.f);
		float vx = (1.f / np) * logf((1.f / es) + sqrtf(1.f / (es * es) + 1.f));
		float kx = (1.f / np) * logf((1.f / es) + sqrtf(1.f / (es * es) - 1.f));
		kx = (expf(kx) + expf(-kx)) / 2.f;
		rp *= ((expf(vx) - expf(-vx)) / 2.f) / kx;
		ip *= ((expf(vx) + expf(-vx)) / 2.f) / kx;
	}
	float t = 2.f * tanf(0.5f);
	float w = 2.f * M_PI * fc;
	float m = rp * rp + ip * ip;
	float d = 4.f - 4.f * rp * t + m * t * t;
	float x0 = t * t / d, x1 = 2.f * x0, x2 = x0;
	float y1 = (8.f - 2.f * m * t * t) / d;
	float y2 = (-4.f - 4.f * rp * t - m * t * t) / d;
	float k = sinf(0.5f - w / 2.f) / sinf(0.5f + w / 2.f);
	d = 1 + y1 * k - y2 * k * k;
	float a[3] = { (x0 - x1 * k + x2 * k * k) / d,
	               (-2.f * x0 * k + x1 + x1 * k * k - 2.f * x2 * k) / d,
	               (x0 * k * k - x1 * k + x2) / d };
	float b[3] = { -0.f, (2.f * k + y1 + y1 * k * k - 2.f * y2 * k) / d, (-(k * k) - y1 * k + y2) / d };
	float sa = 0.f, sb = 0.f;
	for(int i = 0; i < 3; i++) { sa += a[i]; sb += b[i]; }
	float gain = sa / (1.f - sb);
	LpfCoeffs c;
	for(int i = 0; i < 3; i++) { c.A[i] = a[i] / gain; c.B[i] = b[i]; }
	return c;
}

inline uint32_t nco_step(uint32_t centerfreq, uint32_t freq, uint32_t fs) {
	return (uint32_t)(int)(((float)centerfreq - (float)freq) / (float)fs * 256.0f * 65536.0f);
}

// LUT entry for the NCO: {sin[i], cos[i], (sin[i+1]-sin[i]) * 2^-16, (cos[i+1]-cos[i]) * 2^-16}
// so that sincosf_lut()'s v1 + (v2 - v1) * fract becomes (s, c) + (ds, dc) * (float)(phi & 0xffff): one packed FMA.
struct Lut4 { float s, c, ds, dc; };
inline void build_nco_lut(Lut4 out[256]) {
	float s[257], c[257];
	for(uint32_t i = 0; i < 256; i++) sincosf(2.0f * M_PI * (float)i / 256.0f, &s[i], &c[i]);
	s[256] = s[0]; c[256] = c[0];
	for(int i = 0; i < 256; i++) {
		out[i].s = s[i]; out[i].ds = (s[i + 1] - s[i]) * (1.0f / 65536.0f);
		out[i].c = c[i]; out[i].dc = (c[i + 1] - c[i]) * (1.0f / 65536.0f);
	}
}

struct Mat2 { double a, b, c, d; };   // [[a b],[c d]]
inline Mat2 mul(const Mat2 &x, const Mat2 &y) {
	return { x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d };
}
inline Mat2 mpow(Mat2 m, int n) {
	Mat2 r{1, 0, 0, 1};
	while(n > 0) { if(n & 1) r = mul(r, m); m = mul(m, m); n >>= 1; }
	return r;
}

// Everything K1/K2 need, as plain floats (goes to the device by value / constant buffer).
struct BlockForm {
	int   os;                       // oversample = taps per block
	int   run;                      // decimated outputs per thread in K1 (R)
	float g0[kMaxOversample];       // tap j of state component 0: row 0 of T (hap[os-1-j], hap[os-2-j])
	float g1[kMaxOversample];       // tap j of state component 1: row 1 of the same
	float P[4];                     // T M^os T^-1  (row-major 2x2)
	float c0, c1, c2;               // y = c0*t[0] + c1*t[1] + c2*xm[n]
	float cP[kFixW][2];             // (c0,c1) * P^(i+1): fix-up row for the i-th output after a start state
	float Ppow[kFixW + 1][4];       // P^i, i = 0..kFixW
	float Q[6][4];                  // P^(run * 2^d): wave-scan step matrices, d = 0..5
	float Qpow[64][4];              // P^(run * (l+1)), l = 0..63: a carried state's contribution to lane l's end state
	float basis[4];                 // T (row-major): the state the kernels carry is T (v[n], v[n-1]) - see derive_block_form()
};

inline BlockForm derive_block_form(const LpfCoeffs &lp, int os, int run) {
	BlockForm bf{};
	bf.os = os; bf.run = run;
	const double A0 = lp.A[0], A1 = lp.A[1], A2 = lp.A[2], B1 = lp.B[1], B2 = lp.B[2];
	// all-pole impulse response hap[n], n >= -1
	std::vector<double> hap(os + 2);
	auto H = [&](int n) -> double & { return hap[n + 1]; };
	H(-1) = 0.0; H(0) = 1.0;
	for(int n = 1; n <= os; n++) H(n) = B1 * H(n - 1) + B2 * (n >= 2 ? H(n - 2) : 0.0);
	// State basis.  In the basis of the recursion itself, t = (v[n], v[n-1]), the two components are nearly parallel (poles at
	// radius 0.985, 1 degree off the real axis: M^20 = [[15.1, -14.2], [14.7, -13.7]]), and every P t in single precision cancels
	// four digits: 1.2e-4 of the signal's peak at oversample 20 - four times the rounding noise of the reference's own scan and the
	// largest part of what used to separate this path's stream from the reference's.  In the NORMAL form - t' = T t with the
	// columns of T^-1 the real and imaginary parts of M's eigenvector (lambda, 1) - the step matrix is a rotation scaled by
	// |lambda|^os, nothing cancels, and the block form's own rounding falls to 2e-7 of the peak (dev/k1_state_basis.py measures both).
	// Only constants change: taps T (g0, g1), P' = T P T^-1 (and its powers), (c0, c1) T^-1.  Real poles (not this filter): T = 1.
	Mat2 T{1, 0, 0, 1}, Ti{1, 0, 0, 1};
	if(B1 * B1 + 4.0 * B2 < 0.0) {
		const double r = sqrt(-B2), ct = B1 / (2.0 * r), st = sqrt(1.0 - ct * ct);
		Ti = Mat2{r * ct, r * st, 1.0, 0.0};
		T = Mat2{0.0, 1.0, 1.0 / (r * st), -ct / st};
	}
	bf.basis[0] = (float)T.a; bf.basis[1] = (float)T.b; bf.basis[2] = (float)T.c; bf.basis[3] = (float)T.d;
	for(int j = 0; j < os; j++) {
		const double h0 = H(os - 1 - j), h1 = H(os - 2 - j);
		bf.g0[j] = (float)(T.a * h0 + T.b * h1); bf.g1[j] = (float)(T.c * h0 + T.d * h1);
	}
	Mat2 M{B1, B2, 1.0, 0.0};
	Mat2 P = mul(mul(T, mpow(M, os)), Ti);
	bf.P[0] = (float)P.a; bf.P[1] = (float)P.b; bf.P[2] = (float)P.c; bf.P[3] = (float)P.d;
	const double c0v = A0 + A2 / B2, c1v = A1 - A2 * B1 / B2, c2 = -A2 / B2;     // y = c0v v[n] + c1v v[n-1] + c2 xm[n]
	const double c0 = c0v * Ti.a + c1v * Ti.c, c1 = c0v * Ti.b + c1v * Ti.d;     // ... = (c0, c1) t' + c2 xm[n]
	bf.c0 = (float)c0; bf.c1 = (float)c1; bf.c2 = (float)c2;
	Mat2 Pi{1, 0, 0, 1};
	for(int i = 0; i <= kFixW; i++) {
		bf.Ppow[i][0] = (float)Pi.a; bf.Ppow[i][1] = (float)Pi.b; bf.Ppow[i][2] = (float)Pi.c; bf.Ppow[i][3] = (float)Pi.d;
		Pi = mul(Pi, P);
		if(i < kFixW) { bf.cP[i][0] = (float)(c0 * Pi.a + c1 * Pi.c); bf.cP[i][1] = (float)(c0 * Pi.b + c1 * Pi.d); }
	}
	{
		const Mat2 Q1 = mpow(P, run);
		Mat2 Ql = Q1;
		for(int l = 0; l < 64; l++) {
			bf.Qpow[l][0] = (float)Ql.a; bf.Qpow[l][1] = (float)Ql.b; bf.Qpow[l][2] = (float)Ql.c; bf.Qpow[l][3] = (float)Ql.d;
			Ql = mul(Ql, Q1);
		}
	}
	Mat2 Q = mpow(P, run);
	for(int d = 0; d < 6; d++) {
		bf.Q[d][0] = (float)Q.a; bf.Q[d][1] = (float)Q.b; bf.Q[d][2] = (float)Q.c; bf.Q[d][3] = (float)Q.d;
		Q = mul(Q, Q);
	}
	return bf;
}


}  // namespace vdl2
