/*
 * vdl2_oracle.c - TEST INFRASTRUCTURE ONLY (see vdl2_oracle.h).
 *
 * A from-scratch CPU restatement of dumpvdl2's per-channel hot path
 * (reference @ /root/reference, v2.6.0; file:line cited on every function):
 *   src/demod.c      NCO mix, 2-pole Chebyshev IIR, decimation, preamble sync, D8PSK slicer
 *   src/chebyshev.c  filter design
 *   src/decode.c     header code, burst geometry, de-interleave, burst FSM
 *   src/bitstream.c  descrambler, HDLC un-stuffing
 *   src/rs.c + src/libfec/decode_rs.h   RS(255,249) errors-and-erasures decoder
 * The floating-point statements keep the reference's operation order and
 * its float/double promotions (build with -ffp-contract=off, no -ffast-math),
 * so on one channel this is the reference's sequential scan.
 *
 * PARITY PIN (what this oracle has been checked against):
 *  1. the reference's only test vector test/vdl2_model_16b_1050kHz.wav
 *     (a copy is tests/golden/vdl2_model_16b_1050kHz.wav): one burst, header
 *     0x2f7c0e, TL 4029 bits, 3 RS blocks, two FCS-good AVLC frames of 314 and
 *     186 octets containing the two strings the reference's CI greps for
 *     (.github/workflows/build.yml:16-18)  -> tests/test_oracle_golden.py
 *  2. the reference's own RS decoder, compiled unmodified from
 *     src/libfec/{decode_rs_char,init_rs_char}.c into oracle/_ref/libfec_ref.so
 *     (oracle/Makefile) and compared on random error/erasure patterns incl.
 *     beyond-capacity ones                   -> tests/test_oracle_rs.py
 *  3. coefficient known answers recorded from the reference build in
 *     SURVEY.md 8.2 a6                        -> tests/test_oracle_golden.py
 * demod.c/decode.c/bitstream.c/chebyshev.c themselves are NOT buildable in
 * this image (they include a cmake-generated config.h plus glib and libacars
 * headers), so 1-3 are the pin; no stand-in headers were written.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>
#include "vdl2_oracle.h"

/* ---- constants: dumpvdl2.h:37-50, demod.c:37-48, decode.c:45-50 ---- */
#define K_RS_DATA        249
#define K_RS_TOTAL       255
#define K_RS_PAR         6
#define K_TL_BITS        17
#define K_HDR_PAR_BITS   5
#define K_HDR_BITS       25
#define K_PREAMBLE       16
#define K_SPS            10
#define K_RING           160
#define K_SYMRATE        10500
#define K_FIFO_BITS      32768u
#define K_PHERR_BIG      1000.f
#define K_SYNC_SKIP      3
#define K_SYNC_THR       4.f
#define K_MAG_LP         0.9f
#define K_NF_LP          0.85f
#define K_CUTOFF_HZ      8000
#define K_RIPPLE_PCT     0.5f
#define K_MAX_TL         0x3FFFu
#define K_MAX_TL_CORR    0x1FFFu
#define K_LFSR_IV        0x6959u

enum { ST_SEARCH = 0, ST_LOCKED = 1 };          /* DM_INIT / DM_SYNC, dumpvdl2.h:299 */
enum { DS_HEADER = 0, DS_DATA = 1, DS_IDLE = 2 }; /* decoder_states, dumpvdl2.h:300 */

typedef struct {
	/* process_samples() locals, demod.c:289-298 */
	float xr[3], xi[3], yr[3], yi[3];
	int decim_cnt;
	/* vdl2_channel_t, dumpvdl2.h:321-352 */
	float ring[K_RING];
	float prev_phi, prev_slope, slope;
	float pherr[3];
	float ppm_error;
	float mag_lp, mag_nf;
	float frame_pwr;
	int nfcnt, ring_idx, frame_pwr_cnt, sclk, offset_tuning, fec_fixed;
	int dstate, decstate;
	uint32_t freq, nco_phi, nco_dphi, oversample;
	uint32_t want_bits, tl_bits, tl_octets, last_blk_octets, fec_octets, nblocks, syndrome;
	uint16_t lfsr;
	/* bit FIFO (bitstream_t, one byte per bit) */
	uint8_t *bits;
	uint32_t b_start, b_end, b_descr;
	/* bookkeeping that is not in the reference */
	int64_t dsample;            /* decimated samples seen so far - 1 */
	int64_t sync_sample;
	int64_t bursts;
	uint64_t cnt[VDL2O_NUM_COUNTERS];
	/* frames produced by this channel during the current block */
	vdl2o_frame *fr; size_t nfr, capfr;
	uint8_t *oct; size_t noct, capoct;
} chan_t;

struct vdl2o_ctx {
	int nchan, fmt;
	uint32_t oversample;
	float max_ppm;
	float A[3], B[3];
	float sin_t[257], cos_t[257];
	float lrx[K_PREAMBLE], lr_den;
	float u8_levels[256];
	chan_t *ch;
	float *sbuf; uint32_t sbuf_cap, sbuf_len;
	vdl2o_frame *fr; size_t nfr, capfr;
	uint8_t *oct; size_t noct, capoct;
	int trace_chan; float *trace; size_t trace_cap, trace_n;
	float *trace_all; size_t trace_all_cap;   /* [nchan][cap] complex, every channel */
};

/* ======================================================================
 * GF(2^8) / Reed-Solomon: rs.c:27-49, libfec/init_rs.h, libfec/decode_rs.h
 * init_rs_char(8, 0x187, 120, 1, 6, 0): field poly 0x187, first root 120,
 * primitive element step 1, 6 roots, no padding.
 * ==================================================================== */
#define GF_NN 255
#define GF_A0 255   /* log(0) marker */
#define RS_FCR 120
static uint8_t gf_exp[256], gf_log[256], rs_gen[K_RS_PAR + 1];
static int gf_ready;
static pthread_once_t gf_once = PTHREAD_ONCE_INIT;

static inline int mod255(int x) { /* rs-common.h:20-26 */
	while(x >= GF_NN) { x -= GF_NN; x = (x >> 8) + (x & GF_NN); }
	return x;
}

static void gf_setup(void) { /* init_rs.h:52-66 (tables), :86-101 (generator) */
	int sr = 1;
	gf_log[0] = GF_A0; gf_exp[GF_A0] = 0;
	for(int i = 0; i < GF_NN; i++) {
		gf_log[sr] = (uint8_t)i; gf_exp[i] = (uint8_t)sr;
		sr <<= 1;
		if(sr & 0x100) sr ^= 0x187;
		sr &= GF_NN;
	}
	/* g(x) = prod_{i=0..5} (x + alpha^(120+i)), kept in polynomial form, rs_gen[j] = coeff of x^j */
	uint8_t g[K_RS_PAR + 1]; memset(g, 0, sizeof g); g[0] = 1;
	for(int i = 0; i < K_RS_PAR; i++) {
		int root = RS_FCR + i;
		g[i + 1] = 1;
		for(int j = i; j > 0; j--)
			g[j] = g[j - 1] ^ (g[j] ? gf_exp[mod255(gf_log[g[j]] + root)] : 0);
		g[0] = gf_exp[mod255(gf_log[g[0]] + root)];
	}
	memcpy(rs_gen, g, sizeof g);
	gf_ready = 1;
}

static inline uint8_t gf_mul(uint8_t a, uint8_t b) {
	if(a == 0 || b == 0) return 0;
	return gf_exp[mod255(gf_log[a] + gf_log[b])];
}

/* Systematic encoder (no counterpart on the reference's receive path; the
 * parity definition follows from decode_rs.h:82-93: data[0] is the
 * highest-degree coefficient, parity = remainder of data(x)*x^6 / g(x)). */
void vdl2o_rs_encode(const uint8_t data[249], uint8_t parity[6]) {
	pthread_once(&gf_once, gf_setup);
	uint8_t rem[K_RS_PAR]; memset(rem, 0, sizeof rem); /* rem[0] = highest degree */
	for(int i = 0; i < K_RS_DATA; i++) {
		uint8_t fb = data[i] ^ rem[0];
		for(int j = 0; j < K_RS_PAR - 1; j++)
			rem[j] = rem[j + 1] ^ gf_mul(fb, rs_gen[K_RS_PAR - 1 - j]);
		rem[K_RS_PAR - 1] = gf_mul(fb, rs_gen[0]);
	}
	memcpy(parity, rem, K_RS_PAR);
}

/* decode_rs_char(): libfec/decode_rs.h:71-298.  Returns the number of
 * corrected symbols (erasures included) or -1.  era[] lists erased
 * positions; on return it holds the located positions, as in the reference. */
static int rs_decode_block(uint8_t *d, int *era, int n_era) {
	uint8_t lam[K_RS_PAR + 1], syn[K_RS_PAR], b[K_RS_PAR + 1], t[K_RS_PAR + 1], om[K_RS_PAR + 1];
	uint8_t root[K_RS_PAR], reg[K_RS_PAR + 1], loc[K_RS_PAR];
	int count;

	/* syndromes by Horner over the 255 received symbols (:82-93) */
	for(int i = 0; i < K_RS_PAR; i++) syn[i] = d[0];
	for(int j = 1; j < GF_NN; j++)
		for(int i = 0; i < K_RS_PAR; i++)
			syn[i] = (syn[i] == 0) ? d[j] : (uint8_t)(d[j] ^ gf_exp[mod255(gf_log[syn[i]] + (RS_FCR + i))]);
	int any = 0;
	for(int i = 0; i < K_RS_PAR; i++) { any |= syn[i]; syn[i] = gf_log[syn[i]]; } /* index form (:96-100) */
	if(!any) { count = 0; goto done; }                                             /* (:102-108) */

	memset(lam, 0, sizeof lam); lam[0] = 1;
	if(n_era > 0) { /* erasure locator (:113-123) */
		lam[1] = gf_exp[mod255(GF_NN - 1 - era[0])];
		for(int i = 1; i < n_era; i++) {
			int u = mod255(GF_NN - 1 - era[i]);
			for(int j = i + 1; j > 0; j--) {
				uint8_t lg = gf_log[lam[j - 1]];
				if(lg != GF_A0) lam[j] ^= gf_exp[mod255(u + lg)];
			}
		}
	}
	for(int i = 0; i <= K_RS_PAR; i++) b[i] = gf_log[lam[i]];

	/* Berlekamp-Massey (:166-207) */
	int r = n_era, el = n_era;
	while(++r <= K_RS_PAR) {
		uint8_t disc = 0;
		for(int i = 0; i < r; i++)
			if(lam[i] != 0 && syn[r - i - 1] != GF_A0)
				disc ^= gf_exp[mod255(gf_log[lam[i]] + syn[r - i - 1])];
		disc = gf_log[disc];
		if(disc == GF_A0) {
			memmove(&b[1], b, K_RS_PAR); b[0] = GF_A0;
		} else {
			t[0] = lam[0];
			for(int i = 0; i < K_RS_PAR; i++)
				t[i + 1] = (b[i] != GF_A0) ? (uint8_t)(lam[i + 1] ^ gf_exp[mod255(disc + b[i])]) : lam[i + 1];
			if(2 * el <= r + n_era - 1) {
				el = r + n_era - el;
				for(int i = 0; i <= K_RS_PAR; i++)
					b[i] = (lam[i] == 0) ? GF_A0 : (uint8_t)mod255(gf_log[lam[i]] - disc + GF_NN);
			} else {
				memmove(&b[1], b, K_RS_PAR); b[0] = GF_A0;
			}
			memcpy(lam, t, sizeof lam);
		}
	}
	int deg_lam = 0;
	for(int i = 0; i <= K_RS_PAR; i++) { lam[i] = gf_log[lam[i]]; if(lam[i] != GF_A0) deg_lam = i; } /* (:210-215) */

	/* Chien search (:217-240); iprim = 1 so k starts at 0 and steps by 1 */
	memcpy(&reg[1], &lam[1], K_RS_PAR);
	count = 0;
	for(int i = 1, k = 0; i <= GF_NN; i++, k = mod255(k + 1)) {
		uint8_t q = 1;
		for(int j = deg_lam; j > 0; j--)
			if(reg[j] != GF_A0) { reg[j] = (uint8_t)mod255(reg[j] + j); q ^= gf_exp[reg[j]]; }
		if(q != 0) continue;
		root[count] = (uint8_t)i; loc[count] = (uint8_t)k;
		if(++count == deg_lam) break;
	}
	if(deg_lam != count) { count = -1; goto done; } /* (:241-248) */

	/* omega = syn*lambda mod x^6 (:253-261) */
	int deg_om = deg_lam - 1;
	for(int i = 0; i <= deg_om; i++) {
		uint8_t acc = 0;
		for(int j = i; j >= 0; j--)
			if(syn[i - j] != GF_A0 && lam[j] != GF_A0)
				acc ^= gf_exp[mod255(syn[i - j] + lam[j])];
		om[i] = gf_log[acc];
	}
	/* Forney (:267-291) */
	for(int j = count - 1; j >= 0; j--) {
		uint8_t num1 = 0, den = 0;
		for(int i = deg_om; i >= 0; i--)
			if(om[i] != GF_A0) num1 ^= gf_exp[mod255(om[i] + i * root[j])];
		uint8_t num2 = gf_exp[mod255(root[j] * (RS_FCR - 1) + GF_NN)];
		int top = (deg_lam < K_RS_PAR - 1 ? deg_lam : K_RS_PAR - 1) & ~1;
		for(int i = top; i >= 0; i -= 2)
			if(lam[i + 1] != GF_A0) den ^= gf_exp[mod255(lam[i + 1] + i * root[j])];
		if(num1 != 0)
			d[loc[j]] ^= gf_exp[mod255(gf_log[num1] + gf_log[num2] + GF_NN - gf_log[den])];
	}
done:
	if(era != NULL)
		for(int i = 0; i < count; i++) era[i] = loc[i];
	return count;
}

/* rs_verify(): rs.c:32-49 */
int vdl2o_rs_decode(uint8_t block[255], int fec_octets) {
	pthread_once(&gf_once, gf_setup);
	if(fec_octets == 0) return 0;
	int n_era = K_RS_TOTAL - K_RS_DATA - fec_octets;
	if(n_era > 0) {
		int era[K_RS_PAR];
		for(int i = 0; i < n_era; i++) era[i] = K_RS_DATA + fec_octets + i;
		return rs_decode_block(block, era, n_era);
	}
	return rs_decode_block(block, NULL, n_era);
}

/* ======================================================================
 * CRC-16/X.25 step used only by tests and the synthetic generator
 * (same function as crc.c:21-64; bitwise, reflected poly 0x8408)
 * ==================================================================== */
uint16_t vdl2o_crc16(const uint8_t *data, uint32_t len, uint16_t init) {
	uint16_t c = init;
	for(uint32_t i = 0; i < len; i++) {
		c ^= data[i];
		for(int k = 0; k < 8; k++) c = (c & 1) ? (uint16_t)((c >> 1) ^ 0x8408) : (uint16_t)(c >> 1);
	}
	return c;
}

/* ======================================================================
 * The AVLC front door: what the decoder thread does first with a frame
 * (decode.c:466, avlc_parse() avlc.c:163-236, parse_dlc_addr() avlc.c:158-161,
 * reverse() bitstream.c:152-164).  Returns 0 = parsed on, 1 = too short, 2 = bad FCS;
 * *dir: 0 none, 1 air2gnd, 2 air2air, 3 air2all, 4 gnd2air, 5 gnd2gnd, 6 gnd2all.
 * ==================================================================== */
static uint32_t dlc_addr(const uint8_t *b) {
	uint32_t v = (uint32_t)(b[0] >> 1) | ((uint32_t)b[1] << 6) | ((uint32_t)b[2] << 13) | ((uint32_t)(b[3] & 0xfe) << 20);
	/* reverse(v, 28): bit i of the 32-bit word goes to bit 31-i, then the top 28 bits are kept */
	uint32_t r = v; int s = 31;
	for(v >>= 1; v; v >>= 1) { r <<= 1; r |= v & 1; s--; }
	r <<= s;
	r >>= 32 - 28;
	return r & ~(~0u << 28);
}

int vdl2o_avlc_screen(const uint8_t *buf, uint32_t len, uint32_t *dst, uint32_t *src, int *dir) {
	*dst = *src = 0; *dir = 0;
	if(len < 11) return 1;                                    /* MIN_AVLC_LEN, avlc.c:39,168 */
	if(vdl2o_crc16(buf, len, 0xFFFFu) != 0xF0B8u) return 2;  /* GOOD_FCS, avlc.c:40,178-187 */
	*dst = dlc_addr(buf); *src = dlc_addr(buf + 4);
	const unsigned st = (*src >> 24) & 7u, dt = (*dst >> 24) & 7u;   /* a_addr.type, avlc.h:29-42 (little endian) */
	if(st == 1) *dir = (dt == 4 || dt == 5) ? 1 : dt == 1 ? 2 : dt == 7 ? 3 : 0;           /* avlc.c:203-218 */
	else if(st == 4 || st == 5) *dir = dt == 1 ? 4 : (dt == 4 || dt == 5) ? 5 : dt == 7 ? 6 : 0;   /* avlc.c:220-236 */
	return 0;
}

/* ======================================================================
 * Header block code (25,20): decode.c:55-122
 * ==================================================================== */
static const uint32_t hdr_H[K_HDR_PAR_BITS] = { /* parity-check rows, decode.c:55-61 */
	0x001FFF0u, 0x07E1FE8u, 0x18E61E4u, 0x1B6A662u, 0x0D3CAA1u
};
static uint32_t hdr_fix[32];     /* syndrome -> error pattern (decode.c:63-96) */
static uint32_t hdr_weight[32];  /* syndrome -> number of flipped bits (decode.c:98-100) */
static pthread_once_t hdr_once = PTHREAD_ONCE_INIT;

static uint32_t hdr_syndrome(uint32_t w) { /* decode.c:102-117 */
	uint32_t s = 0;
	for(int i = 0; i < K_HDR_PAR_BITS; i++)
		s |= (uint32_t)(__builtin_popcount(w & hdr_H[i]) & 1) << (K_HDR_PAR_BITS - 1 - i);
	return s;
}

/* The reference hard-codes the table; it is the coset-leader table of H:
 * every single-bit pattern owns its syndrome, and the six syndromes left
 * over (3,5,13,18,20,23) are assigned the two-bit patterns listed at
 * decode.c:67,69,77,82,84,87.  Rebuilt here from H plus those six pairs. */
static void hdr_setup(void) {
	memset(hdr_fix, 0, sizeof hdr_fix); memset(hdr_weight, 0, sizeof hdr_weight);
	for(int bit = 0; bit < K_HDR_BITS; bit++) {
		uint32_t e = 1u << bit, s = hdr_syndrome(e);
		hdr_fix[s] = e; hdr_weight[s] = 1;
	}
	static const uint8_t pairs[6][2] = { {23,2}, {23,1}, {24,20}, {23,14}, {23,15}, {24,16} };
	for(int i = 0; i < 6; i++) {
		uint32_t e = (1u << pairs[i][0]) | (1u << pairs[i][1]), s = hdr_syndrome(e);
		hdr_fix[s] = e; hdr_weight[s] = 2;
	}
}

uint32_t vdl2o_header_decode(uint32_t *word) { /* decode.c:111-122 */
	pthread_once(&hdr_once, hdr_setup);
	uint32_t s = hdr_syndrome(*word);
	*word ^= hdr_fix[s];
	return s;
}

uint32_t vdl2o_header_parity(uint32_t upper20) {
	/* low 5 columns of H are the identity, so parity bit i = parity of the upper-20 part of row i */
	uint32_t w = (upper20 & 0xFFFFFu) << K_HDR_PAR_BITS, p = 0;
	for(int i = 0; i < K_HDR_PAR_BITS; i++)
		p |= (uint32_t)(__builtin_popcount(w & hdr_H[i] & ~0x1Fu) & 1) << (K_HDR_PAR_BITS - 1 - i);
	return p;
}

static int fec_octets_for(uint32_t len) { /* get_fec_octetcount(), decode.c:124-133 */
	return len < 3 ? 0 : len < 31 ? 2 : len < 68 ? 4 : 6;
}

/* ======================================================================
 * Chebyshev low-pass design, 2 poles: chebyshev.c:32-119
 * ==================================================================== */
void vdl2o_chebyshev(float fc, float ripple, float Aout[3], float Bout[3]) {
	const int np = 2;
	float rp, ip;
	sincosf((float)(M_PI / (2 * np) + (1 - 1) * M_PI / np), &ip, &rp); /* :35 */
	rp = -rp;
	if(ripple != 0.f) { /* :37-46 */
		float es = sqrtf(powf(100.f / (100.f - ripple), 2.f) - 1.f);
		float vx = (1.f / np) * logf((1.f / es) + sqrtf(1.f / (es * es) + 1.f));
		float kx = (1.f / np) * logf((1.f / es) + sqrtf(1.f / (es * es) - 1.f));
		kx = (expf(kx) + expf(-kx)) / 2.f;
		rp *= ((expf(vx) - expf(-vx)) / 2.f) / kx;
		ip *= ((expf(vx) + expf(-vx)) / 2.f) / kx;
	}
	float t = 2.f * tanf(0.5f);                /* :48 */
	float w = 2.f * M_PI * fc;                 /* :49  (double product narrowed) */
	float m = rp * rp + ip * ip;
	float d = 4.f - 4.f * rp * t + m * t * t;
	float x0 = t * t / d, x1 = 2.f * x0, x2 = x0;
	float y1 = (8.f - 2.f * m * t * t) / d;
	float y2 = (-4.f - 4.f * rp * t - m * t * t) / d;
	float k = sinf(0.5f - w / 2.f) / sinf(0.5f + w / 2.f); /* :58 */
	d = 1 + y1 * k - y2 * k * k;
	float a0 = (x0 - x1 * k + x2 * k * k) / d;
	float a1 = (-2.f * x0 * k + x1 + x1 * k * k - 2.f * x2 * k) / d;
	float a2 = (x0 * k * k - x1 * k + x2) / d;
	float b1 = (2.f * k + y1 + y1 * k * k - 2.f * y2 * k) / d;
	float b2 = (-(k * k) - y1 * k + y2) / d;
	/* chebyshev_lpf_init() cascades the single stage into A/B (:85-101): with one
	 * stage A = (a0,a1,a2), B = (-0, b1, b2); then unity DC gain (:102-110). */
	float A[3] = { a0 * 1.f + a1 * 0.f + a2 * 0.f, a0 * 0.f + a1 * 1.f + a2 * 0.f, a0 * 0.f + a1 * 0.f + a2 * 1.f };
	float B[3] = { -0.f, -(0.f - b1 * 1.f - b2 * 0.f), -(0.f - b1 * 0.f - b2 * 1.f) };
	float sa = 0.f, sb = 0.f;
	for(int i = 0; i < 3; i++) { sa += A[i]; sb += B[i]; }
	float gain = sa / (1.f - sb);
	for(int i = 0; i < 3; i++) { Aout[i] = A[i] / gain; Bout[i] = B[i]; }
}

/* ======================================================================
 * context / channel set-up
 * ==================================================================== */
static void chan_decoder_reset(chan_t *v) { /* decoder_reset(), demod.c:205-211 */
	v->decstate = DS_HEADER;
	v->want_bits = K_HDR_BITS;
	v->fec_fixed = 0;
	v->b_start = v->b_end = v->b_descr = 0;
}

static void chan_demod_reset(chan_t *v) { /* demod_reset(), demod.c:213-220 */
	chan_decoder_reset(v);
	v->sclk = 0;
	v->dstate = ST_SEARCH;
	v->pherr[1] = v->pherr[2] = K_PHERR_BIG;
	v->frame_pwr = 0.f;
	v->frame_pwr_cnt = 0;
}

vdl2o_ctx *vdl2o_create(uint32_t centerfreq, const uint32_t *freqs, int nchan,
		uint32_t oversample, int sample_fmt, float max_ppm) {
	pthread_once(&gf_once, gf_setup);
	pthread_once(&hdr_once, hdr_setup);
	vdl2o_ctx *c = calloc(1, sizeof *c);
	c->nchan = nchan; c->fmt = sample_fmt; c->oversample = oversample; c->max_ppm = max_ppm;
	c->trace_chan = -1;
	uint32_t fs = K_SYMRATE * K_SPS * oversample;                    /* dumpvdl2.c:1073 */
	/* input_lpf_init(), demod.c:367-370 */
	vdl2o_chebyshev((float)K_CUTOFF_HZ / (float)fs, K_RIPPLE_PCT, c->A, c->B);
	/* sincosf_lut_init(), demod.c:372-377 */
	for(uint32_t i = 0; i < 256; i++)
		sincosf(2.0f * M_PI * (float)i / 256.0f, &c->sin_t[i], &c->cos_t[i]);
	c->sin_t[256] = c->sin_t[0]; c->cos_t[256] = c->cos_t[0];
	/* demod_sync_init(), demod.c:84-96 */
	float mean_x = 0.f; c->lr_den = 0.f;
	for(int i = 0; i < K_PREAMBLE; i++) mean_x += i;
	mean_x /= K_PREAMBLE;
	for(int i = 0; i < K_PREAMBLE; i++) {
		c->lrx[i] = i - mean_x;
		c->lr_den += (i - mean_x) * (i - mean_x);
	}
	/* process_buf_uchar_init(), demod.c:349-354 */
	for(int i = 0; i < 256; i++) c->u8_levels[i] = (i - 127.5f) / 127.5f;
	c->ch = calloc((size_t)nchan, sizeof(chan_t));
	for(int k = 0; k < nchan; k++) { /* vdl2_channel_init(), demod.c:379-392 */
		chan_t *v = &c->ch[k];
		v->bits = calloc(K_FIFO_BITS, 1);
		v->mag_nf = 2.0f;
		v->nco_dphi = (uint32_t)(int)(((float)centerfreq - (float)freqs[k]) / (float)fs * 256.0f * 65536.0f);
		v->offset_tuning = (centerfreq != freqs[k]);
		v->oversample = oversample;
		v->freq = freqs[k];
		v->dsample = -1;
		chan_demod_reset(v);
	}
	return c;
}

void vdl2o_destroy(vdl2o_ctx *c) {
	if(!c) return;
	for(int k = 0; k < c->nchan; k++) { free(c->ch[k].bits); free(c->ch[k].fr); free(c->ch[k].oct); }
	free(c->ch); free(c->sbuf); free(c->fr); free(c->oct); free(c);
}

/* ======================================================================
 * burst decoder: decode.c:173-384 + bitstream.c
 * ==================================================================== */
static void chan_emit_frame(const vdl2o_ctx *c, chan_t *v, int idx, const uint8_t *buf, uint32_t len) {
	/* decode_frame(), decode.c:173-194 */
	(void)c;
	if(v->nfr == v->capfr) { v->capfr = v->capfr ? 2 * v->capfr : 16; v->fr = realloc(v->fr, v->capfr * sizeof *v->fr); }
	if(v->noct + len > v->capoct) { while(v->noct + len > v->capoct) v->capoct = v->capoct ? 2 * v->capoct : 4096; v->oct = realloc(v->oct, v->capoct); }
	vdl2o_frame *f = &v->fr[v->nfr++];
	memset(f, 0, sizeof *f);
	f->freq = v->freq;
	f->frame_pwr_dbfs = 10.0f * log10f(v->frame_pwr);
	f->nf_pwr_dbfs = 20.0f * log10f(v->mag_nf + 0.001f);
	f->ppm_error = v->ppm_error;
	f->datalen_octets = v->tl_octets;
	f->synd_weight = hdr_weight[v->syndrome];
	f->num_fec_corrections = v->fec_fixed;
	f->idx = idx;
	f->len = len;
	f->octets_off = v->noct;
	f->burst_ord = v->bursts - 1;
	f->sync_sample = v->sync_sample;
	f->end_sample = v->dsample;
	memcpy(v->oct + v->noct, buf, len);
	v->noct += len;
}

static void fifo_descramble(chan_t *v) { /* bitstream_descramble(), bitstream.c:94-107 */
	if(v->b_descr < v->b_start) v->b_descr = v->b_start;
	uint16_t l = v->lfsr;
	for(uint32_t i = v->b_descr; i < v->b_end; i++) {
		uint8_t bit = (uint8_t)((l ^ (l >> 14)) & 1);
		l = (uint16_t)((l >> 1) | (bit << 14));
		v->bits[i] ^= bit;
	}
	v->lfsr = l;
	v->b_descr = v->b_end;
}

static int fifo_read_octets(chan_t *v, uint8_t *dst, uint32_t n) { /* bitstream_read_lsbfirst(…,8), bitstream.c:70-81 */
	if(v->b_start + 8 * n > v->b_end) return -1;
	for(uint32_t i = 0; i < n; i++) {
		uint8_t o = 0;
		for(int j = 0; j < 8; j++) o |= (uint8_t)((v->bits[v->b_start++] & 1) << j);
		dst[i] = o;
	}
	return 0;
}

/* deinterleave(), decode.c:135-163: scatter `len` octets column-major into rows of 255 */
static int scatter_columns(const uint8_t *in, uint32_t len, uint32_t rows, uint8_t (*out)[K_RS_TOTAL],
		uint32_t width, uint32_t offset) {
	if(rows == 0 || width == 0) return -1;
	uint32_t last = len % width;
	if(last == 0) last = width;
	if(width + offset > K_RS_TOTAL) return -2;
	if(len > rows * width) return -3;
	if(rows > 1 && len - last < (rows - 1) * width) return -4;
	if(last == 0 && len / width < rows) return -5;
	uint32_t r = 0, col = offset;
	last += offset;
	for(uint32_t i = 0; i < len; i++) {
		if(r == rows - 1 && col >= last) { out[r][col] = 0; r = 0; col++; }
		out[r++][col] = in[i];
		if(r == rows) { r = 0; col++; }
	}
	return 0;
}

/* bitstream_copy_next_frame(), bitstream.c:109-150, on a flat bit array.
 * *pos is src->start; end is src->end.  Returns 1 more / 0 last / -1 bad. */
static int next_hdlc_frame(const uint8_t *src, uint32_t *pos, uint32_t end, uint8_t *dst, uint32_t *dst_len) {
	for(;;) {
		int ones = 0, again = 0;
		uint32_t j = 0, dlen = 0, i;
		for(i = *pos; i < end; i++, (*pos)++) {
			uint8_t b = src[i];
			if(b == 0 && ones == 5) { ones = 0; continue; }
			if(b == 1 && ++ones > 6) return -1;
			dst[j] = b;
			if(b == 0) {
				if(ones == 6) {
					if(j == 7) { (*pos)++; again = 1; break; }   /* leading flag */
					if(j < 7) return -1;
					dlen = j - 7; (*pos)++;
					*dst_len = dlen;
					return *pos < end ? 1 : 0;
				}
				ones = 0;
			}
			j++; dlen++;
		}
		if(again) continue;
		*dst_len = dlen;
		return *pos < end ? 1 : 0;
	}
}

static void chan_decode_burst(const vdl2o_ctx *c, chan_t *v) { /* decode_vdl2_burst(), decode.c:196-384 */
	if(v->decstate == DS_HEADER) {
		v->lfsr = K_LFSR_IV;
		fifo_descramble(v);
		if(v->b_start + K_HDR_BITS > v->b_end) { v->cnt[VDL2O_CNT_ERR_NO_HEADER]++; v->decstate = DS_IDLE; return; }
		uint32_t hdr = 0;
		for(int i = 0; i < K_HDR_BITS; i++) hdr |= (uint32_t)(v->bits[v->b_start++] & 1) << (K_HDR_BITS - 1 - i);
		const uint32_t keep = (1u << (K_TL_BITS + K_HDR_PAR_BITS)) - 1;
		hdr &= keep;                                                     /* :209 */
		v->syndrome = vdl2o_header_decode(&hdr);
		if(v->syndrome == 0) v->cnt[VDL2O_CNT_CRC_GOOD]++;
		if((hdr & keep) != hdr) { v->cnt[VDL2O_CNT_CRC_BAD]++; v->decstate = DS_IDLE; return; } /* :215-220 */
		hdr >>= K_HDR_PAR_BITS;
		uint32_t tl = 0;                                                  /* reverse(…,17), :222 */
		for(int i = 0; i < K_TL_BITS; i++) if(hdr & (1u << i)) tl |= 1u << (K_TL_BITS - 1 - i);
		v->tl_bits = tl;
		if((v->syndrome != 0 && tl > K_MAX_TL_CORR) || tl > K_MAX_TL) { v->cnt[VDL2O_CNT_ERR_TOO_LONG]++; v->decstate = DS_IDLE; return; }
		v->tl_octets = tl / 8 + (tl % 8 != 0);
		v->nblocks = v->tl_octets / K_RS_DATA;
		v->fec_octets = v->nblocks * K_RS_PAR;
		v->last_blk_octets = v->tl_octets % K_RS_DATA;
		if(v->last_blk_octets != 0) v->nblocks++;
		v->fec_octets += (uint32_t)fec_octets_for(v->last_blk_octets);
		if(v->last_blk_octets == 0) v->last_blk_octets = K_RS_DATA;       /* :244-245 */
		if(v->fec_octets == 0) { v->cnt[VDL2O_CNT_ERR_NO_FEC]++; v->decstate = DS_IDLE; return; }
		v->want_bits = 8 * (v->tl_octets + v->fec_octets);
		v->decstate = DS_DATA;
		return;
	}
	if(v->decstate != DS_DATA) return;

	fifo_descramble(v);
	uint8_t *data = calloc(v->tl_octets ? v->tl_octets : 1, 1);
	uint8_t *fec = calloc(v->fec_octets, 1);
	uint8_t (*tab)[K_RS_TOTAL] = calloc(v->nblocks, K_RS_TOTAL);
	uint8_t *flat = NULL, *fbits = NULL;
	if(fifo_read_octets(v, data, v->tl_octets) < 0) { v->cnt[VDL2O_CNT_ERR_DATA_TRUNCATED]++; goto out; }
	if(fifo_read_octets(v, fec, v->fec_octets) < 0) { v->cnt[VDL2O_CNT_ERR_FEC_TRUNCATED]++; goto out; }
	if(scatter_columns(data, v->tl_octets, v->nblocks, tab, K_RS_DATA, 0) < 0) { v->cnt[VDL2O_CNT_ERR_DEINTERLEAVE_DATA]++; goto out; }
	uint32_t fec_rows = v->nblocks;
	if(fec_octets_for(v->last_blk_octets) == 0) fec_rows--;              /* :289-291 */
	if(scatter_columns(fec, v->fec_octets, fec_rows, tab, K_RS_PAR, K_RS_DATA) < 0) { v->cnt[VDL2O_CNT_ERR_DEINTERLEAVE_FEC]++; goto out; }

	/* RS per block, then re-serialise LSB-first (:304-334) */
	flat = calloc((size_t)v->nblocks * K_RS_DATA * 8 + 8, 1);
	uint32_t nbits = 0;
	for(uint32_t r = 0; r < v->nblocks; r++) {
		v->cnt[VDL2O_CNT_BLOCKS_PROCESSED]++;
		int npar = K_RS_PAR;
		if(r == v->nblocks - 1) npar = fec_octets_for(v->last_blk_octets);
		int ret = vdl2o_rs_decode(tab[r], npar);
		if(ret < 0) { v->cnt[VDL2O_CNT_ERR_FEC_BAD]++; goto out; }
		v->cnt[VDL2O_CNT_BLOCKS_FEC_OK]++;
		if(ret > 0) v->fec_fixed += ret - (K_RS_PAR - npar);                /* :322 */
		uint32_t n = (r != v->nblocks - 1) ? K_RS_DATA : v->last_blk_octets;
		if(nbits + 8 * n > K_FIFO_BITS) { v->cnt[VDL2O_CNT_ERR_BITSTREAM]++; goto out; }
		for(uint32_t i = 0; i < n; i++)
			for(int j = 0; j < 8; j++) flat[nbits++] = (tab[r][i] >> j) & 1;
	}
	if(v->tl_bits < nbits) nbits = v->tl_bits;                            /* :338-342 */

	fbits = calloc(nbits + 8, 1);
	uint32_t pos = 0;
	int ret, nframes = 0;
	for(;;) {                                                            /* :345-366 */
		uint32_t flen = 0;
		ret = next_hdlc_frame(flat, &pos, nbits, fbits, &flen);
		if(ret < 0) break;
		if(flen % 8 != 0) { v->cnt[VDL2O_CNT_ERR_TRUNCATED_OCTETS]++; goto out; }
		uint32_t fo = flen / 8;
		memset(data, 0, fo);
		for(uint32_t i = 0; i < fo; i++)
			for(int j = 0; j < 8; j++) data[i] |= (uint8_t)(fbits[8 * i + j] << j);
		v->cnt[VDL2O_CNT_MSG_GOOD]++;
		chan_emit_frame(c, v, nframes, data, fo);
		nframes++;
		if(ret == 0) break;
	}
	if(ret < 0) { v->cnt[VDL2O_CNT_ERR_UNSTUFF]++; goto out; }
	if(v->frame_pwr > 1.0f) v->cnt[VDL2O_CNT_MSG_GOOD_LOUD]++;
out:
	free(data); free(fec); free(tab); free(flat); free(fbits);
	v->decstate = DS_IDLE;
}

/* ======================================================================
 * preamble search + D8PSK slicer: demod.c:98-286
 * ==================================================================== */
static float parabola_vertex(float x, int d, float y1, float y2, float y3) { /* calc_para_vertex(), demod.c:98-103 */
	float denom = (float)(d * 2 * d * (-d));
	float a = (x * (y2 - y1) + (x - d) * (y1 - y3) + (x - 2 * d) * (y3 - y2)) / denom;
	float b = (x * x * (y1 - y2) + (x - d) * (x - d) * (y3 - y1) + (x - 2 * d) * (x - 2 * d) * (y2 - y3)) / denom;
	return -b / (2 * a);
}

static const float preamble_phase[K_PREAMBLE] = { /* demod.c:107-124 */
	0 * M_PI / 4, 3 * M_PI / 4, -3 * M_PI / 4, 1 * M_PI / 4, 1 * M_PI / 4, 2 * M_PI / 4, 0 * M_PI / 4, 4 * M_PI / 4,
	-3 * M_PI / 4, 4 * M_PI / 4, -2 * M_PI / 4, 3 * M_PI / 4, 1 * M_PI / 4, -2 * M_PI / 4, -3 * M_PI / 4, 0 * M_PI / 4
};

static int chan_try_sync(const vdl2o_ctx *c, chan_t *v) { /* got_sync(), demod.c:105-198 */
	float e[K_PREAMBLE];
	float mean = 0.f, unwrap = 0.f;
	float prev = mean = e[0] = v->ring[(v->ring_idx + K_SPS) % K_RING] - preamble_phase[0];
	for(int i = 1; i < K_PREAMBLE; i++) {
		float cur = v->ring[(v->ring_idx + (i + 1) * K_SPS) % K_RING] - preamble_phase[i];
		float diff = cur - prev;
		prev = cur;
		if(diff > M_PI) unwrap -= 2.0f * M_PI;
		else if(diff < -M_PI) unwrap += 2.0f * M_PI;
		e[i] = cur + unwrap;
		mean += e[i];
	}
	mean /= K_PREAMBLE;
	for(int i = 0; i < K_PREAMBLE; i++) e[i] -= mean;
	float slope = 0.f;
	for(int i = 0; i < K_PREAMBLE; i++) slope += c->lrx[i] * e[i];
	slope /= c->lr_den;
	float r = 0.f;
	v->pherr[0] = 0.f;
	for(int i = 0; i < K_PREAMBLE; i++) {
		r = e[i] - slope * c->lrx[i];
		v->pherr[0] += r * r;
	}
	if(v->pherr[1] < K_SYNC_THR && v->pherr[0] > v->pherr[1]) {
		float vx = parabola_vertex(v->sclk, K_SYNC_SKIP, v->pherr[2], v->pherr[1], v->pherr[0]);
		v->sclk = -roundf(vx);
		int sp = v->ring_idx - v->sclk;
		if(sp < 0) sp += K_RING;
		v->prev_phi = v->ring[sp];
		v->slope = v->prev_slope;
		v->ppm_error = K_SYMRATE * v->slope / (2.0f * M_PI * v->freq) * 1e+6;
		v->pherr[1] = v->pherr[2] = K_PHERR_BIG;
		if(c->max_ppm && fabsf(v->ppm_error) > c->max_ppm) { v->cnt[VDL2O_CNT_PPM_REJECT]++; return 0; }
		return 1;
	}
	v->pherr[2] = v->pherr[1];
	v->pherr[1] = v->pherr[0];
	v->prev_slope = slope;
	return 0;
}

static void chan_demod(const vdl2o_ctx *c, chan_t *v, float re, float im) { /* demod(), demod.c:222-286 */
	static const uint8_t gray[8] = { 0, 1, 3, 2, 6, 7, 5, 4 };
	if(v->decstate == DS_IDLE) chan_demod_reset(v);
	if(v->dstate == ST_SEARCH) {
		v->ring_idx++; v->ring_idx %= K_RING;
		v->ring[v->ring_idx] = atan2(im, re);
		if(++v->sclk < K_SYNC_SKIP) return;
		v->sclk = 0;
		float mag = hypotf(re, im);
		v->mag_lp = v->mag_lp * K_MAG_LP + mag * (1.0f - K_MAG_LP);
		if(++v->nfcnt == 1000) {
			v->nfcnt = 0;
			v->mag_nf = K_NF_LP * v->mag_nf + (1.0f - K_NF_LP) * fminf(v->mag_lp, v->mag_nf) + 0.0001f;
		}
		if(chan_try_sync(c, v)) {
			v->cnt[VDL2O_CNT_SYNC_GOOD]++;
			v->dstate = ST_LOCKED;
			v->sync_sample = v->dsample;
			v->bursts++;
		}
		return;
	}
	/* ST_LOCKED */
	if(++v->sclk < K_SPS) return;
	v->sclk = 0;
	float phi = atan2(im, re);
	float dphi = phi - v->prev_phi - v->slope;
	if(dphi < 0) dphi += 2.0f * M_PI;
	else if(dphi > 2.0f * M_PI) dphi -= 2.0f * M_PI;
	dphi /= M_PI_4;
	int idx = (int)roundf(dphi) % 8;
	if(idx < 0) { v->cnt[VDL2O_CNT_SLICER_NEG_IDX]++; idx &= 7; } /* reference: out-of-bounds table read */
	float spwr = re * re + im * im;
	v->frame_pwr = (v->frame_pwr * v->frame_pwr_cnt + spwr) / (v->frame_pwr_cnt + 1);
	v->frame_pwr_cnt++;
	v->prev_phi = phi;
	if(v->b_end + 3 > K_FIFO_BITS) { chan_demod_reset(v); return; }        /* bitstream_append_msbfirst, bitstream.c:45-56 */
	for(int j = 2; j >= 0; j--) v->bits[v->b_end++] = (gray[idx] >> j) & 1;
	if(v->b_end - v->b_start >= v->want_bits) chan_decode_burst(c, v);
}

/* ======================================================================
 * the per-sample scan: process_samples(), demod.c:288-337
 * ==================================================================== */
static void chan_scan_block(vdl2o_ctx *c, int k) {
	chan_t *v = &c->ch[k];
	const float *sb = c->sbuf;
	const float A0 = c->A[0], A1 = c->A[1], A2 = c->A[2], B1 = c->B[1], B2 = c->B[2];
	const int tracing = (c->trace_chan == k);
	for(uint32_t i = 0; i < c->sbuf_len;) {
		v->xr[2] = v->xr[1]; v->xr[1] = v->xr[0];
		v->xi[2] = v->xi[1]; v->xi[1] = v->xi[0];
		v->yr[2] = v->yr[1]; v->yr[1] = v->yr[0];
		v->yi[2] = v->yi[1]; v->yi[1] = v->yi[0];
		float re = sb[i++], im = sb[i++];
		if(v->offset_tuning) {
			/* sincosf_lut(), demod.c:58-72 + multiply(), :200-203 */
			uint32_t idx = v->nco_phi >> 16;
			float fr = (float)(v->nco_phi & 0xffff) / 65536.0f;
			float s1 = c->sin_t[idx], s2 = c->sin_t[idx + 1];
			float sn = s1 + (s2 - s1) * fr;
			float c1 = c->cos_t[idx], c2 = c->cos_t[idx + 1];
			float cs = c1 + (c2 - c1) * fr;
			float mr = re * cs - im * sn;
			float mi = im * cs + re * sn;
			re = mr; im = mi;
			v->nco_phi += v->nco_dphi;
			v->nco_phi &= 0xffffff;
		}
		v->xr[0] = re; v->xi[0] = im;
		/* chebyshev_lpf_2pole(), demod.c:74-79 */
		float r = A0 * v->xr[0];
		r += A1 * v->xr[1] + A2 * v->xr[2];
		r += B1 * v->yr[1] + B2 * v->yr[2];
		v->yr[0] = r;
		r = A0 * v->xi[0];
		r += A1 * v->xi[1] + A2 * v->xi[2];
		r += B1 * v->yi[1] + B2 * v->yi[2];
		v->yi[0] = r;
		if(++v->decim_cnt == (int)v->oversample) {
			v->decim_cnt = 0;
			v->dsample++;
			if(c->trace_all && (size_t)v->dsample < c->trace_all_cap) {
				float *t = c->trace_all + 2 * ((size_t)k * c->trace_all_cap + (size_t)v->dsample);
				t[0] = v->yr[0]; t[1] = v->yi[0];
			}
			if(tracing && c->trace_n < c->trace_cap) {
				c->trace[2 * c->trace_n] = v->yr[0]; c->trace[2 * c->trace_n + 1] = v->yi[0]; c->trace_n++;
			}
			chan_demod(c, v, v->yr[0], v->yi[0]);
		}
	}
}

static void gather_block_frames(vdl2o_ctx *c);
typedef struct { vdl2o_ctx *c; int first, step; } scan_job;
static void *scan_thread(void *arg) {
	scan_job *j = arg;
	for(int k = j->first; k < j->c->nchan; k += j->step) chan_scan_block(j->c, k);
	return NULL;
}

void vdl2o_process(vdl2o_ctx *c, const uint8_t *buf, uint32_t len, int nthreads) {
	if(len == 0) return;
	/* process_buf_uchar()/process_buf_short(), demod.c:339-365 */
	uint32_t nfl = (c->fmt == VDL2O_FMT_S16LE) ? len / 2 : len;
	if(nfl + 1 > c->sbuf_cap) { c->sbuf_cap = nfl + 1; c->sbuf = realloc(c->sbuf, (size_t)c->sbuf_cap * sizeof(float)); c->sbuf[nfl] = 0.f; }
	if(c->fmt == VDL2O_FMT_S16LE) {
		const int16_t *p = (const int16_t *)buf;
		for(uint32_t i = 0; i < nfl; i++) c->sbuf[i] = (float)p[i] / 32768.0f;
	} else {
		for(uint32_t i = 0; i < nfl; i++) c->sbuf[i] = c->u8_levels[buf[i]];
	}
	c->sbuf_len = nfl & ~1u;  /* the reference over-reads one stale float on odd counts (SURVEY A-18); callers keep len%4==0 */
	if(nthreads <= 1 || c->nchan == 1) {
		for(int k = 0; k < c->nchan; k++) chan_scan_block(c, k);
	} else {
		if(nthreads > c->nchan) nthreads = c->nchan;
		pthread_t th[nthreads]; scan_job jobs[nthreads];
		for(int t = 0; t < nthreads; t++) { jobs[t] = (scan_job){ c, t, nthreads }; pthread_create(&th[t], NULL, scan_thread, &jobs[t]); }
		for(int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
	}
	gather_block_frames(c);
}

/* ======================================================================
 * whole-capture runs with persistent threads (the CPU baseline of bench.py)
 *
 * VDL2O_RUN_THREAD_PER_CHANNEL is the reference's own threading (dumpvdl2.c:117-135, demod.c:300-301,342-346,356-365):
 * one persistent thread per channel plus the producer, two pthread barriers of count N+1 per block - the producer waits on
 * `demods_ready` until every channel has finished the previous block, converts the new block serially while the channel
 * threads are parked on `samples_ready`, then releases them; after the last block one more wait on `demods_ready`
 * (dumpvdl2.c:1170).  VDL2O_RUN_WORKQUEUE is what a CPU implementation free to restructure would do with the same
 * per-channel scan: `nthreads` persistent workers, the conversion spread over them, channels handed out from a counter.
 * ==================================================================== */
typedef struct {
	vdl2o_ctx *c; int mode, nworkers;
	pthread_barrier_t demods_ready, samples_ready;
	volatile int done;
	const uint8_t *blk; uint32_t blk_len;     /* work-queue mode: raw block being converted */
	volatile int next_chan;
} run_shared;
typedef struct { run_shared *r; int id; } run_worker;

static void convert_range(vdl2o_ctx *c, const uint8_t *buf, uint32_t f0, uint32_t f1) {
	if(c->fmt == VDL2O_FMT_S16LE) {
		const int16_t *p = (const int16_t *)buf;
		for(uint32_t i = f0; i < f1; i++) c->sbuf[i] = (float)p[i] / 32768.0f;      /* demod.c:362-363 */
	} else {
		for(uint32_t i = f0; i < f1; i++) c->sbuf[i] = c->u8_levels[buf[i]];        /* demod.c:344-345 */
	}
}

static void gather_block_frames(vdl2o_ctx *c) {
	for(int k = 0; k < c->nchan; k++) {
		chan_t *v = &c->ch[k];
		for(size_t i = 0; i < v->nfr; i++) {
			if(c->nfr == c->capfr) { c->capfr = c->capfr ? 2 * c->capfr : 64; c->fr = realloc(c->fr, c->capfr * sizeof *c->fr); }
			vdl2o_frame f = v->fr[i];
			if(c->noct + f.len > c->capoct) { while(c->noct + f.len > c->capoct) c->capoct = c->capoct ? 2 * c->capoct : 65536; c->oct = realloc(c->oct, c->capoct); }
			memcpy(c->oct + c->noct, v->oct + f.octets_off, f.len);
			f.octets_off = c->noct; f.chan = k;
			c->noct += f.len;
			c->fr[c->nfr++] = f;
		}
		v->nfr = 0; v->noct = 0;
	}
}

static void *run_channel_thread(void *arg) {      /* process_samples(), demod.c:288-337: one channel, forever */
	run_worker *w = arg; run_shared *r = w->r;
	for(;;) {
		pthread_barrier_wait(&r->demods_ready);
		pthread_barrier_wait(&r->samples_ready);
		if(r->done) break;
		chan_scan_block(r->c, w->id);
	}
	return NULL;
}

static void *run_queue_thread(void *arg) {
	run_worker *w = arg; run_shared *r = w->r; vdl2o_ctx *c = r->c;
	for(;;) {
		pthread_barrier_wait(&r->demods_ready);         /* block handed over (or done) */
		if(r->done) break;
		const uint32_t nfl = (c->fmt == VDL2O_FMT_S16LE) ? r->blk_len / 2 : r->blk_len;
		const uint32_t per = (nfl + (uint32_t)r->nworkers - 1) / (uint32_t)r->nworkers;
		const uint32_t f0 = per * (uint32_t)w->id < nfl ? per * (uint32_t)w->id : nfl, f1 = f0 + per < nfl ? f0 + per : nfl;
		convert_range(c, r->blk, f0, f1);
		pthread_barrier_wait(&r->samples_ready);        /* every slice converted */
		for(;;) {
			const int k = __atomic_fetch_add(&r->next_chan, 1, __ATOMIC_RELAXED);
			if(k >= c->nchan) break;
			chan_scan_block(c, k);
		}
		pthread_barrier_wait(&r->samples_ready);        /* block done */
	}
	return NULL;
}

int vdl2o_run(vdl2o_ctx *c, const uint8_t *buf, uint64_t total_len, uint32_t block_bytes, int mode, int nthreads) {
	if(block_bytes == 0 || (mode != VDL2O_RUN_THREAD_PER_CHANNEL && mode != VDL2O_RUN_WORKQUEUE)) return -1;
	run_shared r; memset(&r, 0, sizeof r);
	r.c = c; r.mode = mode;
	const int nw = mode == VDL2O_RUN_THREAD_PER_CHANNEL ? c->nchan : (nthreads < 1 ? 1 : nthreads);
	r.nworkers = nw;
	const uint32_t maxfl = (c->fmt == VDL2O_FMT_S16LE) ? block_bytes / 2 : block_bytes;
	if(maxfl + 1 > c->sbuf_cap) { c->sbuf_cap = maxfl + 1; c->sbuf = realloc(c->sbuf, (size_t)c->sbuf_cap * sizeof(float)); }
	if(pthread_barrier_init(&r.demods_ready, NULL, (unsigned)nw + 1) || pthread_barrier_init(&r.samples_ready, NULL, (unsigned)nw + 1)) return -2;
	pthread_t *th = calloc((size_t)nw, sizeof *th); run_worker *ws = calloc((size_t)nw, sizeof *ws);
	pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 1 << 20);
	int started = 0;
	for(; started < nw; started++) {
		ws[started] = (run_worker){ &r, started };
		if(pthread_create(&th[started], &at, mode == VDL2O_RUN_THREAD_PER_CHANNEL ? run_channel_thread : run_queue_thread, &ws[started])) break;
	}
	pthread_attr_destroy(&at);
	if(started < nw) { fprintf(stderr, "vdl2o_run: could only start %d of %d threads\n", started, nw); abort(); }
	for(uint64_t off = 0; off < total_len; off += block_bytes) {
		const uint32_t len = (uint32_t)(total_len - off < block_bytes ? total_len - off : block_bytes);
		const uint32_t nfl = (c->fmt == VDL2O_FMT_S16LE) ? len / 2 : len;
		if(mode == VDL2O_RUN_THREAD_PER_CHANNEL) {
			pthread_barrier_wait(&r.demods_ready);            /* demod.c:360: every channel finished the previous block */
			gather_block_frames(c);
			convert_range(c, buf + off, 0, nfl);              /* serial, as in the reference */
			c->sbuf[nfl] = 0.f;
			c->sbuf_len = nfl & ~1u;
			pthread_barrier_wait(&r.samples_ready);           /* demod.c:364 */
		} else {
			r.blk = buf + off; r.blk_len = len; r.next_chan = 0;
			c->sbuf[nfl] = 0.f;
			c->sbuf_len = nfl & ~1u;
			pthread_barrier_wait(&r.demods_ready);
			pthread_barrier_wait(&r.samples_ready);
			pthread_barrier_wait(&r.samples_ready);
			gather_block_frames(c);
		}
	}
	if(mode == VDL2O_RUN_THREAD_PER_CHANNEL) {
		pthread_barrier_wait(&r.demods_ready);                /* dumpvdl2.c:1170: let the channels finish the last block */
		gather_block_frames(c);
		r.done = 1;
		pthread_barrier_wait(&r.samples_ready);
	} else {
		r.done = 1;
		pthread_barrier_wait(&r.demods_ready);
	}
	for(int t = 0; t < nw; t++) pthread_join(th[t], NULL);
	pthread_barrier_destroy(&r.demods_ready); pthread_barrier_destroy(&r.samples_ready);
	free(th); free(ws);
	return 0;
}

size_t vdl2o_num_frames(const vdl2o_ctx *c) { return c->nfr; }
const vdl2o_frame *vdl2o_frames(const vdl2o_ctx *c) { return c->fr; }
const uint8_t *vdl2o_octets(const vdl2o_ctx *c) { return c->oct; }
void vdl2o_clear_frames(vdl2o_ctx *c) { c->nfr = 0; c->noct = 0; }
void vdl2o_counters(const vdl2o_ctx *c, int chan, uint64_t out[VDL2O_NUM_COUNTERS]) { memcpy(out, c->ch[chan].cnt, sizeof c->ch[chan].cnt); }
void vdl2o_get_lpf(const vdl2o_ctx *c, float A[3], float B[3]) { memcpy(A, c->A, sizeof c->A); memcpy(B, c->B, sizeof c->B); }
uint32_t vdl2o_get_dphi(const vdl2o_ctx *c, int chan) { return c->ch[chan].nco_dphi; }
void vdl2o_get_sincos_lut(const vdl2o_ctx *c, float s[257], float co[257]) { memcpy(s, c->sin_t, sizeof c->sin_t); memcpy(co, c->cos_t, sizeof c->cos_t); }
void vdl2o_trace_decimated(vdl2o_ctx *c, int chan, float *dst, size_t cap) { c->trace_chan = chan; c->trace = dst; c->trace_cap = cap; c->trace_n = 0; }
size_t vdl2o_trace_count(const vdl2o_ctx *c) { return c->trace_n; }
void vdl2o_trace_all(vdl2o_ctx *c, float *dst, size_t cap_per_chan) { c->trace_all = dst; c->trace_all_cap = cap_per_chan; }
int64_t vdl2o_decimated_count(const vdl2o_ctx *c, int chan) { return c->ch[chan].dsample + 1; }
