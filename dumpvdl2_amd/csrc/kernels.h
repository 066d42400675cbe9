// kernels.h - the HIP kernels of the hot path (gfx950 / CDNA4, wave64).
//
//   k_chanfir   K1  NCO mix + 2-pole Chebyshev low-pass + decimate  (src/demod.c:58-79,200-203,302-329)
//   k_fixup     K2  segment-start fix-up as a kernel of its own (only with VDL2HIP_NO_FUSE: K1 normally does it itself)
//   k_carry         saves the < oversample input samples left over for the next block
//   k_sync      K3  phase (screening precision) + got_sync() metric of every decimated sample + candidate bitmap
//                   (src/demod.c:232,105-171); the exact double-precision phases only where a preamble is near
//   k_walk      K4  per-channel FSM walker (vdl2_core.h); k_walk_spec + k_walk_stitch: the same walk in speculative segments
//   k_nf        K4b noise-floor replay from the walker's evaluation log (src/demod.c:238-243)
//   k_burst     K5  wave-per-burst decoder (vdl2_core.h)
//   k_frame_finish  wave-per-frame: noise-floor figure + AVLC front-door checks (src/avlc.c:163-236)
//
// K1 is the only kernel that touches every input sample; everything after it runs at
// 1/oversample of that rate.  See DESIGN.md for the block-form derivation and the roofline.
// There is no stored phase stream: atan2 in double (demod.c:232,256) is evaluated where a decision reads a phase - the exact tier
// of K3, the walker, K5 - and K3's screening tier uses a single-precision phase of its own (vdl2_core.h: ChanView::Phi, phase_fast).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "vdl2_core.h"
#include "design.h"

// tuning knobs of K1 (overridable at build time for experiments: dev/gpu_k1_variants.sh)
#ifndef VDL2_K1_UNROLL
#define VDL2_K1_UNROLL 5
#endif

#ifndef VDL2_K1_WAVES_PER_EU
#define VDL2_K1_WAVES_PER_EU 3
#endif

namespace vdl2 {

constexpr int kK1Unroll = VDL2_K1_UNROLL;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ======================================================================
// Referee (vdl2_core.h "Referee"): the reference's own samples of a stretch of one channel's decimated stream.
//
// process_samples() (src/demod.c:302-329) is a sequential scan per channel: sincosf_lut() (:58-72), multiply() (:200-203),
// chebyshev_lpf_2pole() (:74-79) on I and Q, every product and sum rounded to float in exactly that order.  Two such scans over
// the same input started from different filter states become BIT-IDENTICAL after a while - the difference of two fp32
// trajectories of a contracting recursion does not shrink below an ulp, it hits zero: the waiting time is exponentially distributed,
// 1.7e4 input samples on average per component where the channel carries noise, 2.6e4 inside a burst (dev/iir_state_coalescence.c,
// dev/ref_short_runup.py - which also shows that until they meet the two differ by as much as the channeliser's stream does from
// either, 2e-5 .. 1e-4 of the signal: a short run-up buys nothing).  So the reference's trajectory over [n_lo, n_hi] is obtained by
// running ITS arithmetic from `warm` input samples earlier with a zero state.  How often that is not yet the reference's, measured
// on the device against the oracle's stream over 91 773 stretches of a 64-channel capture (dev/gpu_scan_soundness.py,
// profiles/r06_scan_soundness.txt): run-up 2^17: 1.9e-3 of the stretches; 2^18: 2.2e-5; 2^19 and 2^20: none.  The default is
// 3 * 2^16 = 196 608 (VDL2HIP_REF_WARM): a few in 10^4, for half as much again as 2^17 costs (rounds 5 and 6 ran 2^17 and took the
// share for 2.5e-4, from a witness count - see k_ref_scan_multi: a witness started at the same sample sees about half of them).
//
// One wavefront does it.  The part without a recursion - sample conversion, NCO, mixer, the three feed-forward taps - is done
// for 60-odd input samples at a time by the lanes; the recursion y = r0 + (B1 y1 + B2 y2) is uniform: every lane does the same
// packed (I, Q) arithmetic on values handed over with v_readlane.  ~24 ns per input sample: 3.3 ms per scan.
// ======================================================================
#ifndef VDL2_REF_PRIO
#define VDL2_REF_PRIO 3                   // the scan's wavefront among those of its SIMD (s_setprio)
#endif
#ifndef VDL2_K1_PRIO
#define VDL2_K1_PRIO 0
#endif
constexpr int kRefPieces = 6;              // stretches of raw input a feed can reach back into: the feed's own block + the history ring (it may wrap)
constexpr int kRefCache = 256;             // stretches of a channel already made exact (a speculative walker and the stitcher come by the same places; the burst decoder's second pass looks its listed stretches up here: with 64 - until round 6c - a 16 s feed full of weak bursts, 105 listed stretches per channel, pushed its own out before they were read and scanned them again, one by one)
struct RefPiece { const void *p; int64_t s0, n; };   // raw samples with absolute index s0 <= s < s0 + n, contiguous at p
struct RefChan {
	cf32 *y; uint32_t cap, mask;           // the decimated rings, [nchan][cap]
	const uint32_t *dphi;                  // NCO step per channel
	const uint8_t *mix;                    // per channel: offset_tuning (src/demod.c:386: centerfreq != freq)
	const Lut4 *lut;
	float A0, A1, A2, B1, B2;              // src/demod.c:55
	int32_t os, fmt, npiece, kinds;        // kinds: bit k set = requests of kind k (REF_CANDIDATE ...) are served
	RefPiece piece[kRefPieces];            // oldest first, contiguous; the last one is this feed's block
	int64_t warm;                          // input samples the scan starts before the stretch
	unsigned long long *done;              // [nchan][kRefCache] stretches made exact (see ref_exact_window_dev)
	uint32_t *done_n;                      // [nchan] entries written so far
	unsigned long long *dbg; int32_t dbg_chan, dbg_pad;   // development aid (-DVDL2_REF_DEBUG): event log of one channel, dbg[0] = entries written
	uint32_t *stats;                       // [0] scans run, [1] requests answered from the list, [2] requests refused (input no longer held), [3] scans with a shortened run-up, [4 + kind] scans by who asked, [7] channels walked again
};

// one raw sample as process_buf_short() / process_buf_uchar() convert it (src/demod.c:349-365)
__device__ __forceinline__ bool ref_raw_sample(const RefChan &r, int64_t s, float &re, float &im) {
	for(int j = r.npiece - 1; j >= 0; j--) {
		const RefPiece &pc = r.piece[j];
		if(s >= pc.s0 && s < pc.s0 + pc.n) {
			if(r.fmt == 1) { const uint32_t w = ((const uint32_t *)pc.p)[s - pc.s0]; re = (float)(int16_t)(w & 0xffff) / 32768.0f; im = (float)(int16_t)(w >> 16) / 32768.0f; }
			else { const uint16_t w = ((const uint16_t *)pc.p)[s - pc.s0]; re = ((float)(w & 0xff) - 127.5f) / 127.5f; im = ((float)(w >> 8) - 127.5f) / 127.5f; }
			return true;
		}
	}
	re = 0.f; im = 0.f;
	return false;
}

__device__ __forceinline__ float dpp_wave_shr1(float v);
__device__ __forceinline__ float dpp_wave_shr1_keep(float lane0, float v);
#if VDL2_DEVICE_PASS
// (no LDS, no memory traffic inside the recursion: the 64 feed-forward values of a block stay in the lanes that made them and are
// handed to the recursion with v_readlane; the recursion itself is uniform - every lane does the same packed (I, Q) arithmetic.
// The raw samples and NCO table entries of block k + 1 are fetched while block k's recursion runs.)
typedef __attribute__((address_space(1))) const uint32_t ref_gu32;
typedef __attribute__((address_space(1))) const uint16_t ref_gu16;
typedef __attribute__((address_space(1))) const v4f ref_gf4;
__device__ __forceinline__ float ref_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// the list of stretches of a channel made exact: (lo / 256 : 32 bits, length / 256 : 16, launch : 16).
// WHOSE entries may a kernel believe?  The entry is an agent-scope atomic, the samples behind it are plain stores that may still sit
// dirty in the writer's XCD-local L2 until the writing KERNEL ends: an entry counts only if its writer has ended before the reader
// began.  launch = 16 * feed + kind.  Kinds of the walk chain of feed s and when they run (vdl2hip.hip: launch_back / launch_rest):
//   8 scans ahead of the walk        own stream, after the feed's front: BESIDE the walk chains of the feeds before
//   1 speculative / single walk, 2 stitch                 walk stream
//   0 the noted decisions' scans, 3 check                 own stream, after stitch(s): beside the walks of the next one or two feeds
//   4 walk again (after check(s) AND the walks that went ahead of it), 9 / 10 feed s once more from a corrected start - as the first /
//     the second feed behind the one whose second walk corrected it: walk stream
// The walk stream runs (walk ahead by two feeds)  ... S(s) 4(s-2) 9(s-1) 10(s)  S(s+1) 4(s-1) 9(s) 10(s+1)  S(s+2) 4(s) ...  - by one:
// ... S(s) 4(s-1) 9(s)  S(s+1) 4(s) 9(s+1) ... - with 0/3 of feed s anywhere between S(s) and 4(s).  Kinds of the burst stream of feed
// s, after 4(s): 5 burst decoder (first pass), 6 its listed scans, 7 second pass; they run beside the walk chains of the feeds after s
// and beside other feeds' burst kernels.  Hence a reader (feed m, kind mk) believes an entry (feed e, kind ek), d = m - e - rules that
// hold for either depth:
//   4, 9: every walk-chain entry of an earlier feed (check(m - 1) has ended by then); of its own feed 4: everything, 9: {8, 1, 2}.
//   10: d >= 2: yes; d = 1: {8, 1, 2, 9, 10} (check(m - 1) may still run); d = 0: {8, 1, 2}.
//   1, 2, 0, 3: d >= 3: yes; d = 2: {8, 1, 2, 9, 10}; d = 1: {8, 1, 2}; d = 0: reader 1: {8}; 2: {8, 1}; 0: {8, 1, 2}; 3: {8, 1, 2, 0}.
//   burst kernel: walk-chain entries of its own and earlier feeds, and its own feed's earlier burst kinds.
//   the scans ahead of the walk (8) believe nobody; nobody believes an entry of his own launch or a burst kernel of another feed.
// (Round 5 believed every entry of another launch - a burst decoder could pick up the entry of a later feed's scan that was still
// writing.)  Feed numbers wrap at 4096: "earlier" = up to 2047 feeds back; what looks later is scanned again, which is only slower.
// 0xfffe / 0xffff: the test hooks' launches, run with nothing else in flight - they believe everything but themselves and are believed.
__device__ __forceinline__ bool ref_entry_visible(uint32_t entry, uint32_t mine) {
	entry &= 0xffffu; mine &= 0xffffu;
	if(entry == mine) return false;
	if(mine >= 0xfffeu || entry >= 0xfffeu) return true;
	const uint32_t ek = entry & 15u, mk = mine & 15u, d = ((mine >> 4) - (entry >> 4)) & 0xfffu;
	if(mk == 8u || d >= 2048u) return false;
	auto walk_kind = [](uint32_t k) { return k <= 4u || (k >= 8u && k <= 10u); };
	if(!walk_kind(mk)) return walk_kind(ek) || (d == 0u && ek < mk);       // a burst kernel (5, 6, 7)
	if(!walk_kind(ek)) return false;
	const bool early = ek == 8u || ek == 1u || ek == 2u;                        // ends before anything of the next feed's walk begins
	const bool redone = ek == 9u || ek == 10u;
	if(mk == 4u || mk == 9u) return d >= 1u || mk == 4u || early;
	if(mk == 10u) return d >= 2u || early || (d == 1u && redone);
	if(d >= 3u) return true;
	if(d == 2u) return early || redone;
	if(d == 1u) return early;
	switch(mk) {                                                                 // d == 0
		case 1: return ek == 8u;
		case 2: return ek == 8u || ek == 1u;
		case 0: return early;
		default: return early || ek == 0u;                                       // 3
	}
}
__device__ __forceinline__ bool ref_done_lookup(const unsigned long long *done, uint32_t ndv, int64_t n_lo, int64_t n_hi, uint32_t launch, int lane) {
	const uint32_t nd = ndv < (uint32_t)kRefCache ? ndv : (uint32_t)kRefCache;
	bool hit = false;
	for(uint32_t i = (uint32_t)lane; i < nd; i += 64u) {          // (an entry per lane and round: four rounds when the ring is full)
		const unsigned long long e = __hip_atomic_load(done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const int64_t lo = (int64_t)(e >> 32) << 8, hi = lo + ((int64_t)((e >> 16) & 0xffffull) << 8) + 255;
		hit = hit || (lo <= n_lo && n_hi <= hi && ref_entry_visible((uint32_t)(e & 0xffffull), launch));
	}
	return __any(hit) != 0;
}
// ref_window_done(): the stretch as ref_exact_window_dev() would ask for it (whole blocks of 256, as far as the input reaches)
__device__ __forceinline__ bool ref_window_done_dev(const ChanView &v, int64_t n_lo, int64_t n_hi) {
	RefChan *rp = v.ref;
	if(!rp) return false;
	const int npiece = rp->npiece, os = rp->os, c = v.ref_chan;
	if(n_lo < 0) n_lo = 0;
	if(n_hi < n_lo) return true;
	if(npiece <= 0) return false;
	const int64_t in_end = rp->piece[npiece - 1].s0 + rp->piece[npiece - 1].n, last = in_end / os - 1;
	n_lo &= ~255ll;
	if((n_hi | 255) <= last) n_hi |= 255; else if(n_hi < last) n_hi = last;
	return ref_done_lookup(rp->done + (size_t)c * kRefCache, rp->done_n[c], n_lo, n_hi, v.ref_launch, threadIdx.x & 63);
}
__device__ __forceinline__ bool ref_exact_window_dev(const ChanView &v, int64_t n_lo, int64_t n_hi, int kind) {
	#pragma clang fp contract(off)
	RefChan *rp = v.ref;
	if(!rp) return false;
	const int lane = threadIdx.x & 63, c = v.ref_chan;
	// the hook's scalars, once (wave-uniform: they live in SGPRs from here on; nothing below reads the hook again)
	const int os = rp->os, fmt = rp->fmt, npiece = rp->npiece;
	const uint32_t mask = rp->mask, cap = rp->cap;
	const int64_t warm = rp->warm;
	int64_t ps0[kRefPieces], pn[kRefPieces]; const void *pp[kRefPieces];
	#pragma unroll
	for(int j = 0; j < kRefPieces; j++) { ps0[j] = j < npiece ? rp->piece[j].s0 : 0; pn[j] = j < npiece ? rp->piece[j].n : 0; pp[j] = j < npiece ? rp->piece[j].p : nullptr; }
	const float A0 = rp->A0, A1 = rp->A1, A2 = rp->A2;
	const v2f B1 = v2f{rp->B1, rp->B1}, B2 = v2f{rp->B2, rp->B2};
	const uint32_t dphi = rp->dphi[c];
	const bool mix = rp->mix[c] != 0;
	ref_gf4 *lut = (ref_gf4 *)rp->lut;
	unsigned long long *done = rp->done + (size_t)c * kRefCache;
	uint32_t *done_n = rp->done_n + c, *stats = rp->stats;
	__attribute__((address_space(1))) float *yout = (__attribute__((address_space(1))) float *)(rp->y + (size_t)c * cap);
	if(n_lo < 0) n_lo = 0;
	if(n_hi < n_lo) return true;
	const int64_t in_end = ps0[npiece - 1 < 0 ? 0 : npiece - 1] + pn[npiece - 1 < 0 ? 0 : npiece - 1];    // one past the newest raw sample held
	// (whole blocks of 256 samples, as far as the input reaches: the candidates of one preamble ask for overlapping stretches)
	{
		const int64_t last = in_end / os - 1;
		n_lo &= ~255ll;
		if((n_hi | 255) <= last) n_hi |= 255; else if(n_hi < last) n_hi = last;
	}
	// done before?
	if(ref_done_lookup(done, *done_n, n_lo, n_hi, v.ref_launch, lane)) { if(lane == 0) atomicAdd(stats + 1, 1u); return true; }
	const int64_t s_end = (int64_t)os * (n_hi + 1);            // decimated sample k is the filter's output after input sample os (k + 1) - 1
	int64_t s_beg = (int64_t)os * n_lo - warm;
	bool shortened = false;
	if(s_beg < 0) s_beg = 0;                                     // the stream's own start: the reference's state there is zero, exactly
	else {
		if(ps0[0] > s_beg) { shortened = true; s_beg = (ps0[0] + os - 1) / os * os; }      // (the oldest sample held, on the next decimation boundary: nothing that is not held is read)
		if(s_beg > 0 && (int64_t)os * n_lo - s_beg < warm / 4) { if(lane == 0) atomicAdd(stats + 2, 1u); return false; }
	}
	if(npiece <= 0 || in_end < s_end) { if(lane == 0) atomicAdd(stats + 2, 1u); return false; }
	__builtin_amdgcn_s_setprio(VDL2_REF_PRIO);

	const int G = 64 / os, blk = G * os;                        // decimated outputs / input samples per block (os <= kMaxOversample = 32)
	// raw sample s (this lane's of a block) as a 32-bit word, and its NCO table entry
	// (a uniform search for the one stretch a whole block lies in, with a single load behind it, was tried: 23.5 against 22.6 ns per sample)
	auto fetch = [&](int64_t s0, uint32_t &w, v4f &e) {
		const int64_t s = s0 + (lane < blk ? lane : 0);
		w = 0u;
		#pragma unroll
		for(int j = 0; j < kRefPieces; j++) {
			if(j < npiece && s >= ps0[j] && s < ps0[j] + pn[j] && s < s_end) {
				if(fmt == 1) w = ((ref_gu32 *)pp[j])[s - ps0[j]];
				else w = ((ref_gu16 *)pp[j])[s - ps0[j]];
			}
		}
		const uint32_t phi = ((uint32_t)s * dphi) & 0xffffffu;      // applied, then advanced, from 0 at sample 0 (demod.c:313-316); only the low 24 bits count
		e = mix ? lut[phi >> 16] : v4f{0.f, 1.f, 0.f, 0.f};
	};
	// The scan starts on a decimation boundary and goes in blocks of G * os input samples (G = 64 / os decimated outputs: 60 samples
	// at oversample 20), so that inside a block every step and every output is at a fixed place: the recursion is straight-line code,
	// G x os steps of two v_readlane + four packed operations, no branch (a taken branch per step cost more than the arithmetic).
	s_beg -= s_beg % os;
	int64_t k_out = s_beg / os;                                  // the decimated sample the next output is
	v2f y1 = v2f{0.f, 0.f}, y2 = v2f{0.f, 0.f};                 // (yr[1], yi[1]), (yr[2], yi[2]) of demod.c:289-298
	float xm1r = 0.f, xm1i = 0.f, xm2r = 0.f, xm2i = 0.f;       // the mixed samples before the block (uniform)
	// The raw samples and table entries of the next kRefAhead blocks are in flight at any time (a block lasts ~0.8 us, a load that misses
	// the L2 longer: with one block of look-ahead the wavefront waited for memory at every other block - 24 ns per input sample where the
	// bare recursion takes 13, dev/gpu_ubench_scan.hip).  The loop body is kRefAhead blocks of straight-line code, each with a slot of
	// its own, so that the slots stay in the registers they were loaded into; the stretch is padded to whole groups of blocks (the
	// padding reads nothing - fetch() stops at s_end - and stores nothing).
	constexpr int kRefAhead = 3;
	uint32_t w_q[kRefAhead]; v4f e_q[kRefAhead];
	#pragma unroll
	for(int k = 0; k < kRefAhead; k++) fetch(s_beg + (int64_t)k * blk, w_q[k], e_q[k]);
	auto do_block = [&](int64_t sb, auto SLOTC) {
		constexpr int SLOT = decltype(SLOTC)::value;
		const uint32_t w = w_q[SLOT]; const v4f e = e_q[SLOT];
		fetch(sb + (int64_t)kRefAhead * blk, w_q[SLOT], e_q[SLOT]);    // in flight during the next blocks' recursions
		// ---- this lane's sample: conversion (demod.c:349-365), NCO (:58-72), mixer (:200-203) ----
		float re, im;
		if(fmt == 1) { re = (float)(int16_t)(w & 0xffff) / 32768.0f; im = (float)(int16_t)(w >> 16) / 32768.0f; }
		else { re = ((float)(w & 0xff) - 127.5f) / 127.5f; im = ((float)((w >> 8) & 0xff) - 127.5f) / 127.5f; }
		if(mix) {
			const uint32_t phi = ((uint32_t)(sb + lane) * dphi) & 0xffffffu;
			const float F = (float)(phi & 0xffffu);
			const float sn = e.x + e.z * F, cs = e.y + e.w * F;   // v1 + (v2 - v1) * fract (Lut4 = {s, c, ds, dc}: the differences pre-scaled by 2^-16, exactly)
			const float mr = re * cs - im * sn, mi = im * cs + re * sn;
			re = mr; im = mi;
		}
		// in[1], in[2]: the mixed samples of the two lanes before (lanes 0, 1: of the block before)
		float x1r = dpp_wave_shr1(re), x1i = dpp_wave_shr1(im);
		if(lane == 0) { x1r = xm1r; x1i = xm1i; }
		float x2r = dpp_wave_shr1(x1r), x2i = dpp_wave_shr1(x1i);
		if(lane == 0) { x2r = xm2r; x2i = xm2i; }
		xm2r = ref_lane(re, blk - 2); xm2i = ref_lane(im, blk - 2);
		xm1r = ref_lane(re, blk - 1); xm1i = ref_lane(im, blk - 1);
		float fa = A0 * re; fa += A1 * x1r + A2 * x2r;              // r = A0 in0; r += A1 in1 + A2 in2   (demod.c:75-76)
		float fb = A0 * im; fb += A1 * x1i + A2 * x2i;
		// ---- the recursion over the block: r += B1 out1 + B2 out2 (demod.c:77), I and Q as one packed operation; every os-th value is an output ----
		auto run = [&](auto OSC) {
			constexpr int OS = decltype(OSC)::value;
			constexpr int GG = 64 / OS;
			#pragma unroll
			for(int g = 0; g < GG; g++) {
				#pragma unroll
				for(int j = 0; j < OS; j++) {
					const v2f r0 = v2f{ref_lane(fa, g * OS + j), ref_lane(fb, g * OS + j)};
					const v2f yv = r0 + (B1 * y1 + B2 * y2);
					y2 = y1; y1 = yv;
				}
				if(k_out >= n_lo && k_out <= n_hi && lane < 2) yout[2 * ((uint32_t)k_out & mask) + lane] = lane == 0 ? y1.x : y1.y;
				k_out++;
			}
		};
		if(os == 20) run(std::integral_constant<int, 20>());
		else if(os == 10) run(std::integral_constant<int, 10>());
		else if(os == 13) run(std::integral_constant<int, 13>());
		else {
			for(int g = 0; g < G; g++) {
				for(int j = 0; j < os; j++) {
					const v2f r0 = v2f{ref_lane(fa, g * os + j), ref_lane(fb, g * os + j)};
					const v2f yv = r0 + (B1 * y1 + B2 * y2);
					y2 = y1; y1 = yv;
				}
				if(k_out >= n_lo && k_out <= n_hi && lane < 2) yout[2 * ((uint32_t)k_out & mask) + lane] = lane == 0 ? y1.x : y1.y;
				k_out++;
			}
		}
	};
	static_assert(kRefAhead == 3, "the loop below is written out for three slots");
	for(int64_t sb = s_beg; sb < s_end; sb += (int64_t)kRefAhead * blk) {
		do_block(sb, std::integral_constant<int, 0>());
		do_block(sb + blk, std::integral_constant<int, 1>());
		do_block(sb + 2 * (int64_t)blk, std::integral_constant<int, 2>());
	}
	__builtin_amdgcn_s_setprio(0);
	// The new samples are read back by the lanes of THIS wavefront: its stores must have landed (vmcnt(0)) and its CU's vector L1 and
	// scalar cache - neither ever holds dirty data - must not answer with what they held before.  Nothing wider: __threadfence() (an
	// agent-scope release) writes back the whole L2, and an agent-scope INVALIDATE (buffer_inv sc1) drops L2 lines, dirty ones included -
	// the walker's own burst list and state went missing that way.  Other wavefronts see the samples from the next kernel on.
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tbuffer_inv sc0\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (both invalidates are complete before anything is loaded again)
	if(lane == 0) {
		const uint32_t i = atomicAdd(done_n, 1u);
		int64_t len = (n_hi - n_lo) >> 8; if(len > 0xffff) len = 0xffff;     // (whole blocks of 256: n_lo is aligned, n_hi ends a block or the input)
		if(((n_hi + 1) & 255) != 0) len -= 1;                                 // a last, partial block is not promised
		if(len >= 0) __hip_atomic_store(done + (i % (uint32_t)kRefCache), ((unsigned long long)(n_lo >> 8) << 32) | ((unsigned long long)len << 16) | (unsigned long long)(v.ref_launch & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		atomicAdd(stats + 0, 1u); atomicAdd(stats + 4 + kind, 1u);
		if(shortened) atomicAdd(stats + 3, 1u);
	}
	return true;
}
#endif

VDL2_HD void ref_debug_log(const ChanView &v, int tag, int64_t a, float b, float c, float d) {
#if VDL2_DEVICE_PASS && defined(VDL2_REF_DEBUG)
	RefChan *r = v.ref;
	if(!r || !r->dbg || v.ref_chan != r->dbg_chan) return;
	const unsigned long long i = atomicAdd(r->dbg, 1ull);
	if(i < 1000) { unsigned long long *e = r->dbg + 1 + 4 * i; e[0] = ((unsigned long long)tag << 56) | (unsigned long long)(a & 0xffffffffffffffll); e[1] = __float_as_uint(b); e[2] = __float_as_uint(c); e[3] = __float_as_uint(d); }
#else
	(void)v; (void)tag; (void)a; (void)b; (void)c; (void)d;
#endif
}

// the referee's raw-input history: `n` samples (sb bytes each) from src go to the ring at sample position pos (modulo cap samples)
__global__ void k_ref_hist(const uint8_t *src, uint64_t n, uint8_t *ring, uint64_t pos, uint64_t cap, int sb) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const uint64_t d = (pos + i) % cap;
	if(sb == 4) reinterpret_cast<uint32_t *>(ring)[d] = reinterpret_cast<const uint32_t *>(src)[i];
	else reinterpret_cast<uint16_t *>(ring)[d] = reinterpret_cast<const uint16_t *>(src)[i];
}

VDL2_HD __attribute__((always_inline)) bool ref_exact_window(const ChanView &v, int64_t n_lo, int64_t n_hi, void *scratch, int kind);
// test hook: one wavefront makes [n_lo, n_hi] of channel `chan` exact; out[0] = 1 if it could
// (block b of several: channel (chan + b) mod nchan, the stretch moved on by `stride` samples per block)
__global__ __launch_bounds__(64) void k_ref_probe(RefChan *ref, int chan, int nchan, int64_t n_lo, int64_t n_hi, int64_t stride, int *out) {
	__shared__ __align__(16) unsigned char scratch[2048];
	const int c = (chan + (int)blockIdx.x) % nchan;
	const int64_t off = stride * (int64_t)blockIdx.x;
	ChanView v{ ref->y + (size_t)c * ref->cap, nullptr, nullptr, ref->mask, ref, c, 0xffffu };
	const bool ok = ref_exact_window(v, n_lo + off, n_hi + off, scratch, REF_CANDIDATE);
	if(threadIdx.x == 0) out[blockIdx.x] = ok ? 1 : 0;
}

VDL2_HD bool ref_window_done(const ChanView &v, int64_t n_lo, int64_t n_hi) {
#if VDL2_DEVICE_PASS
	return ref_window_done_dev(v, n_lo, n_hi);
#else
	(void)v; (void)n_lo; (void)n_hi;
	return false;
#endif
}
VDL2_HD __attribute__((always_inline)) bool ref_exact_window(const ChanView &v, int64_t n_lo, int64_t n_hi, void *scratch, int kind) {
#if VDL2_DEVICE_PASS
	if(v.ref && !((v.ref->kinds >> kind) & 1)) return false;
	(void)scratch;
	return ref_exact_window_dev(v, n_lo, n_hi, kind);
#else
	(void)v; (void)n_lo; (void)n_hi; (void)scratch; (void)kind;
	return false;
#endif
}

// the slice of BlockForm K1 needs, passed by value so that it lives in the kernarg segment
// (constant address space -> scalar loads into SGPRs)
struct K1Consts {
	float g0[kMaxOversample], g1[kMaxOversample];
	float P[4], c0, c1, c2, pad_;
	float Q[6][4];
	float Pp[kRunMax + 1][4];      // P^i, i <= run
	float cP[kRunMax][2];          // (c0,c1) P^(i+1), i < run
};

inline K1Consts make_k1_consts(const BlockForm &bf) {
	K1Consts k{};
	for(int i = 0; i < kMaxOversample; i++) { k.g0[i] = bf.g0[i]; k.g1[i] = bf.g1[i]; }
	for(int i = 0; i < 4; i++) k.P[i] = bf.P[i];
	k.c0 = bf.c0; k.c1 = bf.c1; k.c2 = bf.c2;
	for(int d = 0; d < 6; d++) for(int i = 0; i < 4; i++) k.Q[d][i] = bf.Q[d][i];
	for(int r = 0; r <= kRunMax; r++) for(int i = 0; i < 4; i++) k.Pp[r][i] = bf.Ppow[r][i];
	for(int r = 0; r < kRunMax; r++) { k.cP[r][0] = bf.cP[r][0]; k.cP[r][1] = bf.cP[r][1]; }
	return k;
}

struct K1Args {
	const void *in;            // raw IQ block of this feed (cs16 or cu8), device memory
	const void *carry;         // raw samples left over from the previous feed
	uint32_t ncarry;           // number of complex samples in carry
	uint64_t nlogical;         // ncarry + samples in `in`
	uint64_t n0;               // absolute index (since stream start) of logical sample 0
	int64_t  k0;               // absolute index of the first decimated output of this feed
	int64_t  D;                // decimated outputs produced by this feed
	int32_t  fmt, nchan, os, nseg, gy;
	int32_t  seg0, seg1;       // this launch does the workgroup segments seg0 <= s < seg1 (seg0 a multiple of 8); one launch: 0, nseg
	const uint32_t *dphi;      // NCO step per channel (24-bit phase)
	const Lut4 *lut;
	K1Consts bf;
	cf32 *y;                   // [nchan][cap]
	float4 *seg_end;           // [nchan][nseg_cap] zero-start state at the end of each workgroup segment
	const float4 *qpow;        // [64] Q^(l+1) row-major 2x2, Q = P^R
	int32_t  tiles;            // tiles per workgroup segment
	// fused fix-up (fuse != 0): K1 also adds the decayed segment-start state to its first tile (one-step look-back), K2 is not launched
	int32_t  fuse;
	const float4 *carry_in;    // [nchan] true filter state at the end of the previous feed
	float4 *carry_out;         // [nchan] ... at the end of this one
	const BlockForm *bfd;      // full tables (cP[kFixW], Ppow[kFixW+1]) in device memory
	unsigned long long *seg_pub; // [nchan][nseg_cap][4]: (epoch << 32 | float bits) of the zero-start segment end state
	uint32_t epoch;
	uint32_t pub_epoch;        // = epoch (differs only when a test forces the look-back to time out)
	uint32_t spin_limit;       // polls (~1 us each) before a consumer stops waiting for the previous segment's state and recomputes it
	uint32_t *sync_timeouts;   // count of workgroups whose look-back wait gave up and took the fall-back (0 with one process per GPU)
	uint32_t cap, mask, nseg_cap;
};

__device__ __forceinline__ void load_sample(const K1Args &a, int64_t s, float &re, float &im) {
	if(a.fmt == 1) {   // S16_LE: (float)v / 32768.0f  (demod.c:362-363)
		const uint32_t *p = (s < (int64_t)a.ncarry) ? (const uint32_t *)a.carry + s : (const uint32_t *)a.in + (s - a.ncarry);
		uint32_t w = *p;
		re = (float)(int16_t)(w & 0xffff) / 32768.0f;
		im = (float)(int16_t)(w >> 16) / 32768.0f;
	} else {           // U8: (i - 127.5f) / 127.5f     (demod.c:349-354)
		const uint16_t *p = (s < (int64_t)a.ncarry) ? (const uint16_t *)a.carry + s : (const uint16_t *)a.in + (s - a.ncarry);
		uint16_t w = *p;
		re = ((float)(w & 0xff) - 127.5f) / 127.5f;
		im = ((float)(w >> 8) - 127.5f) / 127.5f;
	}
}

// Cross-lane moves on the VALU's data-parallel-primitive path (GFX9 DPP controls; checked on the device by
// tests/test_gpu_parity.py::test_dpp_primitives_behave_as_the_scan_assumes).
__device__ __forceinline__ float dpp_row_shr(float v, int d) {     // value of lane - 2^d inside the same row of 16 lanes, else 0
	const int x = __builtin_bit_cast(int, v);
	int r;
	switch(d) {
		case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); break;
		case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); break;
		case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); break;
		default: r = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); break;
	}
	return __builtin_bit_cast(float, r);
}
template<int CTRL, int ROWS>
__device__ __forceinline__ float dpp_bcast(float v) {               // row_bcast:15 (0x142) / row_bcast:31 (0x143) into the rows of ROWS, 0 elsewhere
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shr1_keep(float lane0, float v) {   // value of lane - 1 across the whole wavefront, lane 0: `lane0`
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_wave_shr1(float v) {           // value of lane - 1 across the whole wavefront (lane 0: 0)
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// debug kernel for the test above: out[0..3][lane] = row_shr:4, row_bcast:15 (rows 1,3), row_bcast:31 (rows 2,3), wave_shr:1 of in[lane]
__global__ void k_dpp_probe(const float *in, float *out) {
	const int l = threadIdx.x & 63;
	const float v = in[l];
	out[l] = dpp_row_shr(v, 2);
	out[64 + l] = dpp_bcast<0x142, 0xa>(v);
	out[128 + l] = dpp_bcast<0x143, 0xc>(v);
	out[192 + l] = dpp_wave_shr1(v);
}

// One workgroup = 4 waves that walk `a.tiles` consecutive time tiles (64*R blocks of OS samples each, staged in LDS
// and shared by the waves); each wave owns CR channels; each lane owns R consecutive decimated outputs per tile.
// Within a workgroup's segment the filter state is carried from tile to tile in registers, so the outputs it
// stores are final except for the (decayed) state at the segment start, which K2 adds to the first kFixW of them.
// OS == 0 selects the generic (run-time oversample) build of the same code.
// resident workgroups per CU the channeliser is compiled for (LDS allows 6; dev/gpu_k1_variants.sh sweeps the choices)
#ifndef VDL2_K1_MIN_BLOCKS
#define VDL2_K1_MIN_BLOCKS 6
#endif
#ifndef VDL2_K1_MIN_BLOCKS_CR4
#define VDL2_K1_MIN_BLOCKS_CR4 4
#endif
// optional per-phase cycle probe of the channeliser (development aid, -DVDL2_K1_PROF; compiled out by default): wave 0 of every
// workgroup adds the shader clocks it spends in each phase of a tile to vdl2_k1_prof[phase], the count to [8 + phase]
#ifdef VDL2_K1_PROF
__device__ unsigned long long vdl2_k1_prof[64][16];
// wave-uniform on purpose (every lane of wave 0 does the same scalar arithmetic): the sums live in SGPRs and the kernel's vector
// registers are left alone; 16 atomics per workgroup at its very end, spread over 64 slots
#define K1_BEGIN() unsigned k1_t0_ = (unsigned)__builtin_readcyclecounter(), k1_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k1_tiles_ = 0
#define K1_MARK(k) do { if(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) { const unsigned t_ = (unsigned)__builtin_readcyclecounter(); \
	k1_acc_[k] += t_ - k1_t0_; if((k) == 3) k1_tiles_++; k1_t0_ = t_; } } while(0)
#define K1_END() do { if(threadIdx.x == 0) { for(int k_ = 0; k_ < 8; k_++) atomicAdd(&vdl2_k1_prof[blockIdx.x & 63][k_], (unsigned long long)k1_acc_[k_]); \
	atomicAdd(&vdl2_k1_prof[blockIdx.x & 63][8], 1ull); atomicAdd(&vdl2_k1_prof[blockIdx.x & 63][9], (unsigned long long)k1_tiles_); } } while(0)
#else
#define K1_BEGIN() do {} while(0)
#define K1_MARK(k) do {} while(0)
#define K1_END() do {} while(0)
#endif

// (i - 127.5f) / 127.5f of an unsigned byte (demod.c:349-354) without the division: d = i - 127.5 is exact, q = d * fl(1 / 127.5) is within an
// ulp, one Newton step on the residual d - 127.5 q makes it the correctly rounded quotient - for all 256 values (checked on the device
// against the division: tests/test_gpu_parity.py::test_uint8_conversion_without_the_division)
__device__ __forceinline__ float u8_level(uint32_t i) {
	const float d = (float)i - 127.5f, r = 1.0f / 127.5f;
	const float q = d * r;
	return __builtin_fmaf(__builtin_fmaf(-127.5f, q, d), r, q);
}
__global__ void k_u8_level_probe(float *out) { const uint32_t i = threadIdx.x; out[i] = u8_level(i); out[256 + i] = ((float)i - 127.5f) / 127.5f; }

// U8: the build for unsigned-byte input (the reference's default for --iq-file and what an RTL-SDR delivers): its tiles are fetched a tile
// ahead and converted without per-sample range checks or divisions, as the s16 tiles of the other build are (round 6c: the generic
// staging path cost a 256-channel receiver 18 % of its channeliser: 4.46 against 3.77 ms per 16 s).  Only instantiated where the tile
// prefetch is (kPrefetch below); U8 = false is the code as it was.
template<int OS, int R, int CR, bool U8 = false>
__global__ __launch_bounds__(256, (CR >= 4 ? VDL2_K1_MIN_BLOCKS_CR4 : VDL2_K1_MIN_BLOCKS)) void k_chanfir(K1Args a) {
	static_assert(64 * R == kFixW || R == 1, "the fused fix-up assumes the fix window is the segment's first tile");
	static_assert(R >= 1 && R <= 2, "the run's outputs are held in two register pairs");
	// the small tables are static LDS, so that their addresses are compile-time constants (they fold into the offset field of the
	// ds_read); only the sample tile, whose size depends on the oversampling factor, is dynamic
	__shared__ __align__(16) float4 lut[256];     // NCO look-up
	__shared__ __align__(16) float4 qpow[64];     // Q^(l+1), l = 0..63: what a carry contributes to lane l's end state
	__shared__ __align__(16) float park[4 * 4 * 4 * 4];   // per wave: carried state and the end-of-feed state, [wave][4][CR][4]
	__shared__ int fallback;                      // fused fix-up: some wave of this workgroup gave up waiting for the previous segment's state
	extern __shared__ __align__(16) unsigned char smem[];
	if(VDL2_K1_PRIO) __builtin_amdgcn_s_setprio(VDL2_K1_PRIO);
	const int os = OS ? OS : a.os;
	const int run = R * os;                       // input samples per lane and tile
	float2 *tile = (float2 *)smem;                // [run][65]
	const int tid = threadIdx.x;
	K1_BEGIN();

	// XCD-aware decode of the 1-D block id: workgroups that share a time tile land on one XCD (same L2)
	const int bid = blockIdx.x;
	const int xcd = bid & 7, q = bid >> 3;
	const int gy = q % a.gy;
	const int seg = a.seg0 + (q / a.gy) * 8 + xcd; // workgroup segment = a.tiles tiles
	if(seg >= a.seg1) return;

	// The raw samples of a tile are fetched into registers one tile ahead of their use (specialised builds; cs16 tiles that lie inside
	// the block) - the first tile's right here, in flight together with the table loads below - so that the global-load latency of the
	// staging never sits between two tiles: 6.8 % of K1 at 256 channels, 8.3 % at 32 (profiles/r03_k1_tile_prefetch.txt).  Costs ten
	// registers (114 -> 128, a few loop-invariant values spilled outside the loops): only where four channels per wave leave that room
	// and the tile is a whole number of 256-sample rows.
	constexpr bool kPrefetch = OS != 0 && CR >= 4 && (64 * R * OS + 255) / 256 <= 10 && (64 * R * OS) % 256 == 0;
	static_assert(!U8 || kPrefetch, "the unsigned-byte build exists where the tile prefetch does");
	constexpr int kPre = kPrefetch ? (64 * R * OS) / 256 : 1;
	constexpr bool kPipeGather = kPrefetch;
	uint32_t pre[kPre]; bool have_pre = false;
	#pragma unroll
	for(int k = 0; k < kPre; k++) pre[k] = 0u;
	if(kPrefetch) {
		const int64_t s0 = (int64_t)seg * a.tiles * (64 * R * (OS ? OS : 1));      // first sample of the segment's first tile
		have_pre = a.fmt == (U8 ? 0 : 1) && s0 >= (int64_t)a.ncarry && s0 + 64 * R * (OS ? OS : 1) <= (int64_t)a.nlogical && (int64_t)seg * a.tiles * (64 * R) < a.D;
		if(have_pre) {
			if constexpr(U8) {
				const uint16_t *sn = (const uint16_t *)a.in + (s0 - a.ncarry);
				#pragma unroll
				for(int k = 0; k < kPre; k++) pre[k] = sn[tid + 256 * k];
			} else {
			const uint32_t *sn = (const uint32_t *)a.in + (s0 - a.ncarry);
			#pragma unroll
			for(int k = 0; k < kPre; k++) pre[k] = sn[tid + 256 * k];
			}
		}
	}
	lut[tid] = ((const float4 *)a.lut)[tid];
	if(tid < 64) qpow[tid] = a.qpow[tid];
	if(tid == 0) fallback = 0;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave-uniform, so per-channel values stay in SGPRs
	const int cbase = (gy * 4 + wave) * CR;
	const bool wave_active = cbase < a.nchan;
	const K1Consts &bf = a.bf;
	const int tile_n = 64 * run;                  // input samples per tile
	const int L = 64 * R;                         // decimated outputs per tile

	uint32_t dph[CR];
	float4 *carry = (float4 *)(park + wave * (4 * CR * 4));    // state carried into the current tile (zero at the segment start)
	float4 *svp = carry + CR;                                   // zero-start state after the feed's last valid block
	float4 *sendp = svp + CR;                                   // zero-start state at the end of the segment (what seg_end gets)
	float4 *tsp = sendp + CR;                                   // true state at the segment start (fused fix-up)
	// fused: tile 0's outputs wait for the segment-start state - in registers (16 of them with four channels per wave), or, where the
	// registers are worth more than a second trip through L2 (VDL2_K1_HOLD_IN_MEMORY), in the output stream itself: written as they are,
	// read back by the lane that wrote them, corrected and written again at the very end
#ifdef VDL2_K1_HOLD_IN_MEMORY
	constexpr bool kHoldRegs = !(OS != 0 && CR >= 4);
#else
	constexpr bool kHoldRegs = true;
#endif
	float hold[CR][4];
	#pragma unroll
	for(int c = 0; c < CR; c++) hold[c][0] = hold[c][1] = hold[c][2] = hold[c][3] = 0.f;
	#pragma unroll
	for(int c = 0; c < CR; c++) {
		const int ch = cbase + c < a.nchan ? cbase + c : a.nchan - 1;
		dph[c] = a.dphi[ch];
		if(lane == 0) carry[c] = make_float4(0.f, 0.f, 0.f, 0.f);
	}
	const float P0 = bf.P[0], P1 = bf.P[1], P2 = bf.P[2], P3 = bf.P[3];
	const float c0 = bf.c0, c1 = bf.c1, c2 = bf.c2;

	for(int ts = 0; ts < a.tiles; ts++) {
		const int64_t tix = (int64_t)seg * a.tiles + ts;       // tile index within the feed
		const int64_t kbase = tix * L;                           // feed-local index of the tile's first output
		if(kbase >= a.D) break;
		const int64_t sbase = tix * tile_n;
		K1_MARK(ts ? 4 : 0);                                     // 0: prologue (tables, first prefetch); 4: scan, outputs, carry of the previous tile
		if(ts) __syncthreads();                                  // everyone is done with the previous tile
		K1_MARK(5);                                              // 5: waiting for the workgroup's other waves before the tile is overwritten
		const bool fast_now = a.fmt == (U8 ? 0 : 1) && sbase >= (int64_t)a.ncarry && sbase + tile_n <= (int64_t)a.nlogical;
		if(kPrefetch && U8 && fast_now) {
			// (the unsigned-byte build: the same, two bytes per sample)
			const uint16_t *src = (const uint16_t *)a.in + (sbase - a.ncarry);
			#pragma unroll
			for(int k = 0; k < kPre; k++) {
				const int t = tid + 256 * k;
				const uint32_t w = have_pre ? pre[k] : (uint32_t)src[t];
				const int l = t / run, m = t - l * run;
				tile[m * 65 + l] = make_float2(u8_level(w & 0xffu), u8_level((w >> 8) & 0xffu));
			}
			const int64_t snext = sbase + tile_n;
			have_pre = ts + 1 < a.tiles && (tix + 1) * L < a.D && snext + tile_n <= (int64_t)a.nlogical;
			if(have_pre) {
				const uint16_t *sn = (const uint16_t *)a.in + (snext - a.ncarry);
				#pragma unroll
				for(int k = 0; k < kPre; k++) pre[k] = sn[tid + 256 * k];
			}
		} else if(kPrefetch && fast_now) {
			// the usual case - a cs16 tile that lies entirely inside this feed's block: no per-sample range or carry checks
			const uint32_t *src = (const uint32_t *)a.in + (sbase - a.ncarry);
			#pragma unroll
			for(int k = 0; k < kPre; k++) {
				const int t = tid + 256 * k;
				const uint32_t w = have_pre ? pre[k] : src[t];
				const int l = t / run, m = t - l * run;
				tile[m * 65 + l] = make_float2((float)(int16_t)(w & 0xffff) / 32768.0f, (float)(int16_t)(w >> 16) / 32768.0f);
			}
			const int64_t snext = sbase + tile_n;
			have_pre = ts + 1 < a.tiles && (tix + 1) * L < a.D && snext + tile_n <= (int64_t)a.nlogical;
			if(have_pre) {
				const uint32_t *sn = (const uint32_t *)a.in + (snext - a.ncarry);
				#pragma unroll
				for(int k = 0; k < kPre; k++) pre[k] = sn[tid + 256 * k];
			}
		} else if(fast_now) {
			// (builds without the prefetch) a cs16 tile that lies entirely inside this feed's block: no per-sample range or carry checks
			const uint32_t *src = (const uint32_t *)a.in + (sbase - a.ncarry);
			for(int t = tid; t < tile_n; t += 256) {
				const uint32_t w = src[t];
				const int l = t / run, m = t - l * run;
				tile[m * 65 + l] = make_float2((float)(int16_t)(w & 0xffff) / 32768.0f, (float)(int16_t)(w >> 16) / 32768.0f);
			}
		} else {
			for(int t = tid; t < tile_n; t += 256) {
				const int64_t sidx = sbase + t;
				float re = 0.f, im = 0.f;
				if(sidx < (int64_t)a.nlogical) load_sample(a, sidx, re, im);
				const int l = t / run, m = t - l * run;
				tile[m * 65 + l] = make_float2(re, im);
			}
		}
		K1_MARK(1);                                              // 1: staging (conversion + LDS stores + issue of the next tile's loads)
		__syncthreads();
		K1_MARK(2);                                              // 2: waiting for the staging of the other waves
		if(!wave_active) continue;

		uint32_t ph[CR];
		const uint32_t nabs = (uint32_t)((a.n0 + (uint64_t)sbase + (uint64_t)lane * run) & 0xffffffu);
		#pragma unroll
		for(int c = 0; c < CR; c++) ph[c] = nabs * dph[c];
		// last valid block of this tile (tile-local index) and who owns it
		const int64_t rem = a.D - kbase;
		const int blast = rem >= L ? L - 1 : (int)rem - 1;
		const int lb = blast / R, ib = blast - lb * R;

		float t0r[CR], t0i[CR], t1r[CR], t1i[CR];     // running (zero-start) state of this lane's run
		float ya[CR][2], yb[CR][2];                   // zero-start outputs of the run (R <= 2 kept in registers)
		#pragma unroll
		for(int c = 0; c < CR; c++) {
			t0r[c] = t0i[c] = t1r[c] = t1i[c] = 0.f;
			ya[c][0] = ya[c][1] = yb[c][0] = yb[c][1] = 0.f;
		}

		// The block loop stays rolled: one iteration = OS samples x CR channels of straight-line code.
		#pragma unroll 1
		for(int i = 0; i < R; i++) {
			// (re, im) pairs throughout, so that the mix and the two tap sums are packed FP32 operations
			v2f A0[CR], A1[CR], M[CR];
			#pragma unroll
			for(int c = 0; c < CR; c++) { A0[c] = v2f{0.f, 0.f}; A1[c] = v2f{0.f, 0.f}; M[c] = v2f{0.f, 0.f}; }
			const float2 *trow = tile + (size_t)(i * os) * 65 + lane;
			// Partially unrolled on purpose: a fully unrolled run makes the scheduler hoist every LUT
			// gather (4 VGPRs each) to the top and spill.  Taps come from scalar loads (uniform index).
			// the LUT gather of the NEXT channel-sample is issued before the current one is worked on (four channels per wave: the
			// compiler otherwise waits for every gather right behind its issue; 1.9 % of K1 at 256 channels, 2.6 % at 32,
			// profiles/r03_k1_variants.txt - the sample loop is bound by the LDS gather rate, so there is not much to hide)
			float4 en = kPipeGather ? lut[(ph[0] >> 16) & 0xffu] : make_float4(0.f, 0.f, 0.f, 0.f);
			#pragma unroll kK1Unroll
			for(int j = 0; j < os; j++) {
				const float2 x = trow[j * 65];
				const v2f X = v2f{x.x, x.y}, Xr = v2f{-x.y, x.x};          // x and i*x, shared by the channels
				const float g0 = bf.g0[j], g1 = bf.g1[j];
				#pragma unroll
				for(int c = 0; c < CR; c++) {
					const uint32_t p = ph[c];
					const float F = (float)(p & 0xffffu);                 // sincosf_lut(): fract * 65536
					float4 e;
					if(kPipeGather) {
						e = en;
						ph[c] = p + dph[c];
						en = lut[(ph[(c + 1) % CR] >> 16) & 0xffu];       // (c+1, j), or (0, j+1) - whose phase has just been advanced
					} else e = lut[(p >> 16) & 0xffu];
					const v2f sc = __builtin_elementwise_fma(v2f{e.z, e.w}, v2f{F, F}, v2f{e.x, e.y});   // (sin, cos)
					// multiply(): (re*cos - im*sin, im*cos + re*sin) = cos * x + sin * (i x), the product rounded as before
					const v2f m = __builtin_elementwise_fma(v2f{sc.y, sc.y}, X, v2f{sc.x, sc.x} * Xr);
					A0[c] = __builtin_elementwise_fma(v2f{g0, g0}, m, A0[c]);
					A1[c] = __builtin_elementwise_fma(v2f{g1, g1}, m, A1[c]);
					M[c] = m;
					if(!kPipeGather) ph[c] = p + dph[c];
				}
			}
			float a0r[CR], a0i[CR], a1r[CR], a1i[CR], lr[CR], li[CR];
			#pragma unroll
			for(int c = 0; c < CR; c++) { a0r[c] = A0[c].x; a0i[c] = A0[c].y; a1r[c] = A1[c].x; a1i[c] = A1[c].y; lr[c] = M[c].x; li[c] = M[c].y; }
			#pragma unroll
			for(int c = 0; c < CR; c++) {
				// state update t <- P t + acc, then y = c0*v[n] + c1*v[n-1] + c2*xm[n] (zero-start part)
				const float n0r = __builtin_fmaf(P0, t0r[c], __builtin_fmaf(P1, t1r[c], a0r[c]));
				const float n0i = __builtin_fmaf(P0, t0i[c], __builtin_fmaf(P1, t1i[c], a0i[c]));
				const float n1r = __builtin_fmaf(P2, t0r[c], __builtin_fmaf(P3, t1r[c], a1r[c]));
				const float n1i = __builtin_fmaf(P2, t0i[c], __builtin_fmaf(P3, t1i[c], a1i[c]));
				t0r[c] = n0r; t0i[c] = n0i; t1r[c] = n1r; t1i[c] = n1i;
				const float yr = __builtin_fmaf(c0, n0r, __builtin_fmaf(c1, n1r, c2 * lr[c]));
				const float yi = __builtin_fmaf(c0, n0i, __builtin_fmaf(c1, n1i, c2 * li[c]));
				if(i == ib && lane == lb) svp[c] = make_float4(n0r, n0i, n1r, n1i);
				if(i == 0) { ya[c][0] = yr; ya[c][1] = yi; } else { yb[c][0] = yr; yb[c][1] = yi; }
			}
		}

		K1_MARK(3);                                              // 3: the sample loop and the block updates
		// wave-level scan of the lane end states: X_l = Q X_{l-1} + E_l, Q = P^R, X_-1 = carry - on the cross-lane data paths
		// of the VALU (DPP), no LDS traffic (ds_bpermute) and no lane-range selects:
		//   4 Kogge-Stone steps inside each row of 16 lanes (row_shr:2^d; a lane whose source lies outside its row reads 0),
		//   rows 1 and 3 take the end state of rows 0 and 2 (row_bcast:15) times Q^(p+1), p = lane & 15,
		//   rows 2 and 3 take the end state of lane 31 (row_bcast:31) times Q^(lane-31),
		// then every lane adds what the carried state of the previous tile contributes, Q^(lane+1) carry.
		#pragma unroll
		for(int d = 0; d < 4; d++) {
			const float q0 = bf.Q[d][0], q1 = bf.Q[d][1], q2 = bf.Q[d][2], q3 = bf.Q[d][3];
			#pragma unroll
			for(int c = 0; c < CR; c++) {
				const float o0r = dpp_row_shr(t0r[c], d), o0i = dpp_row_shr(t0i[c], d);
				const float o1r = dpp_row_shr(t1r[c], d), o1i = dpp_row_shr(t1i[c], d);
				t0r[c] = __builtin_fmaf(q0, o0r, __builtin_fmaf(q1, o1r, t0r[c])); t0i[c] = __builtin_fmaf(q0, o0i, __builtin_fmaf(q1, o1i, t0i[c]));
				t1r[c] = __builtin_fmaf(q2, o0r, __builtin_fmaf(q3, o1r, t1r[c])); t1i[c] = __builtin_fmaf(q2, o0i, __builtin_fmaf(q3, o1i, t1i[c]));
			}
		}
		{
			const float4 qa = qpow[lane & 15];                  // Q^(p+1): only rows 1 and 3 receive a non-zero operand
			#pragma unroll
			for(int c = 0; c < CR; c++) {
				const float o0r = dpp_bcast<0x142, 0xa>(t0r[c]), o0i = dpp_bcast<0x142, 0xa>(t0i[c]);
				const float o1r = dpp_bcast<0x142, 0xa>(t1r[c]), o1i = dpp_bcast<0x142, 0xa>(t1i[c]);
				t0r[c] = __builtin_fmaf(qa.x, o0r, __builtin_fmaf(qa.y, o1r, t0r[c])); t0i[c] = __builtin_fmaf(qa.x, o0i, __builtin_fmaf(qa.y, o1i, t0i[c]));
				t1r[c] = __builtin_fmaf(qa.z, o0r, __builtin_fmaf(qa.w, o1r, t1r[c])); t1i[c] = __builtin_fmaf(qa.z, o0i, __builtin_fmaf(qa.w, o1i, t1i[c]));
			}
			const float4 qb = qpow[(lane - 32) & 63];           // Q^(lane-31) for lanes 32..63 (rows 2 and 3)
			#pragma unroll
			for(int c = 0; c < CR; c++) {
				const float o0r = dpp_bcast<0x143, 0xc>(t0r[c]), o0i = dpp_bcast<0x143, 0xc>(t0i[c]);
				const float o1r = dpp_bcast<0x143, 0xc>(t1r[c]), o1i = dpp_bcast<0x143, 0xc>(t1i[c]);
				t0r[c] = __builtin_fmaf(qb.x, o0r, __builtin_fmaf(qb.y, o1r, t0r[c])); t0i[c] = __builtin_fmaf(qb.x, o0i, __builtin_fmaf(qb.y, o1i, t0i[c]));
				t1r[c] = __builtin_fmaf(qb.z, o0r, __builtin_fmaf(qb.w, o1r, t1r[c])); t1i[c] = __builtin_fmaf(qb.z, o0i, __builtin_fmaf(qb.w, o1i, t1i[c]));
			}
		}
		const float4 qp = qpow[lane];
		#pragma unroll
		for(int c = 0; c < CR; c++) {
			// add what the carried state contributes, then each lane needs the state at the START of its run
			const float4 cy = carry[c];
			t0r[c] = __builtin_fmaf(qp.x, cy.x, __builtin_fmaf(qp.y, cy.z, t0r[c])); t0i[c] = __builtin_fmaf(qp.x, cy.y, __builtin_fmaf(qp.y, cy.w, t0i[c]));
			t1r[c] = __builtin_fmaf(qp.z, cy.x, __builtin_fmaf(qp.w, cy.z, t1r[c])); t1i[c] = __builtin_fmaf(qp.z, cy.y, __builtin_fmaf(qp.w, cy.w, t1i[c]));
			float T0r = dpp_wave_shr1(t0r[c]), T0i = dpp_wave_shr1(t0i[c]), T1r = dpp_wave_shr1(t1r[c]), T1i = dpp_wave_shr1(t1i[c]);
			if(lane == 0) { T0r = cy.x; T0i = cy.y; T1r = cy.z; T1i = cy.w; }
			const bool cvalid = cbase + c < a.nchan;
			const int64_t kloc = kbase + (int64_t)lane * R;
			cf32 *yout = a.y + (size_t)(cbase + c) * a.cap;
			// outputs of the run, completed with the decayed run-start state (cP[i] = (c0,c1) P^(i+1))
			const float f0r = __builtin_fmaf(bf.cP[0][0], T0r, __builtin_fmaf(bf.cP[0][1], T1r, ya[c][0])), f0i = __builtin_fmaf(bf.cP[0][0], T0i, __builtin_fmaf(bf.cP[0][1], T1i, ya[c][1]));
			if(R > 1) {
				const float f1r = __builtin_fmaf(bf.cP[1][0], T0r, __builtin_fmaf(bf.cP[1][1], T1r, yb[c][0])), f1i = __builtin_fmaf(bf.cP[1][0], T0i, __builtin_fmaf(bf.cP[1][1], T1i, yb[c][1]));
				const uint32_t s0 = (uint32_t)(a.k0 + kloc) & a.mask;
				if(kHoldRegs && a.fuse && ts == 0) {
					hold[c][0] = f0r; hold[c][1] = f0i; hold[c][2] = f1r; hold[c][3] = f1i;      // stored after the fix-up below
				} else {
					if(cvalid && kloc + 1 < a.D && (s0 & 1u) == 0) {
						// both outputs of the run in one 16-byte store: a wavefront writes 1 KiB of contiguous, fully used lines
						*reinterpret_cast<float4 *>(yout + s0) = make_float4(f0r, f0i, f1r, f1i);
					} else {
						if(cvalid && kloc < a.D) yout[s0] = cf32{f0r, f0i};
						if(cvalid && kloc + 1 < a.D) yout[(uint32_t)(a.k0 + kloc + 1) & a.mask] = cf32{f1r, f1i};
					}
				}
			} else {
				if(kHoldRegs && a.fuse && ts == 0) { hold[c][0] = f0r; hold[c][1] = f0i; }
				else {
					if(cvalid && kloc < a.D) yout[(uint32_t)(a.k0 + kloc) & a.mask] = cf32{f0r, f0i};
				}
			}
			if(cvalid && lane == lb && (rem <= L || ts == a.tiles - 1)) {
				// state at the end of the segment's valid part, with zero state at the segment start
				const float *Pp = bf.Pp[ib + 1];
				float4 e;
				const float4 sv = svp[c];
				e.x = sv.x + (Pp[0] * T0r + Pp[1] * T1r); e.y = sv.y + (Pp[0] * T0i + Pp[1] * T1i);
				e.z = sv.z + (Pp[2] * T0r + Pp[3] * T1r); e.w = sv.w + (Pp[2] * T0i + Pp[3] * T1i);
				a.seg_end[(size_t)(cbase + c) * a.nseg_cap + seg] = e;
				sendp[c] = e;
				if(a.fuse) {
					// publish for the next segment's workgroup: each word carries the feed's epoch, so no flag and no fence
					// (an agent-scope release would write back the whole L2 of this XCD)
					unsigned long long *pub = a.seg_pub + ((size_t)(cbase + c) * a.nseg_cap + seg) * 4;
					const unsigned long long ep = (unsigned long long)a.pub_epoch << 32;
					__hip_atomic_store(pub + 0, ep | __float_as_uint(e.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(pub + 1, ep | __float_as_uint(e.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(pub + 2, ep | __float_as_uint(e.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(pub + 3, ep | __float_as_uint(e.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			// carry into the next tile = state at the end of lane 63's run
			if(lane == 63) carry[c] = make_float4(t0r[c], t0i[c], t1r[c], t1i[c]);
		}
	}
	K1_MARK(4);
	if(!a.fuse) { K1_END(); return; }

	// ---- fused K2: the first tile's outputs get the decayed state of the segment start ----
	// The state at the start of segment s is the zero-start state at the end of segment s-1 (a segment is >= kFixW blocks
	// long, older history has decayed below fp32 resolution), which the workgroup of s-1 publishes before it waits for
	// anything itself (in the epilogue of its last tile): a one-step look-back, no chain.  Workgroups are dispatched in
	// block-id order and s-1 always has a smaller block id, so the producer is running or done whenever a consumer waits -
	// as long as the launch's workgroups stay resident in that order.  Where they do not (a GPU time-sliced between processes
	// restores saved waves in any order), a consumer that has waited spin_limit polls stops waiting and WORKS THE STATE OUT ITSELF
	// (below): nothing fails, the workgroup just does one more tile.
	bool timed_out = false;
	if(wave_active) {
		if(seg == 0) {
			if(lane < CR) tsp[lane] = a.carry_in[cbase + lane < a.nchan ? cbase + lane : a.nchan - 1];
		} else if(lane < 4 * CR) {
			const int c = lane >> 2, ch = cbase + c < a.nchan ? cbase + c : a.nchan - 1;
			const unsigned long long *src = a.seg_pub + ((size_t)ch * a.nseg_cap + seg - 1) * 4 + (lane & 3);
			unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for(int spins = 0; (uint32_t)(w >> 32) != a.epoch; spins++) {
				if((uint32_t)spins > a.spin_limit) { timed_out = true; break; }
				__builtin_amdgcn_s_sleep(8);
				w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			reinterpret_cast<float *>(tsp)[lane] = __uint_as_float((uint32_t)w);
		}
	}
	// the whole workgroup has to agree (the fall-back stages a tile together)
	if(timed_out) fallback = 1;
	__syncthreads();
	K1_MARK(6);                                                  // 6: look-back wait
	if(fallback) {
		// Fall-back: the zero-start state at the end of the previous segment's LAST TILE is, to fp32 resolution, the state the
		// producer would have published (what lies further back has decayed by P^128 ~ 1e-17).  Stage that tile, let every lane run its
		// blocks from zero (the arithmetic of the main loop), weigh lane l's end state with Q^(63-l) and sum over the wavefront.
		// Differs from the published value in rounding only (a sum instead of a scan); rare, so it is written for clarity, not speed.
		if(tid == 0) atomicAdd(a.sync_timeouts, 1u);
		const int64_t sbase = ((int64_t)seg * a.tiles - 1) * tile_n;
		for(int t = tid; t < tile_n; t += 256) {
			const int64_t sidx = sbase + t;
			float re = 0.f, im = 0.f;
			if(sidx >= 0 && sidx < (int64_t)a.nlogical) load_sample(a, sidx, re, im);
			const int l = t / run, m = t - l * run;
			tile[m * 65 + l] = make_float2(re, im);
		}
		__syncthreads();
		if(wave_active) {
			const uint32_t nabs = (uint32_t)((a.n0 + (uint64_t)sbase + (uint64_t)lane * run) & 0xffffffu);
			const float4 qw = lane < 63 ? qpow[62 - lane] : make_float4(1.f, 0.f, 0.f, 1.f);      // Q^(63 - lane)
			#pragma unroll 1
			for(int c = 0; c < CR; c++) {
				uint32_t p = nabs * dph[c];
				float u0r = 0.f, u0i = 0.f, u1r = 0.f, u1i = 0.f;
				#pragma unroll 1
				for(int i = 0; i < R; i++) {
					v2f A0 = v2f{0.f, 0.f}, A1 = v2f{0.f, 0.f};
					const float2 *trow = tile + (size_t)(i * os) * 65 + lane;
					#pragma unroll 1
					for(int j = 0; j < os; j++) {
						const float2 x = trow[j * 65];
						const v2f X = v2f{x.x, x.y}, Xr = v2f{-x.y, x.x};
						const float F = (float)(p & 0xffffu);
						const float4 e = lut[(p >> 16) & 0xffu];
						const v2f sc = __builtin_elementwise_fma(v2f{e.z, e.w}, v2f{F, F}, v2f{e.x, e.y});
						const v2f m = __builtin_elementwise_fma(v2f{sc.y, sc.y}, X, v2f{sc.x, sc.x} * Xr);
						A0 = __builtin_elementwise_fma(v2f{bf.g0[j], bf.g0[j]}, m, A0);
						A1 = __builtin_elementwise_fma(v2f{bf.g1[j], bf.g1[j]}, m, A1);
						p += dph[c];
					}
					const float n0r = __builtin_fmaf(P0, u0r, __builtin_fmaf(P1, u1r, A0.x)), n0i = __builtin_fmaf(P0, u0i, __builtin_fmaf(P1, u1i, A0.y));
					const float n1r = __builtin_fmaf(P2, u0r, __builtin_fmaf(P3, u1r, A1.x)), n1i = __builtin_fmaf(P2, u0i, __builtin_fmaf(P3, u1i, A1.y));
					u0r = n0r; u0i = n0i; u1r = n1r; u1i = n1i;
				}
				float s0r = qw.x * u0r + qw.y * u1r, s0i = qw.x * u0i + qw.y * u1i, s1r = qw.z * u0r + qw.w * u1r, s1i = qw.z * u0i + qw.w * u1i;
				#pragma unroll
				for(int d = 1; d < 64; d <<= 1) { s0r += __shfl_xor(s0r, d); s0i += __shfl_xor(s0i, d); s1r += __shfl_xor(s1r, d); s1i += __shfl_xor(s1i, d); }
				if(lane == 0) tsp[c] = make_float4(s0r, s0i, s1r, s1i);
			}
		}
	}
	if(!wave_active) return;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const int64_t seglen = (int64_t)a.tiles * L;
	const int64_t kloc0 = (int64_t)seg * seglen + (int64_t)lane * R;          // this lane's outputs in the segment's first tile
	#pragma unroll
	for(int c = 0; c < CR; c++) {
		const bool cvalid = cbase + c < a.nchan;
		const float4 ts = tsp[c];
		const int i0 = lane * R;
		// exactly K2's arithmetic: v += cP[i][0] * ts.(x|y) + cP[i][1] * ts.(z|w)
		float v0r = hold[c][0], v0i = hold[c][1], v1r = hold[c][2], v1i = hold[c][3];
		if(!kHoldRegs) {
			const cf32 *yin = a.y + (size_t)(cbase + c) * a.cap;
			const uint32_t r0 = (uint32_t)(a.k0 + kloc0) & a.mask;
			if(R > 1 && cvalid && kloc0 + 1 < a.D && (r0 & 1u) == 0) { const float4 q4 = *reinterpret_cast<const float4 *>(yin + r0); v0r = q4.x; v0i = q4.y; v1r = q4.z; v1i = q4.w; }
			else {
				if(cvalid && kloc0 < a.D) { const cf32 q2 = yin[r0]; v0r = q2.re; v0i = q2.im; }
				if(R > 1 && cvalid && kloc0 + 1 < a.D) { const cf32 q2 = yin[(uint32_t)(a.k0 + kloc0 + 1) & a.mask]; v1r = q2.re; v1i = q2.im; }
			}
		}
		v0r += a.bfd->cP[i0][0] * ts.x + a.bfd->cP[i0][1] * ts.z; v0i += a.bfd->cP[i0][0] * ts.y + a.bfd->cP[i0][1] * ts.w;
		if(R > 1) { v1r += a.bfd->cP[i0 + 1][0] * ts.x + a.bfd->cP[i0 + 1][1] * ts.z; v1i += a.bfd->cP[i0 + 1][0] * ts.y + a.bfd->cP[i0 + 1][1] * ts.w; }
		cf32 *yout = a.y + (size_t)(cbase + c) * a.cap;
		const uint32_t s0 = (uint32_t)(a.k0 + kloc0) & a.mask;
		if(R > 1 && cvalid && kloc0 + 1 < a.D && (s0 & 1u) == 0) *reinterpret_cast<float4 *>(yout + s0) = make_float4(v0r, v0i, v1r, v1i);
		else {
			if(cvalid && kloc0 < a.D) yout[s0] = cf32{v0r, v0i};
			if(R > 1 && cvalid && kloc0 + 1 < a.D) yout[(uint32_t)(a.k0 + kloc0 + 1) & a.mask] = cf32{v1r, v1i};
		}
		// the filter state handed to the next feed (K2's k == D-1 branch)
		if(cvalid && lane == 0 && (int64_t)(seg + 1) * seglen >= a.D) {
			const int64_t len = a.D - (int64_t)seg * seglen;
			float4 e = sendp[c];
			if(len <= kFixW) {
				const float *Pp = a.bfd->Ppow[len];
				e.x += Pp[0] * ts.x + Pp[1] * ts.z; e.y += Pp[0] * ts.y + Pp[1] * ts.w;
				e.z += Pp[2] * ts.x + Pp[3] * ts.z; e.w += Pp[2] * ts.y + Pp[3] * ts.w;
			}
			a.carry_out[cbase + c] = e;
		}
	}
	K1_MARK(7);                                                  // 7: fix-up of the first tile's outputs
	K1_END();
}

struct K2Args {
	cf32 *y; const float4 *seg_end; const float4 *carry_in; float4 *carry_out;
	const BlockForm *bf;
	int64_t k0, D; uint32_t cap, mask, nseg_cap; int32_t seglen;
};

// K2 (VDL2HIP_NO_FUSE only): add the decayed segment-start state to the first kFixW outputs of every workgroup segment.
__global__ __launch_bounds__(256) void k_fixup(K2Args a) {
	const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int c = blockIdx.y;
	if(k >= a.D) return;
	const BlockForm &bf = *a.bf;
	const uint32_t slot = (uint32_t)(a.k0 + k) & a.mask;
	const int seg = (int)(k / a.seglen), i = (int)(k - (int64_t)seg * a.seglen);
	if(i < kFixW) {
		cf32 v = a.y[(size_t)c * a.cap + slot];
		const float4 ts = seg ? a.seg_end[(size_t)c * a.nseg_cap + seg - 1] : a.carry_in[c];
		v.re += bf.cP[i][0] * ts.x + bf.cP[i][1] * ts.z;
		v.im += bf.cP[i][0] * ts.y + bf.cP[i][1] * ts.w;
		a.y[(size_t)c * a.cap + slot] = v;
	}
	if(k == a.D - 1) {   // filter state handed to the next feed
		const int len = i + 1;
		float4 e = a.seg_end[(size_t)c * a.nseg_cap + seg];
		if(len <= kFixW) {
			const float4 ts = seg ? a.seg_end[(size_t)c * a.nseg_cap + seg - 1] : a.carry_in[c];
			const float *Pp = bf.Ppow[len];
			e.x += Pp[0] * ts.x + Pp[1] * ts.z; e.y += Pp[0] * ts.y + Pp[1] * ts.w;
			e.z += Pp[2] * ts.x + Pp[3] * ts.z; e.w += Pp[2] * ts.y + Pp[3] * ts.w;
		}
		a.carry_out[c] = e;
	}
}

// keep the input samples that did not fill a whole decimation block (process_samples()' cnt, demod.c:322)
__global__ void k_carry(K1Args a, void *carry_out, uint32_t nrem) {
	const uint32_t i = threadIdx.x;
	if(i >= nrem) return;
	const int64_t s = (int64_t)(a.nlogical - nrem) + i;
	if(a.fmt == 1) {
		const uint32_t *p = (s < (int64_t)a.ncarry) ? (const uint32_t *)a.carry + s : (const uint32_t *)a.in + (s - a.ncarry);
		((uint32_t *)carry_out)[i] = *p;
	} else {
		const uint16_t *p = (s < (int64_t)a.ncarry) ? (const uint16_t *)a.carry + s : (const uint16_t *)a.in + (s - a.ncarry);
		((uint16_t *)carry_out)[i] = *p;
	}
}

// the per-feed counters of the output control block start at zero - the record and octet counters behind the shares the burst
// decoder's wavefronts own from the start (vdl2_core.h: burst_reserve_*); the capacities stay
__device__ __forceinline__ void reset_out_ctl(OutCtl *ctl, uint32_t k5_waves) {
	ctl->nbursts = 0; ctl->overflow = 0; ctl->nvalid = 0; ctl->pool_out_used = 0;
	ctl->nframes = burst_reserve_initial_frames(k5_waves); ctl->pool_used = burst_reserve_initial_pool(k5_waves);
}
__global__ void k_reset_ctl(OutCtl *ctl, uint32_t k5_waves) { reset_out_ctl(ctl, k5_waves); }   // for a feed too short to have a front

struct K3Args {
	const cf32 *y; cf32 *pf; uint64_t *cand; uint64_t *flag; const Tables *tab;
	int64_t nbase, k1;        // first sample to (re)compute (multiple of 64); one past the last valid sample
	uint32_t cap, mask;
	int32_t wpl;              // exact tier: flag words scanned per lane (1..kK3bWordsPerLane)
	OutCtl *ctl; uint32_t k5_waves;   // the feed's output control block, reset here (the last kernel of the front, so that no copy has to do it)
	// referee (nullptr: off): the feed's hook - written here from `refv`, for the same reason - and what the candidate verdict needs
	RefChan *ref; RefChan refv; float max_ppm; const float *ppm_thr; int32_t ref_on; uint32_t *rq_n, *rq_flag; RefBad *rq_bad; ScanReq *pq; uint32_t pq_cap; uint32_t *rq_flag2;      // rq_n, rq_flag, rq_bad: the feed's list of decisions to check / its "walk again" flags / the decisions that fell, reset here;      // (ref != nullptr, ref_on == 0: the hook is written, the verdicts are the plain ones)
};

// K3: got_sync() metric (contiguous ring) + the candidate bitmap, in two tiers and two kernels.
//
// k_sync_screen - every decimated sample.  A block turns kK3Tile consecutive outputs (+150 back) into screening-precision
// phases (phase_fast, in turns) in LDS and gives every sample the cheap screening value of the metric (vdl2_core.h: same unwrap
// decisions, running sums, float only).  The windows of samples n, n+10, ..., n+10(S-1) (S = kK3Share = 8) read the same taps
// one place apart, so one lane takes those S: for the 12 taps before the early exit it reads 11+S phases for what would be 12 S,
// forms the tap-to-tap differences once, and walks the S windows one after the other (each with the wavefront-wide early exit).
// Ten consecutive lanes cover 10 S consecutive samples; 320 threads cover the tile.  Verdicts go through LDS to be regrouped into words of 64 consecutive samples.  Output: one flag
// bit per sample - "the exact value may be under the threshold".
//
// k_sync_exact4 (and its older form k_sync_exact) - only where a flag is set (on noise 3e-5 of the samples): the reference's arithmetic - atan2 in double on the
// 16 taps, the double-precision unwrap, the centred regression - for the flagged samples and 3 samples either side (those are
// y1/y3 of calc_para_vertex and the right-hand side of the candidate test), stored in pf, and the candidate bit
// pherr(n-3) < 4 && pherr(n) > pherr(n-3) of every sample.  The walker reads the metric nowhere else.  A sample whose right
// neighbour has not arrived yet is computed exactly; the next feed redoes the last partial bitmap word anyway.
#ifndef VDL2_K3_SHARE
#define VDL2_K3_SHARE 8              // windows (10 samples apart) per lane; tile = 320 threads x share.  8 against 4 (round 2): 27 phases read
                                     // for what would be 8 x 16, a halo of 150 on 2 560 samples instead of on 1 280: K3a 5 % faster (profiles/r03_k3_share8.txt)
#endif
constexpr int kK3Share = VDL2_K3_SHARE, kK3Threads = 320, kK3Tile = kK3Threads * kK3Share;
static_assert(kK3Tile == kK3Threads * kK3Share && kK3Tile % 64 == 0 && kK3Threads % 10 == 0 && (kK3Tile / 64) % (kK3Threads / 64) == 0, "screen tile");

__global__ __launch_bounds__(kK3Threads) void k_sync_screen(K3Args a) {
	if(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.rq_n) a.rq_n[3] = 0u;      // (the exact tier's list of stretches to make exact ahead of the walk)
	__shared__ float tile[kK3Tile + 150];           // screening phases of samples nblk-150 .. nblk+kK3Tile-1
	__shared__ uint8_t verdict[kK3Tile];
	const int c = blockIdx.y, tid = threadIdx.x;
	const int64_t nblk = a.nbase + (int64_t)blockIdx.x * kK3Tile;
	const cf32 *y = a.y + (size_t)c * a.cap;
	constexpr int kFull = (kK3Tile + 150) / kK3Threads, kRest = (kK3Tile + 150) % kK3Threads;
	if(nblk >= 150 && nblk + kK3Tile <= a.k1) {
		// the whole tile and its history exist (all but the first and last block of a channel): no per-sample range tests, ring
		// offsets in 32 bits, all loads in flight before the first phase is worked out
		const uint32_t r0 = (uint32_t)(nblk - 150) + tid;
		cf32 v[kFull + 1];
		#pragma unroll
		for(int k = 0; k <= kFull; k++) if(k < kFull || tid < kRest) v[k] = y[(r0 + kK3Threads * k) & a.mask];
		#pragma unroll
		for(int k = 0; k <= kFull; k++) if(k < kFull || tid < kRest) tile[tid + kK3Threads * k] = phase_fast(v[k]);
	} else {
		for(int j = tid; j < kK3Tile + 150; j += kK3Threads) {
			const int64_t t = nblk - 150 + j;
			tile[j] = (t < 0 || t >= a.k1) ? 0.f : phase_fast(y[(uint32_t)t & a.mask]);
		}
	}
	__syncthreads();
	// this lane's windows: samples nblk + s0 + 10 q, q = 0..3; tap i of window q is tile[s0 + 10 (q + i)]
	const int s0 = 10 * kK3Share * (tid / 10) + tid % 10;
	constexpr int kEarlyPh = kScreenEarly + kK3Share - 1;             // phases the first kScreenEarly taps of the lane's windows touch
	float ph[kEarlyPh], d[kEarlyPh];
	#pragma unroll
	for(int k = 0; k < kEarlyPh; k++) ph[k] = tile[s0 + 10 * k];
	#pragma unroll
	for(int k = 1; k < kEarlyPh; k++) d[k] = ph[k] - ph[k - 1];
	#pragma unroll
	for(int q = 0; q < kK3Share; q++) {
		// the first kScreenEarly taps bound the value from below: most wavefronts stop here
		ScreenAcc acc;
		screen_begin(acc);
		screen_taps(&d[q], 1, kScreenEarly, acc);
		float ps = screen_value(acc, kScreenEarly);
		if(__any(ps < kScreenEarlyThr)) {
			float pl[kPreamble - kScreenEarly + 1], dl[kPreamble];
			#pragma unroll
			for(int j = 0; j <= kPreamble - kScreenEarly; j++) pl[j] = tile[s0 + 10 * (q + kScreenEarly - 1 + j)];
			#pragma unroll
			for(int j = 0; j < kPreamble - kScreenEarly; j++) dl[kScreenEarly + j] = pl[j + 1] - pl[j];
			screen_taps(dl, kScreenEarly, kPreamble, acc);
			ps = screen_value(acc, kPreamble);
		}
		verdict[s0 + 10 * q] = !(ps >= kScreenThr);      // a NaN (sample too small for phase_fast) goes to the exact tier
	}
	__syncthreads();
	#pragma unroll
	for(int j = 0; j < (kK3Tile / 64) / (kK3Threads / 64); j++) {
		const int word = (tid >> 6) * ((kK3Tile / 64) / (kK3Threads / 64)) + j;
		const int64_t n = nblk + 64 * word + (tid & 63);
		const unsigned long long bits = __ballot(n < a.k1 && verdict[64 * word + (tid & 63)] != 0);
		if((tid & 63) == 0 && n < a.k1) a.flag[(size_t)c * (a.cap >> 6) + ((uint32_t)(n >> 6) & (a.mask >> 6))] = bits;
	}
}

constexpr int kK3bWordsPerLane = 4;      // at most; fewer when that leaves the chip short of wavefronts (few channels)
// The exact tier, a word at a time.  A word with work is taken by the whole wavefront: the 224 samples its evaluations can read (the
// word and 160 before it) are loaded once, coalesced, and turned into exact phases (atan2 in double, four per lane) - and, for the
// referee, into the bounds on their errors (ref_eps2_of: a sample and its three predecessors) - in LDS; then every sample that has
// work is ONE lane's: sixteen phases out of LDS, the reference's metric on them in its own operation order (sync_metric), the
// referee's error figure beside it.  Same atan2, same metric, bit-identical values as the four-lanes-per-sample form it replaces
// (which read every tap from memory - five loads per tap with the referee - and kept three lanes in four idle during the metric).
// FULL: sync_metric_ref()'s `full` (receivers that scan ahead of the walk; a template so that the other form is the old code exactly)
template<bool FULL>
__global__ __launch_bounds__(256, 4) void k_sync_exact4(K3Args a) {
	if(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { reset_out_ctl(a.ctl, a.k5_waves); if(a.ref) *a.ref = a.refv; if(a.rq_n) { a.rq_n[0] = 0u; a.rq_n[1] = 0u; a.rq_n[2] = 0u; a.rq_n[4] = 0u; a.rq_n[5] = 0u; a.rq_n[6] = 0u; } }   // (rq_n[1], [2]: the burst decoder's lists, BurstDefer)
	if(blockIdx.x == 0 && threadIdx.x == 0 && a.rq_flag) { a.rq_flag[blockIdx.y] = 0u; a.rq_bad[blockIdx.y].n = 0u; if(a.rq_flag2) a.rq_flag2[blockIdx.y] = 0u; }
	constexpr int kBack = 160, kSpan = kBack + 64;        // staged samples: base - 160 .. base + 63
	// [wave][6 + bit]: metric of sample word*64 + bit (entries 0..5 = the six samples before the word), its slope, and - for the
	// referee - its error figure E and its value with the one discontinuity taken the other way (vdl2_core.h: sync_metric_ref)
	__shared__ float psh[4][64 + 6], fsh[4][64 + 6], esh[4][64 + 6], ash[4][64 + 6], bsh[4][64 + 6];
	__shared__ float phs[4][kSpan], e2s[4][kSpan];
	__shared__ uint64_t s_need[4][64 * kK3bWordsPerLane];
	__shared__ uint8_t s_fprev[4][64 * kK3bWordsPerLane];
	__shared__ uint16_t s_list[4][64 * kK3bWordsPerLane];
	const int c = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const cf32 *y = a.y + (size_t)c * a.cap;
	const uint64_t *flag = a.flag + (size_t)c * (a.cap >> 6);
	uint64_t *cand = a.cand + (size_t)c * (a.cap >> 6);
	const uint32_t wmask = a.mask >> 6;
	const Tables &T = *a.tab;
	const bool ref_on = a.ref != nullptr && a.ref_on != 0;
	const float ppm_thr = ref_on ? a.ppm_thr[c] : 0.f;
	const int64_t w0 = a.nbase >> 6, w1 = (a.k1 + 63) >> 6;
	const int64_t wb = w0 + ((int64_t)blockIdx.x * 4 + wave) * (64 * a.wpl);   // first word of this wavefront
	int nwork = 0;
	for(int g = 0; g < a.wpl; g++) {
		const int64_t w = wb + 64 * g + lane;
		uint64_t need = 0, fprev = 0, fpp = 0, f0 = 0;
		if(w < w1) {
			f0 = flag[(uint32_t)w & wmask];
			fprev = w > 0 ? flag[(uint32_t)(w - 1) & wmask] : 0ull;      // words before nbase hold the previous feed's flags
			fpp = w > 1 ? flag[(uint32_t)(w - 2) & wmask] : 0ull;
			const uint64_t fnext = w + 1 < w1 ? flag[(uint32_t)(w + 1) & wmask] : 0ull;
			need = f0 | (f0 << 3) | (f0 >> 3) | (fprev >> 61) | (fnext << 61);
			const int64_t base = w << 6;
			if(a.k1 - 3 < base + 64) {                                      // right neighbour n+3 not there yet
				const int64_t lo = a.k1 - 3 - base;
				need |= lo <= 0 ? ~0ull : (~0ull << lo);
			}
			if(a.k1 < base + 64) need &= (a.k1 - base <= 0) ? 0ull : (~0ull >> (64 - (a.k1 - base)));   // samples that exist
			if(need == 0) cand[(uint32_t)w & wmask] = 0;
		}
		// which of the six samples before the word have a tabulated metric: the last six bits of the previous word's work mask
		const uint64_t need_prev = fprev | (fprev << 3) | (fprev >> 3) | (fpp >> 61) | (f0 << 61);
		s_need[wave][64 * g + lane] = need; s_fprev[wave][64 * g + lane] = (uint8_t)(need_prev >> 58);
		const unsigned long long busy = __ballot(need != 0);
		if(need != 0) s_list[wave][nwork + __builtin_popcountll(busy & ((1ull << lane) - 1ull))] = (uint16_t)(64 * g + lane);
		nwork += __builtin_popcountll(busy);
	}
	WAVE_SYNC();
	float *ps = psh[wave], *fs = fsh[wave], *es = esh[wave], *as = ash[wave], *bs = bsh[wave], *ph = phs[wave], *e2 = e2s[wave];
	#pragma unroll 1
	for(int it = 0; it < nwork; it++) {
		const int idx = (int)s_list[wave][it];
		const uint64_t needj = s_need[wave][idx];
		const uint32_t fprevj = s_fprev[wave][idx] & 63u;                 // which of the six samples before the word are tabulated (bit k: sample -6 + k)
		const int64_t wj = wb + idx, r0 = (wj << 6) - kBack;
		// ---- stage: phases (and |y|^2, for the error bounds) of samples r0 .. r0 + 223, four per lane, all loads first ----
		{
			cf32 yv[4]; bool in[4];
			#pragma unroll
			for(int k = 0; k < 4; k++) {
				const int64_t t = r0 + lane + 64 * k;
				in[k] = 64 * k + lane < kSpan && t >= 0 && t < a.k1;
				yv[k] = in[k] ? y[(uint32_t)t & a.mask] : cf32{0.f, 0.f};
			}
			#pragma unroll
			for(int k = 0; k < 4; k++) if(64 * k + lane < kSpan) {
				ph[64 * k + lane] = in[k] ? phase_of(yv[k]) : 0.f;
				e2[64 * k + lane] = in[k] ? yv[k].re * yv[k].re + yv[k].im * yv[k].im : -1.f;     // (-1: outside the stream - its phase is exactly the reference's)
			}
		}
		ps[6 + lane] = kPherrBig; if(lane < 6) ps[lane] = kPherrBig;
		WAVE_SYNC();
		if(ref_on) {
			// |y|^2 -> the squared bound on the phase error (ref_eps2_of), in place: every lane reads its four and their predecessors first
			float ev[4];
			#pragma unroll
			for(int k = 0; k < 4; k++) {
				const int i = 64 * k + lane;
				ev[k] = 0.f;
				if(i >= 3 && i < kSpan) {
					const float m0 = e2[i], m1 = e2[i - 1], m2 = e2[i - 2], m3 = e2[i - 3];
					ev[k] = m0 < 0.f ? 0.f : ref_eps2_of(m0, fmaxf(m1, 0.f), fmaxf(m2, 0.f), fmaxf(m3, 0.f));
				}
			}
			WAVE_SYNC();
			#pragma unroll
			for(int k = 0; k < 4; k++) if(64 * k + lane < kSpan) e2[64 * k + lane] = ev[k];
			WAVE_SYNC();
		}
		// ---- one lane per sample with work: the word's own (bit = lane), then the six before it (lanes 0..5) ----
		#pragma unroll 1
		for(int round = 0; round < 2; round++) {
			const int bit = round == 0 ? lane : lane - 6;
			const bool todo = round == 0 ? ((needj >> lane) & 1ull) != 0 : (lane < 6 && ((fprevj >> lane) & 1u) && (wj << 6) + bit >= 0);
			if(todo) {
				const int i0 = bit + kBack - 150;                            // staged index of tap 0
				float p[kPreamble];
				#pragma unroll
				for(int i = 0; i < kPreamble; i++) p[i] = ph[i0 + 10 * i];
				float pv, fv, E = 0.f, pa, pb;
				if(ref_on) sync_metric_ref(p, e2 + i0, 10, T, pv, fv, E, pa, pb, FULL);
				else { sync_metric(p, T, pv, fv); pa = pb = pv; }
				ps[6 + bit] = pv; fs[6 + bit] = fv; es[6 + bit] = E; as[6 + bit] = pa; bs[6 + bit] = pb;
			}
		}
		WAVE_SYNC();
		// the verdict of every sample of the word as a candidate: got_sync() may fire there (bitmap) / some decision of the fire is
		// within the margin of the stream's error (sign of the tabulated metric: the walker has it checked on the reference's samples)
		const int64_t n = (wj << 6) + lane;
		const float p0 = ps[6 + lane], p3 = ps[3 + lane], p6 = ps[lane];
		int vd = 0;
		if(n >= 3 && n < a.k1) {
			if(!ref_on) vd = is_candidate(p3, p0) ? 1 : 0;
			else vd = ref_candidate_verdict(ref_pherr_range(p0, as[6 + lane], bs[6 + lane], es[6 + lane]), ref_pherr_range(p3, as[3 + lane], bs[3 + lane], es[3 + lane]), fs[3 + lane], es[3 + lane],
			                                ref_pherr_range(p6, as[lane], bs[lane], es[lane]), a.max_ppm, ppm_thr);
		}
		if((needj >> lane) & 1ull) a.pf[(size_t)c * a.cap + ((uint32_t)n & a.mask)] = cf32{ (vd & 2) ? -p0 : p0, fs[6 + lane] };
		const unsigned long long bits = __ballot((vd & 1) != 0);
		if(lane == 0) cand[(uint32_t)wj & wmask] = bits;
		if(a.pq) {
			// the marked candidates of the word: the stretch their decisions read goes on the list of those made exact before the walk
			const unsigned long long mk = __ballot((vd & 2) != 0 && ((needj >> lane) & 1ull) != 0);
			if(mk && lane == 0) {
				const int64_t nf = (wj << 6) + __builtin_ctzll(mk), nl = (wj << 6) + 63 - __builtin_clzll(mk);
				int64_t lo = (nf - kRefPre) & ~255ll, hi = (nl + kRefPost) | 255; if(lo < 0) lo = 0; if(hi > a.k1 - 1) hi = a.k1 - 1;
				const uint32_t i = atomicAdd(a.rq_n + 3, 1u);
				if(i < a.pq_cap) a.pq[i] = ScanReq{ c, REF_CANDIDATE, lo, hi };
			}
		}
		WAVE_SYNC();
	}
}

struct K4Args {
	const cf32 *y; const cf32 *pf; const uint64_t *cand; const Tables *tab;
	WalkState *ws; unsigned long long *cnt; Burst *bursts; uint32_t *nb_chan; uint32_t cap_bursts_chan; OutCtl *ctl; const uint32_t *freq;
	EvalChunk *log; uint32_t *nlog; uint32_t cap_log;
	int64_t k_end; float max_ppm; uint32_t cap, mask; int32_t chan_first, nchan;
	const float *ppm_thr;      // per channel: ppm_gate_threshold(freq, max_ppm)
	RefChan *ref;              // referee hook of this feed (nullptr: off)
	uint32_t ref_launch;       // ... and a number that tells this launch from the others (k_walk_stitch: + 1, k_ref_verify: + 2, k_walk_again: + 3)
	// optimistic mode (rq != nullptr; vdl2_core.h: ref_verify): the feed's list of decisions to check, the "walk again" flag per channel,
	// and where the state and counters a channel's walk starts from are kept
	RefReq *rq; uint32_t *rq_n; uint32_t rq_cap; uint32_t *rq_flag; WalkState *ws_snap; unsigned long long *cnt_snap; RefBad *rq_bad; int32_t ref_pre;
	int32_t force_again;       // test hook (vdl2hip_debug_option "force_again"): the check flags EVERY channel, so that every channel's state and counters go back to the snapshot and the feed is stitched a second time
	// The walk of the NEXT feed does not wait for this feed's check (vdl2hip.hip: launch_rest): when it has run already, a channel
	// walked again (k_walk_stitch, again = 2) leaves its end state and counters in ws_tmp / cnt_tmp instead of the live rows and compares
	// them with what the next feed's walk started from (its snapshot: ws_snap_next / cnt_snap_next).  The same - all but always - and
	// the next feed's walk stands; else the snapshot is corrected, rq_flag2_next[c] set, and the next feed is stitched once more for
	// that channel from the corrected snapshot (again = 3, its own rq_flag2).
	WalkState *ws_tmp; unsigned long long *cnt_tmp; WalkState *ws_snap_next; unsigned long long *cnt_snap_next; uint32_t *rq_flag2, *rq_flag2_next;
	uint32_t *rq_flag2_next2;  // walk ahead by two feeds: the feed after the next has been walked as well and is stitched once more too (again = 4 for the next feed: into the snapshot the feed after it starts from; 3 for that one)
	int32_t force_mismatch;    // test hook: every channel walked again is taken to have ended differently (the next feed is redone for it)
};

__global__ __launch_bounds__(64, 4) void k_walk(K4Args a) {
	__shared__ WalkShared sh;
	const int c = blockIdx.x;
	ChanView v{ a.y + (size_t)c * a.cap, a.pf + (size_t)c * a.cap, a.cand + (size_t)c * (a.cap >> 6), a.mask, a.ref, c, a.ref_launch, a.rq, a.rq_n, a.rq_cap, a.rq_flag, a.rq_bad ? a.rq_bad + c : nullptr, a.ref_pre != 0 };
	EvalLog lg{ a.log + (size_t)c * a.cap_log, a.nlog + c };
	walk_channel(c, a.freq[c], a.max_ppm, a.ppm_thr[c], a.k_end, *a.tab, v, &a.ws[c], a.cnt + (size_t)c * kNumCounters,
	             a.bursts + (size_t)c * a.cap_bursts_chan, a.cap_bursts_chan, a.nb_chan + c, a.ctl, lg, sh, WalkSnap{ a.rq ? a.ws_snap : nullptr, a.cnt_snap });
}

// Referee, optimistic mode (vdl2_core.h: ref_verify): the decisions the walk noted are checked on the reference's own samples, a
// wavefront each, all at once - a scan takes ~3 ms, and the walk does not wait for it ...
__global__ __launch_bounds__(64, 4) void k_ref_verify(K4Args a) {
	__shared__ float lds[64];
	const uint32_t nreq = *a.rq_n < a.rq_cap ? *a.rq_n : a.rq_cap;
	if(a.force_again && blockIdx.x == 0) for(int c = (int)threadIdx.x; c < a.nchan; c += 64) a.rq_flag[c] = 1u;
	for(uint32_t i = blockIdx.x; i < nreq; i += gridDim.x) {
		const RefReq r = a.rq[i];
		const int c = r.chan;
		ChanView v{ a.y + (size_t)c * a.cap, a.pf + (size_t)c * a.cap, a.cand + (size_t)c * (a.cap >> 6), a.mask, a.ref, c, a.ref_launch + 2u };
		if(!ref_verify(r, a.freq[c], a.max_ppm, a.ppm_thr[c], a.k_end, *a.tab, v, lds) && threadIdx.x == 0) {
			a.rq_flag[c] = 1u;
			const uint32_t k = atomicAdd(&a.rq_bad[c].n, 1u);
			if(k < (uint32_t)kRefBad) a.rq_bad[c].at[k] = 4 * r.n + r.kind;
		}
		WAVE_SYNC();
	}
}
// ... and a channel one of whose decisions did not stand is walked again, from the state the feed started with
__global__ __launch_bounds__(64, 4) void k_walk_again(K4Args a) {
	__shared__ WalkShared sh;
	const int c = blockIdx.x;
	if(!a.rq_flag[c]) return;
	ChanView v{ a.y + (size_t)c * a.cap, a.pf + (size_t)c * a.cap, a.cand + (size_t)c * (a.cap >> 6), a.mask, a.ref, c, a.ref_launch + 3u };
	EvalLog lg{ a.log + (size_t)c * a.cap_log, a.nlog + c };
	walk_again(c, a.freq[c], a.max_ppm, a.ppm_thr[c], a.k_end, *a.tab, v, &a.ws[c], a.cnt + (size_t)c * kNumCounters,
	           a.bursts + (size_t)c * a.cap_bursts_chan, a.cap_bursts_chan, a.nb_chan + c, a.ctl, lg, sh, WalkSnap{ a.ws_snap, a.cnt_snap });
	if(threadIdx.x == 0) atomicAdd(a.ref->stats + 7, 1u);
}

// K4 in segments (vdl2_core.h "Speculative segments"): grid.x = 1 + 3*(nseg-1) walks per channel, then one stitcher per channel
struct K4sArgs {
	K4Args k; SpecOut *spec; uint32_t spec_stride; int32_t nseg; int64_t k0, seglen; uint32_t *seg_stats;
	int32_t again;             // k_walk_stitch, second launch (referee, optimistic mode): only the channels flagged "walk again" (vdl2_core.h: stitch_channel)
};

// Four walks per workgroup, one per wavefront (they share nothing): see k_nf_replay for why the back-end kernels that run
// beside the channeliser come in workgroups of four waves.
constexpr int kWalkWaves = 4;
__global__ __launch_bounds__(64 * kWalkWaves, 4) void k_walk_spec(K4sArgs s) {
	__shared__ WalkShared shw[kWalkWaves];
	const K4Args &a = s.k;
	const int wave = threadIdx.x >> 6;
	const int c = blockIdx.y, x = blockIdx.x * kWalkWaves + wave;
	if(x >= 1 + 3 * (s.nseg - 1)) return;
	WalkShared &sh = shw[wave];
	ChanView v{ a.y + (size_t)c * a.cap, a.pf + (size_t)c * a.cap, a.cand + (size_t)c * (a.cap >> 6), a.mask, a.ref, c, a.ref_launch, a.rq, a.rq_n, a.rq_cap, a.rq_flag, a.rq_bad ? a.rq_bad + c : nullptr, a.ref_pre != 0 };
	if(x == 0) return;         // (segment 0 is walked from the real state by the stitcher)
	{
		const int seg = 1 + (x - 1) / 3, r = (x - 1) % 3;
		const int64_t b = s.k0 + (int64_t)seg * s.seglen, kn = seg + 1 < s.nseg ? b + s.seglen : a.k_end;
		spec_walk(c, a.freq[c], a.max_ppm, a.ppm_thr[c], b, r, kn, *a.tab, v, s.spec + (size_t)c * s.spec_stride + (x - 1), sh);
	}
}

// (64 threads are launched; the declared bound of 256 threads x 4 waves per SIMD is what gives the kernel the 128-register
// budget: with the true bound the compiler sizes the budget by the LDS-limited occupancy and takes 166, and a wave that needs
// more registers than one channeliser wave frees (128) waits for two of them to retire at once - measured 25 us alone, 1.6 ms
// beside the channeliser.  The same for k_burst.)
// Two channels per workgroup (dynamic LDS, as in k_burst: four would be 67 KB).
constexpr int kStitchWaves = 2;
struct StitchLds { WalkShared sh; StitchShared ss; };
__global__ __launch_bounds__(256, 4) void k_walk_stitch(K4sArgs s) {
	extern __shared__ __align__(16) unsigned char k4_lds[];       // StitchLds[kStitchWaves]
	const K4Args &a = s.k;
	const int wave = threadIdx.x >> 6, c = blockIdx.x * kStitchWaves + wave;      // a channel per wavefront
	if(c >= a.nchan) return;
	// again: 0 the feed's walk; 1 a flagged channel once more, nothing walked after this feed yet (state and counters: the live rows);
	// 2 the same when the next feed HAS been walked (K4Args: ws_tmp ...); 3 this feed once more for a channel whose start state the
	// previous feed's second walk has corrected
	// 4: as 3 when the feed AFTER this one has been walked too (walk ahead by two): the end state and counters go where that feed starts
	// from (its snapshot: ws_snap_next), and it is stitched once more itself (3; its flag was set together with this feed's); 5: as 3,
	// as the second feed behind the corrected one (launch kind 10: ref_entry_visible())
	const int mode = s.again;
	if((mode == 1 || mode == 2) && !a.rq_flag[c]) return;
	if(mode >= 3 && !a.rq_flag2[c]) return;
	StitchLds &lds = reinterpret_cast<StitchLds *>(k4_lds)[wave];
	// (launch kinds, ref_entry_visible(): a.ref_launch is kind 1; stitch 2, walk again 4, the corrected feed's second walk 9 / 10)
	ChanView v{ a.y + (size_t)c * a.cap, a.pf + (size_t)c * a.cap, a.cand + (size_t)c * (a.cap >> 6), a.mask, a.ref, c, a.ref_launch + (mode == 0 ? 1u : mode == 5 ? 9u : mode >= 3 ? 8u : 3u), mode ? nullptr : a.rq, a.rq_n, a.rq_cap, a.rq_flag,
	            // (mode 3, 4, 5: this feed's check may not have run yet - no speculative walk that noted decisions is adopted: spec_requests_stand() without a list)
	            (a.rq_bad && mode < 3) ? a.rq_bad + c : nullptr, a.ref_pre != 0 };
	EvalLog lg{ a.log + (size_t)c * a.cap_log, a.nlog + c };
	WalkState *gstate = mode == 2 ? &a.ws_tmp[c] : mode == 4 ? &a.ws_snap_next[c] : &a.ws[c];
	unsigned long long *cnt = (mode == 2 ? a.cnt_tmp : mode == 4 ? a.cnt_snap_next : a.cnt) + (size_t)c * kNumCounters;
	stitch_channel(c, a.freq[c], a.max_ppm, a.ppm_thr[c], s.k0, s.seglen, s.nseg, a.k_end, *a.tab, v, gstate, cnt,
	               a.bursts + (size_t)c * a.cap_bursts_chan, a.cap_bursts_chan, a.nb_chan + c, a.ctl, lg,
	               s.spec + (size_t)c * s.spec_stride, lds.sh, lds.ss, s.seg_stats + 2 * c, WalkSnap{ (a.rq || mode) ? a.ws_snap : nullptr, a.cnt_snap }, mode != 0);
	if(mode && (threadIdx.x & 63) == 0) atomicAdd(a.ref->stats + 7, 1u);
	if(mode == 2) {
		// the channel's end state and counters after the second walk against what the next feed's walk started from
		WAVE_SYNC_GLOBAL();
		const int lane = threadIdx.x & 63;
		bool same = true;
		if(lane == 0) same = walk_state_equal(a.ws_tmp[c], a.ws_snap_next[c]) && !a.force_mismatch;
		if(lane < kNumCounters) same = same && cnt[lane] == a.cnt_snap_next[(size_t)c * kNumCounters + lane];
		if(__any(!same)) {
			if(lane == 0) { a.ws_snap_next[c] = a.ws_tmp[c]; a.rq_flag2_next[c] = 1u; if(a.rq_flag2_next2) a.rq_flag2_next2[c] = 1u; atomicAdd(a.ref->stats + 8, 1u); }
			if(lane < kNumCounters) a.cnt_snap_next[(size_t)c * kNumCounters + lane] = cnt[lane];
		}
	}
	if(mode >= 3 && (threadIdx.x & 63) == 0) a.rq_flag2[c] = 0u;
}

struct K4bArgs {
	const cf32 *y; NfState *nf; EvalChunk *log; uint32_t *nlog; int64_t *sc_first; int64_t *sc_cum;
	NfFeed *feed; float *lpbuf; float *ring; uint32_t ring_mask; uint32_t cap, mask, cap_log, cap_comb, cap_hist; int32_t nchan;
};

// K4b: noise-floor replay from the walker's evaluation log, in three small passes
constexpr int kNfWaves = 4;                // wavefronts per workgroup in the three passes: a channel (passes 1, 3) or a group of updates (pass 2) each
__global__ __launch_bounds__(64 * kNfWaves, 4) void k_nf_prepare(K4bArgs a) {
	extern __shared__ __align__(16) unsigned char nf_lds[];      // NfShared[kNfWaves]; dynamic, so that the compiler sizes the register budget by the launch bound and not by the LDS-limited occupancy (see k_burst)
	NfShared *shw = reinterpret_cast<NfShared *>(nf_lds);
	const int wave = threadIdx.x >> 6, c = blockIdx.x * kNfWaves + wave;
	if(c >= a.nchan) return;
	EvalLog lg{ a.log + (size_t)c * a.cap_log, a.nlog + c };
	NfScratch sc{ a.sc_first + (size_t)c * (a.cap_comb + 1), a.sc_cum + (size_t)c * (a.cap_comb + 1) };
	nf_prepare(&a.nf[c], lg, sc, a.cap_comb, &a.feed[c], shw[wave]);
}

// Four wavefronts per workgroup, each with a group of updates of its own: a workgroup then takes exactly the room one
// channeliser workgroup leaves on a CU (a wave per SIMD inside its 120 registers, 37 KB of LDS), where single-wave workgroups
// each kept a whole channeliser workgroup out for as long as they lived (measured: the replay beside the channeliser cost the
// front 0.29 ms per 256-channel step).
__global__ __launch_bounds__(64 * kNfWaves, 4) void k_nf_replay(K4bArgs a) {
	extern __shared__ __align__(16) unsigned char nf_lds[];      // NfShared[kNfWaves]; dynamic, so that the compiler sizes the register budget by the launch bound and not by the LDS-limited occupancy (see k_burst)
	NfShared *shw = reinterpret_cast<NfShared *>(nf_lds);
	const int c = blockIdx.y, wave = threadIdx.x >> 6;
	ChanView v{ a.y + (size_t)c * a.cap, nullptr, nullptr, a.mask };
	NfScratch sc{ a.sc_first + (size_t)c * (a.cap_comb + 1), a.sc_cum + (size_t)c * (a.cap_comb + 1) };
	const NfFeed fd = a.feed[c];
	for(int64_t g = (int64_t)blockIdx.x * kNfWaves + wave; fd.u0 + 1 + kNfGroup * g <= fd.u1; g += (int64_t)gridDim.x * kNfWaves) {
		nf_replay_group(v, sc, fd, g, a.lpbuf + (size_t)c * a.cap_hist, a.cap_hist, shw[wave]);
		WAVE_SYNC();
	}
}

__global__ __launch_bounds__(64 * kNfWaves, 4) void k_nf_finish(K4bArgs a) {
	extern __shared__ __align__(16) unsigned char nf_lds[];      // NfShared[kNfWaves]; dynamic, so that the compiler sizes the register budget by the launch bound and not by the LDS-limited occupancy (see k_burst)
	NfShared *shw = reinterpret_cast<NfShared *>(nf_lds);
	const int wave = threadIdx.x >> 6, c = blockIdx.x * kNfWaves + wave;
	if(c >= a.nchan) return;
	NfScratch sc{ a.sc_first + (size_t)c * (a.cap_comb + 1), a.sc_cum + (size_t)c * (a.cap_comb + 1) };
	nf_finish(&a.nf[c], sc, a.feed[c], a.lpbuf + (size_t)c * a.cap_hist, a.ring + (size_t)c * (a.ring_mask + 1), a.ring_mask, a.cap_hist, shw[wave]);
}

// The three passes in one kernel, a wavefront per channel: for short blocks, where there are one or two updates per channel and
// three launches would cost more than the work (the reference's own block size, dumpvdl2.h:48: 4 000 decimated samples)
__device__ __forceinline__ void nf_all_body(const K4bArgs &a, unsigned char *lds, int block) {
	NfShared *shw = reinterpret_cast<NfShared *>(lds);
	const int wave = threadIdx.x >> 6, c = block * kNfWaves + wave;
	if(c >= a.nchan) return;
	ChanView v{ a.y + (size_t)c * a.cap, nullptr, nullptr, a.mask };
	EvalLog lg{ a.log + (size_t)c * a.cap_log, a.nlog + c };
	NfScratch sc{ a.sc_first + (size_t)c * (a.cap_comb + 1), a.sc_cum + (size_t)c * (a.cap_comb + 1) };
	nf_prepare(&a.nf[c], lg, sc, a.cap_comb, &a.feed[c], shw[wave]);
	WAVE_SYNC_GLOBAL();                    // the combined chunk list and the feed record are read back by other lanes
	const NfFeed fd = a.feed[c];
	for(int64_t g = 0; fd.u0 + 1 + kNfGroup * g <= fd.u1; g++) {
		nf_replay_group(v, sc, fd, g, a.lpbuf + (size_t)c * a.cap_hist, a.cap_hist, shw[wave]);
		WAVE_SYNC();
	}
	WAVE_SYNC_GLOBAL();                    // ... and so are the replayed mag_lp values
	nf_finish(&a.nf[c], sc, fd, a.lpbuf + (size_t)c * a.cap_hist, a.ring + (size_t)c * (a.ring_mask + 1), a.ring_mask, a.cap_hist, shw[wave]);
}
__global__ __launch_bounds__(64 * kNfWaves, 4) void k_nf_all(K4bArgs a) {
	extern __shared__ __align__(16) unsigned char nf_lds[];
	nf_all_body(a, nf_lds, (int)blockIdx.x);
}

struct K5Args {
	const cf32 *y; const Tables *tab; unsigned long long *cnt;
	const Burst *bursts; const uint32_t *nb_chan; uint32_t cap_bursts_chan; int32_t nchan;   // nb_chan[c]: bursts the walker has listed for channel c (its own list: no atomics on its critical path)
	OutFrame *frames; uint8_t *pool; OutCtl *ctl; const uint32_t *freq;
	uint32_t cap, mask;
	RefChan *ref;              // referee hook of this feed (nullptr: off)
	uint32_t ref_launch;
	BurstDefer df;             // pass 0: the referee's scans on the spot; 1: bursts that need one are listed; 2: the listed bursts
};

// ======================================================================
// Many scans side by side.  ref_exact_window_dev() spends a whole wavefront on one recursion - 64 lanes doing the same
// arithmetic for 3.3 ms: with a few hundred requests per feed that was 8 % (config4) to 28 % (config4_bursty) of the channeliser's
// time.  Here a workgroup takes kScanLanes requests at once: kScanProd producer wavefronts do the part without a recursion for one
// request and 64 input samples at a time (lane = sample), into LDS; one consumer wavefront runs the recursions, a step of all of
// them per turn.  Same arithmetic, operation for operation.
// Round 6: the consumer's lane is one COMPONENT of one request (lanes 0-15: I of requests 0-15, lanes 16-31: Q) and its arithmetic
// plain fp32 - v_mul, v_add, v_mul, v_add per step, the chain of three dependent operations 21 shader clocks - where round 5 ran a
// request per lane in packed (I, Q) arithmetic: dependent v_pk_* operations take 31 clocks per step on the bare recursion
// (dev/gpu_ubench_scan.hip, profiles/r06_ubench_scan.txt) and the kernel took 65.  The feed-forward values lie in LDS as
// [component][request][sample], so a lane fetches four steps with one ds_read_b128.  With the consumer at 23 clocks per step the
// producers became the bound: a wavefront alone on its SIMD issues one instruction per 4+ clocks, ~90 instructions per request and
// block of 60 samples -> two requests per producer (8 producers, 16 requests per workgroup) keep pace with the consumer, four did not
// (measured with either side switched off: consumer 1.35 ms per 136 192-sample scan, producers of four requests 2.3 ms).
// ======================================================================
constexpr int kScanLanes = 16, kScanProd = 8, kScanBlock = 64, kScanRow = kScanBlock + 4;
#ifndef VDL2_SCAN_WAVES
#define VDL2_SCAN_WAVES 11
#endif
constexpr int kScanWaves = VDL2_SCAN_WAVES;   // wavefronts launched per workgroup.  11: wavefronts 4 and 8 - which share SIMD 0 with the consumer (wavefront w runs on SIMD w mod 4) - leave at once, the 8 producers are wavefronts 1-3, 5-7, 9, 10: the consumer's chain of dependent operations has its SIMD to itself.  9: no idle wavefronts   // (row of 68 floats: 16-byte aligned, consecutive requests 4 banks apart)
struct ScanShared {
	alignas(16) float buf[2][2][kScanLanes][kScanRow];   // [block parity][component][request][sample] feed-forward values
	int64_t s_beg[kScanLanes], len[kScanLanes], n_lo[kScanLanes], n_hi[kScanLanes];
	uint32_t dphi[kScanLanes]; int32_t chan[kScanLanes], kind[kScanLanes]; uint32_t flags[kScanLanes];   // flags: 1 mix, 2 shortened run-up
	int64_t lmax, smin;
	v4f lut[256];                                // the NCO table
};
// retry_q / retry_n / retry_cap: where a scan that has NOT met its witness by its stretch's first output (see the consumer) is listed
// instead of being published; the host queues a second launch behind the first that takes that list as `sq` with warm_mul = 4 - the
// same scan from four times further back (nullptr: nothing is listed, an unmet scan is published as it is and counted).
// OS: the oversampling factor if it is 20 or 10 (a block is then 60 input samples - whole decimation periods - and the consumer's
// steps are straight-line code with the outputs at fixed places: a compare + branch per step cost as much as the arithmetic), else 0
template<int FMT, int OS>
__global__ __launch_bounds__(64 * kScanWaves) void k_ref_scan_multi(RefChan *rp, uint32_t launch, const ScanReq *sq, const RefReq *rq, const uint32_t *n_ptr, uint32_t cap, int64_t k_end,
                      ScanReq *retry_q, uint32_t *retry_n, uint32_t retry_cap, int32_t warm_mul) {
#if VDL2_DEVICE_PASS
	#pragma clang fp contract(off)
	__shared__ ScanShared sh;
	constexpr int BLK = OS ? (kScanBlock / (OS ? OS : 1)) * OS : kScanBlock;    // input samples per block
	const uint32_t ntot = *n_ptr < cap ? *n_ptr : cap;
	const uint32_t base = blockIdx.x * (uint32_t)kScanLanes;
	if(base >= ntot) return;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int os = rp->os, npiece = rp->npiece;
	const uint32_t mask = rp->mask, cap_y = rp->cap;
	const int64_t warm = rp->warm * (int64_t)(warm_mul > 1 ? warm_mul : 1);
	int64_t ps0[kRefPieces], pn[kRefPieces]; const void *pp[kRefPieces];
	#pragma unroll
	for(int j = 0; j < kRefPieces; j++) { ps0[j] = j < npiece ? rp->piece[j].s0 : 0; pn[j] = j < npiece ? rp->piece[j].n : 0; pp[j] = j < npiece ? rp->piece[j].p : nullptr; }
	const float A0 = rp->A0, A1 = rp->A1, A2 = rp->A2;
	uint32_t *stats = rp->stats;
	// ---- the requests of this workgroup: stretch, run-up, what has been done before (as ref_exact_window_dev) ----
	// (the whole first wavefront: four lanes per request - lanes r, r + 16, r + 32, r + 48 work out the same values and share the look-up in
	// the channel's ring of finished stretches, a quarter of its entries each; lane r alone counts and writes)
	static_assert(kScanLanes == 16, "four lanes of the first wavefront per request");
	if(threadIdx.x < 64) {
		const int r = threadIdx.x & (kScanLanes - 1), part = threadIdx.x >> 4;
		int64_t n_lo = 0, n_hi = -1, s_beg = 0, len = 0; int c = 0, kind = 0; uint32_t fl = 0;
		if(base + r < ntot && npiece > 0) {
			if(sq) { const ScanReq q = sq[base + r]; c = q.chan; kind = q.kind; n_lo = q.lo; n_hi = q.hi; }
			else { const RefReq q = rq[base + r]; c = q.chan; kind = q.kind; ref_request_window(q, k_end, n_lo, n_hi); }
			const int64_t in_end = ps0[npiece - 1] + pn[npiece - 1];
			if(n_lo < 0) n_lo = 0;
			if(kind == REF_STALE) kind = REF_CANDIDATE;                  // (asked for by the candidate search: counted and switched with it)
			bool go = n_hi >= n_lo && ((rp->kinds >> kind) & 1);
			if(go) {
				const int64_t last = in_end / os - 1;
				n_lo &= ~255ll;
				if((n_hi | 255) <= last) n_hi |= 255; else if(n_hi < last) n_hi = last;
				const unsigned long long *done = rp->done + (size_t)c * kRefCache;
				const uint32_t ndv = rp->done_n[c], nd = ndv < (uint32_t)kRefCache ? ndv : (uint32_t)kRefCache;
				bool hit = false;
				for(uint32_t i = (uint32_t)part; i < nd; i += 4u) {
					const unsigned long long e = __hip_atomic_load(done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					const int64_t lo = (int64_t)(e >> 32) << 8, hi = lo + ((int64_t)((e >> 16) & 0xffffull) << 8) + 255;
					hit = hit || (lo <= n_lo && n_hi <= hi && ref_entry_visible((uint32_t)(e & 0xffffull), launch));
				}
				// (the four lanes of a request have taken the same branches up to here: its partners are active whenever a lane is)
				// (no short cut: a lane that skipped the exchange would not be read by its partners)
				int h = hit ? 1 : 0;
				h |= __shfl_xor(h, 16);
				h |= __shfl_xor(h, 32);
				hit = h != 0;
				if(hit) { if(part == 0) atomicAdd(stats + 1, 1u); go = false; }
			}
			if(go) {
				const int64_t s_end = (int64_t)os * (n_hi + 1);
				s_beg = (int64_t)os * n_lo - warm;
				bool refuse = in_end < s_end;
				if(s_beg < 0) s_beg = 0;
				else {
					if(ps0[0] > s_beg) { fl |= 2u; s_beg = (ps0[0] + os - 1) / os * os; }
					// (a retry goes with what is held - it was not refused the first time)
					if(s_beg > 0 && (int64_t)os * n_lo - s_beg < rp->warm / 4) refuse = true;
				}
				if(refuse) { if(part == 0) atomicAdd(stats + 2, 1u); go = false; }
				else {
					s_beg -= s_beg % os;
					len = s_end - s_beg;
					if(rp->mix[c]) fl |= 1u;
				}
			}
			if(!go) len = 0;
		}
		if(part == 0) {
			sh.s_beg[r] = s_beg; sh.len[r] = len; sh.n_lo[r] = n_lo; sh.n_hi[r] = n_hi; sh.chan[r] = c; sh.kind[r] = kind; sh.flags[r] = fl;
			sh.dphi[r] = len ? rp->dphi[c] : 0u;
		}
	}
	__syncthreads();
	if(threadIdx.x < kScanLanes) {
		// the same stretch twice in one workgroup (the candidates of one preamble): once is enough
		const int r = threadIdx.x;
		bool dup = false;
		for(int q = 0; q < r; q++) dup = dup || (sh.len[q] && sh.chan[q] == sh.chan[r] && sh.n_lo[q] == sh.n_lo[r] && sh.n_hi[q] >= sh.n_hi[r]);
		if(dup && sh.len[r]) { atomicAdd(stats + 1, 1u); sh.len[r] = 0; }
	}
	__syncthreads();
	if(threadIdx.x == 0) {
		int64_t m = 0, lo = INT64_MAX;
		for(int r = 0; r < kScanLanes; r++) { m = sh.len[r] > m ? sh.len[r] : m; if(sh.len[r] > 0 && sh.s_beg[r] < lo) lo = sh.s_beg[r]; }
		sh.lmax = m; sh.smin = lo;
		for(int r = 0; r < kScanLanes; r++) if(sh.len[r] <= 0) sh.s_beg[r] = lo;      // (idle lanes: any position in range)
	}
	__syncthreads();
	const int64_t lmax = sh.lmax;
	if(lmax <= 0) return;
	__builtin_amdgcn_s_setprio(3);                               // (a few dozen wavefronts on the whole device, on the walk's critical path)
	const int64_t nblk = ((lmax + BLK - 1) / BLK + 2) / 4 * 4 + 1;   // blocks; 1 + a multiple of 4 (the producers' loop; a block too many computes on zeros and stores nothing)
	for(int i = threadIdx.x; i < 256; i += blockDim.x) sh.lut[i] = ((ref_gf4 *)rp->lut)[i];
	__syncthreads();

	// a barrier that waits for LDS traffic only (the producers' raw-sample loads for the block after next stay in flight across it -
	// __syncthreads() would wait for them)
	auto uni64 = [](int64_t v) -> int64_t { return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v)); };
	auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
	constexpr int NQ = kScanLanes / kScanProd;                     // requests per producer wavefront
	// producer number of this wavefront (-1: none)
	const int pidx = kScanWaves == 1 + kScanProd ? wave - 1 : (wave == 0 || (wave & 3) == 0) ? -1 : wave - 1 - (wave > 4) - (wave > 8);
	if(wave > 0 && pidx < 0) return;                              // (an idle wavefront: it has taken part in the set-up's barriers; s_barrier counts the wavefronts still there)
	if(wave > 0) {
#if defined(VDL2_SCAN_EXP) && VDL2_SCAN_EXP == 3
		__builtin_amdgcn_s_setprio(1);
#endif
		// ---- a producer: requests pidx, pidx + kScanProd, ...; what it needs of each in registers (uniform values: where the
		// request's sample 0 would lie if the stretch of raw input it is in went on for ever, how far that stretch does go) ----
		const uint8_t *q_ptr[NQ]; uint32_t q_pend[NQ], q_ph0[NQ], q_len[NQ], q_dphi[NQ];
		int64_t q_beg[NQ];
		float pre[NQ], pim[NQ];                                       // a request's mixed samples of the block before (lanes 62, 63 are read)
		// the raw samples of the next PF blocks are in flight at any time (block b's in wq[b % PF]): a block lasts the consumer ~0.6 us,
		// a load that misses the L2 takes longer - with one block of look-ahead the producers waited for memory, and the consumer for them
		constexpr int PF = 4;
		uint32_t wq[PF][NQ];
		constexpr int SB = FMT == 1 ? 4 : 2;                          // bytes per raw sample
		#pragma unroll
		for(int q = 0; q < NQ; q++) {
			const int r = pidx + q * kScanProd;
			q_beg[q] = uni64(sh.s_beg[r]);
			q_ph0[q] = (uint32_t)q_beg[q];
			q_len[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sh.len[r]);
			// a channel that is not mixed (src/demod.c:312: offset_tuning == 0) gets the phase step 0: table entry 0 with fraction 0 is
			// (sin, cos) = (0, 1) exactly, and multiplying by it leaves a sample exactly as it is (the conversions never produce a negative zero)
			q_dphi[q] = (__builtin_amdgcn_readfirstlane((int)sh.flags[r]) & 1) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.dphi[r]) : 0u;
			pre[q] = 0.f; pim[q] = 0.f; q_ptr[q] = nullptr; q_pend[q] = 0u;
		}
		// the stretch of raw input that holds sample p of request q (uniform; a handful of times per scan)
		auto locate = [&](int q, uint32_t p) {
			const int64_t s = q_beg[q] + p;
			q_ptr[q] = nullptr; q_pend[q] = p;                           // (not held: zeros, one block at a time)
			for(int j = 0; j < npiece; j++) {
				const int64_t a0 = rp->piece[j].s0, an = rp->piece[j].n;
				if(s >= a0 && s < a0 + an) {
					q_ptr[q] = (const uint8_t *)rp->piece[j].p - (a0 - q_beg[q]) * SB;
					const int64_t e = a0 + an - q_beg[q];
					q_pend[q] = e > 0xffffffffll ? 0xffffffffu : (uint32_t)e;
				}
			}
		};
		// the raw sample of request q and block b for this lane (0 past the request's end)
		auto fetch = [&](int q, uint32_t b) -> uint32_t {
			const uint32_t p0 = b * (uint32_t)BLK;
			if(p0 >= q_len[q]) return 0u;                                // (uniform)
			const uint32_t n = q_len[q] - p0 < (uint32_t)BLK ? q_len[q] - p0 : (uint32_t)BLK;
			if(p0 + n > q_pend[q]) {
				// the block reaches past the stretch: lane by lane (the scan's first block, and where two stretches meet)
				uint32_t w = 0u;
				for(uint32_t i = 0; i < n; i++) {
					if(p0 + i >= q_pend[q]) locate(q, p0 + i);
					if(q_ptr[q] && (uint32_t)lane == i) w = FMT == 1 ? *(ref_gu32 *)(q_ptr[q] + (size_t)(p0 + i) * SB) : (uint32_t)*(ref_gu16 *)(q_ptr[q] + (size_t)(p0 + i) * SB);
					if(!q_ptr[q]) q_pend[q] = p0 + i + 1;
				}
				return w;
			}
			uint32_t w = 0u;
			if((uint32_t)lane < n) w = FMT == 1 ? *(ref_gu32 *)(q_ptr[q] + (size_t)(p0 + (uint32_t)lane) * SB) : (uint32_t)*(ref_gu16 *)(q_ptr[q] + (size_t)(p0 + (uint32_t)lane) * SB);
			return w;
		};
		// block b of all its requests (the raw samples are in wq[SLOT]): the NCO table entries first, all of them, then the arithmetic;
		// the raw samples of block b + PF are asked for before either
		auto produce = [&](uint32_t b, auto SLOTC) {
			constexpr int SLOT = decltype(SLOTC)::value;
#if defined(VDL2_SCAN_EXP) && VDL2_SCAN_EXP == 1
			return;
#endif
			uint32_t wc[NQ]; v4f eq[NQ]; uint32_t phq[NQ];
			#pragma unroll
			for(int q = 0; q < NQ; q++) { wc[q] = wq[SLOT][q]; wq[SLOT][q] = fetch(q, b + PF); }
			#pragma unroll
			for(int q = 0; q < NQ; q++) {
				phq[q] = ((q_ph0[q] + b * (uint32_t)BLK + (uint32_t)lane) * q_dphi[q]) & 0xffffffu;
				eq[q] = sh.lut[phq[q] >> 16];
			}
			// (straight-line for all its requests, so that their dependent chains - conversion, mixer, lane shifts, taps - interleave:
			// a request that is through has zeros for raw samples and produces values nobody stores)
			float re[NQ], im[NQ];
			#pragma unroll
			for(int q = 0; q < NQ; q++) {
				const uint32_t w = wc[q];
				if(FMT == 1) { re[q] = (float)(int16_t)(w & 0xffff) / 32768.0f; im[q] = (float)(int16_t)(w >> 16) / 32768.0f; }
				else { re[q] = ((float)(w & 0xff) - 127.5f) / 127.5f; im[q] = ((float)((w >> 8) & 0xff) - 127.5f) / 127.5f; }
				const float F = (float)(phq[q] & 0xffffu);
				const float sn = eq[q].x + eq[q].z * F, cs = eq[q].y + eq[q].w * F;
				const float mr = re[q] * cs - im[q] * sn, mi = im[q] * cs + re[q] * sn;
				re[q] = mr; im[q] = mi;
			}
			#pragma unroll
			for(int q = 0; q < NQ; q++) {
				// in[1], in[2] (demod.c:75): the lane before's, lane 0 takes the block before's last ones (wave_shr:1 leaves lane 0 what it held)
				const float x1r = dpp_wave_shr1_keep(ref_lane(pre[q], BLK - 1), re[q]), x1i = dpp_wave_shr1_keep(ref_lane(pim[q], BLK - 1), im[q]);
				const float x2r = dpp_wave_shr1_keep(ref_lane(pre[q], BLK - 2), x1r), x2i = dpp_wave_shr1_keep(ref_lane(pim[q], BLK - 2), x1i);
				pre[q] = re[q]; pim[q] = im[q];
				float fa = A0 * re[q]; fa += A1 * x1r + A2 * x2r;
				float fb = A0 * im[q]; fb += A1 * x1i + A2 * x2i;
				sh.buf[b & 1][0][pidx + q * kScanProd][lane] = fa;
				sh.buf[b & 1][1][pidx + q * kScanProd][lane] = fb;
			}
		};
		#pragma unroll
		for(int k = 0; k < PF; k++) {
			#pragma unroll
			for(int q = 0; q < NQ; q++) wq[k][q] = fetch(q, (uint32_t)k);
		}
		produce(0u, std::integral_constant<int, 0>());
		lds_barrier();
		// blocks 1 .. nblk - 1, each followed by a barrier, and one barrier more (the consumer's last block).  nblk - 1 is a multiple of
		// PF (see nblk): the loop body is PF blocks of straight-line code, block b1 + k in slot (1 + k) % PF, and the slots stay in the
		// registers they were loaded into (a conditional tail made the compiler shuffle them with v_mov, each of which waits for its load)
		static_assert(PF == 4, "the loop below is written out for four slots");
		for(uint32_t b1 = 1; b1 < (uint32_t)nblk; b1 += PF) {
			produce(b1, std::integral_constant<int, 1>()); lds_barrier();
			produce(b1 + 1, std::integral_constant<int, 2>()); lds_barrier();
			produce(b1 + 2, std::integral_constant<int, 3>()); lds_barrier();
			produce(b1 + 3, std::integral_constant<int, 0>()); lds_barrier();
		}
		lds_barrier();
		return;
	}
	// ---- the consumer: lane = (component, request) ----
	lds_barrier();
	// Lanes 32-63 are the WITNESS: the same requests over the same feed-forward values from ANOTHER state.  Two fp32 trajectories of
	// this recursion become bit-identical after a while (1.6e4 samples on average) and stay so; that the run-up was long enough for the
	// zero-start trajectory to have become the reference's is what "the reference's own samples" rests on - a trajectory that has NOT
	// met its witness by the stretch's first output has certainly not forgotten where it started.  The converse does not hold: the two
	// start at the same sample from states close to each other and often meet each other before either meets the reference's (of 174
	// stretches that differed from the oracle's after 2^17 samples the witness had flagged 37, and it flagged 52 that did not differ:
	// profiles/r06_scan_soundness.txt) - it is a monitor, the run-up's length is the guarantee.  Costs nothing: the lanes were idle.  Not met: counted (stats[9] -> vdl2hip_stats.referee_unmet); the stretch is then within the rounding noise of the
	// reference's, like the channeliser's own, not bit for bit it.  (A scan from the stream's very start begins in the reference's
	// state, zero, exactly: its witness starts there too.)  Round 6: such a scan is listed and run again from four times further back
	// (retry_q), where the history ring reaches that far.
	const int r = lane & (kScanLanes - 1), comp = (lane / kScanLanes) & 1;
	const bool witness = lane >= 2 * kScanLanes;
	const float b1 = rp->B1, b2 = rp->B2;
	float y1 = (witness && sh.s_beg[r] > 0) ? 0.25f : 0.f, t2 = b2 * ((witness && sh.s_beg[r] > 0) ? 0.125f : 0.f);   // out[1] of demod.c:289-298 (this lane's component) and B2 * out[2], the product a step early
	float w_y1 = 0.f, w_t2 = 0.f;                                   // the state when the stretch's first output is due
	const int64_t my_lo = sh.n_lo[r], my_hi = (sh.len[r] > 0 && !witness) ? sh.n_hi[r] : -1;
	int64_t k_out = sh.s_beg[r] / os;                                // the decimated sample this lane's next output is
	__attribute__((address_space(1))) float *yout = (__attribute__((address_space(1))) float *)(rp->y + (size_t)sh.chan[r] * cap_y) + comp;
	int cnt = 0;                                                     // input samples since the last output (uniform: every request starts on a decimation boundary)
	if constexpr(OS != 0) {
		// whole decimation periods per block: chunks of 20 steps, the outputs at fixed places
		constexpr int CH = 20, NV = CH / 4;
		static_assert(BLK % CH == 0 && CH % OS == 0, "chunks of whole decimation periods");
		(void)cnt;
		for(int64_t b = 0; b < nblk; b++) {
			const float *row = &sh.buf[b & 1][comp][r][0];
			v4f cur[NV], nxt[NV];
			#pragma unroll
			for(int j = 0; j < NV; j++) cur[j] = *reinterpret_cast<const v4f *>(row + 4 * j);
			#pragma unroll
			for(int j0 = 0; j0 < BLK; j0 += CH) {
				if(j0 + CH < BLK) {
					#pragma unroll
					for(int j = 0; j < NV; j++) nxt[j] = *reinterpret_cast<const v4f *>(row + j0 + CH + 4 * j);
				}
				#pragma unroll
				for(int j = 0; j < CH; j++) {
#if defined(VDL2_SCAN_EXP) && VDL2_SCAN_EXP == 2
					if(j) continue;
#endif
					const float m = b1 * y1;
					const float sm = m + t2;
					t2 = b2 * y1;
					y1 = cur[j >> 2][j & 3] + sm;
					if((j + 1) % OS == 0) {
						if(k_out >= my_lo && k_out <= my_hi) yout[2 * ((uint32_t)k_out & mask)] = y1;
						if(k_out == my_lo) { w_y1 = y1; w_t2 = t2; }
						k_out++;
					}
				}
				#pragma unroll
				for(int j = 0; j < NV; j++) cur[j] = nxt[j];
			}
			lds_barrier();
		}
	} else {
	constexpr int kChunk = 16;
	for(int64_t b = 0; b < nblk; b++) {
		const float *row = &sh.buf[b & 1][comp][r][0];
		v4f cur[kChunk / 4], nxt[kChunk / 4];
		#pragma unroll
		for(int j = 0; j < kChunk / 4; j++) cur[j] = *reinterpret_cast<const v4f *>(row + 4 * j);
		#pragma unroll
		for(int j0 = 0; j0 < BLK; j0 += kChunk) {
			// the next chunk's values are asked for before this chunk's steps: the LDS latency hides behind the recursion
			if(j0 + kChunk < BLK) {
				#pragma unroll
				for(int j = 0; j < kChunk / 4; j++) nxt[j] = *reinterpret_cast<const v4f *>(row + j0 + kChunk + 4 * j);
			}
			#pragma unroll
			for(int j = 0; j < kChunk; j++) {
				// r = r0 + (B1 * out1 + B2 * out2): three roundings in the reference's order (demod.c:77)
				const float m = b1 * y1;
				const float sm = m + t2;
				t2 = b2 * y1;
				y1 = cur[j >> 2][j & 3] + sm;
				if(__builtin_expect(++cnt == os, 0)) {
					cnt = 0;
					if(k_out >= my_lo && k_out <= my_hi) yout[2 * ((uint32_t)k_out & mask)] = y1;
					if(k_out == my_lo) { w_y1 = y1; w_t2 = t2; }
					k_out++;
				}
			}
			#pragma unroll
			for(int j = 0; j < kChunk / 4; j++) cur[j] = nxt[j];
		}
		lds_barrier();
	}
	}
	// main against witness (lane ^ 32), then I and Q together (lane ^ 16)
	bool unmet = __float_as_uint(__shfl_xor(w_y1, 32)) != __float_as_uint(w_y1) || __float_as_uint(__shfl_xor(w_t2, 32)) != __float_as_uint(w_t2);
	unmet = unmet || __shfl_xor((int)unmet, kScanLanes) != 0;
	if(lane < kScanLanes && sh.len[r] > 0) {
		const int c = sh.chan[r];
		const int64_t n_lo = sh.n_lo[r], n_hi = sh.n_hi[r];
		bool listed = false;
		if(unmet && retry_q) {
			const uint32_t k = atomicAdd(retry_n, 1u);
			if(k < retry_cap) { retry_q[k] = ScanReq{ c, sh.kind[r], n_lo, n_hi }; listed = true; atomicAdd(stats + 10, 1u); }
		}
		if(unmet && !listed) atomicAdd(stats + 9, 1u);
		const uint32_t i = listed ? 0u : atomicAdd(rp->done_n + c, 1u);
		int64_t len = (n_hi - n_lo) >> 8; if(len > 0xffff) len = 0xffff;
		if(((n_hi + 1) & 255) != 0) len -= 1;
		if(listed) len = -1;                                          // (not published: the retry does that)
		if(len >= 0) __hip_atomic_store(rp->done + (size_t)c * kRefCache + (i % (uint32_t)kRefCache), ((unsigned long long)(n_lo >> 8) << 32) | ((unsigned long long)len << 16) | (unsigned long long)(launch & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		atomicAdd(stats + 0, 1u); atomicAdd(stats + 4 + sh.kind[r], 1u);
		if(sh.flags[r] & 2u) atomicAdd(stats + 3, 1u);
	}
#else
	(void)rp; (void)launch; (void)sq; (void)rq; (void)n_ptr; (void)cap; (void)k_end;
#endif
}

// Two bursts per workgroup, one per wavefront (they share nothing; four would need more LDS than a channeliser workgroup
// leaves on a CU): see k_nf_replay.
#ifndef VDL2_K5_WAVES
#define VDL2_K5_WAVES 2
#endif
constexpr int kBurstWaves = VDL2_K5_WAVES;
constexpr int kK5MaxChan = 1024;          // most channels a receiver may have (vdl2hip_create): a wavefront of the burst decoder keeps all their burst-list offsets in LDS
// (the LDS is dynamic so that the compiler does not see its size: it would size the register budget by the LDS-limited occupancy
// and take 169, more than a channeliser wave leaves)
// (the body of k_burst: `block` of `nblocks` workgroups of kBurstWaves wavefronts)
__device__ __forceinline__ void burst_body(const K5Args &a, unsigned char *k5_lds, uint32_t block, uint32_t nblocks) {
	if((threadIdx.x >> 6) >= kBurstWaves) return;
	BurstShared &sh = reinterpret_cast<BurstShared *>(k5_lds)[threadIdx.x >> 6];
	uint32_t *bb = reinterpret_cast<uint32_t *>(k5_lds + sizeof(BurstShared) * kBurstWaves) + (size_t)(threadIdx.x >> 6) * (kK5MaxChan + 1);
	const int lane = threadIdx.x & 63;
	const uint32_t wave_id = block * kBurstWaves + (threadIdx.x >> 6);
	// Every wavefront turns the per-channel burst counts into offsets for itself, in LDS (one load per lane and 64 channels, one scan):
	// finding a burst's channel is then a search in LDS, not eight dependent trips to memory per burst - and no kernel of its own has to
	// run between the walker and this one.
	uint32_t total = 0;
	for(int c0 = 0; c0 < a.nchan; c0 += 64) {
		const uint32_t v = c0 + lane < a.nchan ? a.nb_chan[c0 + lane] : 0u;
		uint32_t inc = v;
		#pragma unroll
		for(int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if(lane >= d) inc += o; }
		if(c0 + lane < a.nchan) bb[c0 + lane] = total + inc - v;
		total += __shfl(inc, 63);
	}
	if(lane == 0) { bb[a.nchan] = total; if(wave_id == 0 && a.df.pass != 2) a.ctl->nbursts = total; }
	WAVE_SYNC();
	if(a.df.pass == 2) {
		// the bursts the first pass has listed; its wavefronts own the initial shares of the output, this pass's go to the counters
		const uint32_t nd = *a.df.dq_n < a.df.dq_cap ? *a.df.dq_n : a.df.dq_cap;
		if(wave_id >= nd) return;
		burst_shared_init(*a.tab, wave_id, a.ctl, sh, true);
		for(uint32_t i = wave_id; i < nd; i += nblocks * kBurstWaves) {
			const uint32_t g = a.df.dq[i];
			int lo = 0, hi = a.nchan;
			while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(bb[mid] <= g) lo = mid; else hi = mid; }
			const int c = lo;
			const Burst b = a.bursts[(size_t)c * a.cap_bursts_chan + (g - bb[c])];
			ChanView v{ a.y + (size_t)c * a.cap, nullptr, nullptr, a.mask, a.ref, c, a.ref_launch };
			decode_burst(b, a.freq[c], *a.tab, v, a.cnt + (size_t)c * kNumCounters, a.frames, a.pool, a.ctl, sh, &a.df, g);
			WAVE_SYNC();
		}
		burst_reserve_done(a.frames, sh);
		return;
	}
	if(wave_id >= total) {
		// nothing to decode: this wavefront's share of the output (vdl2_core.h: burst_reserve_*) stays empty
		const uint32_t slot = wave_id * (uint32_t)kResSlots + (uint32_t)lane;
		if(lane < kResSlots && slot < a.ctl->cap_frames) { OutFrame &f = a.frames[slot]; f.chan = -1; f.len = 0; f.pool_off = 0; f.nf_upd = 0; }
		return;
	}
	burst_shared_init(*a.tab, wave_id, a.ctl, sh);
	for(uint32_t g = wave_id; g < total; g += nblocks * kBurstWaves) {
		int lo = 0, hi = a.nchan;                       // channel c with bb[c] <= g < bb[c+1]
		while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(bb[mid] <= g) lo = mid; else hi = mid; }
		const int c = lo;
		const Burst b = a.bursts[(size_t)c * a.cap_bursts_chan + (g - bb[c])];
		ChanView v{ a.y + (size_t)c * a.cap, nullptr, nullptr, a.mask, a.ref, c, a.ref_launch };
		decode_burst(b, a.freq[c], *a.tab, v, a.cnt + (size_t)c * kNumCounters, a.frames, a.pool, a.ctl, sh, a.df.pass ? &a.df : nullptr, g);
		WAVE_SYNC();
	}
	burst_reserve_done(a.frames, sh);
}

__global__ __launch_bounds__(256, 4) void k_burst(K5Args a) {
	extern __shared__ __align__(16) unsigned char k5_lds[];       // BurstShared[kBurstWaves], then the per-channel burst offsets
	burst_body(a, k5_lds, blockIdx.x, gridDim.x);
}

// Short feeds: the noise floor (needs the walker's evaluation log) and the burst decoder (needs its burst lists) do not need each other,
// so they run as ONE launch of two kinds of workgroups - the first nf_blocks are k_nf_all's, the rest k_burst's - instead of one kernel
// after the other on the short feed's single stream.  (Block size 64 * kNfWaves; the burst decoder's workgroups use their first
// kBurstWaves wavefronts.)
static_assert(kNfWaves >= kBurstWaves, "a workgroup of the joint launch holds either kind");
__global__ __launch_bounds__(64 * kNfWaves, 4) void k_nf_burst(K4bArgs nf, K5Args k5, uint32_t nf_blocks) {
	extern __shared__ __align__(16) unsigned char back_lds[];
	if(blockIdx.x < nf_blocks) nf_all_body(nf, back_lds, (int)blockIdx.x);
	else burst_body(k5, back_lds, blockIdx.x - nf_blocks, gridDim.x - nf_blocks);
}

// After K4b and K5 have both finished: the noise-floor figure and the AVLC front-door checks of every frame (one wavefront per
// frame), and the feed's output made compact.  The burst decoder's wavefronts write into shares of the record array and the octet
// pool that they own (vdl2_core.h: burst_reserve_*), which leaves tombstone records and unused octets behind; a wavefront here takes
// 64 records at a time, counts the real ones, reserves their places in the delivered arrays with ONE pair of atomics, and writes each
// finished frame - record and octets - there.  The host copies exactly what is delivered.
constexpr int kFrameWaves = 4, kFrameChunk = 16;
// what one small copy brings the host together with the control block (most blocks of a live receiver hold a handful of frames)
constexpr int kMailFrames = 8, kMailPool = 2048;
struct OutMail { OutCtl ctl; OutFrame frames[kMailFrames]; uint8_t pool[kMailPool]; };
__global__ __launch_bounds__(64 * kFrameWaves) void k_frame_finish(OutFrame *frames, const uint8_t *pool, OutMail *mail, const Tables *tab,
		unsigned long long *acnt, const float *ring, uint32_t ring_mask, OutFrame *frames_out, uint8_t *pool_out) {
	__shared__ FrameShared shw[kFrameWaves];
	__shared__ uint32_t s_off[kFrameWaves][64];
	OutCtl *ctl = &mail->ctl;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t n = ctl->nframes < ctl->cap_frames ? ctl->nframes : ctl->cap_frames;
	const uint32_t first = (blockIdx.x * kFrameWaves + wave) * (uint32_t)kFrameChunk;
	if(first >= n) return;
	FrameShared &sh = shw[wave];
	bool init = false;
	for(uint32_t c0 = first; c0 < n; c0 += gridDim.x * kFrameWaves * (uint32_t)kFrameChunk) {
		const uint32_t i = c0 + (uint32_t)lane;
		const bool valid = lane < kFrameChunk && i < n && frames[i].chan >= 0;     // not a tombstone
		const uint32_t padded = valid ? (frames[i].len + 3u) & ~3u : 0u;
		const unsigned long long vm = __ballot(valid);
		if(!vm) continue;
		uint32_t inc = padded;                                       // inclusive scan of the octet space over the chunk
		#pragma unroll
		for(int d = 1; d < kFrameChunk; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if(lane >= d) inc += o; }
		s_off[wave][lane] = inc - padded;
		const uint32_t tot = __shfl(inc, kFrameChunk - 1), cnt = (uint32_t)__popcll(vm);
		uint32_t slot0 = 0, off0 = 0;
		if(lane == 0) { slot0 = atomicAdd(&ctl->nvalid, cnt); off0 = atomicAdd(&ctl->pool_out_used, tot); }
		if(!init) { frame_shared_init(*tab, sh); init = true; }      // (while the atomics are under way)
		slot0 = __shfl(slot0, 0); off0 = __shfl(off0, 0);
		WAVE_SYNC();
		uint32_t rank = 0;
		for(unsigned long long m = vm; m; m &= m - 1, rank++) {
			const int l = __builtin_ctzll(m);
			OutFrame &f = frames[c0 + (uint32_t)l];
			const int c = f.chan;
			finish_frame(f, pool, *tab, acnt + (size_t)c * kNumAvlcCounters, ring + (size_t)c * (ring_mask + 1), ring_mask, sh);
			// deliver: the record with its octets' new place, the octets themselves (finish_frame() has left them in sh.buf)
			const uint32_t len = f.len, src_off = f.pool_off, dst_off = off0 + s_off[wave][l], slot = slot0 + rank;
			const bool in_mail = slot < (uint32_t)kMailFrames && dst_off + len <= (uint32_t)kMailPool;
			if(lane == 0) { OutFrame g = f; g.pool_off = dst_off; frames_out[slot] = g; if(slot < (uint32_t)kMailFrames) mail->frames[slot] = g; }
			for(uint32_t k = (uint32_t)lane; k < len; k += 64) {
				const uint8_t o = k < (uint32_t)sizeof sh.buf ? sh.buf[k] : pool[src_off + k];
				pool_out[dst_off + k] = o;
				if(in_mail) mail->pool[dst_off + k] = o;
			}
			WAVE_SYNC();
		}
	}
}

}  // namespace vdl2
