// vdl2_core.h - the burst-level logic of the hot path, written once in a
// "wave-phase" style so that the very same source is (a) the body of the HIP
// kernels (one 64-lane wavefront per channel / per burst) and (b) compilable
// with plain g++ for the CPU unit tests (tests/hostsim), where a phase is an
// ordinary loop over 64 lanes.  There is no CPU product path: hostsim is a
// test of this file, not something the library can fall back to.
//
// Style rules:
//   * control flow between phases is wave-uniform (decided from shared scalars);
//   * WAVE_FOR(l) ... WAVE_END runs its body once per lane; lanes talk through
//     shared arrays (LDS) only;
//   * LANE0 ... LANE0_END runs a sequential section on one lane and publishes
//     its results through shared memory.
//
// Reference functions restated here (reference v2.6.0):
//   got_sync()/calc_para_vertex()      src/demod.c:98-198
//   demod() DM_INIT / DM_SYNC          src/demod.c:222-286
//   decode_vdl2_burst()/decode_frame() src/decode.c:173-384
//   decode_header()/get_fec_octetcount()/deinterleave()  src/decode.c:102-163
//   bitstream_descramble()/bitstream_copy_next_frame()   src/bitstream.c:94-150
//   rs_verify() + decode_rs_char()     src/rs.c:32-49, src/libfec/decode_rs.h:71-298
#pragma once
#include <cstdint>
#include <cmath>

#if defined(__HIP_DEVICE_COMPILE__)
#define VDL2_DEVICE_PASS 1
#else
#define VDL2_DEVICE_PASS 0
#endif

#if defined(__HIPCC__)
#define VDL2_HD __host__ __device__ inline
#else
#define VDL2_HD inline
#endif

#if VDL2_DEVICE_PASS
#define VDL2_LANE() ((int)(threadIdx.x & 63))
// Between two phases the lanes of ONE wavefront exchange data through LDS.  A wavefront's LDS instructions are executed in the order
// they are issued, so all that is needed is that the compiler keeps that order: a wavefront-scope fence.  (A workgroup-scope fence
// also waits for every outstanding GLOBAL access of the wave - vmcnt(0) - i.e. each phase that had fired a counter atomic or a store
// and forgotten it paid a trip to memory at its end: half of the burst decoder's time.)  WAVE_SYNC_GLOBAL() is that stronger form, for
// the one place where a lane reads from global memory what another lane of its wavefront has written in the same kernel.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while(0)
#define WAVE_SYNC_GLOBAL() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while(0)
#define WAVE_FOR(l) { const int l = VDL2_LANE();
#define WAVE_END } WAVE_SYNC();
#define LANE0 if(VDL2_LANE() == 0) {
#define LANE0_END } WAVE_SYNC();
#else
#define WAVE_SYNC() do {} while(0)
#define WAVE_SYNC_GLOBAL() do {} while(0)
// Host build: the lanes of a phase run one after the other.  A phase must not depend on that order (on the device the
// lanes run together): -DVDL2_HOST_REVERSE_LANES runs them backwards, and the CPU tests must give the same answers.
#ifdef VDL2_HOST_REVERSE_LANES
#define WAVE_FOR(l) for(int l##_fw_ = 0; l##_fw_ < 64; l##_fw_++) { const int l = 63 - l##_fw_;
#else
#define WAVE_FOR(l) for(int l = 0; l < 64; l++) {
#endif
#define WAVE_END }
#define LANE0 {
#define LANE0_END }
#endif

#if VDL2_DEVICE_PASS
#define VDL2_CNT_ADD(arr, which, val) atomicAdd(&(arr)[which], (unsigned long long)(val))
#else
#define VDL2_CNT_ADD(arr, which, val) ((arr)[which] += (unsigned long long)(val))
#endif

// optional per-phase cycle probes of the walker and the burst decoder (development aid, -DVDL2_K5_PROF; compiled out by default).
// A wavefront keeps its sums in wave-uniform registers and adds them to one of 64 global slots once, when it is done: no atomics
// inside what is being measured (round 3's probe did two atomics on one address per mark - thousands of waves queueing on them
// were most of what it then reported).
#if defined(__HIPCC__) && defined(VDL2_K5_PROF)
__device__ unsigned long long vdl2_k5_prof[64][16];
__device__ unsigned long long vdl2_k4_prof[64][24];
__device__ unsigned long long vdl2_nf_prof[64][16];
#endif
#if VDL2_DEVICE_PASS && defined(VDL2_K5_PROF)
#define K4_BEGIN() unsigned long long k4_t0_ = __builtin_readcyclecounter(); unsigned k4_acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, k4_n_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define K4_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); k4_acc_[k] += (unsigned)(t_ - k4_t0_); k4_n_[k]++; k4_t0_ = t_; } while(0)
#define K4_END() do { if(VDL2_LANE() == 0) { for(int k_ = 0; k_ < 10; k_++) { atomicAdd(&vdl2_k4_prof[blockIdx.x & 63][k_], (unsigned long long)k4_acc_[k_]); \
	atomicAdd(&vdl2_k4_prof[blockIdx.x & 63][12 + k_], (unsigned long long)k4_n_[k_]); } } } while(0)
#define K5_BEGIN() unsigned long long k5_t0_ = __builtin_readcyclecounter(); unsigned k5_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define K5_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); k5_acc_[k] += (unsigned)(t_ - k5_t0_); k5_t0_ = t_; } while(0)
#define K5_END() do { if(VDL2_LANE() == 0) { for(int k_ = 0; k_ < 8; k_++) atomicAdd(&vdl2_k5_prof[blockIdx.x & 63][k_], (unsigned long long)k5_acc_[k_]); \
	atomicAdd(&vdl2_k5_prof[blockIdx.x & 63][8], 1ull); } } while(0)
#define NF_BEGIN() unsigned long long nf_t0_ = __builtin_readcyclecounter(); unsigned nf_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define NF_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); nf_acc_[k] += (unsigned)(t_ - nf_t0_); nf_t0_ = t_; } while(0)
#define NF_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define NF_END() do { if(VDL2_LANE() == 0) { for(int k_ = 0; k_ < 8; k_++) atomicAdd(&vdl2_nf_prof[blockIdx.x & 63][k_], (unsigned long long)nf_acc_[k_]); \
	atomicAdd(&vdl2_nf_prof[blockIdx.x & 63][8], 1ull); } } while(0)
#else
#define K4_BEGIN() do {} while(0)
#define K4_MARK(k) do {} while(0)
#define K4_END() do {} while(0)
#define K5_MARK(k) do {} while(0)
#define K5_BEGIN() do {} while(0)
#define K5_END() do {} while(0)
#define NF_BEGIN() do {} while(0)
#define NF_MARK(k) do {} while(0)
#define NF_WAIT_LOADS() do {} while(0)
#define NF_END() do {} while(0)
#endif

namespace vdl2 {

// ---- constants (dumpvdl2.h:37-50, demod.c:37-48, decode.c:45-50) ----
constexpr int kRsK = 249, kRsN = 255, kRsPar = 6;
constexpr int kHdrBits = 25, kTlBits = 17, kHdrParBits = 5;
constexpr int kPreamble = 16, kSpsDec = 10, kSyncSkip = 3;
constexpr float kPherrBig = 1000.f, kSyncThr = 4.f;
constexpr float kPiBelow = 0x1.921fb4p+1f; // largest float < M_PI: (double)x > M_PI  <=>  x > kPiBelow for float x
constexpr uint32_t kMaxTl = 0x3FFFu, kMaxTlCorr = 0x1FFFu;
constexpr uint32_t kLfsrIv = 0x6959u;
constexpr int kMaxSyms = 5632;            // >= ceil((8*(2048+52)+25)/3) = 5609
constexpr int kPrbsBits = kMaxSyms * 3;
constexpr int kMaxOctets = 2112;          // 2048 data + 52 FEC, rounded up
constexpr int kMaxBlocks = 9;
constexpr int kMaxWords = 512;            // 16 384 bits of corrected data
constexpr int kMaxTerm = 2048;            // a flag needs 8 bits
constexpr int kFreshAfter = 156;          // evaluations at n >= a+156: n, n-3, n-6 touch only the current interval
constexpr int kNumIv = 32;                // DM_INIT interval history (160/6 < 32 intervals can matter)
constexpr int kNfTail = 64;               // evaluation chunks remembered across feeds for the noise-floor lookback
constexpr int kLpTerms = 256;             // 0.9^256 ~ 2e-12: below fp32 resolution of mag_lp
constexpr int kNumCounters = 20;
constexpr int kNumAvlcCounters = 10;
constexpr int kMinAvlcLen = 11;           // avlc.c:39
constexpr uint32_t kGoodFcs = 0xF0B8u;    // avlc.c:40
// segmented walk (see "Speculative segments" below)
constexpr int kCleanAfter = 320;          // search state this far into an interval no longer depends on anything before the interval
constexpr int kSpecBack = 4096;           // a speculative walker pretends its DM_INIT interval started this far before its segment
constexpr int kSpecBursts = 48;           // per speculative segment; more than that and the segment is simply walked for real
constexpr int kSpecLog = 128;
constexpr int kSpecReq = 8;              // decisions within the margin a speculative segment may note (more: it gives up)
constexpr int kMaxSeg = 32;
constexpr int kCandWin = 512;             // candidate-bitmap words (64 samples each) a walker loads per pass and keeps in LDS

enum { CNT_SYNC_GOOD = 0, CNT_CRC_GOOD, CNT_CRC_BAD, CNT_ERR_NO_HEADER, CNT_ERR_TOO_LONG, CNT_ERR_NO_FEC,
       CNT_ERR_DATA_TRUNCATED, CNT_ERR_FEC_TRUNCATED, CNT_ERR_DEINTERLEAVE_DATA, CNT_ERR_DEINTERLEAVE_FEC,
       CNT_ERR_FEC_BAD, CNT_ERR_BITSTREAM, CNT_ERR_TRUNCATED_OCTETS, CNT_ERR_UNSTUFF, CNT_BLOCKS_PROCESSED,
       CNT_BLOCKS_FEC_OK, CNT_MSG_GOOD, CNT_MSG_GOOD_LOUD, CNT_PPM_REJECT, CNT_SLICER_NEG_IDX };
// the per-channel counters of the AVLC front door (decode.c:466, avlc.c:170-233)
enum { ACNT_PROCESSED = 0, ACNT_TOO_SHORT, ACNT_GOOD, ACNT_BAD_FCS, ACNT_AIR2GND, ACNT_AIR2AIR, ACNT_AIR2ALL, ACNT_GND2AIR, ACNT_GND2GND, ACNT_GND2ALL };
enum { AVLC_OK = 0, AVLC_TOO_SHORT = 1, AVLC_BAD_FCS = 2 };

// Read-only tables, built on the host once per context (tables.h) and kept in device memory.
struct Tables {
	float    pr_phase[kPreamble];      // demod.c:107-124
	float    lrx[kPreamble];           // demod.c:84-96
	float    lr_den;
	float    pad_[3];
	uint8_t  gray[8];                  // demod.c:223
	uint32_t hdr_H[kHdrParBits];       // decode.c:55-61
	uint32_t hdr_fix[32];              // decode.c:63-96
	uint32_t hdr_weight[32];           // decode.c:98-100
	uint8_t  gf_exp[512];              // alpha^i, doubled: exp[a+b] valid for a,b <= 254
	uint8_t  gf_log[256];              // log(0) = 255 marker
	uint8_t  prbs[kPrbsBits];          // bitstream.c:94-107 from LFSR_IV, one bit per byte
	uint16_t crc16[256];               // crc.c:23-57: reflected CRC-16-CCITT (x^16+x^12+x^5+1), one step per octet
	uint8_t  prbs_oct[kMaxOctets];     // the eight PRBS bits that cover data/FEC octet i (stream bits 25+8i ..), packed LSB first like the octet itself
};

struct cf32 { float re, im; };
VDL2_HD float phase_of(cf32 y);

// The referee (see "Referee" below): what a kernel needs to turn a stretch of one channel's decimated stream into the reference's
// own samples, bit for bit.  Defined by the build: kernels.h for the device (raw input + the sequential scan of demod.c:302-329),
// tests/hostsim for the CPU tests (a trace of exact samples).  nullptr: the samples are taken as they are.
struct RefChan;

// One channel's decimated-rate streams, ring-addressed by absolute sample index.  There is no stored phase stream: the
// reference's atan2 (double, narrowed to float: demod.c:232,256) is evaluated where a decision reads a phase - the exact
// tier of the sync kernel, the walker, the burst decoder - which is a few percent of the samples; the sync kernel's
// screening tier works on a cheap single-precision phase of its own (phase_fast()).
// the noted decisions of a channel and feed that did not stand: 4 n + kind each; n > kRefBad: more than the list holds, or some went unnoted
constexpr int kRefBad = 6;
struct RefBad { uint32_t n, pad_; int64_t at[kRefBad]; };
struct ChanView {
	const cf32     *y;                 // filtered + decimated samples (lp_re, lp_im)
	const cf32     *pf;                // {pherr[0], freq_err} of got_sync() evaluated at n (contiguous ring) - valid where a preamble is near
	const uint64_t *cand;              // bit n: pf[n-3].p < 4 && pf[n].p > pf[n-3].p
	uint32_t        mask;              // capacity - 1 (capacity is a power of two)
	RefChan        *ref = nullptr;     // referee hook of this channel (nullptr: none)
	int32_t         ref_chan = 0;      // ... and the channel's index there (device build: one hook per feed, shared by the channels)
	uint32_t        ref_launch = 0;    // ... and which kernel launch this is (device build: see ref_exact_window_dev)
	// optimistic mode (rq != nullptr): a decision within the margin is TAKEN on the samples as they are and noted here, to be checked on
	// the reference's own afterwards by all the wavefronts it takes at once (ref_verify); a channel one of whose decisions does not
	// stand is walked again from the feed's start.  rq_flag: per channel, "walk again" (nullptr: a speculative walk - it gives up instead)
	struct RefReq  *rq = nullptr; uint32_t *rq_n = nullptr; uint32_t rq_cap = 0; uint32_t *rq_flag = nullptr;
	struct RefBad  *rq_bad = nullptr;  // this channel's noted decisions that did NOT stand (written by the check, read when the channel is stitched again)
	// the stretches around the marked candidates have been made the reference's own BEFORE the walk (the exact sync tier lists them,
	// k_ref_scan_multi runs between the front and the walk): a marked candidate is then decided on the spot - by speculative walks too,
	// which only look whether the stretch is done (ref_window_done) - and the walk's chain from feed to feed holds no scan for it
	bool            ref_pre = false;
	VDL2_HD cf32  Y(int64_t n) const { return y[(uint32_t)n & mask]; }
	VDL2_HD float Phi(int64_t n) const { return n < 0 ? 0.f : phase_of(y[(uint32_t)n & mask]); }   // atan2(lp_im, lp_re); 0 before the stream starts
	VDL2_HD cf32  PF(int64_t n) const { return pf[(uint32_t)n & mask]; }
	VDL2_HD uint64_t Cand(int64_t word) const { return cand[(uint32_t)word & (mask >> 6)]; }
};

// A burst whose header decoded; consumed by decode_burst().
struct Burst {
	int32_t  chan, nsym;
	int64_t  t_first;                  // sample of the first symbol after the unique word
	int64_t  sync_sample, end_sample, ord;
	float    prev_phi0, vdphi, ppm;
	float    vdphi_err;                // referee: bound on |vdphi - the reference's| (0: taken on the reference's own samples; < 0: minus the bound, and the slope cannot be worked out again - its taps reach through the interval history)
	int64_t  prev_n;                   // sample whose phase is prev_phi0 (-1: before the stream, phase 0; -2: unknown)
	uint32_t tl_bits, syndrome;        // syndrome: bits 0-4 the header syndrome, bits 8.. the weight of the pattern it corrects (metadata->synd_weight)
	int64_t  nf_upd;                   // number of mag_nf updates that preceded the sync (v->mag_nf at decode_frame())
	int64_t  sync_evals;               // got_sync() evaluations executed up to and including the one that fired (nf_upd = sync_evals / 1000)
};

struct OutFrame {
	int32_t  chan, idx;
	uint32_t len, pool_off;
	uint32_t synd_weight, datalen_octets;
	int32_t  num_fec_corrections;
	float    frame_pwr_dbfs, nf_pwr_dbfs, ppm_error;
	int64_t  burst_ord, sync_sample, end_sample;
	int64_t  nf_upd;                   // Burst::nf_upd: which entry of the noise-floor ring stamp_noise_floor() turns into nf_pwr_dbfs
	uint32_t avlc_status, dst_addr, src_addr, pad_;   // finish_frame(): avlc_parse()'s first checks
};

struct OutCtl {
	uint32_t nbursts, nframes, pool_used, overflow;
	uint32_t cap_bursts, cap_frames, cap_pool, cap_log;
	uint32_t nvalid, pool_out_used;    // what k_frame_finish delivers: records without the tombstones, octets without the holes (the host copies these)
	uint32_t pad_[2];
};

// A stretch of executed got_sync() evaluations: samples first, first+3, ..., first+3*(count-1).
struct EvalChunk { int64_t first; int64_t count; };

// Per-channel evaluation log of one feed (written by the walker, consumed by the noise-floor kernel)
struct EvalLog { EvalChunk *chunks; uint32_t *n; };

// Persistent per-channel noise-floor state (v->mag_nf, v->nfcnt and enough history to replay v->mag_lp)
struct NfState {
	float   mag_nf;                    // after `updates` updates
	int32_t ntail;
	int64_t evals;                     // evaluations accounted so far
	int64_t tail_ord;                  // ordinal (0-based) of the first evaluation of tail[0]
	EvalChunk tail[kNfTail];           // most recent chunks, oldest first, covering >= kLpTerms evaluations when available
};

VDL2_HD void nf_state_init(NfState &s) { s.mag_nf = 2.0f; s.ntail = 0; s.evals = 0; s.tail_ord = 0; }

// Persistent per-channel FSM state of the walker (what vdl2_channel_t carries between samples).
struct WalkState {
	int64_t a;                         // first sample of the current DM_INIT interval
	int64_t e;                         // next got_sync() evaluation
	int64_t e0;                        // first evaluation of the current grid run
	int64_t bursts;                    // syncs accepted so far
	float   pherr1, pherr2, prev_dphi; // v->pherr[1], v->pherr[2], v->prev_dphi
	int64_t evals;                     // got_sync() evaluations executed so far (v->nfcnt = evals % 1000)
	int32_t mode;                      // 0 search, 1 waiting for header symbols, 2 waiting for burst end
	int32_t niv, pad_;
	int64_t iva[kNumIv], ivb[kNumIv];  // earlier DM_INIT intervals, most recent first
	Burst   pb;                        // burst in progress
};

VDL2_HD uint32_t float_bits_of(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
#define FLOAT_BITS(f) float_bits_of(f)
// the same state, as far as anything that follows can depend on it (intervals beyond niv and a burst that is not in progress are leftovers)
VDL2_HD bool burst_equal(const Burst &x, const Burst &y) {
	return x.chan == y.chan && x.nsym == y.nsym && x.t_first == y.t_first && x.sync_sample == y.sync_sample && x.end_sample == y.end_sample && x.ord == y.ord
		&& FLOAT_BITS(x.prev_phi0) == FLOAT_BITS(y.prev_phi0) && FLOAT_BITS(x.vdphi) == FLOAT_BITS(y.vdphi) && FLOAT_BITS(x.ppm) == FLOAT_BITS(y.ppm) && FLOAT_BITS(x.vdphi_err) == FLOAT_BITS(y.vdphi_err)
		&& x.prev_n == y.prev_n && x.tl_bits == y.tl_bits && x.syndrome == y.syndrome && x.nf_upd == y.nf_upd && x.sync_evals == y.sync_evals;
}
VDL2_HD bool walk_state_equal(const WalkState &x, const WalkState &y) {
	if(x.a != y.a || x.e != y.e || x.e0 != y.e0 || x.bursts != y.bursts || x.evals != y.evals || x.mode != y.mode || x.niv != y.niv) return false;
	if(FLOAT_BITS(x.pherr1) != FLOAT_BITS(y.pherr1) || FLOAT_BITS(x.pherr2) != FLOAT_BITS(y.pherr2) || FLOAT_BITS(x.prev_dphi) != FLOAT_BITS(y.prev_dphi)) return false;
	for(int i = 0; i < x.niv && i < kNumIv; i++) if(x.iva[i] != y.iva[i] || x.ivb[i] != y.ivb[i]) return false;
	return x.mode == 0 || burst_equal(x.pb, y.pb);
}

VDL2_HD void walk_state_init(WalkState &s) {
	s.a = 0; s.e = 2; s.e0 = 2; s.bursts = 0;
	s.pherr1 = s.pherr2 = kPherrBig; s.prev_dphi = 0.f;
	s.evals = 0; s.mode = 0; s.niv = 0; s.pad_ = 0;
}

// ======================================================================
// element-wise pieces
// ======================================================================
// atan2() in double for demod.c:232,256 ("atan2(im, re)" narrowed to float).  Same value as libm's after the
// narrowing: |error| <= ~3e-16 (about 2 double ulps), so the float result differs from a correctly rounded
// double atan2 only when that lands within 2 ulp_double of a float rounding boundary (~1e-8 of samples; libm and
// OCML differ from each other just as often).  Argument reduction: with t = min/max in [0,1] and c = k/8 the
// nearest eighth, atan(t) = atan(c) + atan((min - c*max)/(max + c*min)), |u| <= 1/16 -> 8-term odd series.
// One f64 division instead of the ~100-instruction library routine: K2 is bound by it.
VDL2_HD double atan2_f64(double y, double x) {
	const double ax = fabs(x), ay = fabs(y);
	const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
#if VDL2_DEVICE_PASS
	// the device only ever sees float samples widened to double: finite, and 0 or >= 1.4e-45 in magnitude.  atan2(+-0, x):
	// +-0 for x > 0 or x = +0, +-pi for x < 0 or x = -0 - no library routine in the kernels
	if(!(mx > 0.0)) return copysign(signbit(x) ? 3.141592653589793 : 0.0, y);
	const float r = (float)mn / (float)mx;
#else
	if(!(mx > 0.0) || !(mx < 1.0e300)) return atan2(y, x);         // zeros, infinities, NaN: leave to the library
	const float r = (float)mn / (float)mx;
	if(!(r >= 0.f && r <= 1.f)) return atan2(y, x);                // operands outside the float range (never for float inputs)
#endif
	const int k = (int)(r * 8.0f + 0.5f);
	// atan(k/8), k = 0..8, picked with selects: an indexed table would be a per-lane memory load in the middle of what is, in
	// the walker and the exact tier of the sync kernel, a latency-bound dependent chain
	const double Ak = k < 4 ? (k < 2 ? (k == 0 ? 0.0 : 0.12435499454676144) : (k == 2 ? 0.24497866312686414 : 0.35877067027057225))
	                        : (k < 6 ? (k == 4 ? 0.4636476090008061 : 0.5585993153435624)
	                                 : (k == 6 ? 0.6435011087932844 : (k == 7 ? 0.7188299996216245 : 0.7853981633974483)));
	const double c = 0.125 * (double)k;
	const double u = fma(-c, mx, mn) / fma(c, mn, mx);
	const double z = u * u;
	double p = -1.0 / 15.0;
	p = fma(p, z, 1.0 / 13.0); p = fma(p, z, -1.0 / 11.0); p = fma(p, z, 1.0 / 9.0); p = fma(p, z, -1.0 / 7.0);
	p = fma(p, z, 1.0 / 5.0); p = fma(p, z, -1.0 / 3.0);
	double a = Ak + fma(p * z, u, u);
	if(ay > ax) a = 1.5707963267948966 - a;
	if(x < 0.0) a = 3.141592653589793 - a;
	return copysign(a, y);
}
VDL2_HD float phase_of(cf32 y) { return (float)atan2_f64((double)y.im, (double)y.re); }

// Single-precision phase for the sync kernel's screening tier only, in TURNS (-0.5, 0.5]: odd minimax polynomial of degree 13
// on min/max with the 1/2pi folded into its coefficients, error < 1e-7 turn = 6e-7 rad (tests/test_phase.py).  No decision is
// taken on it: a window whose screening value is near the threshold, or in which one of its unwrap decisions could go the other
// way with the exact phases, is redone exactly (kScreenGuard).  Turns, because the unique word's phases are exact eighths then
// and the unwrap "subtract the nearest whole turn" is a v_rndne and a subtraction.
VDL2_HD float phase_fast(cf32 y) {
	const float ax = fabsf(y.re), ay = fabsf(y.im);
	const bool steep = ay > ax;
#if VDL2_DEVICE_PASS
	// branch-free, 19 instructions.  A zero sample gives 0 (v_mul_legacy: 0 * inf = 0); a sample so small that v_rcp overflows
	// gives a NaN, which the screening kernel reads as "flag it"
	const float mx = steep ? ay : ax, mn = __builtin_amdgcn_fmed3f(ax, ay, 0.f);
	float t;
	// (the s_nop is the wait state a VALU read of a transcendental result needs on gfx940+; the compiler does not look inside)
	asm("s_nop 0\n\tv_mul_legacy_f32 %0, %1, %2" : "=v"(t) : "v"(mn), "v"(__builtin_amdgcn_rcpf(mx)));
#else
	const float mx = steep ? ay : ax, mn = steep ? ax : ay;
	if(!(mx > 0.f)) return 0.f;
	const float t = mn / mx;
#endif
	const float z = t * t;
	constexpr double k = 0.15915494309189535;     // 1 / 2 pi
	float p = (float)(0.006811664905399084 * k);
	p = fmaf(p, z, (float)(-0.03360380604863167 * k)); p = fmaf(p, z, (float)(0.07962316274642944 * k)); p = fmaf(p, z, (float)(-0.13233311474323273 * k));
	p = fmaf(p, z, (float)(0.19807806611061096 * k)); p = fmaf(p, z, (float)(-0.3331736624240875 * k)); p = fmaf(p, z, (float)(0.9999961256980896 * k));
	float a = p * t;
	a = steep ? 0.25f - a : a;
	a = y.re < 0.f ? 0.5f - a : a;
	return copysignf(a, y.im);
}

// hypotf() as glibc evaluates it (double intermediate), demod.c:238
VDL2_HD float mag_of(cf32 y) { return (float)sqrt((double)y.re * (double)y.re + (double)y.im * (double)y.im); }

// got_sync() up to the threshold test: demod.c:129-171.  ph[i] = phase 150-10i samples ago.
VDL2_HD void sync_metric(const float *ph, const Tables &T, float &pherr, float &slope_out) {
	float e[kPreamble];
	float mean = 0.f, unwrap = 0.f;
	float prev = mean = e[0] = ph[0] - T.pr_phase[0];
	for(int i = 1; i < kPreamble; i++) {
		float cur = ph[i] - T.pr_phase[i];
		float diff = cur - prev;
		prev = cur;
		// demod.c:137-141: "if(errdiff > M_PI) unwrap -= 2.0f * M_PI; else if(errdiff < -M_PI) unwrap += ..." with the
		// float operands promoted to double.  (double)x > M_PI  <=>  x > kPiBelow for a float x, and adding 0.0 in
		// double and narrowing back leaves unwrap unchanged, so the update is written without branches.
		const double step = diff > kPiBelow ? -(2.0f * M_PI) : (diff < -kPiBelow ? (2.0f * M_PI) : 0.0);
		unwrap = (float)((double)unwrap + step);
		e[i] = cur + unwrap;
		mean += e[i];
	}
	mean /= kPreamble;
	for(int i = 0; i < kPreamble; i++) e[i] -= mean;
	float slope = 0.f;
	for(int i = 0; i < kPreamble; i++) slope += T.lrx[i] * e[i];
	slope /= T.lr_den;
	float acc = 0.f;
	for(int i = 0; i < kPreamble; i++) {
		float r = e[i] - slope * T.lrx[i];
		acc += r * r;
	}
	pherr = acc; slope_out = slope;
}

// Screening form of sync_metric(): the same unwrap decisions, taken on single-precision phases in turns, and the residual from
// running sums, p = S2 - S0^2/16 - S1^2/den.  It works on the raw differences d[i] = phase(tap i) - phase(tap i-1) - which the
// windows n, n+10, n+20, ... share, so a lane that screens several of those forms them once - and the unique word's own phase
// steps, which are exact eighths of a turn:
//   u = d[i] - dq[i]/8                    the reference's cur[i] - cur[i-1]                              (demod.c:133-136)
//   w = u - clamp(rint(u), -1, 1)         one step of -+2 pi when the difference is beyond +-pi          (demod.c:137-141)
//   e[i] = e[i-1] + w                     = cur[i] + unwrap
// It differs from the exact value by rounding only (< 0.2 rad^2 for any phase sequence), which is all the sync kernel needs to
// know that a sample is nowhere near the threshold kSyncThr.  screen_value() is in rad^2 like the exact metric.
constexpr float kScreenThr = 5.5f;
// The running sums are raw moments (sum e, sum i*e, sum e^2), so the value over the first n taps - the residual of the best
// line through those n points, a lower bound of the residual over all 16 - is available along the way: the sync kernel stops
// after kScreenEarly taps when no lane of the wavefront is still under the threshold (97 % of them on noise or data).
constexpr int kScreenEarly = 12;
constexpr float kScreenEarlyThr = 5.8f;   // early bound + its rounding slack must stay above kScreenThr
// An unwrap decision - "is the difference of two taps beyond half a turn" - could differ from the one the reference takes on
// its own phases when |u| is within the combined error of 0.5: two phase_fast() errors (2e-7 turn), the roundings of d and u
// (1.8e-7 turn) and the reference's own three roundings in radians (1e-6 rad = 1.6e-7 turn).  A window where some |u| comes
// within kScreenGuard of 0.5 is treated as flagged.  (|u| = 1.5, where the clamp starts to matter, is not a decision of the
// reference - it never takes a second step - and needs no guard: rint() and the clamp give the same w on both sides of it.)
constexpr float kScreenGuard = 3.2e-6f;   // turns (2e-5 rad)
// cumulative unique-word phases in eighths of a turn (tables.h: q[]) differenced: dq[i] = q[i] - q[i-1]
VDL2_HD constexpr int screen_dq(int i) {
	constexpr int dq[kPreamble] = { 0, 3, -6, 4, 0, 1, -2, 4, -7, 7, -6, 5, -2, -3, -1, 3 };
	return dq[i];
}
// gmax: largest |w| over the taps whose |u| cannot reach 1.5 (|d| <= 1, |dq| < 4), where 0.5 - |w| is the distance from the
// decision; gmin: smallest ||u| - 0.5| over the others
struct ScreenAcc { float e, m0, m1, m2, gmax, gmin; };

// tap 0: e[0] = phase of tap 0 in the reference - taken as 0 here: the residual of a least-squares line does not change when a
// constant is added to every point, the unwrap decisions never look at e[0], and three instructions per window go away
VDL2_HD void screen_begin(ScreenAcc &a) {
	a.e = 0.f; a.m0 = 0.f; a.m1 = 0.f; a.m2 = 0.f; a.gmax = 0.f; a.gmin = 1.f;
}

// taps i0 <= i < i1 (i0 >= 1), d[i] as above
VDL2_HD void screen_taps(const float *d, int i0, int i1, ScreenAcc &a) {
	for(int i = i0; i < i1; i++) {
		const int dq = screen_dq(i);
		const float u = dq == 0 ? d[i] : d[i] - 0.125f * (float)dq;
		float q = rintf(u);
		float w;
		if(dq > -4 && dq < 4) {
			w = u - q;                                // |u| < 1.5: rint() is already within +-1
			a.gmax = fmaxf(a.gmax, fabsf(w));
		} else {
#if VDL2_DEVICE_PASS
			q = __builtin_amdgcn_fmed3f(q, -1.f, 1.f);
#else
			q = q < -1.f ? -1.f : (q > 1.f ? 1.f : q);
#endif
			w = u - q;
			a.gmin = fminf(a.gmin, fabsf(fabsf(u) - 0.5f));
		}
		a.e += w;
		a.m0 += a.e; a.m1 = fmaf((float)i, a.e, a.m1); a.m2 = fmaf(a.e, a.e, a.m2);
	}
}

// residual of the least-squares line through the first n points: m2 - m0^2/n - (m1 - xbar m0)^2 / Sxx, Sxx = n(n^2-1)/12
// (n is a constant at every call site: the reciprocals fold, no division is executed - it is a screening value), in rad^2
VDL2_HD float screen_value(const ScreenAcc &a, int n) {
	const float xbar = 0.5f * (float)(n - 1), inv_n = 1.0f / (float)n, inv_sxx = 12.0f / (float)(n * (n * n - 1));
	const float c = a.m1 - xbar * a.m0;
	const float v = (a.m2 - a.m0 * a.m0 * inv_n - c * c * inv_sxx) * (float)(4.0 * M_PI * M_PI);
	// an unwrap decision too close to call: let the exact tier look
	return (a.gmax > 0.5f - kScreenGuard || a.gmin < kScreenGuard) ? 0.f : v;
}

// ph[i]: phase_fast() of tap i (turns)
VDL2_HD float sync_metric_screen(const float *ph, int ntaps = kPreamble) {
	float d[kPreamble];
	for(int i = 1; i < ntaps; i++) d[i] = ph[i] - ph[i - 1];
	ScreenAcc a;
	screen_begin(a);
	screen_taps(d, 1, ntaps, a);
	return screen_value(a, ntaps);
}

// calc_para_vertex(v->sclk = 0, SYNC_SKIP, y1, y2, y3): demod.c:98-103,178
VDL2_HD float parabola_vertex(float y1, float y2, float y3) {
	const float x = 0.f; const int d = kSyncSkip;
	float denom = (float)(d * 2 * d * (-d));
	float a = (x * (y2 - y1) + (x - d) * (y1 - y3) + (x - 2 * d) * (y3 - y2)) / denom;
	float b = (x * x * (y1 - y2) + (x - d) * (x - d) * (y3 - y1) + (x - 2 * d) * (x - 2 * d) * (y2 - y3)) / denom;
	return -b / (2 * a);
}

// The --max-ppm gate (demod.c:185,190-192): ppm_error = 10500 * dphi / (2 pi freq) * 1e6 in the reference's float/double mix.
VDL2_HD float ppm_of(float vdphi, uint32_t freq) { return (float)((double)(10500 * vdphi) / (2.0f * M_PI * (double)freq) * 1e+6); }
// |ppm_of(x)| is even and non-decreasing in |x| (a float product by a positive constant, an exact widening, a division by and a
// product with positive constants, a narrowing: every step rounds monotonically), so the gate "fabsf(ppm) > max_ppm" is a
// comparison of |x| with one number: the largest float the gate lets pass.  Found once per walk by bisection on the bit pattern.
VDL2_HD float ppm_gate_threshold(uint32_t freq, float max_ppm) {
	uint32_t lo = 0u, hi = 0x7f800000u;            // passes (ppm 0) / does not (infinite)
	while(hi - lo > 1u) {
		const uint32_t mid = lo + (hi - lo) / 2u;
		float x;
#if VDL2_DEVICE_PASS
		x = __uint_as_float(mid);
#else
		__builtin_memcpy(&x, &mid, 4);
#endif
		if(fabsf(ppm_of(x, freq)) > max_ppm) hi = mid; else lo = mid;
	}
	float t;
#if VDL2_DEVICE_PASS
	t = __uint_as_float(lo);
#else
	__builtin_memcpy(&t, &lo, 4);
#endif
	return t;
}

// One D8PSK decision: demod.c:256-264.  Returns the phase-step index 0..7.
VDL2_HD int slice_symbol(float phi, float prev_phi, float vdphi, int &neg) {
	float dphi = phi - prev_phi - vdphi;
	if(dphi < 0) dphi = (float)((double)dphi + 2.0f * M_PI);
	else if((double)dphi > 2.0f * M_PI) dphi = (float)((double)dphi - 2.0f * M_PI);
	dphi = (float)((double)dphi / M_PI_4);
	int idx = (int)roundf(dphi) % 8;
	if(idx < 0) { neg++; idx &= 7; }   // the reference indexes graycode[] out of bounds here
	return idx;
}

// got_sync() metric at decimated sample n assuming the phase ring holds the 160 most recent
// samples contiguously (taps n-150, n-140, ..., n): what the sync kernel's exact tier tabulates.
VDL2_HD cf32 metric_contiguous(const ChanView &v, int64_t n, const Tables &T) {
	float ph[kPreamble];
	for(int i = 0; i < kPreamble; i++) ph[i] = v.Phi(n - 150 + 10 * i);
	cf32 r;
	sync_metric(ph, T, r.re, r.im);
	return r;
}

// candidate flag: the only places where got_sync() can succeed on a contiguous ring (demod.c:173)
VDL2_HD bool is_candidate(float p_prev, float p_now) { return p_prev < kSyncThr && p_now > p_prev; }

VDL2_HD int fec_octets_for(uint32_t len) { return len < 3 ? 0 : len < 31 ? 2 : len < 68 ? 4 : 6; }  // decode.c:124-133

VDL2_HD int popc32(uint32_t v) { return __builtin_popcount(v); }
VDL2_HD int ctz32(uint32_t v) { return __builtin_ctz(v); }
VDL2_HD uint32_t parity32(uint32_t v) { v ^= v >> 16; v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1; }

struct Geometry { uint32_t tl_bits, octets, nblocks, last_len, fec_octets, want_bits, syndrome; int status; };
enum { HDR_OK = 0, HDR_CRC_BAD = 1, HDR_TOO_LONG = 2, HDR_NO_FEC = 3 };

// Header word (25 bits, MSB first) -> burst geometry: decode.c:209-258
VDL2_HD Geometry header_to_geometry(uint32_t hdr, const uint32_t *tH, const uint32_t *tfix) {
	Geometry g{};
	const uint32_t keep = (1u << (kTlBits + kHdrParBits)) - 1;
	hdr &= keep;
	uint32_t s = 0;
	for(int i = 0; i < kHdrParBits; i++) s |= parity32(hdr & tH[i]) << (kHdrParBits - 1 - i);
	hdr ^= tfix[s];
	g.syndrome = s;
	if((hdr & keep) != hdr) { g.status = HDR_CRC_BAD; return g; }
	hdr >>= kHdrParBits;
	uint32_t tl = 0;
	for(int i = 0; i < kTlBits; i++) if(hdr & (1u << i)) tl |= 1u << (kTlBits - 1 - i);
	g.tl_bits = tl;
	if((s != 0 && tl > kMaxTlCorr) || tl > kMaxTl) { g.status = HDR_TOO_LONG; return g; }
	g.octets = tl / 8 + (tl % 8 != 0);
	g.nblocks = g.octets / kRsK;
	g.fec_octets = g.nblocks * kRsPar;
	g.last_len = g.octets % kRsK;
	if(g.last_len != 0) g.nblocks++;
	g.fec_octets += (uint32_t)fec_octets_for(g.last_len);
	if(g.last_len == 0) g.last_len = kRsK;
	if(g.fec_octets == 0) { g.status = HDR_NO_FEC; return g; }
	g.want_bits = 8 * (g.octets + g.fec_octets);
	g.status = HDR_OK;
	return g;
}

// ======================================================================
// Referee.  The channeliser evaluates the reference's filter in block form; its output is what exact arithmetic gives, and
// differs from the reference's own fp32 scan (demod.c:302-329) by that scan's rounding noise: <= 1.4e-4 of the largest
// sample within the last 8 (measured over 1e7 samples of the fuzz and bench workloads; rms 1e-5).  Every decision the
// reference takes on those samples - candidate test, parabola vertex, --max-ppm gate, symbol slicer - is therefore taken
// here WITH A MARGIN: when the decision could come out differently for some stream within that distance, the wavefront asks
// ref_exact_window() for the reference's own samples of the stretch the decision reads (the scan re-run sequentially in the
// reference's operation order from RefChan::warm input samples back - 196 608 by default: two such scans started from different
// states are bit-identical after 1.7e4 - 2.6e4 samples on average, and a zero-start scan had not yet become the reference's after
// 2^17 in 1.9e-3 of the stretches measured, after 2^18 in 2.2e-5 (kernels.h)) and takes the decision again on those.  A decision is then either robust against the stream's error or taken on
// samples that are the reference's own with that probability; where they are not yet, they are within its rounding noise of them.
// ======================================================================
constexpr float kRefKappa = 3.0e-4f;       // bound used for |y - y_ref| / (largest |y| among the samples a decision reads): 2x the worst seen
constexpr float kRefBig = 1.0e30f;
constexpr int   kRefPre = 156, kRefPost = 96;   // a marginal candidate at n: evaluations n-6, n-3, n read n-156..n; sync point + 9 header symbols end before n+96

// the stretch [n_lo, n_hi] of the channel's decimated stream becomes the reference's own; wave-uniform call, `scratch`: >= 2 KiB of
// LDS the caller can spare.  false: not possible (no referee, or the raw input is no longer held) - the caller keeps its decision.
VDL2_HD void ref_debug_log(const ChanView &v, int tag, int64_t a, float b, float c, float d);   // development aid (a no-op unless the build provides one)
enum { REF_CANDIDATE = 0, REF_HEADER = 1, REF_SYMBOLS = 2, REF_STALE = 3 };   // who asks (statistics; a test hook can switch a kind off)
// a decision taken on the channeliser's samples although it lies within the margin (optimistic mode)
struct RefReq {
	int32_t  chan, kind;               // REF_CANDIDATE / REF_HEADER
	int64_t  n;                        // the candidate / the sync sample
	int64_t  t_first, prev_n;          // header: the first symbol's sample, the sample whose phase is prev_phi0
	float    vdphi, vdphi_err, prev_phi0, pad_;
	uint32_t code, pad2_;              // what was decided.  Candidate: ref_candidate_code().  Header: the nine symbols, three bits each
};
// got_sync()'s outcome at a candidate: bit 0 it fires (demod.c:173), bit 1 it passes the --max-ppm gate (:190), bits 8-15 sclk + 64 (:179)
VDL2_HD uint32_t ref_candidate_code(float y1, float y2, float y3, float prevd, float max_ppm, float ppm_thr) {
	if(!is_candidate(y2, y3)) return 0u;
	const int sclk = (int)(-roundf(parabola_vertex(y1, y2, y3)));
	const uint32_t pass = !(max_ppm != 0.f && fabsf(prevd) > ppm_thr) ? 2u : 0u;
	return 1u | pass | ((uint32_t)((sclk + 64) & 0xff) << 8);
}
// one lane (inside a LANE0 section).  false: the list is full
VDL2_HD bool ref_log_request(const ChanView &v, const RefReq &r) {
#if VDL2_DEVICE_PASS
	const uint32_t i = atomicAdd(v.rq_n, 1u);
#else
	const uint32_t i = (*v.rq_n)++;
#endif
	if(i >= v.rq_cap) return false;
	v.rq[i] = r;
	return true;
}
VDL2_HD bool ref_exact_window(const ChanView &v, int64_t n_lo, int64_t n_hi, void *scratch, int kind);
VDL2_HD bool ref_window_done(const ChanView &v, int64_t n_lo, int64_t n_hi);      // has another launch made the stretch exact?  (never scans)

// squared bound on the phase error of decimated sample n: the stream's error there is at most kRefKappa x the largest of the
// sample and its three predecessors (the scan's rounding noise is an exponentially weighted average of |y| over ~2 decimated
// samples; measured: <= 1.5e-4 of that maximum, rms 1e-5), hence its phase's kRefKappa * max / |y|.  0 outside the stream
// (phase exactly the reference's), kRefBig for a sample that is exactly zero inside it.
VDL2_HD float ref_eps2_of(float m2_0, float m2_1, float m2_2, float m2_3) {
	const float a = fmaxf(fmaxf(m2_0, m2_1), fmaxf(m2_2, m2_3));
	return m2_0 > 0.f ? kRefKappa * kRefKappa * a / m2_0 : kRefBig;
}
VDL2_HD float ref_eps2(const ChanView &v, int64_t n) {
	if(n < 0) return 0.f;
	float m2[4] = {0.f, 0.f, 0.f, 0.f};
	cf32 q[4];
	for(int k = 0; k < 4; k++) q[k] = n - k >= 0 ? v.Y(n - k) : cf32{0.f, 0.f};     // loads first
	for(int k = 0; k < 4; k++) m2[k] = q[k].re * q[k].re + q[k].im * q[k].im;
	return ref_eps2_of(m2[0], m2[1], m2[2], m2[3]);
}
// how far a got_sync() metric value can be from the reference's: |dp| <= 2 sqrt(p) E + E^2 when the 16 phases move by a vector of
// length E (mean and slope removal are projections), E = kappa A sqrt(sum 1/|y_i|^2), A = the largest tap
// E sums the taps' WORST-CASE bounds as if all sixteen errors were at their maximum and lined up with the residual; they are
// independent, and mostly thirty times smaller: over 25 000 preamble-like windows of the fuzz and bench captures the metric moved by
// at most 0.040 of that figure (rms 0.006), the slope by at most 0.034 (dev/ref_margin_calibration.py).  kRefSum = 2x the worst seen.
constexpr float kRefSum = 0.085f;
VDL2_HD float ref_pherr_margin(float p, float E) { const float e = kRefSum * E; return 2.0f * sqrtf(p) * e + e * e + 4e-6f * p + 1e-6f; }
// ... and the slope: |df| <= E sqrt(sum lrx^2) / lr_den = E / sqrt(340)
VDL2_HD float ref_slope_margin(float E) { return 0.0543f * kRefSum * E + 1e-7f; }

// could -roundf(calc_para_vertex()) come out differently for y1, y2, y3 anywhere in their boxes?  (the vertex is a ratio of two
// forms linear in each y: over a box its extremes are at corners unless the denominator changes sign, which the corners show too)
struct RefRange { float lo, hi; };
VDL2_HD bool ref_vertex_marginal(RefRange y1, RefRange y2, RefRange y3) {
	// (the three values' errors are independent: the vertex moves by the root of the sum of the squares of what each alone does, taken
	// from its own end points; a value that "cannot be told" - [0, big] - or a vertex that runs away decides it outright)
	const float c1 = 0.5f * (y1.lo + y1.hi), c2 = 0.5f * (y2.lo + y2.hi), c3 = 0.5f * (y3.lo + y3.hi);
	if(y1.hi - y1.lo > 100.f || y2.hi - y2.lo > 100.f || y3.hi - y3.lo > 100.f) return true;
	const float v0 = parabola_vertex(c1, c2, c3);
	float d2 = 0.f;
	for(int k = 0; k < 3; k++) {
		const float a = parabola_vertex(k == 0 ? y1.lo : c1, k == 1 ? y2.lo : c2, k == 2 ? y3.lo : c3);
		const float b = parabola_vertex(k == 0 ? y1.hi : c1, k == 1 ? y2.hi : c2, k == 2 ? y3.hi : c3);
		if(!(a == a) || !(b == b) || fabsf(a) > 1.0e6f || fabsf(b) > 1.0e6f) return true;
		const float d = fmaxf(fabsf(a - v0), fabsf(b - v0));
		d2 += d * d;
	}
	if(!(v0 == v0)) return true;
	const float d = sqrtf(d2) + 1e-5f;
	return roundf(v0 - d) != roundf(v0 + d);
}

// the values the reference's metric can have where this path's is p: [p - m, p + m], widened to the value with one unwrap decision
// taken the other way (palt, see sync_metric_ref) where such a decision hangs on the stream's error
VDL2_HD RefRange ref_pherr_range(float p, float alo, float ahi, float E) {
	if(!(p < kPherrBig)) return RefRange{ kPherrBig, kPherrBig };       // not tabulated / not part of the run: exactly PHERR_MAX
	if(E >= kRefBig) return RefRange{ 0.f, kRefBig };
	const float a = p < alo ? p : alo, b = p > ahi ? p : ahi;
	return RefRange{ a - ref_pherr_margin(a, E), b + ref_pherr_margin(b, E) };
}

// got_sync()'s verdict at a candidate, given what K3's exact tier knows there: the metric at n, n-3 (with its slope f3 and error
// figure E3) and n-6 (kPherrBig when that evaluation is not part of the run, or was not tabulated: then the walker does not use it either).
// bit 0: the candidate may fire; bit 1: some decision of the fire (candidate test, vertex with either y1, gate) is within the margin
VDL2_HD int ref_candidate_verdict(RefRange r0, RefRange r3, float f3, float E3, RefRange r6, float max_ppm, float ppm_thr) {
	if(!(r3.lo < kPherrBig) || !(r0.lo < kPherrBig)) return 0;
	const bool possibly = r3.lo < kSyncThr && r0.hi > r3.lo;
	if(!possibly) return 0;
	const bool surely = r3.hi < kSyncThr && r0.lo > r3.hi;
	bool marg = !surely;
	if(!marg) {
		marg = ref_vertex_marginal(RefRange{ kPherrBig, kPherrBig }, r3, r0) || (r6.lo < kPherrBig && ref_vertex_marginal(r6, r3, r0));
		if(!marg && max_ppm != 0.f) marg = fabsf(fabsf(f3) - ppm_thr) <= ref_slope_margin(E3);
	}
	return 1 | (marg ? 2 : 0);
}

// sync_metric() with the unwrap decision at tap `flip` taken the other way (the value the reference gets when that decision,
// which hangs on a phase difference within the stream's error of +-pi, goes the other way on its samples; flip < 0: as it is)
VDL2_HD float sync_metric_flipped(const float *ph, const Tables &T, int flip, int cut = -1) {      // cut: the tap that reads -phase (atan2's branch cut)
	float e[kPreamble];
	float mean = 0.f, unwrap = 0.f;
	float prev = mean = e[0] = (cut == 0 ? -ph[0] : ph[0]) - T.pr_phase[0];
	for(int i = 1; i < kPreamble; i++) {
		float cur = (i == cut ? -ph[i] : ph[i]) - T.pr_phase[i];
		float diff = cur - prev;
		prev = cur;
		double step = diff > kPiBelow ? -(2.0f * M_PI) : (diff < -kPiBelow ? (2.0f * M_PI) : 0.0);
		if(i == flip) step = step != 0.0 ? 0.0 : (diff > 0.f ? -(2.0f * M_PI) : (2.0f * M_PI));
		unwrap = (float)((double)unwrap + step);
		e[i] = cur + unwrap;
		mean += e[i];
	}
	mean /= kPreamble;
	for(int i = 0; i < kPreamble; i++) e[i] -= mean;
	float slope = 0.f;
	for(int i = 0; i < kPreamble; i++) slope += T.lrx[i] * e[i];
	slope /= T.lr_den;
	float acc = 0.f;
	for(int i = 0; i < kPreamble; i++) { float r = e[i] - slope * T.lrx[i]; acc += r * r; }
	return acc;
}

VDL2_HD void sync_metric_two(const float *ph, const Tables &T, const int *ev, const int *kind, float pherr, float &lo, float &hi);
// The values the metric takes when ONE or TWO unwrap decisions (taps ev[0], ev[1]; n of them) go the other way, without running the
// metric again: a decision taken the other way moves the unwrapped phases from its tap on by +-2 pi, a step H_i; mean and slope
// removal are a projection R, so the residual moves by +-2 pi R H_i and
//   p' = p + 2 (2 pi) sum s_i (r . H_i) + (2 pi)^2 sum s_i s_j (H_i . R H_j),   r . H_i = the sum of the residuals from tap i on,
//   H_i . R H_j = 16 - max(i, j) - (16 - i)(16 - j) / 16 - L_i L_j / lr_den,   L_i = sum of lrx from tap i on = i (16 - i) / 2.
// Equal to sync_metric_flipped()'s value up to rounding (a few 1e-5 on values of tens to hundreds: the caller's range is widened by
// what that could be); a tenth of its work - these windows are most of what the exact sync tier's margins cost: inside a burst the phase
// steps between taps are multiples of pi / 4 and one in eight is +-pi to within the noise.  lo / hi: the smallest / largest value.
VDL2_HD void sync_metric_unwrap_alts(const float *ph, const Tables &T, const int *ev, int n, float pherr, float &lo, float &hi) {
	float e[kPreamble];
	float mean = 0.f, unwrap = 0.f;
	float sgn[2] = {0.f, 0.f};
	float prev = mean = e[0] = ph[0] - T.pr_phase[0];
	for(int i = 1; i < kPreamble; i++) {
		const float cur = ph[i] - T.pr_phase[i], diff = cur - prev;
		prev = cur;
		const double step = diff > kPiBelow ? -(2.0f * M_PI) : (diff < -kPiBelow ? (2.0f * M_PI) : 0.0);
		// the other way: a step that was taken is not (+-1 turn back), one that was not is taken in the direction of the difference
		for(int k = 0; k < n; k++) if(ev[k] == i) sgn[k] = step != 0.0 ? (step < 0.0 ? 1.f : -1.f) : (diff > 0.f ? -1.f : 1.f);
		unwrap = (float)((double)unwrap + step);
		e[i] = cur + unwrap;
		mean += e[i];
	}
	mean /= kPreamble;
	float slope = 0.f;
	for(int i = 0; i < kPreamble; i++) { e[i] -= mean; slope += T.lrx[i] * e[i]; }
	slope /= T.lr_den;
	float rho[2] = {0.f, 0.f};
	for(int i = kPreamble - 1; i >= 1; i--) {
		const float r = e[i] - slope * T.lrx[i];
		for(int k = 0; k < n; k++) if(i >= ev[k]) rho[k] += r;
	}
	const float tp = (float)(2.0 * M_PI);
	auto G = [&](int i, int j) { const float Li = 0.5f * (float)(i * (kPreamble - i)), Lj = 0.5f * (float)(j * (kPreamble - j));
	                             return (float)(kPreamble - (i > j ? i : j)) - (float)((kPreamble - i) * (kPreamble - j)) / (float)kPreamble - Li * Lj / T.lr_den; };
	lo = hi = pherr;
	for(int c = 1; c < (1 << n); c++) {
		float v = pherr;
		for(int k = 0; k < n; k++) if((c >> k) & 1) {
			v += 2.0f * tp * sgn[k] * rho[k] + tp * tp * G(ev[k], ev[k]);
			for(int m = 0; m < k; m++) if((c >> m) & 1) v += 2.0f * tp * tp * sgn[k] * sgn[m] * G(ev[k], ev[m]);
		}
		const float slack = 1e-3f + 2e-5f * fabsf(v);
		lo = v - slack < lo ? v - slack : lo; hi = v + slack > hi ? v + slack : hi;
	}
	if(lo < 0.f) lo = 0.f;
}
// sync_metric() plus what the referee needs: E (see ref_pherr_margin) and palt - the value the reference gets when the ONE
// discontinuity within the stream's error goes the other way on its samples: an unwrap decision (a phase difference within the
// margin of +-pi), or a tap whose phase is within the margin of atan2()'s branch cut (it then reads +pi for -pi: with the
// reference's single unwrap step per tap the metric is not continuous there).  palt = pherr when there is none; with several, or
// a tap that is exactly zero, E = kRefBig: "cannot tell".  eps2[i]: ref_eps2() of tap i
// `full`: one or two unwrap decisions within the margin are worked out (sync_metric_unwrap_alts) - the commonest kind by far: 93 % of
// config4's marked candidates were windows with two of them, inside bursts, whose metric is 20-200 whichever way the two go and which
// round 5 gave up on ("cannot tell": two unwrap decisions could not both be taken the other way).  On for receivers that scan the
// stretches around EVERY marked candidate ahead of the walk (<= 64 channels: a tenth of the scans, their step 10-40 % shorter); off -
// the old verdicts, `template` in the exact sync tier so that its code is the old code - for the others, whose walk visits few of
// those windows (159 -> 121 scans per config4 block) and whose front paid 4-5 % for the extra arithmetic and registers
// (profiles/r06_fewer_marks_ab.txt).
VDL2_HD void sync_metric_ref(const float *ph, const float *eps2, int estride, const Tables &T, float &pherr, float &slope_out, float &E_out, float &alo, float &ahi, bool full = true) {      // eps2[i * estride]; [alo, ahi]: the alternatives' values
	sync_metric(ph, T, pherr, slope_out);
	float s = 0.f; bool big = false; int nev = 0, flip = -1, cut = -1; int ev[2] = {0, 0}, kind[2] = {0, 0};
	float eprev = sqrtf(eps2[0]), cprev = ph[0] - T.pr_phase[0];
	for(int i = 0; i < kPreamble; i++) {
		const float q = eps2[i * estride];
		s += q; big = big || q >= kRefBig;
		if(kPiBelow - fabsf(ph[i]) <= sqrtf(q) + 1e-6f && q > 0.f) { if(nev < 2) { ev[nev] = i; kind[nev] = 1; } nev++; cut = i; }
	}
	for(int i = 1; i < kPreamble; i++) {
		const float cur = ph[i] - T.pr_phase[i], diff = cur - cprev, e = sqrtf(eps2[i * estride]);
		if(e + eprev > 0.f && fabsf(fabsf(diff) - kPiBelow) <= e + eprev + 4e-6f) { if(nev < 2) { ev[nev] = i; kind[nev] = 0; } nev++; flip = i; }   // (two taps that are the reference's own: its decision, exactly)
		cprev = cur; eprev = e;
	}
	alo = ahi = pherr;
	if(s == 0.f && !big) { E_out = 0.f; return; }                 // (every tap is the reference's own)
	if(full && nev >= 1 && nev <= 2 && !big && !kind[0] && (nev == 1 || !kind[1])) sync_metric_unwrap_alts(ph, T, ev, nev, pherr, alo, ahi);
	else if(nev == 1 && !big) alo = ahi = sync_metric_flipped(ph, T, flip, cut);
	else if(nev == 2 && !big) { sync_metric_two(ph, T, ev, kind, pherr, alo, ahi); if(ahi >= kRefBig) big = true; }
	E_out = (big || nev > 2) ? kRefBig : sqrtf(s);
}
// ... with TWO discontinuities within the margin (noise windows with a faded tap or two: half of what "cannot be told" on the bench
// workloads): the smallest and the largest of the four values the reference can get.  (ev[k]: tap, kind[k]: 0 unwrap decision, 1 branch cut)
VDL2_HD void sync_metric_two(const float *ph, const Tables &T, const int *ev, const int *kind, float pherr, float &lo, float &hi) {
	lo = hi = pherr;
	for(int c = 1; c < 4; c++) {
		float q[kPreamble];
		for(int i = 0; i < kPreamble; i++) q[i] = ph[i];
		int flip = -1;
		for(int k = 0; k < 2; k++) if((c >> k) & 1) { if(kind[k]) q[ev[k]] = -q[ev[k]]; else flip = ev[k]; }
		// (two unwrap decisions cannot both go through sync_metric_flipped(): "cannot tell" here - sync_metric_ref's `full` form works
		// them out another way, sync_metric_unwrap_alts)
		float v;
		if((c == 3) && !kind[0] && !kind[1]) { lo = 0.f; hi = kRefBig; return; }
		v = sync_metric_flipped(q, T, flip, -1);
		lo = v < lo ? v : lo; hi = v > hi ? v : hi;
	}
}

// One D8PSK decision with its margin: the distance (radians) of slice_symbol()'s argument from the nearest decision boundary ...
VDL2_HD float ref_symbol_dist(float phi, float prev_phi, float vdphi) {
	float dphi = phi - prev_phi - vdphi;
	if(dphi < 0) dphi = (float)((double)dphi + 2.0f * M_PI);
	else if((double)dphi > 2.0f * M_PI) dphi = (float)((double)dphi - 2.0f * M_PI);
	dphi = (float)((double)dphi / M_PI_4);
	const float fr = dphi - floorf(dphi);                       // decision boundaries at k + 0.5
	return fabsf(fr - 0.5f) * (float)M_PI_4;
}
// 1 / |y|^2 of a symbol's sample (the burst decoder bounds the stream's error there by kRefKappa x the larger of the decision's two
// samples and twice the burst's rms instead of looking at the sample's predecessors: it reads thousands of symbols)
VDL2_HD float ref_inv_mag2(cf32 y) { const float m2 = y.re * y.re + y.im * y.im; return m2 > 0.f ? 1.0f / m2 : kRefBig; }
// ... and whether the decision could come out differently when the two phases and the carrier slope move by e_sum in all
VDL2_HD bool ref_symbol_marginal(float phi, float prev_phi, float vdphi, float e_sum) { return ref_symbol_dist(phi, prev_phi, vdphi) <= e_sum + 2e-6f; }
// ======================================================================
// Walker: the per-channel sequential FSM of demod()/got_sync(), hopping
// between the sparse places where something can happen.
// ======================================================================
struct WalkShared {
	WalkState st;
	float p[64], f[64];
	float vring[320];
	int32_t found[64];
	int32_t flag[64];
	int32_t sym[16];
	int32_t neg[16];
	// scalars published by LANE0 sections
	float u_y1, u_y2, u_y3, u_prevd;
	// open chunk of the evaluation log
	int64_t lg_first, lg_count; uint32_t lg_n;
	// speculative loads made together with a candidate's metric values (one memory round trip instead of three)
	float spec[48]; int64_t spec_n; int64_t vring_a;
	// small read-only tables staged once per launch
	uint32_t t_H[kHdrParBits], t_fix[32]; uint8_t t_gray[8], t_prbs[32];
	uint32_t nb;                       // bursts emitted by this channel in this feed
	int64_t first_fire;                // sample of the first got_sync() success (any outcome) since walk_load(); INT64_MAX if none
	uint32_t cap_log;                  // ctl->cap_log, read once (walk_load): the log's bookkeeping then never waits for global memory
	// the candidate words of the last bitmap pass (words cw0 <= w < cw_end) and the metric values around the last fire (samples
	// wbase .. wbase+63): a preamble - above all a neighbour's, which the --max-ppm gate drops - sets a cluster of candidate bits,
	// and the fires after the first of a cluster are then decided by one lane out of LDS, without another trip to memory
	uint64_t cw[kCandWin]; int64_t cw0, cw_end;
	uint64_t nzw[kCandWin / 64];       // bit i: cw[i] != 0 (a lane owns the byte of its eight words): one lane finds the next word worth a look without reading the empty ones
	float wre[64], wim[64]; int64_t wbase;
	int32_t u_fire; int64_t u_n;
	// referee: the candidate whose metric values have been redone on the reference's own samples, and those values (evaluations x_n,
	// x_n - 3, x_n - 6); x_hdr: sync sample of the burst whose header symbols have been; u_verr: bound on the error of u_prevd
	int64_t x_n, x_hdr; float x_p0, x_p3, x_f3, x_p6, u_verr; int32_t x_ok;
	float hph[16], him2[16];           // header: phases and 1/|y|^2 of the sync point and the nine symbols
	int64_t x_lo, x_hi;                // the stretch this wavefront has had made exact last (the candidates of one preamble ask for overlapping ones)
	// referee, evaluations near an interval start (their taps reach through the interval history): squared phase bounds of the staged
	// ring, the evaluations' values / ranges / error figures (two before the pass + 64), which of them are within the margin, and the
	// interval start whose ring has been made exact (up to sample xs_hi)
	// (they live in the candidate-word cache `cw`, which such a pass gives up: StaleRef)
	int64_t xs_a, xs_hi;
};

// lowest lane whose flag is set, or -1.  Call from wave-uniform code after a WAVE_END.
#if VDL2_DEVICE_PASS
VDL2_HD int wave_first_flag(const int32_t *flags) {
	const unsigned long long b = __ballot(flags[VDL2_LANE()] != 0);
	return b ? (int)__ffsll((long long)b) - 1 : -1;
}
#else
VDL2_HD int wave_first_flag(const int32_t *flags) {
	for(int l = 0; l < 64; l++) if(flags[l]) return l;
	return -1;
}
#endif

// number of lanes whose flag is set.  Call from wave-uniform code after a WAVE_END.
#if VDL2_DEVICE_PASS
VDL2_HD int wave_count_flags(const int32_t *flags) { return (int)__popcll(__ballot(flags[VDL2_LANE()] != 0)); }
#else
VDL2_HD int wave_count_flags(const int32_t *flags) { int n = 0; for(int l = 0; l < 64; l++) n += flags[l] != 0; return n; }
#endif

// Wave-wide primitives on 64 values that the lanes have left in LDS.  Call from wave-uniform code after a WAVE_END (the device
// versions read the calling lane's own entry and combine over the wavefront in registers; the host versions are plain loops).
#if VDL2_DEVICE_PASS
// a[l] <- sum of a[0..l-1]; returns the total
VDL2_HD uint32_t wave_excl_scan64(uint32_t *a) {
	const int l = VDL2_LANE();
	const uint32_t v = a[l];
	uint32_t inc = v;
	for(int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if(l >= d) inc += o; }
	a[l] = inc - v;
	const uint32_t tot = __shfl(inc, 63);
	WAVE_SYNC();
	return tot;
}
// smallest a[l] over the wavefront
VDL2_HD uint32_t wave_min64(const uint32_t *a) {
	uint32_t v = a[VDL2_LANE()];
	for(int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_xor(v, d); v = o < v ? o : v; }
	return v;
}
#else
VDL2_HD uint32_t wave_excl_scan64(uint32_t *a) {
	uint32_t acc = 0;
	for(int l = 0; l < 64; l++) { const uint32_t v = a[l]; a[l] = acc; acc += v; }
	return acc;
}
VDL2_HD uint32_t wave_min64(const uint32_t *a) {
	uint32_t m = a[0];
	for(int l = 1; l < 64; l++) if(a[l] < m) m = a[l];
	return m;
}
#endif

struct StaleRef { float veps[320]; float sp[66], sf[66], slo[66], shi[66], sE[66]; int32_t smarg[64]; };
// absolute index of the DM_INIT sample d steps before n (n inside the current interval); -1 = before the stream
VDL2_HD int64_t seq_index(const WalkState &st, int64_t n, int d) {
	if((int64_t)d <= n - st.a) return n - d;
	int64_t r = d - (n - st.a) - 1;
	for(int i = 0; i < st.niv; i++) {
		int64_t len = st.ivb[i] - st.iva[i] + 1;
		if(r < len) return st.ivb[i] - r;
		r -= len;
	}
	return -1;
}

VDL2_HD void push_interval(WalkState &st, int64_t a, int64_t b) {
	int n = st.niv < kNumIv ? st.niv : kNumIv - 1;
	for(int i = n; i > 0; i--) { st.iva[i] = st.iva[i - 1]; st.ivb[i] = st.ivb[i - 1]; }
	st.iva[0] = a; st.ivb[0] = b;
	if(st.niv < kNumIv) st.niv++;
}

// `count` evaluations starting at `first` (step 3) are being executed: note them for the noise-floor
// kernel (demod.c:238-243 is replayed there, off the walker's critical path).  One lane (call inside a LANE0 section).
VDL2_HD void log_evals_lane0(WalkShared &sh, const EvalLog &lg, OutCtl *ctl, int64_t first, int64_t count) {
	if(count <= 0) return;
	// the open chunk lives in LDS; a chunk is written out (a plain store, nothing waits for it) only when
	// the next one cannot be merged into it
	const int64_t lc = sh.lg_count, lf = sh.lg_first;
	if(lc > 0 && lf + 3 * lc == first) sh.lg_count = lc + count;
	else {
		if(lc > 0) {
			const uint32_t ln = sh.lg_n;
			if(ln < sh.cap_log) { lg.chunks[ln].first = lf; lg.chunks[ln].count = lc; sh.lg_n = ln + 1; }
			else ctl->overflow = 1;  // pathological storm of grid shifts: noise floor becomes approximate
		}
		sh.lg_first = first; sh.lg_count = count;
	}
	sh.st.evals += count;
}
VDL2_HD void log_evals(WalkShared &sh, const EvalLog &lg, OutCtl *ctl, int64_t first, int64_t count) {
	if(count <= 0) return;
	LANE0
		log_evals_lane0(sh, lg, ctl, first, count);
	LANE0_END
}

// demod_reset() after a burst or a rejected header: a new DM_INIT interval starts at sample `a`
VDL2_HD void restart_search(WalkState &st, int64_t a) {
	st.a = a; st.e = st.e0 = a + 2;
	st.pherr1 = st.pherr2 = kPherrBig;
	st.mode = 0;
}

// Start of a walk: bring the channel's FSM state and the small tables into LDS.  `resume`: continue the burst list of
// this feed (a previous walk_store() left its count in *nbursts_out) instead of starting it.
VDL2_HD void walk_load(const WalkState *gstate, const EvalLog &lg, const uint32_t *nbursts_out, bool resume, const Tables &T, const OutCtl *ctl, WalkShared &sh) {
	LANE0
		sh.st = *gstate;
		sh.cap_log = ctl->cap_log;
		sh.lg_n = resume ? *lg.n : 0; sh.lg_first = 0; sh.lg_count = 0;
		sh.spec_n = -1; sh.vring_a = -1; sh.nb = resume ? *nbursts_out : 0;
		sh.first_fire = INT64_MAX;
		sh.cw0 = 0; sh.cw_end = 0; sh.wbase = 0; sh.u_fire = 0; sh.u_n = 0;
		sh.x_n = -1; sh.x_hdr = -1; sh.u_verr = 0.f; sh.x_ok = 0; sh.x_lo = 0; sh.x_hi = -1; sh.xs_a = -1; sh.xs_hi = -1;
	LANE0_END
	WAVE_FOR(l)
		if(l < kHdrParBits) sh.t_H[l] = T.hdr_H[l];
		if(l < 32) { sh.t_fix[l] = T.hdr_fix[l]; sh.t_prbs[l] = T.prbs[l]; }
		if(l < 8) sh.t_gray[l] = T.gray[l];
	WAVE_END
}

// the open evaluation chunk goes to the log
VDL2_HD void walk_flush_log(WalkShared &sh, const EvalLog &lg, OutCtl *ctl) {
	LANE0
		if(sh.lg_count > 0) {
			if(sh.lg_n < sh.cap_log) { lg.chunks[sh.lg_n].first = sh.lg_first; lg.chunks[sh.lg_n].count = sh.lg_count; sh.lg_n++; }
			else ctl->overflow = 1;
			sh.lg_count = 0;
		}
	LANE0_END
}

VDL2_HD void walk_store(WalkShared &sh, WalkState *gstate, const EvalLog &lg, OutCtl *ctl, uint32_t *nbursts_out) {
	walk_flush_log(sh, lg, ctl);
	LANE0
		*gstate = sh.st;
		*lg.n = sh.lg_n;
		*nbursts_out = sh.nb;
	LANE0_END
}

// search state that depends on nothing but the sample streams and the evaluation grid (sh.st.e modulo 3)
VDL2_HD bool walk_clean(const WalkState &st) { return st.mode == 0 && st.e >= st.a + kCleanAfter && st.e >= st.e0 + 6; }

// Advance the FSM held in sh.st up to (not including) decimated sample k_end.  With `stop_clean` the walk also stops as
// soon as the state is "clean" (walk_clean()).  Stopping anywhere is exact: it is what a feed boundary does.
// `spec`: a speculative walk (spec_walk()) - it does not call the referee: at a decision within the margin it gives up (its result is
// marked unusable, the stitcher walks that segment for real), so that a scan is run once, by the walk that counts, not by three hypotheses.
// (always inlined: as a function of its own on the device - it has grown past the inliner's patience - the walker kernels fault)
VDL2_HD __attribute__((always_inline)) void walk_run(int chan, uint32_t freq, float max_ppm, float ppm_thr, int64_t k_end, bool stop_clean, const Tables &T,
		const ChanView &v, unsigned long long *cnt, Burst *bursts, uint32_t cap_bursts, OutCtl *ctl, const EvalLog &lg, WalkShared &sh, bool spec = false) {
	K4_BEGIN();
	LANE0
		// a speculative gather left by an earlier call was bounded by that call's k_end (phases beyond it read as zero): the
		// stitcher calls this function several times per feed with growing k_end
		sh.spec_n = -1;
	LANE0_END
	K4_MARK(0);
	for(;;) {
		if(sh.st.mode == 0) {
			const int64_t e = sh.st.e;
			// evaluations run up to k_lim; sample reads stay bounded by k_end (what has been written)
			int64_t k_lim = k_end;
			if(stop_clean) {
				int64_t c = sh.st.a + kCleanAfter; if(c < sh.st.e0 + 6) c = sh.st.e0 + 6;
				if(e >= c) break;
				if(c < k_lim) k_lim = c;
			}
			if(e >= k_lim) break;
			int fired = 0;
			int64_t fire_n = 0;
			if(e < sh.st.a + kFreshAfter) {
				// ---- explicit evaluations near an interval start (ring still holds pre-burst samples) ----
				int64_t lim = sh.st.a + kFreshAfter; if(lim > k_lim) lim = k_lim;
				int64_t nb = (lim - e + 2) / 3; if(nb > 64) nb = 64;
				// stage the reference's phase ring as it stands around this interval start: vring[160+t] = sample a+t,
				// vring[159-r] = the (r+1)-th DM_INIT sample before a (through the interval history)
				const int64_t a0 = sh.st.a;
				// Referee: these evaluations are not tabulated (K3 knows contiguous windows only), so their margins are worked out here,
				// from the phase bounds of the staged samples; when the first evaluation that may fire is within its margin the ring is
				// made the reference's own (the stretches it reads, through the interval history) and the pass is done again
				const int64_t fwd_hi = (a0 + 159 < k_end - 1) ? a0 + 159 : k_end - 1;
				const bool ref_here = v.ref != nullptr;
				const bool exact_ring = ref_here && sh.xs_a == a0 && sh.xs_hi >= fwd_hi;      // (made the reference's own by an earlier turn of this loop)
				StaleRef &sr = *reinterpret_cast<StaleRef *>(sh.cw);
				static_assert(sizeof(StaleRef) <= sizeof(sh.cw), "the referee's scratch of a pass near an interval start fits the candidate-word cache");
				if(ref_here) {
					LANE0
						sh.cw0 = 0; sh.cw_end = 0;                 // (the cached words are gone)
					LANE0_END
				}
				WAVE_FOR(l)
					cf32 yv[5]; bool ok[5]; int64_t nn[5];
					for(int q = 0; q < 5; q++) {                  // the five loads of a lane first, then the five phases
						const int j = l + 64 * q;
						int64_t n;
						if(j >= 160) { n = a0 + (j - 160); ok[q] = n < k_end; }
						else { n = seq_index(sh.st, a0, 160 - j); ok[q] = n >= 0; }
						nn[q] = n;
						yv[q] = ok[q] ? v.Y(n) : cf32{0.f, 0.f};
					}
					for(int q = 0; q < 5; q++) sh.vring[l + 64 * q] = ok[q] ? phase_of(yv[q]) : 0.f;
					if(ref_here && !exact_ring) for(int q = 0; q < 5; q++) sr.veps[l + 64 * q] = ok[q] ? ref_eps2(v, nn[q]) : 0.f;
					if(l == 0) sh.vring_a = a0;
				WAVE_END
				if(!ref_here) {
					WAVE_FOR(l)
						if(l < nb) {
							const int t = (int)(e + 3 * l - a0);
							float ph[kPreamble];
							for(int i = 0; i < kPreamble; i++) ph[i] = sh.vring[160 + t - 150 + 10 * i];
							sync_metric(ph, T, sh.p[l], sh.f[l]);
						}
					WAVE_END
					WAVE_FOR(l)
						const float pm1 = l ? sh.p[l - 1] : sh.st.pherr1;
						sh.flag[l] = (l < nb && pm1 < kSyncThr && sh.p[l] > pm1) ? 1 : 0;
					WAVE_END
				} else {
					// evaluations e - 6, e - 3 (as far as they belong to this run: else PHERR_MAX, exactly) and the nb of the pass
					const int64_t e0r = sh.st.e0;
					for(int base = 0; base < (int)nb + 2; base += 64) {
						WAVE_FOR(l)
							const int i = base + l;
							if(i < (int)nb + 2) {
								const int64_t pos = e + 3 * (int64_t)(i - 2);
								if(pos >= e0r && pos >= a0) {
									const int t = (int)(pos - a0);
									float ph[kPreamble], e2[kPreamble];
									for(int k = 0; k < kPreamble; k++) { ph[k] = sh.vring[160 + t - 150 + 10 * k]; e2[k] = exact_ring ? 0.f : sr.veps[160 + t - 150 + 10 * k]; }
									float pv, fv, E = 0.f, alo, ahi;
									RefRange r;
									if(exact_ring) { sync_metric(ph, T, pv, fv); r = RefRange{ pv, pv }; }
									else { sync_metric_ref(ph, e2, 1, T, pv, fv, E, alo, ahi, v.ref_pre); r = ref_pherr_range(pv, alo, ahi, E); }
									sr.sp[i] = pv; sr.sf[i] = fv; sr.sE[i] = E; sr.slo[i] = r.lo; sr.shi[i] = r.hi;
								} else { sr.sp[i] = kPherrBig; sr.sf[i] = 0.f; sr.sE[i] = 0.f; sr.slo[i] = kPherrBig; sr.shi[i] = kPherrBig; }
							}
						WAVE_END
					}
					WAVE_FOR(l)
						int fl = 0, mg = 0;
						if(l < nb) {
							const int i = l + 2;
							if(exact_ring) fl = (sr.sp[i - 1] < kSyncThr && sr.sp[i] > sr.sp[i - 1]) ? 1 : 0;
							else {
								const int vd = ref_candidate_verdict(RefRange{ sr.slo[i], sr.shi[i] }, RefRange{ sr.slo[i - 1], sr.shi[i - 1] }, sr.sf[i - 1], sr.sE[i - 1],
								                                     RefRange{ sr.slo[i - 2], sr.shi[i - 2] }, max_ppm, ppm_thr);
								fl = vd & 1; mg = (vd >> 1) & 1;
							}
							sh.p[l] = sr.sp[i]; sh.f[l] = sr.sf[i];
						}
						sh.flag[l] = fl; sr.smarg[l] = mg;
					WAVE_END
					LANE0
						// (pherr[1], pherr[2] and prev_dphi as this run's evaluations before the pass left them: the same values by definition,
						// worked out on the ring as it is now)
						sh.st.pherr1 = sr.sp[1]; sh.st.pherr2 = sr.sp[0];
						if(sr.sp[1] < kPherrBig) sh.st.prev_dphi = sr.sf[1];
					LANE0_END
					const int jm = wave_first_flag(sh.flag);
					if(jm >= 0 && sr.smarg[jm]) {
						// the first evaluation that may fire hangs on the reference's rounding
#if !VDL2_DEVICE_PASS && defined(VDL2_HOST_DEBUG)
						if(getenv("HOSTSIM_DEBUG_STALE")) fprintf(stderr, "stale chan %d a0 %lld e %lld jm %d spec %d: r6=[%g,%g] r3=[%g,%g] r0=[%g,%g] f3=%g E3=%g E0=%g\n", chan, (long long)a0, (long long)e, jm, (int)spec,
							sr.slo[jm], sr.shi[jm], sr.slo[jm + 1], sr.shi[jm + 1], sr.slo[jm + 2], sr.shi[jm + 2], sr.sf[jm + 1], sr.sE[jm + 1], sr.sE[jm + 2]);
#endif
						if(spec) { LANE0 ctl->overflow = 1; LANE0_END break; }
						bool ok = false;
						// the stretches the ring reads: the current interval's part, then the history's (runs of DM_INIT samples between bursts);
						// optimistic mode: they are noted, the channel is flagged - it is walked again, asking on the spot, and finds them done -
						// and this walk goes on with the samples as they are
						const bool note = v.rq != nullptr;
						if(note) {
							LANE0
								v.rq_flag[chan] = 1u;
							LANE0_END
						}
						{
							int64_t wl[8], wh[8]; int nw = 1;
							wl[0] = a0; wh[0] = fwd_hi;
							int64_t run_hi = -1, run_lo = -1;
							for(int r = 1; r <= 160; r++) {
								const int64_t n = seq_index(sh.st, a0, r);
								if(n < 0) break;
								if(run_hi < 0) { run_hi = n; run_lo = n; }
								else if(n >= run_lo - 512) run_lo = n;                    // (the same stretch, or near enough to be one scan)
								else { if(nw < 8) { wl[nw] = run_lo; wh[nw] = run_hi; nw++; } run_hi = n; run_lo = n; }
							}
							if(run_hi >= 0 && nw < 8) { wl[nw] = run_lo; wh[nw] = run_hi; nw++; }
							ok = !note;
							for(int i = 0; i < nw; i++) {
								if(note) {
									LANE0
										RefReq rq{};
										rq.chan = chan; rq.kind = REF_STALE; rq.n = wl[i]; rq.t_first = wh[i];
										(void)ref_log_request(v, rq);
									LANE0_END
								} else if(ok) ok = ref_exact_window(v, wl[i], wh[i], sh.cw, REF_CANDIDATE);
							}
							if(!note) {
								LANE0
									sh.cw0 = 0; sh.cw_end = 0;
									if(ok) { sh.xs_a = a0; sh.xs_hi = fwd_hi; }
								LANE0_END
							}
						}
						if(ok) continue;                            // the same pass again, on the reference's own samples
						// (not made exact: the plain test on the values as they are)
						WAVE_FOR(l)
							const float pm1 = l ? sh.p[l - 1] : sr.sp[1];
							sh.flag[l] = (l < nb && pm1 < kSyncThr && sh.p[l] > pm1) ? 1 : 0;
						WAVE_END
					}
				}
				const int jf = wave_first_flag(sh.flag);
				const int64_t nexec = jf >= 0 ? jf + 1 : nb;
				LANE0
					WalkState &st = sh.st;
					if(jf >= 0) {
						sh.u_y3 = sh.p[jf];
						sh.u_y2 = jf >= 1 ? sh.p[jf - 1] : st.pherr1;
						sh.u_y1 = jf >= 2 ? sh.p[jf - 2] : (jf == 1 ? st.pherr1 : st.pherr2);
						sh.u_prevd = jf >= 1 ? sh.f[jf - 1] : st.prev_dphi;
						// (taps through the interval history: the burst decoder cannot redo this slope - 0 when the ring is the reference's own,
						// else minus the bound)
						sh.u_verr = exact_ring ? 0.f : (ref_here ? -ref_slope_margin(sr.sE[jf + 1]) : -1.f);
					} else {
						const float o1 = st.pherr1;
						st.pherr1 = sh.p[nb - 1];
						st.pherr2 = nb >= 2 ? sh.p[nb - 2] : o1;
						st.prev_dphi = sh.f[nb - 1];
						st.e = e + 3 * nb;
					}
				LANE0_END
				K4_MARK(1);
				log_evals(sh, lg, ctl, e, nexec);
				K4_MARK(2);
				fired = jf >= 0;
				fire_n = e + 3 * (int64_t)(jf >= 0 ? jf : 0);
			} else {
				// ---- hop over the candidate bitmap (got_sync() can only fire where a bit is set) ----
				// the first evaluation of a run cannot fire (pherr[1] is still PHERR_MAX): start at max(e, e0+3)
				const int64_t start = e > sh.st.e0 + 3 ? e : sh.st.e0 + 3;
				int64_t w0 = start >> 6;
				const int64_t wend = (k_lim + 63) >> 6;
				while(w0 < wend && !fired) {
					// the words of the previous pass are still in LDS: a search that starts among them costs no trip to memory
					constexpr int kWpl = kCandWin / 64;                  // words per lane and pass (consecutive: the lowest lane with a hit has the first hit)
					static_assert(kWpl == 8, "a lane's words are one byte of the non-zero map");
					const bool cached = w0 >= sh.cw0 && w0 < sh.cw_end;
					const int64_t wb = cached ? sh.cw0 : w0;
					const int64_t wlast = cached ? sh.cw_end : (wb + kCandWin < wend ? wb + kCandWin : wend);
					WAVE_FOR(l)
						int32_t hit = -1;
						uint64_t wd[kWpl];
						if(cached) { for(int q = 0; q < kWpl; q++) wd[q] = wb + kWpl * l + q < wlast ? sh.cw[kWpl * l + q] : 0ull; }
						else {
							for(int q = 0; q < kWpl; q++) { const int64_t w = wb + kWpl * l + q; wd[q] = w < wlast ? v.Cand(w) : 0ull; }
							for(int q = 0; q < kWpl; q++) sh.cw[kWpl * l + q] = wd[q];
							uint32_t nz = 0;
							for(int q = 0; q < kWpl; q++) nz |= (wd[q] != 0ull ? 1u : 0u) << q;
							reinterpret_cast<uint8_t *>(sh.nzw)[l] = (uint8_t)nz;
						}
						for(int q = 0; q < kWpl && hit < 0; q++) {
							const int64_t w = wb + kWpl * l + q;
							uint64_t bits = wd[q];
							if(bits) {
								// keep bits with index >= start, < k_lim, and congruent to e modulo 3
								const int64_t base = w << 6;
								const int r = (((int)(e - base)) % 3 + 3) % 3;   // first bit position on the grid (|e - base| << 2^31)
								bits &= 0x9249249249249249ull << r;                // bits r, r+3, ...
								if(base < start) bits &= (start - base >= 64) ? 0ull : (~0ull << (start - base));
								if(base + 64 > k_lim) bits &= (k_lim - base <= 0) ? 0ull : (~0ull >> (64 - (k_lim - base)));
								if(bits) hit = 64 * q + __builtin_ctzll(bits);
							}
						}
						sh.found[l] = hit;
						sh.flag[l] = hit >= 0;
					WAVE_END
					if(!cached) {
						LANE0
							sh.cw0 = wb; sh.cw_end = wlast;
						LANE0_END
					}
					const int lf = wave_first_flag(sh.flag);
					if(lf >= 0) { fired = 1; fire_n = ((wb + kWpl * lf) << 6) + sh.found[lf]; }
					else w0 = wlast;
				}
				K4_MARK(3);
				if(fired) {
					// The metric values around the fire in one load: y1, y2, y3 of calc_para_vertex() and the slope of the evaluation
					// before it are all the --max-ppm gate needs, and a fire the gate drops needs nothing else.  One lane then takes
					// this fire and the ones that follow it inside the window (a cluster of candidate bits around one preamble) until
					// one passes the gate - only that one needs phases.
					const int64_t wbase0 = fire_n - 6;
					WAVE_FOR(l)
						const cf32 pv = v.PF(wbase0 + l);
						sh.wre[l] = pv.re; sh.wim[l] = pv.im;
					WAVE_END
					K4_MARK(8);
					LANE0
						// (the search state and the open chunk of the evaluation log are held in registers while this lane works through the
						// fires: every access to them in LDS would be a wait of its own)
						WalkState &st = sh.st;
						int64_t n = fire_n, e_cur = e, kl = k_lim;
						int64_t wbase = wbase0, whi = wbase0 + 63;        // samples whose metric values are in sh.wre / sh.wim (a candidate n needs n - 6 .. n)
						const int64_t cw0 = sh.cw0, ncw = sh.cw_end - sh.cw0, clean_at = st.a + kCleanAfter;
						int64_t e0 = st.e0, evals = st.evals, lgf = sh.lg_first, lgc = sh.lg_count;
						uint32_t lgn = sh.lg_n, nrej = 0;
						const uint32_t cap_log = sh.cap_log;
						int pending = 0; bool moved = false;
						if(sh.first_fire == INT64_MAX) sh.first_fire = n;
						for(;;) {
							// Referee: the sign of the tabulated pherr marks a candidate some decision of which - the candidate test itself, the
							// vertex, the gate - is within the margin of the stream's error (K3's exact tier: ref_candidate_verdict()); it is
							// redone on the reference's own samples before anything is decided (below: sh.u_fire == 2), and comes by here again
							const bool exact_here = sh.x_n == n;
							const bool marked = v.ref && !exact_here && sh.wre[n - wbase] < 0.f;
							if(marked && (!v.rq || v.ref_pre)) { pending = 2; break; }
#ifdef VDL2_REF_DEBUG
							ref_debug_log(v, exact_here ? 2 : 1, n, sh.wre[n - wbase], exact_here ? sh.x_p3 : sh.wre[n - 3 - wbase], (float)(e0 % 3));
#endif
							float y1, y2, y3, prevd;
							if(exact_here) { y1 = (n - 6 >= e0) ? sh.x_p6 : kPherrBig; y2 = sh.x_p3; y3 = sh.x_p0; prevd = sh.x_f3; }
							else { y1 = (n - 6 >= e0) ? fabsf(sh.wre[n - 6 - wbase]) : kPherrBig; y2 = fabsf(sh.wre[n - 3 - wbase]); y3 = fabsf(sh.wre[n - wbase]); prevd = sh.wim[n - 3 - wbase]; }
							{   // log_evals_lane0() on the register copies: evaluations e_cur, e_cur + 3, ..., n
								const int64_t count = (n - e_cur) / 3 + 1;
								if(lgc > 0 && lgf + 3 * lgc == e_cur) lgc += count;
								else {
									if(lgc > 0) {
										if(lgn < cap_log) { lg.chunks[lgn].first = lgf; lg.chunks[lgn].count = lgc; lgn++; }
										else ctl->overflow = 1;
									}
									lgf = e_cur; lgc = count;
								}
								evals += count;
							}
							moved = true;
							int64_t e2;
							if(marked) {
								// optimistic mode: decided on the samples as they are, noted for the check on the reference's own
								RefReq rq{};
								rq.chan = chan; rq.kind = REF_CANDIDATE; rq.n = n;
								rq.code = ref_candidate_code(y1, y2, y3, prevd, max_ppm, ppm_thr) | ((n - 6 >= e0) ? 0u : 0x10000u);
								if(!ref_log_request(v, rq)) { if(v.rq_flag) { v.rq_flag[chan] = 1u; if(v.rq_bad) v.rq_bad->n = (uint32_t)kRefBad + 1u; } else ctl->overflow = 1; }
							}
							if((exact_here || marked) && !is_candidate(y2, y3)) {
								// on these samples the evaluation does not fire (the bitmap holds "may fire"): the run goes on
								e2 = n + 3; e_cur = e2;
							} else {
								if(!(max_ppm != 0.f && fabsf(prevd) > ppm_thr)) {      // = fabsf(ppm_of(prevd, freq)) > max_ppm (ppm_gate_threshold())
									sh.u_y1 = y1; sh.u_y2 = y2; sh.u_y3 = y3; sh.u_prevd = prevd;
									sh.u_verr = (exact_here && sh.x_ok) ? 0.f : 1.f;      // 1: bound worked out from the taps of n - 3 (below)
									pending = 1;
									break;
								}
								// demod.c:190-192: dropped by the gate; v->sclk keeps the vertex value, which shifts the evaluation grid (demod.c:179,233)
								const int sclk = (int)(-roundf(parabola_vertex(y1, y2, y3)));
								nrej++;
								int64_t step = 3 - sclk; if(step < 1) step = 1;
								e2 = n + step;
								e0 = e2; e_cur = e2;
							}
							// the next candidate on the (new) grid, if the words in LDS reach it
							kl = k_end;
							if(stop_clean) { int64_t c = clean_at; if(c < e0 + 6) c = e0 + 6; if(c < kl) kl = c; }
							const int64_t s2 = e2 > e0 + 3 ? e2 : e0 + 3;      // the first evaluation of a run cannot fire
							int64_t n2 = -1;
							for(int64_t i = (s2 >> 6) - cw0; i >= 0 && i < ncw; ) {
								const uint64_t nzm = sh.nzw[i >> 6] >> (i & 63);       // non-zero words from word i to the end of its group of 64
								if(!nzm) { i = (i | 63) + 1; continue; }
								i += __builtin_ctzll(nzm);
								if(i >= ncw) break;
								const int64_t base = (cw0 + i) << 6;
								if(base >= kl) break;
								uint64_t bits = sh.cw[i];
								const int r = (((int)(e2 - base)) % 3 + 3) % 3;
								bits &= 0x9249249249249249ull << r;
								if(base < s2) bits &= (s2 - base >= 64) ? 0ull : (~0ull << (s2 - base));
								if(base + 64 > kl) bits &= (kl - base <= 0) ? 0ull : (~0ull >> (64 - (kl - base)));
								if(bits) { n2 = base + __builtin_ctzll(bits); break; }
								i++;
							}
							if(n2 < 0) break;
							if(n2 > whi || n2 - 6 < wbase) {
								// beyond the window of metric values, but its word is in LDS: this lane fetches the three values the next fire
								// needs by itself - cheaper than sending the whole wavefront round the search again
								const cf32 p1 = v.PF(n2 - 6), p2 = v.PF(n2 - 3), p3 = v.PF(n2);
								wbase = n2 - 6; whi = n2;
								sh.wre[0] = p1.re; sh.wre[3] = p2.re; sh.wim[3] = p2.im; sh.wre[6] = p3.re;
							}
							n = n2;
						}
						sh.lg_first = lgf; sh.lg_count = lgc; sh.lg_n = lgn; st.evals = evals;
						if(nrej) {
							VDL2_CNT_ADD(cnt, CNT_PPM_REJECT, nrej);
							st.pherr1 = st.pherr2 = kPherrBig;
							st.e0 = e0;
						}
						if(moved && pending != 1) st.e = e_cur;
						sh.u_fire = pending; sh.u_n = n;
					LANE0_END
					if(sh.u_fire == 2) {
						// ---- referee: the candidate at u_n on the reference's own samples ----
						if(spec && !v.ref_pre) { LANE0 ctl->overflow = 1; LANE0_END break; }
						const int64_t n = sh.u_n;
						int64_t lo = (n - kRefPre) & ~255ll, hi = (n + kRefPost) | 255; if(lo < 0) lo = 0; if(hi > k_end - 1) hi = k_end - 1;   // (whole blocks of 256)
						bool ok = sh.x_lo <= lo && hi <= sh.x_hi;
						if(!ok && spec) {
							// a speculative walk does not scan: the stretch has been made exact ahead of the walk, or the walk gives up
							ok = ref_window_done(v, lo, hi);
							if(!ok) { LANE0 ctl->overflow = 1; LANE0_END break; }
							LANE0
								sh.x_lo = lo; sh.x_hi = hi;
							LANE0_END
						} else if(!ok) {
							ok = ref_exact_window(v, lo, hi, sh.cw, REF_CANDIDATE);      // (the word cache is the scratch: reloaded when the search comes by again)
							LANE0
								if(ok) { sh.x_lo = lo; sh.x_hi = hi; }
								sh.cw0 = 0; sh.cw_end = 0;
							LANE0_END
						}
						WAVE_FOR(l)
							if(l < 48) sh.spec[l] = v.Phi(n - 3 * (l >> 4) - 150 + 10 * (l & 15));
						WAVE_END
						WAVE_FOR(l)
							if(l < 3) sync_metric(&sh.spec[16 * l], T, sh.p[l], sh.f[l]);
						WAVE_END
#ifdef VDL2_REF_DEBUG
						LANE0
							ref_debug_log(v, 3, n, sh.p[0], sh.p[1], sh.p[2]);
							ref_debug_log(v, 4, n, sh.spec[0], sh.spec[15], sh.spec[47]);
						LANE0_END
#endif
						LANE0
							sh.spec_n = -1;
							sh.x_n = n; sh.x_p0 = sh.p[0]; sh.x_p3 = sh.p[1]; sh.x_f3 = sh.f[1]; sh.x_p6 = sh.p[2]; sh.x_ok = ok ? 1 : 0;
						LANE0_END
						continue;       // the search resumes at st.e and finds this candidate again, now decided on those values
					}
					fired = sh.u_fire;
					fire_n = sh.u_n;
					K4_MARK(2);
					if(fired) {
						const int64_t n = fire_n;
						// this one synchronises: the four possible sync-point phases (sclk = 2..5) and the nine header-symbol phases for
						// each of them, in one round trip
						WAVE_FOR(l)
							float val = 0.f;
							if(l < 4) val = v.Phi(n - 2 - l);
							else if(l < 40) { const int sc = 2 + (l - 4) / 9, m = (l - 4) % 9; const int64_t t = n + kSpsDec - sc + (int64_t)kSpsDec * m; if(t < k_end) val = v.Phi(t); }
							if(l < 40) sh.spec[l] = val;
							else if(l < 56 && v.ref) sh.him2[l - 40] = ref_eps2(v, n - 3 - 150 + 10 * (l - 40));   // referee: the taps of evaluation n - 3, whose slope becomes the burst's carrier offset
						WAVE_END
						LANE0
							sh.spec_n = n;
							if(v.ref && sh.u_verr == 1.f) {
								float si = 0.f;
								for(int i = 0; i < kPreamble; i++) si += sh.him2[i];
								sh.u_verr = ref_slope_margin(sqrtf(si));
							}
						LANE0_END
					}
				} else {
					// nothing up to k_lim: park just past the last evaluation that exists.  v->pherr[1], pherr[2] and prev_dphi
					// as that evaluation leaves them are computed here (one round trip for the 32 phases): the sync kernel stores
					// metric values only where a preamble is near, and an arbitrary stopping place is not one
					const int64_t cnt_ev = (k_lim - 1 - e) / 3 + 1;     // e < k_lim here
					const int64_t nl = e + 3 * (cnt_ev - 1);
					log_evals(sh, lg, ctl, e, cnt_ev);
					K4_MARK(2);
					WAVE_FOR(l)
						if(l < 32) sh.spec[l] = v.Phi(nl - 3 * (l >> 4) - 150 + 10 * (l & 15));   // one round trip for both windows
					WAVE_END
					WAVE_FOR(l)
						if(l < 2) sync_metric(&sh.spec[16 * l], T, sh.p[l], sh.f[l]);
					WAVE_END
					LANE0
						sh.spec_n = -1;                 // sh.spec no longer holds a fire-path gather
					LANE0_END
					LANE0
						sh.st.pherr1 = sh.p[0];
						sh.st.pherr2 = (nl - 3 >= sh.st.e0) ? sh.p[1] : kPherrBig;
						sh.st.prev_dphi = sh.f[0];
						sh.st.e = nl + 3;
					LANE0_END
					K4_MARK(9);
				}
			}
			if(fired) {
				K4_MARK(4);
				// ---- got_sync() success branch: demod.c:173-193 ----
				LANE0
					WalkState &st = sh.st;
					const int64_t n = fire_n;
					if(sh.first_fire == INT64_MAX) sh.first_fire = n;
					float vx = parabola_vertex(sh.u_y1, sh.u_y2, sh.u_y3);
					int sclk = (int)(-roundf(vx));
					const int64_t prev_n = sclk >= 0 ? seq_index(st, n, sclk) : -2;
					float prev_phi0;
					if(sh.spec_n == n && sclk >= 2 && sclk <= 5) prev_phi0 = sh.spec[sclk - 2];
					else if(sh.vring_a == st.a && sclk >= 0 && 160 + (n - st.a) - sclk >= 0 && 160 + (n - st.a) - sclk < 320) prev_phi0 = sh.vring[160 + (n - st.a) - sclk];
					else prev_phi0 = v.Phi(seq_index(st, n, sclk));
					float vdphi = sh.u_prevd;
					float ppm = ppm_of(vdphi, freq);
					st.pherr1 = st.pherr2 = kPherrBig;
					if(max_ppm != 0.f && fabsf(ppm) > max_ppm) {
						VDL2_CNT_ADD(cnt, CNT_PPM_REJECT, 1);
						int64_t step = 3 - sclk; if(step < 1) step = 1;   // v->sclk keeps the vertex value: demod.c:179,233
						st.e = st.e0 = n + step;
					} else {
						VDL2_CNT_ADD(cnt, CNT_SYNC_GOOD, 1);
						push_interval(st, st.a, n);
						st.pb.chan = chan; st.pb.nsym = 0;
						st.pb.t_first = n + (kSpsDec - sclk);
						st.pb.sync_sample = n; st.pb.end_sample = 0; st.pb.ord = st.bursts++;
						st.pb.prev_phi0 = prev_phi0; st.pb.vdphi = vdphi; st.pb.ppm = ppm; st.pb.nf_upd = st.evals / 1000; st.pb.sync_evals = st.evals;
						st.pb.vdphi_err = v.ref ? sh.u_verr : 0.f; st.pb.prev_n = prev_n;
						st.pb.tl_bits = 0; st.pb.syndrome = 0;
						st.mode = 1;
					}
				LANE0_END
			}
			K4_MARK(5);
		} else if(sh.st.mode == 1) {
			// ---- header: 9 symbols = 27 bits, of which 25 are the header (decode.c:198-258) ----
			const int64_t t8 = sh.st.pb.t_first + 8 * kSpsDec;
			if(t8 >= k_end) break;
			// phases (and, for the referee, magnitudes) of the sync point and the nine symbols, from the stream as it stands now
			const int64_t ns = sh.st.pb.sync_sample;
			WAVE_FOR(l)
				if(l < 10) {
					const int64_t t = l ? sh.st.pb.t_first + (int64_t)(l - 1) * kSpsDec : sh.st.pb.prev_n;
					sh.hph[l] = t < 0 ? (t == -1 ? 0.f : sh.st.pb.prev_phi0) : v.Phi(t);
					sh.him2[l] = !v.ref ? 0.f : t == -2 ? kRefBig : ref_eps2(v, t);
				}
			WAVE_END
			if(v.ref && sh.x_hdr != ns) {
				// referee: a header symbol within the margin of a decision boundary (demod.c:256-264) is sliced on the reference's own samples
				WAVE_FOR(l)
					int fl = 0;
					if(l < 9) {
						const float ev = fabsf(sh.st.pb.vdphi_err);
						fl = ref_symbol_marginal(sh.hph[l + 1], sh.hph[l], sh.st.pb.vdphi, sqrtf(sh.him2[l]) + sqrtf(sh.him2[l + 1]) + ev);
					}
					sh.flag[l] = fl;
				WAVE_END
				const bool hdr_marginal = wave_first_flag(sh.flag) >= 0;
				if(hdr_marginal && v.rq) {
					LANE0
						// optimistic mode: sliced on the samples as they are (below), noted for the check on the reference's own
						int neg_ = 0; uint32_t code = 0;
						for(int l = 0; l < 9; l++) code |= (uint32_t)slice_symbol(sh.hph[l + 1], sh.hph[l], sh.st.pb.vdphi, neg_) << (3 * l);
						RefReq rq{};
						rq.chan = chan; rq.kind = REF_HEADER; rq.n = ns; rq.t_first = sh.st.pb.t_first; rq.prev_n = sh.st.pb.prev_n;
						rq.vdphi = sh.st.pb.vdphi; rq.vdphi_err = sh.st.pb.vdphi_err; rq.prev_phi0 = sh.st.pb.prev_phi0; rq.code = code;
						if(!ref_log_request(v, rq)) { if(v.rq_flag) { v.rq_flag[chan] = 1u; if(v.rq_bad) v.rq_bad->n = (uint32_t)kRefBad + 1u; } else ctl->overflow = 1; }
						sh.x_hdr = ns;
					LANE0_END
				} else if(hdr_marginal) {
					if(spec) { LANE0 ctl->overflow = 1; LANE0_END break; }
					int64_t lo = ns - kRefPre; if(sh.st.pb.prev_n >= 0 && sh.st.pb.prev_n < lo) lo = sh.st.pb.prev_n;
					const bool ok = ref_exact_window(v, lo, t8, sh.cw, REF_HEADER);
					const bool redo_slope = ok && sh.st.pb.vdphi_err > 0.f;          // a fire on a contiguous ring: the slope of evaluation ns - 3 again
					if(redo_slope) {
						WAVE_FOR(l)
							if(l < 16) sh.spec[l] = v.Phi(ns - 3 - 150 + 10 * l);
						WAVE_END
					}
					LANE0
						sh.cw0 = 0; sh.cw_end = 0; sh.spec_n = -1; sh.x_hdr = ns;
						if(redo_slope) {
							float p_, f_;
							sync_metric(sh.spec, T, p_, f_);
							sh.st.pb.vdphi = f_; sh.st.pb.ppm = ppm_of(f_, freq); sh.st.pb.vdphi_err = 0.f;
						}
					LANE0_END
					continue;                     // the symbols are read and sliced again
				}
			}
#if !VDL2_DEVICE_PASS && defined(VDL2_HOST_DEBUG)
			if(getenv("HOSTSIM_DEBUG_HDR")) { fprintf(stderr, "hdr chan %d sync %lld t_first %lld vdphi %.9g verr %g x_hdr %lld:", chan, (long long)ns, (long long)sh.st.pb.t_first, sh.st.pb.vdphi, sh.st.pb.vdphi_err, (long long)sh.x_hdr);
				for(int l = 0; l < 9; l++) fprintf(stderr, " [%.6f d=%.2e e=%.2e]", sh.hph[l + 1], ref_symbol_dist(sh.hph[l + 1], sh.hph[l], sh.st.pb.vdphi), sqrtf(sh.him2[l]) + sqrtf(sh.him2[l + 1]));
				fprintf(stderr, "\n"); }
#endif
			WAVE_FOR(l)
				if(l < 9) {
					int neg = 0;
					sh.sym[l] = sh.t_gray[slice_symbol(sh.hph[l + 1], sh.hph[l], sh.st.pb.vdphi, neg)];
					sh.neg[l] = neg;
				}
			WAVE_END
			LANE0
				WalkState &st = sh.st;
				uint32_t hdr = 0;
				for(int b = 0; b < kHdrBits; b++) {
					uint32_t bit = ((uint32_t)sh.sym[b / 3] >> (2 - b % 3)) & 1u;
					bit ^= sh.t_prbs[b];
					hdr |= bit << (kHdrBits - 1 - b);
				}
				Geometry g = header_to_geometry(hdr, sh.t_H, sh.t_fix);
				if(g.syndrome == 0) VDL2_CNT_ADD(cnt, CNT_CRC_GOOD, 1);
				if(g.status != HDR_OK) {
					int negs = 0; for(int i = 0; i < 9; i++) negs += sh.neg[i];
					VDL2_CNT_ADD(cnt, CNT_SLICER_NEG_IDX, negs);
					VDL2_CNT_ADD(cnt, g.status == HDR_CRC_BAD ? CNT_CRC_BAD : g.status == HDR_TOO_LONG ? CNT_ERR_TOO_LONG : CNT_ERR_NO_FEC, 1);
					restart_search(st, t8 + 1);
				} else {
					st.pb.tl_bits = g.tl_bits; st.pb.syndrome = g.syndrome | ((uint32_t)popc32(sh.t_fix[g.syndrome]) << 8);   // + synd_weight[] (decode.c:98-100) = bits the pattern flips
					st.pb.nsym = (int32_t)((g.want_bits + kHdrBits + 2) / 3);
					st.pb.end_sample = st.pb.t_first + (int64_t)(st.pb.nsym - 1) * kSpsDec;
					st.mode = 2;
				}
			LANE0_END
			K4_MARK(6);
		} else {
			// ---- burst body: wait until its last symbol has arrived, then hand it to the burst decoder ----
			if(sh.st.pb.end_sample >= k_end) break;
			LANE0
				WalkState &st = sh.st;
				if(sh.nb < cap_bursts) bursts[sh.nb++] = st.pb; else ctl->overflow = 1;
				restart_search(st, st.pb.end_sample + 1);
			LANE0_END
		}
	}
	K4_MARK(7);
	K4_END();
}

// Process one channel up to (not including) decimated sample k_end.
// what a channel's walk of one feed starts from, kept so that the walk can be done again (optimistic mode: ref_verify / walk_again)
struct WalkSnap { WalkState *ws; unsigned long long *cnt; };
VDL2_HD void walk_snapshot(const WalkSnap &snap, int chan, const WalkState *gstate, const unsigned long long *cnt) {
	if(!snap.ws) return;
	LANE0
		snap.ws[chan] = *gstate;
	LANE0_END
	WAVE_FOR(l)
		if(l < kNumCounters) snap.cnt[(size_t)chan * kNumCounters + l] = cnt[l];
	WAVE_END
}
VDL2_HD __attribute__((always_inline)) void walk_channel(int chan, uint32_t freq, float max_ppm, float ppm_thr, int64_t k_end, const Tables &T,
		const ChanView &v, WalkState *gstate, unsigned long long *cnt, Burst *bursts, uint32_t cap_bursts, uint32_t *nbursts_out,
		OutCtl *ctl, const EvalLog &lg, WalkShared &sh, WalkSnap snap = WalkSnap{nullptr, nullptr}) {
	walk_snapshot(snap, chan, gstate, cnt);
	walk_load(gstate, lg, nbursts_out, false, T, ctl, sh);
	walk_run(chan, freq, max_ppm, ppm_thr, k_end, false, T, v, cnt, bursts, cap_bursts, ctl, lg, sh);
	walk_store(sh, gstate, lg, ctl, nbursts_out);
}

// ======================================================================
// Speculative segments.  One channel's feed is cut into segments [b_s, b_s+1).  Segment 0 is walked
// from the channel's real state.  For every later boundary three speculative walkers start in the
// only kind of state a *clean* search can be in there - mode 0, far into its DM_INIT interval, next
// evaluation at b_s + r (r = 0,1,2: the 3-sample evaluation grid) - and record what they do up to
// b_s+1 in private buffers.  The stitcher then goes through the boundaries in order: when the real
// state at b_s is clean, the speculative walk of the matching grid phase IS what the reference's FSM
// does next (every later decision of a clean search depends only on the sample streams and the
// grid), so its output is appended and its end state adopted; when it is not (a burst or a fresh
// interval straddles the boundary) the real walker continues until the state is clean, which is
// normally one burst later, and the speculative walk is joined there if it had not done anything
// yet; otherwise the segment is walked for real.  Nothing is approximated: a speculative result is
// either provably the real one or thrown away.
// ======================================================================
struct SpecHead {                      // what the stitcher needs to decide and chain; staged in LDS for all segments
	int64_t n_first;                   // first got_sync() success of the speculative walk (INT64_MAX: none)
	int64_t a, e, e0, evals, bursts;   // end state (evals, bursts relative to the segment start)
	float   pherr1, pherr2, prev_dphi;
	int32_t mode, niv;
	uint32_t nb, nlog, ok;
	uint32_t nreq, pad_;               // decisions within the margin taken along the way (optimistic mode): SpecOut::req
	EvalChunk c_first, c_last;         // first and last chunk of the evaluation log
};

struct SpecOut {
	SpecHead h;
	WalkState st;
	unsigned long long cnt[kNumCounters];
	OutCtl ctl;
	uint32_t nlog, pad_;
	Burst bursts[kSpecBursts];
	EvalChunk chunks[kSpecLog];
	RefReq req[kSpecReq]; uint32_t nreq, pad2_;
};

struct StitchShared {
	SpecHead head[(kMaxSeg - 1) * 3];
	// accepted speculative segments whose bursts / log chunks / counters are copied at the end
	int32_t  job_src[kMaxSeg]; uint32_t job_nb[kMaxSeg], job_dstb[kMaxSeg], job_logn[kMaxSeg], job_dstlog[kMaxSeg];
	int64_t  job_base_b[kMaxSeg], job_base_e[kMaxSeg];
	int32_t  njobs;
	// parts of the adopted end state that still live in a SpecOut (fetched only if a real walk needs them)
	int32_t  hist_src; int64_t hist_a, hist_sent;
	int32_t  pb_src;   int64_t pb_base_b, pb_base_e;
	int32_t  accepted, walked;           // statistics: segments adopted / walked for real
	int32_t  u_ok;
	uint32_t cap_log;                    // ctl->cap_log, read once
};

// one speculative walk: segment [b, k_end), evaluation grid phase r
VDL2_HD __attribute__((always_inline)) void spec_walk(int chan, uint32_t freq, float max_ppm, float ppm_thr, int64_t b, int r, int64_t k_end, const Tables &T,
		const ChanView &v, SpecOut *o, WalkShared &sh) {
	EvalLog lg{ o->chunks, &o->nlog };
	LANE0
		o->nlog = 0; o->nreq = 0;
		o->ctl.nbursts = o->ctl.nframes = o->ctl.pool_used = o->ctl.overflow = 0;
		o->ctl.cap_bursts = kSpecBursts; o->ctl.cap_frames = 0; o->ctl.cap_pool = 0; o->ctl.cap_log = kSpecLog;
		walk_state_init(o->st);
		o->st.a = b - kSpecBack; o->st.e0 = o->st.a + 2; o->st.e = b + r;
	LANE0_END
	WAVE_FOR(l)
		if(l < kNumCounters) o->cnt[l] = 0;
	WAVE_END
	WAVE_SYNC_GLOBAL();                    // lane 0's counter atomics below must find the zeros the other lanes have just stored
	uint32_t nb_dummy = 0;
	walk_load(&o->st, lg, &nb_dummy, false, T, &o->ctl, sh);
	ChanView vs = v;                       // (optimistic mode: what the walk notes goes to the segment's own list; the stitcher passes it on if it adopts the walk)
	if(v.rq) { vs.rq = o->req; vs.rq_n = &o->nreq; vs.rq_cap = kSpecReq; vs.rq_flag = nullptr; }
	walk_run(chan, freq, max_ppm, ppm_thr, k_end, false, T, vs, o->cnt, o->bursts, kSpecBursts, &o->ctl, lg, sh, true);
	walk_flush_log(sh, lg, &o->ctl);
	LANE0
		const WalkState &st = sh.st;
		o->st = st; o->nlog = sh.lg_n;
		SpecHead h;
		h.n_first = sh.first_fire;
		h.a = st.a; h.e = st.e; h.e0 = st.e0; h.evals = st.evals; h.bursts = st.bursts;
		h.pherr1 = st.pherr1; h.pherr2 = st.pherr2; h.prev_dphi = st.prev_dphi;
		h.mode = st.mode; h.niv = st.niv;
		h.nb = sh.nb; h.nlog = sh.lg_n; h.ok = o->ctl.overflow ? 0u : 1u;
		h.nreq = o->nreq < (uint32_t)kSpecReq ? o->nreq : (uint32_t)kSpecReq; h.pad_ = 0;
		h.c_first.first = h.c_first.count = 0; h.c_last = h.c_first;
		if(sh.lg_n > 0) { h.c_first = o->chunks[0]; h.c_last = o->chunks[sh.lg_n - 1]; }
		o->h = h;
	LANE0_END
}

// adopt speculative segment `idx` (boundary b, grid phase r) if the real state in sh.st allows it
// (again: the channel is stitched a second time because a noted decision did not stand - a walk that noted decisions is adopted
// only if it was adopted the first time, so that they were checked, and none of them is on the channel's list of those that fell)
VDL2_HD bool spec_requests_stand(const SpecOut &o, uint32_t checked, const RefBad *B) {
	if(!B || !checked || B->n > (uint32_t)kRefBad) return false;
	const uint32_t nr = o.nreq < (uint32_t)kSpecReq ? o.nreq : (uint32_t)kSpecReq;
	for(uint32_t k = 0; k < nr; k++)
		for(uint32_t i = 0; i < B->n; i++)
			if(4 * o.req[k].n + o.req[k].kind == B->at[i]) return false;
	return true;
}
VDL2_HD bool stitch_try_accept(int64_t b, int64_t kn, int seg, uint32_t cap_bursts, OutCtl *ctl, const EvalLog &lg, WalkShared &sh, StitchShared &ss, const SpecOut *again_spec = nullptr, const RefBad *bad = nullptr) {
	LANE0
		ss.u_ok = 0;
		WalkState &st = sh.st;
		if(walk_clean(st) && st.e >= b && st.e < kn) {
			const int r = (int)((st.e - b) % 3);
			const int idx = (seg - 1) * 3 + r;
			const SpecHead &H = ss.head[idx];
			const int64_t pre = (st.e - (b + r)) / 3;          // evaluations of the speculative walk that precede the join
			if(H.ok && !(again_spec && H.nreq && !spec_requests_stand(again_spec[idx], H.pad_, bad)) && H.n_first >= st.e && ss.njobs < kMaxSeg && (H.nlog == 0 || H.c_first.count > pre)) {
				const int64_t base_e = st.evals - pre, base_b = st.bursts;
				const int64_t sent_a = b - kSpecBack;
				const int j = ss.njobs++;
				ss.job_src[j] = idx; ss.job_base_b[j] = base_b; ss.job_base_e[j] = base_e;
				// --- evaluation log: first chunk joins the open one, last chunk stays open, the middle is copied later
				ss.job_logn[j] = 0; ss.job_dstlog[j] = sh.lg_n;
				if(H.nlog > 0) {
					EvalChunk c0 = H.c_first; c0.first += 3 * pre; c0.count -= pre;
					if(sh.lg_count > 0 && sh.lg_first + 3 * sh.lg_count == c0.first) { c0.first = sh.lg_first; c0.count += sh.lg_count; }
					else if(sh.lg_count > 0) {
						if(sh.lg_n < ss.cap_log) { lg.chunks[sh.lg_n].first = sh.lg_first; lg.chunks[sh.lg_n].count = sh.lg_count; sh.lg_n++; }
						else ctl->overflow = 1;
					}
					if(H.nlog == 1) { sh.lg_first = c0.first; sh.lg_count = c0.count; }
					else {
						if(sh.lg_n < ss.cap_log) { lg.chunks[sh.lg_n] = c0; sh.lg_n++; } else ctl->overflow = 1;
						uint32_t mid = H.nlog - 2;
						if(sh.lg_n + mid > ss.cap_log) { mid = ss.cap_log > sh.lg_n ? ss.cap_log - sh.lg_n : 0; ctl->overflow = 1; }
						ss.job_logn[j] = mid; ss.job_dstlog[j] = sh.lg_n; sh.lg_n += mid;
						sh.lg_first = H.c_last.first; sh.lg_count = H.c_last.count;
					}
				}
				// --- bursts
				uint32_t nb = H.nb;
				if(sh.nb + nb > cap_bursts) { nb = cap_bursts > sh.nb ? cap_bursts - sh.nb : 0; ctl->overflow = 1; }
				ss.job_nb[j] = nb; ss.job_dstb[j] = sh.nb; sh.nb += nb;
				// --- state
				if(H.niv > 0) { ss.hist_src = idx; ss.hist_a = st.a; ss.hist_sent = sent_a; }
				if(H.a != sent_a) st.a = H.a;
				if(H.e0 != sent_a + 2) st.e0 = H.e0;
				st.e = H.e; st.pherr1 = H.pherr1; st.pherr2 = H.pherr2; st.prev_dphi = H.prev_dphi;
				st.mode = H.mode; st.evals = base_e + H.evals; st.bursts = base_b + H.bursts;
				if(H.mode != 0) { ss.pb_src = idx; ss.pb_base_b = base_b; ss.pb_base_e = base_e; } else ss.pb_src = -1;
				sh.spec_n = -1; sh.vring_a = -1;
				ss.accepted++;
				ss.u_ok = 1;
			}
		}
	LANE0_END
	return ss.u_ok != 0;
}

// before the real walker runs again: fetch the interval history / pending burst an adopted end state left in its SpecOut.
// Only the adopted walk's own intervals are taken: its oldest one is longer than kCleanAfter, and the 160-sample look-back
// of a fresh interval (seq_index()) can never get past an interval that long.
VDL2_HD void stitch_materialize(const SpecOut *spec, WalkShared &sh, StitchShared &ss) {
	if(ss.hist_src >= 0) {
		const SpecOut *o = spec + ss.hist_src;
		const int niv = ss.head[ss.hist_src].niv;
		WAVE_FOR(l)
			if(l < niv && l < kNumIv) {
				const int64_t ia = o->st.iva[l];
				sh.st.iva[l] = ia == ss.hist_sent ? ss.hist_a : ia;
				sh.st.ivb[l] = o->st.ivb[l];
			}
		WAVE_END
		LANE0
			sh.st.niv = niv; ss.hist_src = -1;
		LANE0_END
	}
	if(ss.pb_src >= 0) {
		LANE0
			Burst pb = spec[ss.pb_src].st.pb;
			pb.ord += ss.pb_base_b; pb.sync_evals += ss.pb_base_e; pb.nf_upd = pb.sync_evals / 1000;
			sh.st.pb = pb; ss.pb_src = -1;
		LANE0_END
	}
}

// The feed [k0, k_end) of one channel in nseg segments of seglen samples; spec[(s-1)*3 + r] holds the speculative walks of
// segments 1..nseg-1.  Segment 0 is walked from the real state right here (the walk that may call the referee is this one
// wavefront per channel, in this kernel, and no other).
VDL2_HD __attribute__((always_inline)) void stitch_channel(int chan, uint32_t freq, float max_ppm, float ppm_thr, int64_t k0, int64_t seglen, int nseg, int64_t k_end, const Tables &T,
		const ChanView &v, WalkState *gstate, unsigned long long *cnt, Burst *bursts, uint32_t cap_bursts, uint32_t *nbursts_out,
		OutCtl *ctl, const EvalLog &lg, const SpecOut *spec, WalkShared &sh, StitchShared &ss, uint32_t *seg_stats, WalkSnap snap = WalkSnap{nullptr, nullptr}, bool again = false) {
	// `again` (optimistic mode, after the check): a channel one of whose noted decisions did not stand - its state and counters go
	// back to what the feed started from (snap) and the feed is stitched once more, the referee asked on the spot (v.rq == nullptr); a
	// speculative walk that noted decisions of its own is not adopted (it is walked for real: what it asks for has been made exact by the
	// check), the others still are - they took no decision within the margin
	if(again) {
		LANE0
			*gstate = snap.ws[chan];
		LANE0_END
		WAVE_FOR(l)
			if(l < kNumCounters) cnt[l] = snap.cnt[(size_t)chan * kNumCounters + l];
		WAVE_END
		WAVE_SYNC_GLOBAL();
	} else
	walk_snapshot(snap, chan, gstate, cnt);
	walk_load(gstate, lg, nbursts_out, false, T, ctl, sh);
	walk_run(chan, freq, max_ppm, ppm_thr, k0 + seglen < k_end ? k0 + seglen : k_end, false, T, v, cnt, bursts, cap_bursts, ctl, lg, sh);
	LANE0
		ss.njobs = 0; ss.hist_src = -1; ss.pb_src = -1; ss.accepted = 0; ss.walked = 0; ss.cap_log = ctl->cap_log;
	LANE0_END
	const int nspec = (nseg - 1) * 3;
	WAVE_FOR(l)
		for(int i = l; i < nspec; i += 64) ss.head[i] = spec[i].h;
	WAVE_END
	for(int s = 1; s < nseg; s++) {
		const int64_t b = k0 + (int64_t)s * seglen;
		const int64_t kn = s + 1 < nseg ? b + seglen : k_end;
		if(stitch_try_accept(b, kn, s, cap_bursts, ctl, lg, sh, ss, again ? spec : nullptr, v.rq_bad)) continue;
		stitch_materialize(spec, sh, ss);
		walk_run(chan, freq, max_ppm, ppm_thr, kn, true, T, v, cnt, bursts, cap_bursts, ctl, lg, sh);
		if(stitch_try_accept(b, kn, s, cap_bursts, ctl, lg, sh, ss, again ? spec : nullptr, v.rq_bad)) continue;
		walk_run(chan, freq, max_ppm, ppm_thr, kn, false, T, v, cnt, bursts, cap_bursts, ctl, lg, sh);
		LANE0
			ss.walked++;
		LANE0_END
	}
	stitch_materialize(spec, sh, ss);
	// copy what the adopted segments produced
	const int nj = ss.njobs;
	WAVE_FOR(l)
		// bursts: flattened over the jobs
		uint32_t tot = 0;
		for(int j = 0; j < nj; j++) tot += ss.job_nb[j];
		for(uint32_t t = (uint32_t)l; t < tot; t += 64) {
			uint32_t off = t; int j = 0;
			while(off >= ss.job_nb[j]) { off -= ss.job_nb[j]; j++; }
			Burst x = spec[ss.job_src[j]].bursts[off];
			x.ord += ss.job_base_b[j]; x.sync_evals += ss.job_base_e[j]; x.nf_upd = x.sync_evals / 1000;
			bursts[ss.job_dstb[j] + off] = x;
		}
		uint32_t totl = 0;
		for(int j = 0; j < nj; j++) totl += ss.job_logn[j];
		for(uint32_t t = (uint32_t)l; t < totl; t += 64) {
			uint32_t off = t; int j = 0;
			while(off >= ss.job_logn[j]) { off -= ss.job_logn[j]; j++; }
			lg.chunks[ss.job_dstlog[j] + off] = spec[ss.job_src[j]].chunks[1 + off];
		}
		if(l < kNumCounters) {
			unsigned long long acc = 0;
			for(int j = 0; j < nj; j++) acc += spec[ss.job_src[j]].cnt[l];
			if(acc) VDL2_CNT_ADD(cnt, l, acc);
		}
		if(v.rq && l < kSpecReq) {
			// the decisions within the margin the adopted walks took: onto the feed's list, to be checked
			for(int j = 0; j < nj; j++)
				if((uint32_t)l < ss.head[ss.job_src[j]].nreq && !ref_log_request(v, spec[ss.job_src[j]].req[l])) { v.rq_flag[chan] = 1u; if(v.rq_bad) v.rq_bad->n = (uint32_t)kRefBad + 1u; }
		}
	WAVE_END
	walk_store(sh, gstate, lg, ctl, nbursts_out);
	LANE0
		// (an adopted walk's noted decisions are on the feed's list now: if the channel is stitched again, the walk may be adopted again
		// provided they all stood)
		if(v.rq) for(int j = 0; j < nj; j++) if(ss.head[ss.job_src[j]].nreq) const_cast<SpecOut *>(spec)[ss.job_src[j]].h.pad_ = 1u;
		if(seg_stats) { seg_stats[0] += (uint32_t)ss.accepted; seg_stats[1] += (uint32_t)ss.walked; }
	LANE0_END
}

// ======================================================================
// Referee, optimistic mode: the check.  One wavefront per noted decision: the stretch the decision reads becomes the reference's
// own (ref_exact_window), the decision is taken again on it - the reference's - and compared with what the walk did.  They agree
// all but once in a hundred (the margin is wide, the stream's error small): then nothing else happens; otherwise the channel is
// walked again from the feed's start (walk_again), in the synchronous mode, over samples that are by then exact where it matters.
// `lds`: >= 64 floats.  true: the decision stands (or cannot be checked: the raw input is no longer held).
// ======================================================================
// the stretch a noted decision reads
VDL2_HD void ref_request_window(const RefReq &r, int64_t k_end, int64_t &lo, int64_t &hi) {
	if(r.kind == REF_CANDIDATE) {
		lo = (r.n - kRefPre) & ~255ll; hi = (r.n + kRefPost) | 255; if(lo < 0) lo = 0; if(hi > k_end - 1) hi = k_end - 1;
	} else if(r.kind == REF_STALE) {                               // a stretch the ring of an interval start reads: [n, t_first], as the walk will ask for it
		lo = r.n; hi = r.t_first;
	} else {                                                        // header: sync point + nine symbols, and the carrier slope they are sliced with
		lo = r.n - kRefPre; if(r.prev_n >= 0 && r.prev_n < lo) lo = r.prev_n;
		hi = r.t_first + 8 * kSpsDec;
	}
}
VDL2_HD __attribute__((always_inline)) bool ref_verify(const RefReq &r, uint32_t freq, float max_ppm, float ppm_thr, int64_t k_end, const Tables &T, const ChanView &v, float *lds) {
	int64_t lo, hi;
	ref_request_window(r, k_end, lo, hi);
	if(r.kind == REF_STALE) {
		// nothing to compare: the walk has flagged the channel itself; the stretch is made exact for the walk-again to find
		(void)ref_exact_window(v, lo, hi, nullptr, REF_CANDIDATE);
		return true;
	}
	if(r.kind == REF_CANDIDATE) {
		const int64_t n = r.n;
		if(!ref_exact_window(v, lo, hi, nullptr, REF_CANDIDATE)) return true;
		WAVE_FOR(l)
			if(l < 48) lds[l] = v.Phi(n - 3 * (l >> 4) - 150 + 10 * (l & 15));
		WAVE_END
		WAVE_FOR(l)
			if(l < 3) sync_metric(&lds[16 * l], T, lds[48 + l], lds[52 + l]);
		WAVE_END
		LANE0
			const float y1 = (r.code & 0x10000u) ? kPherrBig : lds[50];
			lds[56] = ((ref_candidate_code(y1, lds[49], lds[48], lds[53], max_ppm, ppm_thr) ^ r.code) & 0xffffu) == 0u ? 1.f : 0.f;
		LANE0_END
		return lds[56] != 0.f;
	}
	// header: sync point + nine symbols, and the carrier slope they are sliced with
	const int64_t ns = r.n;
	if(!ref_exact_window(v, lo, hi, nullptr, REF_HEADER)) return true;
	WAVE_FOR(l)
		if(l < 10) { const int64_t t = l ? r.t_first + (int64_t)(l - 1) * kSpsDec : r.prev_n; lds[l] = t < 0 ? (t == -1 ? 0.f : r.prev_phi0) : v.Phi(t); }
		else if(l >= 16 && l < 32) lds[l] = v.Phi(ns - 3 - 150 + 10 * (l - 16));
	WAVE_END
	LANE0
		float vd = r.vdphi;
		if(r.vdphi_err > 0.f) { float p_; sync_metric(&lds[16], T, p_, vd); }      // (a fire on a contiguous ring: the slope of evaluation ns - 3 again)
		int neg_ = 0; uint32_t code = 0;
		for(int l = 0; l < 9; l++) code |= (uint32_t)slice_symbol(lds[l + 1], lds[l], vd, neg_) << (3 * l);
		lds[56] = code == r.code ? 1.f : 0.f;
	LANE0_END
	(void)freq;
	return lds[56] != 0.f;
}

// a channel whose walk took a decision that does not stand: its state and counters back to what the feed started from, then the
// whole feed again, sequentially, with the referee asked on the spot (most of what it asks for has been made exact by the check)
VDL2_HD __attribute__((always_inline)) void walk_again(int chan, uint32_t freq, float max_ppm, float ppm_thr, int64_t k_end, const Tables &T, const ChanView &v, WalkState *gstate,
		unsigned long long *cnt, Burst *bursts, uint32_t cap_bursts, uint32_t *nbursts_out, OutCtl *ctl, const EvalLog &lg, WalkShared &sh, const WalkSnap &snap) {
	LANE0
		*gstate = snap.ws[chan];
	LANE0_END
	WAVE_FOR(l)
		if(l < kNumCounters) cnt[l] = snap.cnt[(size_t)chan * kNumCounters + l];
	WAVE_END
	WAVE_SYNC_GLOBAL();
	ChanView vs = v; vs.rq = nullptr; vs.rq_n = nullptr; vs.rq_cap = 0; vs.rq_flag = nullptr;
	walk_channel(chan, freq, max_ppm, ppm_thr, k_end, T, vs, gstate, cnt, bursts, cap_bursts, nbursts_out, ctl, lg, sh);
}

// ======================================================================
// Noise floor: v->mag_lp / v->mag_nf of demod.c:238-243, replayed per feed from the walker's
// evaluation log.  Update U happens at the (1000*U)-th evaluation; v->mag_lp there is the
// recurrence mag_lp = mag_lp*0.9 + mag*0.1 over the preceding evaluations, replayed over the last
// kLpTerms of them (0.9^256 ~ 2e-12, below fp32 resolution) in the reference's order.
// ======================================================================
constexpr int kNfGroup = 32;               // updates a wavefront replays per pass: one lane per update in the recurrence (the cheap part)
constexpr int kNfSeg = 64;                 // evaluations gathered between two stretches of the recurrence (one per lane)
constexpr int kNfSlice = 191;              // chunks of the list a group of updates keeps in LDS (a group that touches more reads the list from memory)
static_assert(kLpTerms % kNfSeg == 0, "the replay window is a whole number of gather segments");
struct NfShared {
	alignas(16) float mags[kNfGroup][kNfSeg + 1];      // [update][evaluation of the segment, newest first]; +1: row padding keeps the per-lane replay off one LDS bank
	float   lp[kNfGroup];                               // the recurrence between segments
	int64_t pos[kNfGroup], avail[kNfGroup];             // sample of the evaluation that triggers the update; evaluations before it in the same chunk
	int32_t ci[kNfGroup];                               // its chunk (-1: nothing to replay)
	// the stretch of the combined chunk list the group's updates can touch, staged once (nf_replay_group): sl_cum[i] = cum[sl_lo + i]
	int64_t sl_cum[kNfSlice + 1], sl_first[kNfSlice];
	int32_t flag[64], idx[64];
};

struct NfScratch { int64_t *first; int64_t *cum; };   // combined (tail + feed) chunk list: first sample, ordinal of first evaluation

// what the three noise-floor passes of one feed share (per channel)
struct NfFeed { int64_t ev0, ev1, u0, u1, begin_ord; uint32_t ncomb, pad_; };

// pass 1: combined chunk list = remembered tail + this feed's log, with evaluation ordinals (a prefix sum, 64 chunks at a time)
VDL2_HD void nf_prepare(const NfState *g, const EvalLog &lg, const NfScratch &sc, uint32_t cap_comb, NfFeed *fd, NfShared &sh) {
	int64_t *cnt64 = reinterpret_cast<int64_t *>(&sh.mags[0][0]);   // 64 counts, then 64 running ordinals, then one total
	const uint32_t nlog = *lg.n;
	const uint32_t ntail = (uint32_t)g->ntail;
	uint32_t ntot = ntail + nlog; if(ntot > cap_comb) ntot = cap_comb;
	LANE0
		cnt64[128] = g->tail_ord;
	LANE0_END
	for(uint32_t base = 0; base < ntot; base += 64) {
		WAVE_FOR(l)
			const uint32_t k = base + (uint32_t)l;
			if(k < ntot) {
				const EvalChunk ch = k < ntail ? g->tail[k] : lg.chunks[k - ntail];
				sc.first[k] = ch.first; cnt64[l] = ch.count;
			}
		WAVE_END
		LANE0
			int64_t ord = cnt64[128];
			const uint32_t m = ntot - base < 64 ? ntot - base : 64;
			for(uint32_t i = 0; i < m; i++) { cnt64[64 + i] = ord; ord += cnt64[i]; }
			cnt64[128] = ord;
		LANE0_END
		WAVE_FOR(l)
			const uint32_t k = base + (uint32_t)l;
			if(k < ntot) sc.cum[k] = cnt64[64 + l];
		WAVE_END
	}
	LANE0
		const int64_t ord = cnt64[128];
		sc.cum[ntot] = ord;
		fd->ncomb = ntot; fd->ev0 = g->evals; fd->ev1 = ord; fd->u0 = g->evals / 1000; fd->u1 = ord / 1000;
		fd->begin_ord = ntot ? g->tail_ord : ord;
	LANE0_END
}

// pass 2 (one wavefront per group of kNfGroup updates): v->mag_lp at each update of the group.
// The 256 evaluations before an update are gathered 64 at a time, oldest segment first, by the whole wavefront - lane l takes
// the l-th newest evaluation of the segment, so a load instruction reads 64 evaluations 3 samples apart of one update (1.5 KiB
// of one channel's ring), 16 updates' worth in flight - then one lane per update runs 64 steps of its recurrence out of LDS.
// An evaluation's sample follows from the update's own chunk by arithmetic (the stretches of search between bursts are
// thousands of evaluations long); only a lane whose evaluation lies before the chunk's first walks the chunk list.
VDL2_HD void nf_replay_group(const ChanView &v, const NfScratch &sc, const NfFeed &fd, int64_t group, float *lpbuf, uint32_t cap_hist, NfShared &sh) {
	const int64_t ubase = fd.u0 + 1 + (int64_t)kNfGroup * group;      // first (1-based, global) update of this group
	if(ubase > fd.u1) return;
	const int ncomb = (int)fd.ncomb;
	const int nupd = fd.u1 - ubase + 1 < kNfGroup ? (int)(fd.u1 - ubase + 1) : kNfGroup;
	// The chunks this group can touch - from the one that holds the oldest evaluation its first update replays to the one that holds
	// its last update's trigger - go to LDS once: the lanes then find their evaluations' samples there instead of walking the list in
	// memory, one dependent load after the other (the --max-ppm gate starts a new chunk with every preamble it drops, so the 256
	// evaluations before an update regularly span several).  The first chunk is found by all 64 lanes probing the list at once
	// (three rounds for 8 000 chunks where a binary search takes thirteen).
	const int64_t o_first = 1000 * ubase - 1 - (kLpTerms - 1), o_lastgrp = 1000 * (ubase + nupd - 1) - 1;
	NF_BEGIN();
	int c_lo = 0;
	if(ncomb > 0 && o_first > fd.begin_ord) {
		int lo = 0, hi = ncomb;                                     // cum[lo] <= o_first < cum[hi]  (cum[0] = begin_ord, cum[ncomb] = all evaluations so far)
		while(hi - lo > 1) {
			WAVE_FOR(l)
				int idx = lo + (int)(((int64_t)(hi - lo) * (l + 1)) / 65);
				if(idx <= lo) idx = lo + 1;
				if(idx >= hi) idx = hi - 1;
				sh.idx[l] = idx; sh.flag[l] = sc.cum[idx] <= o_first;
			WAVE_END
			const int k = wave_count_flags(sh.flag);               // the probes ascend: the first k hold, the others do not
			const int nlo = k > 0 ? sh.idx[k - 1] : lo, nhi = k < 64 ? sh.idx[k] : hi;
			WAVE_SYNC();
			lo = nlo; hi = nhi;
		}
		c_lo = lo;
	}
	NF_MARK(0);
	int nsl = ncomb - c_lo < kNfSlice ? ncomb - c_lo : kNfSlice;   // chunks staged
	WAVE_FOR(l)
		for(int i = l; i <= nsl; i += 64) sh.sl_cum[i] = sc.cum[c_lo + i];
		for(int i = l; i < nsl; i += 64) sh.sl_first[i] = sc.first[c_lo + i];
	WAVE_END
	// does the staged stretch reach the group's last trigger?  (else: the list is read from memory, as it always was)
	const bool staged = ncomb > 0 && (c_lo + nsl == ncomb || sh.sl_cum[nsl] > o_lastgrp);
	const int64_t *cum = staged ? sh.sl_cum - c_lo : sc.cum, *first = staged ? sh.sl_first - c_lo : sc.first;
	const int c_min = staged ? c_lo : 0, c_end = staged ? c_lo + nsl : ncomb;
	NF_MARK(1);
	WAVE_FOR(l)
		if(l < nupd) {
			const int64_t o_last = 1000 * (ubase + l) - 1;               // ordinal of the evaluation that triggers the update
			int ci = -1; int64_t pos = 0, avail = 0;
			if(o_last >= fd.begin_ord && ncomb > 0) {
				int lo = c_min, hi = c_end;
				while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(cum[mid] <= o_last) lo = mid; else hi = mid; }
				ci = lo; avail = o_last - cum[ci]; pos = first[ci] + 3 * avail;
			}
			sh.ci[l] = ci; sh.pos[l] = pos; sh.avail[l] = avail; sh.lp[l] = 0.f;
		}
	WAVE_END
	NF_MARK(2);
	for(int seg = kLpTerms / kNfSeg - 1; seg >= 0; seg--) {
		WAVE_FOR(l)
			const int64_t t = (int64_t)kNfSeg * seg + l;                  // this lane's evaluation: the t-th before the triggering one
			constexpr int kBatch = 16;
			for(int u0 = 0; u0 < nupd; u0 += kBatch) {
				int64_t ps[kBatch]; cf32 yv[kBatch];
				for(int q = 0; q < kBatch; q++) {
					const int u = u0 + q;
					ps[q] = -1;
					if(u < nupd && sh.ci[u] >= 0) {
						if(t <= sh.avail[u]) ps[q] = sh.pos[u] - 3 * t;
						else {                                                  // in an earlier chunk, if anywhere
							int64_t rem = t - sh.avail[u] - 1;                    // evaluations to skip, counted back from the end of chunk ci-1
							for(int cj = sh.ci[u] - 1; cj >= c_min; cj--) {       // (below c_min only what no update of the group replays)
								const int64_t len = cum[cj + 1] - cum[cj];
								if(rem < len) { ps[q] = first[cj] + 3 * (len - 1 - rem); break; }
								rem -= len;
							}
						}
					}
				}
				NF_MARK(3);
				for(int q = 0; q < kBatch; q++) yv[q] = ps[q] >= 0 ? v.Y(ps[q]) : cf32{0.f, 0.f};
				NF_WAIT_LOADS(); NF_MARK(4);
				for(int q = 0; q < kBatch; q++) if(u0 + q < nupd) sh.mags[u0 + q][l] = ps[q] >= 0 ? mag_of(yv[q]) : -1.f;
				NF_MARK(5);
			}
		WAVE_END
		WAVE_FOR(l)
			if(l < nupd) {
				float lp = sh.lp[l];
				for(int j = kNfSeg - 1; j >= 0; j--) {
					const float mg = sh.mags[l][j];
					if(mg >= 0.f) lp = lp * 0.9f + mg * (1.0f - 0.9f);
				}
				sh.lp[l] = lp;
			}
		WAVE_END
		NF_MARK(6);
	}
	WAVE_FOR(l)
		const int64_t i = ubase + l - fd.u0;
		if(l < nupd && i < (int64_t)cap_hist) lpbuf[i] = sh.lp[l];
	WAVE_END
	NF_MARK(7);
	NF_END();
}

// pass 3: the mag_nf chain over this feed's updates (sequential, so it runs out of LDS) into the per-channel history
// ring - ring[U & ring_mask] = v->mag_nf after U updates, which is what decode_frame() reports for a burst that
// synchronised after U updates - and the state for the next feed.
VDL2_HD void nf_finish(NfState *g, const NfScratch &sc, const NfFeed &fd, const float *lpbuf, float *ring, uint32_t ring_mask,
		uint32_t cap_hist, NfShared &sh) {
	float *buf = &sh.mags[0][0];
	constexpr int kBatch = 2048;
	int64_t nupd = fd.u1 - fd.u0;                       // updates of this feed
	if(nupd > (int64_t)cap_hist - 1) nupd = (int64_t)cap_hist - 1;
	LANE0
		buf[kBatch] = g->mag_nf;
		ring[(uint32_t)fd.u0 & ring_mask] = g->mag_nf;
	LANE0_END
	for(int64_t i0 = 1; i0 <= nupd; i0 += kBatch) {
		const int m = (int)(nupd - i0 + 1 < kBatch ? nupd - i0 + 1 : kBatch);
		WAVE_FOR(l)
			for(int i = l; i < m; i += 64) buf[i] = lpbuf[i0 + i];
		WAVE_END
		LANE0
			float nf = buf[kBatch];
			for(int i = 0; i < m; i++) {
				nf = 0.85f * nf + (1.0f - 0.85f) * fminf(buf[i], nf) + 0.0001f;
				buf[i] = nf;
			}
			buf[kBatch] = nf;
		LANE0_END
		WAVE_FOR(l)
			for(int i = l; i < m; i += 64) ring[(uint32_t)(fd.u0 + i0 + i) & ring_mask] = buf[i];
		WAVE_END
	}
	LANE0
		const float nf = buf[kBatch];
		g->mag_nf = nf;
		g->evals = fd.ev1;
		// keep the newest chunks that cover the last kLpTerms evaluations
		const int ncomb = (int)fd.ncomb;
		int first_keep = ncomb;
		while(first_keep > 0 && (fd.ev1 - sc.cum[first_keep] < kLpTerms) && (ncomb - first_keep < kNfTail)) first_keep--;
		int nt = 0;
		for(int i = first_keep; i < ncomb; i++, nt++) { g->tail[nt].first = sc.first[i]; g->tail[nt].count = sc.cum[i + 1] - sc.cum[i]; }
		g->ntail = nt;
		g->tail_ord = ncomb ? sc.cum[first_keep] : fd.ev1;
	LANE0_END
}

// v->mag_nf as decode_frame() sees it when a frame is output (decode.c:374): its value at the time of the sync, looked up
// in the history ring.  Kept out of the burst decoder so that the noise-floor replay and the burst decoder of a feed can
// run side by side.
VDL2_HD void stamp_noise_floor(OutFrame &f, const float *ring, uint32_t ring_mask) {
	f.nf_pwr_dbfs = 20.0f * log10f(ring[(uint32_t)f.nf_upd & ring_mask] + 0.001f);
}

// ======================================================================
// Frame finishing (one wavefront per frame, after K4b and K5): the noise-floor figure, and the first thing the
// reference's decoder thread does with a frame (decode.c:466, avlc_parse() avlc.c:163-236): count it, drop it if it is
// shorter than 11 octets or its FCS residue is not 0xF0B8, read the two link addresses and classify the direction.
// ======================================================================
struct FrameShared {
	alignas(16) uint8_t buf[kMaxOctets + 64];
	uint16_t tab[4][256];              // slicing tables: tab[k][b] = FCS register after octet b and k zero octets
};

// once per workgroup: tab[0] is the reference's table (crc.c:23-57), tab[k] = tab[k-1] advanced by one zero octet
VDL2_HD void frame_shared_init(const Tables &T, FrameShared &sh) {
	WAVE_FOR(l)
		for(int b = l; b < 256; b += 64) sh.tab[0][b] = T.crc16[b];
	WAVE_END
	WAVE_FOR(l)
		for(int b = l; b < 256; b += 64) {
			uint16_t c = sh.tab[0][b];
			for(int k = 1; k < 4; k++) { c = (uint16_t)((c >> 8) ^ sh.tab[0][c & 0xffu]); sh.tab[k][b] = c; }
		}
	WAVE_END
}

// parse_dlc_addr() (avlc.c:158-161): 4 address octets -> 28-bit value, bits reversed (addr:24 | type:3 | status:1)
VDL2_HD uint32_t avlc_addr(const uint8_t *b) {
	uint32_t v = (uint32_t)(b[0] >> 1) | ((uint32_t)b[1] << 6) | ((uint32_t)b[2] << 13) | ((uint32_t)(b[3] & 0xfe) << 20);
	uint32_t r = 0;
	for(int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);    // reverse(v, 28) of bitstream.c:152-164
	return (r >> 4) & 0x0FFFFFFFu;
}

VDL2_HD void finish_frame(OutFrame &f, const uint8_t *pool, const Tables &T, unsigned long long *acnt, const float *ring, uint32_t ring_mask, FrameShared &sh) {
	const uint32_t len = f.len;
	const uint8_t *src = pool + f.pool_off;
	const uint32_t n = len < (uint32_t)sizeof sh.buf ? len : (uint32_t)sizeof sh.buf;
	WAVE_FOR(l)
		for(uint32_t i = l; i < n; i += 64) sh.buf[i] = src[i];
	WAVE_END
	LANE0
		stamp_noise_floor(f, ring, ring_mask);
		VDL2_CNT_ADD(acnt, ACNT_PROCESSED, 1);
		uint32_t status = AVLC_OK, dst = 0, sa = 0;
		if(len < (uint32_t)kMinAvlcLen) { status = AVLC_TOO_SHORT; VDL2_CNT_ADD(acnt, ACNT_TOO_SHORT, 1); }
		else {
			// crc16_ccitt(buf, len, 0xFFFF) (crc.c:59-63), four octets per step: the 16-bit register is used up by the
			// first two, so the four table reads of a step do not depend on each other
			uint32_t crc = 0xFFFFu;
			uint32_t i = 0;
			for(; i + 4 <= n; i += 4) {
				const uint32_t w = *reinterpret_cast<const uint32_t *>(sh.buf + i);   // octet i in the low byte (little endian)
				const uint32_t x = crc ^ (w & 0xffffu);
				crc = (uint32_t)(sh.tab[3][x & 0xffu] ^ sh.tab[2][x >> 8] ^ sh.tab[1][(w >> 16) & 0xffu] ^ sh.tab[0][w >> 24]);
			}
			for(; i < len; i++) {
				const uint8_t b = i < n ? sh.buf[i] : src[i];
				crc = (crc >> 8) ^ sh.tab[0][(crc ^ b) & 0xffu];
			}
			if(crc != kGoodFcs) { status = AVLC_BAD_FCS; VDL2_CNT_ADD(acnt, ACNT_BAD_FCS, 1); }
			else {
				VDL2_CNT_ADD(acnt, ACNT_GOOD, 1);
				dst = avlc_addr(sh.buf); sa = avlc_addr(sh.buf + 4);
				const uint32_t st = (sa >> 24) & 7u, dt = (dst >> 24) & 7u;  // a_addr.type
				const bool s_air = st == 1, s_gnd = st == 4 || st == 5;
				const int to = dt == 1 ? 0 : (dt == 4 || dt == 5) ? 1 : dt == 7 ? 2 : -1;    // aircraft / ground / all
				if(s_air && to >= 0) VDL2_CNT_ADD(acnt, to == 1 ? ACNT_AIR2GND : to == 0 ? ACNT_AIR2AIR : ACNT_AIR2ALL, 1);
				if(s_gnd && to >= 0) VDL2_CNT_ADD(acnt, to == 0 ? ACNT_GND2AIR : to == 1 ? ACNT_GND2GND : ACNT_GND2ALL, 1);
			}
		}
		f.avlc_status = status; f.dst_addr = dst; f.src_addr = sa; f.pad_ = 0;
	LANE0_END
}

// ======================================================================
// Burst decoder: one wavefront per burst
// ======================================================================
struct BurstShared {
	// sym/oct are dead once the RS table is filled; tpos/tG are first written by the un-stuffer: they share storage,
	// which keeps a workgroup of this latency-bound kernel from taking LDS that the channeliser of the next feed wants
	union {
		struct {
			uint8_t sym[kMaxSyms];     // 3-bit Gray values, one per symbol
			uint8_t oct[kMaxOctets];   // received data + FEC octets (still interleaved)
		};
		struct {
			uint16_t tpos[kMaxTerm];   // terminator positions, ascending
			uint16_t tG[kMaxTerm];     // kept bits before each terminator
		};
	};
	uint8_t tab[kMaxBlocks * 256];     // de-interleaved RS blocks, row stride 256
	uint32_t xw[kMaxWords + 1];        // corrected data as 32-bit words, stream bit 32w+k = bit k of xw[w]
	uint32_t keptw[kMaxWords + 1];     // bits that survive zero-deletion (valid and not a stuffed zero)
	uint32_t termw[kMaxWords];         // flag terminators (a 0 after exactly six 1s)
	uint16_t cumk[kMaxWords + 1];      // kept bits before word w
	uint16_t cumt[kMaxWords + 1];      // terminators before word w
	uint32_t lanek[64], lanet[64];
	int32_t  laneerr[64];
	int32_t  flag_err[64];
	uint8_t synp[kRsPar][64];          // per-lane syndrome partials
	uint8_t syn[8];
	uint8_t rs_lam[8];
	int32_t rs_deg;
	uint32_t rs_hits[64];
	uint8_t gf_exp[512], gf_log[256];  // LDS copies of the field tables
	uint8_t gam[2][8];                 // erasure locators of short blocks: the missing parity octets sit at fixed places, [0]: two of them, [1]: four
	float   pw[64];
	float   qmin[64];                  // referee: per lane, the smallest (distance from a decision boundary) / (1/|y_cur| + 1/|y_prev|) among its symbols
	int32_t neg[64];
	int32_t u_ret, u_kind, u_ok, u_negs, u_defer;
	uint32_t u_pm[4];                  // referee: the pieces of the burst that hold a marked symbol's samples
	uint32_t u_k, u_sprev, u_lastend, u_S, u_L, u_off, u_capf, u_capp;
	float u_pwr, u_pwr_db;
	uint32_t res_slot, res_nslot, res_off, res_npool, res_capf, res_capp;   // the wavefront's reserve of frame records and octet space (burst_reserve_*)
	uint32_t fr_S[64], fr_len[64], fr_cum[65];   // the frames of one pass of step 6: first kept bit, octets, start of their (padded) pool space
};

// decode_rs_char() on one 255-symbol row (libfec/decode_rs.h:71-298) behind rs_verify() (rs.c:32-49), split in
// wave phases: syndromes (all lanes), erasure locator + Berlekamp-Massey (one lane), Chien search (all lanes),
// Forney (one lane).  The single-lane parts are written with fully unrolled, statically indexed loops so the
// small polynomials live in registers.  Result (corrected-symbol count or -1) in sh.u_ret.
VDL2_HD int gf_mod255(int x) { while(x >= 255) x -= 255; return x; }

VDL2_HD void rs_decode_row(uint8_t *row, int npar, BurstShared &sh) {
	const uint8_t *EXP = sh.gf_exp, *LOG = sh.gf_log;
	const int n_era = kRsPar - npar;
	if(npar == 0) { LANE0 sh.u_ret = 0; LANE0_END return; }
	// syndromes S_i = sum_j d[j] * alpha^((120+i)*(254-j)): the value the Horner loop of decode_rs.h:82-93 produces
	WAVE_FOR(l)
		uint8_t acc[kRsPar] = {0, 0, 0, 0, 0, 0};
		for(int j = l; j < kRsN; j += 64) {
			const uint8_t d = row[j];
			if(d) {
				const int lg = LOG[d], pw = 254 - j;
				for(int i = 0; i < kRsPar; i++) acc[i] ^= EXP[lg + ((120 + i) * pw) % 255];
			}
		}
		for(int i = 0; i < kRsPar; i++) sh.synp[i][l] = acc[i];
	WAVE_END
	WAVE_FOR(l)
		if(l < kRsPar) { uint8_t a = 0; for(int q = 0; q < 64; q++) a ^= sh.synp[l][q]; sh.syn[l] = a; }
	WAVE_END
	const int any = sh.syn[0] | sh.syn[1] | sh.syn[2] | sh.syn[3] | sh.syn[4] | sh.syn[5];
	if(!any) { LANE0 sh.u_ret = 0; LANE0_END return; }           // decode_rs.h:102-108: a codeword, data untouched
	if(n_era > 0) {
		// A short block arrives with its missing parity octets zero-filled and declared erased (rs.c:36-44, decode.c:279-280): its
		// syndromes are never zero, and the reference goes through the whole decoder to "correct" octets nobody reads.  When the
		// block has no other error, every discrepancy of the Berlekamp-Massey loop (decode_rs.h:166-207, r > no_eras) is zero - it is
		// coefficient r-1 of syndrome x erasure locator - so lambda stays the erasure locator, the Chien search finds exactly the
		// erased places, and decode_rs_char() returns their number with the data columns untouched.  Those coefficients are checked
		// here, a lane each; only a block that does have an error takes the long way.
		const uint8_t *gam = sh.gam[n_era == 2 ? 0 : 1];
		WAVE_FOR(l)
			if(l >= n_era && l < kRsPar) {
				int acc = 0;
				for(int i = 0; i <= n_era; i++) {
					const int g = gam[i], sy = sh.syn[l - i];
					if(g && sy) acc ^= EXP[LOG[g] + LOG[sy]];
				}
				sh.synp[0][l] = (uint8_t)acc;
			}
		WAVE_END
		int disc = 0;
		for(int m = n_era; m < kRsPar; m++) disc |= sh.synp[0][m];
		if(!disc) { LANE0 sh.u_ret = n_era; LANE0_END return; }
	}

	LANE0
		const int A0 = 255;
		int syn[kRsPar], lam[kRsPar + 1], b[kRsPar + 1], t[kRsPar + 1];
		for(int i = 0; i < kRsPar; i++) syn[i] = LOG[sh.syn[i]];                // index form (:96-100)
		for(int i = 0; i <= kRsPar; i++) lam[i] = 0;
		lam[0] = 1;
		if(n_era > 0) {                                                           // erasure locator (:113-123)
			lam[1] = EXP[gf_mod255(254 - (kRsK + npar))];
			for(int i = 1; i < kRsPar; i++) {
				if(i < n_era) {
					const int u = gf_mod255(254 - (kRsK + npar + i));
					for(int j = kRsPar; j > 0; j--) {
						if(j <= i + 1) { const int lg = LOG[lam[j - 1]]; if(lg != A0) lam[j] ^= EXP[u + lg]; }
					}
				}
			}
		}
		for(int i = 0; i <= kRsPar; i++) b[i] = LOG[lam[i]];
		int el = n_era;
		for(int r = 1; r <= kRsPar; r++) {                                      // Berlekamp-Massey (:166-207)
			if(r > n_era) {
				int disc = 0;
				for(int i = 0; i < kRsPar; i++) {
					if(i < r) { const int sy = syn[r - i - 1 < 0 ? 0 : r - i - 1]; if(lam[i] != 0 && sy != A0) disc ^= EXP[LOG[lam[i]] + sy]; }
				}
				disc = LOG[disc];
				if(disc == A0) {
					for(int i = kRsPar; i > 0; i--) b[i] = b[i - 1];
					b[0] = A0;
				} else {
					t[0] = lam[0];
					for(int i = 0; i < kRsPar; i++) t[i + 1] = (b[i] != A0) ? (lam[i + 1] ^ EXP[disc + b[i]]) : lam[i + 1];
					if(2 * el <= r + n_era - 1) {
						el = r + n_era - el;
						for(int i = 0; i <= kRsPar; i++) b[i] = (lam[i] == 0) ? A0 : gf_mod255(LOG[lam[i]] - disc + 255);
					} else {
						for(int i = kRsPar; i > 0; i--) b[i] = b[i - 1];
						b[0] = A0;
					}
					for(int i = 0; i <= kRsPar; i++) lam[i] = t[i];
				}
			}
		}
		int deg = 0;
		for(int i = 0; i <= kRsPar; i++) { lam[i] = LOG[lam[i]]; if(lam[i] != A0) deg = i; sh.rs_lam[i] = (uint8_t)lam[i]; }   // (:210-215)
		sh.rs_deg = deg;
	LANE0_END

	// Chien search (:217-240): position i (1..255) is a root iff 1 + sum_j alpha^(lambda_j + j*i) == 0
	WAVE_FOR(l)
		uint32_t hits = 0;
		for(int q = 0; q < 4; q++) {
			const int i = 1 + l + 64 * q;
			if(i <= 255) {
				int qv = 1;
				for(int j = 1; j <= kRsPar; j++) {
					const int lj = sh.rs_lam[j];
					if(j <= sh.rs_deg && lj != 255) qv ^= EXP[gf_mod255(lj + (j * i) % 255)];
				}
				if(qv == 0) hits |= 1u << q;
			}
		}
		sh.rs_hits[l] = hits;
	WAVE_END

	LANE0
		const int A0 = 255, FCR = 120;
		const int deg = sh.rs_deg;
		int root[kRsPar], count = 0;
		for(int i = 0; i < kRsPar; i++) root[i] = 0;
		for(int q = 0; q < 4; q++)
			for(int l = 0; l < 64; l++)
				if((sh.rs_hits[l] >> q) & 1u) {
					const int i = 1 + l + 64 * q;
					for(int c = 0; c < kRsPar; c++) if(c == count) root[c] = i;
					count++;
				}
		int ret;
		if(deg != count) ret = -1;                                                // (:241-248)
		else {
			int lam[kRsPar + 1], syn[kRsPar], om[kRsPar];
			for(int i = 0; i <= kRsPar; i++) lam[i] = sh.rs_lam[i];
			for(int i = 0; i < kRsPar; i++) syn[i] = LOG[sh.syn[i]];
			const int deg_om = deg - 1;
			for(int i = 0; i < kRsPar; i++) {                                     // omega = syn*lambda mod x^6 (:253-261)
				int acc = 0;
				for(int j = 0; j < kRsPar; j++)
					if(i <= deg_om && j <= i) { const int sy = syn[i - j < 0 ? 0 : i - j]; if(sy != A0 && lam[j] != A0) acc ^= EXP[sy + lam[j]]; }
				om[i] = LOG[acc];
			}
			for(int c = 0; c < kRsPar; c++) {                                     // Forney (:267-291)
				if(c < count) {
					const int rt = root[c];
					int num1 = 0, den = 0;
					for(int i = 0; i < kRsPar; i++) if(i <= deg_om && om[i] != A0) num1 ^= EXP[gf_mod255(om[i] + (i * rt) % 255)];
					const int num2 = EXP[gf_mod255((rt * (FCR - 1)) % 255 + 255)];
					const int top = (deg < kRsPar - 1 ? deg : kRsPar - 1) & ~1;
					for(int i = 0; i <= 4; i += 2) if(i <= top && lam[i + 1] != A0) den ^= EXP[gf_mod255(lam[i + 1] + (i * rt) % 255)];
					// loc = k of decode_rs.h:224 with iprim = 1: k = i - 1
					if(num1 != 0) row[rt - 1] ^= EXP[gf_mod255(LOG[num1] + LOG[num2] + 255 - LOG[den])];
				}
			}
			ret = count;
		}
		sh.u_ret = ret;
	LANE0_END
}


// Frame records and octet space.  Wavefront w of the burst decoder OWNS records w*kResSlots .. and octets w*kResPool .. of the feed's
// output (the feed-wide counters start behind those of all wavefronts: burst_reserve_initial_*) and hands them out to its bursts
// itself; only when a list of frames does not fit what is left does it go to the counters, for what the list needs plus a fresh
// reserve.  No atomic on the common path: a returning atomic on a counter that every wavefront of the chip is after was the longest
// wait of the whole burst (tens of thousands of clocks).  What a wavefront has not used when it is done - or gives up for a fresh
// reserve - stays behind as tombstone records (chan = -1: skipped by k_frame_finish and by the host) and unused octets.
constexpr int kResSlots = 8, kResPool = 1024;
VDL2_HD uint32_t burst_reserve_initial_frames(uint32_t nwaves) { return nwaves * (uint32_t)kResSlots; }
VDL2_HD uint32_t burst_reserve_initial_pool(uint32_t nwaves) { return nwaves * (uint32_t)kResPool; }
// one lane (call inside a LANE0 section): the unused records of the reserve become tombstones
VDL2_HD void burst_reserve_release(OutFrame *frames, BurstShared &sh) {
	for(uint32_t i = 0; i < sh.res_nslot; i++) {
		const uint32_t slot = sh.res_slot + i;
		if(slot < sh.res_capf) { OutFrame &f = frames[slot]; f.chan = -1; f.len = 0; f.pool_off = 0; f.nf_upd = 0; }
	}
	sh.res_nslot = 0; sh.res_npool = 0;
}
// when the wavefront has decoded its last burst
VDL2_HD void burst_reserve_done(OutFrame *frames, BurstShared &sh) {
	LANE0
		burst_reserve_release(frames, sh);
	LANE0_END
}

// once per wavefront (`wave` of the launch's wavefronts): LDS copies of the field tables (they do not change from burst to burst), the
// erasure locators of short blocks, the wavefront's own share of the output
VDL2_HD void burst_shared_init(const Tables &T, uint32_t wave, const OutCtl *ctl, BurstShared &sh, bool no_reserve = false) {
	WAVE_FOR(l)
		for(int i = l; i < 512; i += 64) sh.gf_exp[i] = T.gf_exp[i];
		for(int i = l; i < 256; i += 64) sh.gf_log[i] = T.gf_log[i];
	WAVE_END
	WAVE_FOR(l)
		if(l < 2) {
			// the erasure locator decode_rs.h:113-123 builds for the 2 (l = 0) or 4 (l = 1) parity octets a short block does not
			// carry: they are always the last ones of the row, so the polynomial depends on their number only
			const int n_era = 2 + 2 * l, npar = kRsPar - n_era;
			int lam[kRsPar + 1];
			for(int i = 0; i <= kRsPar; i++) lam[i] = 0;
			lam[0] = 1;
			lam[1] = sh.gf_exp[gf_mod255(254 - (kRsK + npar))];
			for(int i = 1; i < n_era; i++) {
				const int u = gf_mod255(254 - (kRsK + npar + i));
				for(int j = i + 1; j > 0; j--) { const int lg = sh.gf_log[lam[j - 1]]; if(lg != 255) lam[j] ^= sh.gf_exp[u + lg]; }
			}
			for(int i = 0; i <= kRsPar; i++) sh.gam[l][i] = (uint8_t)lam[i];
		}
		if(l == 0) {
			sh.res_slot = wave * (uint32_t)kResSlots; sh.res_nslot = (uint32_t)kResSlots; sh.res_off = wave * (uint32_t)kResPool; sh.res_npool = (uint32_t)kResPool;
			if(no_reserve) { sh.res_nslot = 0; sh.res_npool = 0; }        // (the second pass over deferred bursts: those shares are the first pass's)
			sh.res_capf = ctl->cap_frames; sh.res_capp = ctl->cap_pool;
		}
	WAVE_END
}

// Referee, long feeds: a burst some of whose symbols have to be sliced on the reference's own samples is not held up for the
// scans (milliseconds each, one after the other) - the first pass lists it and the stretches it needs (pass 1), k_ref_scan makes
// them all at once, a wavefront each, and a second pass decodes the listed bursts (pass 2: every stretch it asks for is then done).
constexpr int kRefPieceBits = 10;           // a burst's stretches are asked for in pieces of 2^10 decimated samples, each scanned on its own (a scan is its run-up, 2^17 input samples, plus its stretch: with pieces of 2^12 - 82 000 input samples - a listed burst's scans took 2.2 ms instead of 1.6)
struct ScanReq { int32_t chan, kind; int64_t lo, hi; };
struct BurstDefer { uint32_t *dq, *dq_n; uint32_t dq_cap; ScanReq *sq; uint32_t *sq_n; uint32_t sq_cap; int32_t pass; };

// decode_vdl2_burst() DEC_DATA branch + decode_frame(): decode.c:259-380, 173-194.  burst_shared_init() first.
// (df, tag: see BurstDefer - tag is what pass 1 lists the burst as)
VDL2_HD __attribute__((always_inline)) void decode_burst(const Burst &b, uint32_t freq, const Tables &T, const ChanView &v, unsigned long long *cnt,
		OutFrame *frames, uint8_t *pool, OutCtl *ctl, BurstShared &sh, const BurstDefer *df = nullptr, uint32_t tag = 0) {
	// geometry again from TL (decode.c:233-256)
	const uint32_t octets = b.tl_bits / 8 + (b.tl_bits % 8 != 0);
	uint32_t nblk = octets / kRsK, last = octets % kRsK;
	uint32_t fec = nblk * kRsPar;
	if(last != 0) nblk++;
	fec += (uint32_t)fec_octets_for(last);
	if(last == 0) last = kRsK;
	const int npar_last = fec_octets_for(last);
	const int nsym = b.nsym;
	K5_BEGIN();

	// 1. slice every symbol (demod.c:252-274); decisions are independent because prev_phi is the raw phase.  A lane takes
	//    a run of consecutive symbols, so that the phase of a symbol (one double-precision atan2, evaluated here: there is no
	//    stored phase stream) also serves as the next symbol's prev_phi.
	//    Referee: a symbol whose decision is within the margin of the stream's error is marked (bit 7 of its entry) and sliced
	//    again below, on the reference's own samples.
	const int per_lane = (nsym + 63) / 64;
	const bool ref_on = v.ref != nullptr;
	float vdphi = b.vdphi, ppm = b.ppm;
	const float verr = fabsf(b.vdphi_err);
	WAVE_FOR(l)
		float pw = 0.f; int neg = 0; float qmin = kRefBig;
		const int m0 = l * per_lane, m1 = m0 + per_lane < nsym ? m0 + per_lane : nsym;
		float prev = b.prev_phi0, pim2 = 0.f, pm2 = 0.f;              // phase, 1/|y|^2 and |y|^2 of the previous symbol's sample
		if(m0 < nsym) {
			const int64_t tp = m0 > 0 ? b.t_first + (int64_t)(m0 - 1) * kSpsDec : b.prev_n;
			if(tp >= 0) { const cf32 yp = v.Y(tp); if(m0 > 0) prev = phase_of(yp); pim2 = ref_inv_mag2(yp); pm2 = yp.re * yp.re + yp.im * yp.im; }
			else if(tp == -2) pim2 = kRefBig;
		}
		for(int mb = m0; mb < m1; mb += 4) {
			cf32 yv[4];
			for(int q = 0; q < 4; q++) yv[q] = mb + q < m1 ? v.Y(b.t_first + (int64_t)(mb + q) * kSpsDec) : cf32{0.f, 0.f};   // loads first
			for(int q = 0; q < 4 && mb + q < m1; q++) {
				const float cur = phase_of(yv[q]);
				int neg1 = 0;
				const int idx = slice_symbol(cur, prev, vdphi, neg1);
				uint8_t sy = (uint8_t)((idx ^ (idx >> 1)) | (neg1 << 6));   // graycode[] of demod.c:223 (= Tables::gray, tests/test_design.py), without the table; bit 6: counted in `neg`
				neg += neg1;
				const float m2 = yv[q].re * yv[q].re + yv[q].im * yv[q].im;
				if(ref_on) {
					const float im2 = ref_inv_mag2(yv[q]), a2 = fmaxf(m2, pm2);
					const float d = ref_symbol_dist(cur, prev, vdphi) - verr - 2e-6f, w = sqrtf(im2) + sqrtf(pim2);
					if(d <= kRefKappa * sqrtf(a2) * w) sy |= 0x80;
					const float qv = w > 0.f ? d / w : kRefBig;
					qmin = qv < qmin ? qv : qmin;
					pim2 = im2; pm2 = m2;
				}
				sh.sym[mb + q] = sy;
				prev = cur;
				pw += m2;
			}
		}
		sh.pw[l] = pw; sh.neg[l] = neg; sh.qmin[l] = qmin;
	WAVE_END
	LANE0
		// frame_pwr: the reference keeps a running mean updated per symbol (demod.c:266-268); the mean
		// of the same terms is formed here from 64 partial sums (differs only in rounding).
		double s = 0.0; int negs = 0;
		for(int l = 0; l < 64; l++) { s += (double)sh.pw[l]; negs += sh.neg[l]; }
		sh.u_pwr = (float)(s / (double)nsym);
		sh.u_negs = negs;
	LANE0_END
	if(ref_on) {
		// the error bound scales with the signal around the sample: the two samples of the decision (above) or the burst's rms -
		// a lane one of whose symbols could be within THAT bound looks at its run again; then the first and last marked symbol
		const float athr = kRefKappa * 2.0f * sqrtf(sh.u_pwr);
		WAVE_FOR(l)
			const int m0 = l * per_lane, m1 = m0 + per_lane < nsym ? m0 + per_lane : nsym;
			uint32_t first = 0xffffffffu, last = 0u;
			if(m0 < nsym) {
				const bool again = sh.qmin[l] <= athr;
				float prev = b.prev_phi0, pim2 = 0.f;
				if(again) {
					const int64_t tp = m0 > 0 ? b.t_first + (int64_t)(m0 - 1) * kSpsDec : b.prev_n;
					if(tp >= 0) { const cf32 yp = v.Y(tp); if(m0 > 0) prev = phase_of(yp); pim2 = ref_inv_mag2(yp); }
					else if(tp == -2) pim2 = kRefBig;
				}
				for(int m = m0; m < m1; m++) {
					if(again) {
						const cf32 y = v.Y(b.t_first + (int64_t)m * kSpsDec);
						const float cur = phase_of(y), im2 = ref_inv_mag2(y);
						if(ref_symbol_dist(cur, prev, vdphi) - verr - 2e-6f <= athr * (sqrtf(im2) + sqrtf(pim2))) sh.sym[m] |= 0x80;
						prev = cur; pim2 = im2;
					}
					if(sh.sym[m] & 0x80) { if(first == 0xffffffffu) first = (uint32_t)m; last = (uint32_t)m; }
				}
			}
			sh.lanek[l] = first; sh.lanet[l] = ~last;
		WAVE_END
		const uint32_t mfirst = wave_min64(sh.lanek), mlast = ~wave_min64(sh.lanet);
		WAVE_SYNC();
		if(mfirst != 0xffffffffu) {
			// ---- referee: the marked symbols (and, if it is not the reference's yet, the carrier slope) on the reference's own samples ----
			// A scan costs its run-up (2^17 input samples) plus the stretch, so not the whole burst is asked for but the pieces of it
			// that hold a marked symbol's two samples or the slope's taps (which lie before the burst).
			const bool redo_slope = b.vdphi_err > 0.f;
			const int64_t t_lo = mfirst > 0 ? b.t_first + (int64_t)(mfirst - 1) * kSpsDec : (b.prev_n >= 0 ? b.prev_n : b.t_first);
			const int64_t hi = b.t_first + (int64_t)mlast * kSpsDec;
			const int64_t s_lo = b.sync_sample - kRefPre;
			const int64_t w_lo = (redo_slope && s_lo < t_lo) ? (s_lo < 0 ? 0 : s_lo) : t_lo;
			const int64_t P0 = w_lo >> kRefPieceBits;
			const bool wide = (hi >> kRefPieceBits) - P0 >= 128;             // (longer than any burst of the bench workloads: one stretch then)
			const int pass = df ? df->pass : 0;
			LANE0
				uint32_t pm[4] = { 0u, 0u, 0u, 0u };
				if(wide) pm[0] = 1u;
				else {
					if(redo_slope) for(int64_t q = ((s_lo < 0 ? 0 : s_lo) >> kRefPieceBits) - P0; q <= (b.sync_sample >> kRefPieceBits) - P0; q++) if(q >= 0 && q < 128) pm[q >> 5] |= 1u << (q & 31);
					for(int l = 0; l < 64; l++) {
						const uint32_t first = sh.lanek[l], last = ~sh.lanet[l];
						if(first == 0xffffffffu) continue;
						const int64_t a_ = first > 0 ? b.t_first + (int64_t)(first - 1) * kSpsDec : (b.prev_n >= 0 ? b.prev_n : b.t_first);
						const int64_t b_ = b.t_first + (int64_t)last * kSpsDec;
						for(int64_t q = (a_ >> kRefPieceBits) - P0; q <= (b_ >> kRefPieceBits) - P0; q++) if(q >= 0 && q < 128) pm[q >> 5] |= 1u << (q & 31);
					}
				}
				for(int i = 0; i < 4; i++) sh.u_pm[i] = pm[i];
				sh.u_defer = 0;
				if(pass == 1) {
					// list the burst and its pieces; a burst that does not get on the list is done here and now
#if VDL2_DEVICE_PASS
					const uint32_t i = atomicAdd(df->dq_n, 1u);
#else
					const uint32_t i = (*df->dq_n)++;
#endif
					if(i < df->dq_cap) {
						df->dq[i] = tag; sh.u_defer = 1;
						uint32_t np = 0;
						for(int q = 0; q < 4; q++) np += (uint32_t)popc32(pm[q]);
#if VDL2_DEVICE_PASS
						uint32_t j = atomicAdd(df->sq_n, np);
#else
						uint32_t j = *df->sq_n; *df->sq_n += np;
#endif
						for(int q = 0; q < 128; q++) {
							if(!((pm[q >> 5] >> (q & 31)) & 1u)) continue;
							int64_t lo_ = (P0 + q) << kRefPieceBits, hi_ = ((P0 + q + 1) << kRefPieceBits) - 1;
							if(wide) { lo_ = w_lo; hi_ = hi; }
							if(lo_ < w_lo) lo_ = w_lo;
							if(hi_ > hi) hi_ = hi;                                   // (the slope's taps lie before the burst's first symbol: below hi)
							if(j < df->sq_cap) df->sq[j] = ScanReq{ v.ref_chan, REF_SYMBOLS, lo_, hi_ };      // (a piece that finds no room is scanned by pass 2 itself)
							j++;
						}
					}
				}
			LANE0_END
			if(sh.u_defer) { K5_END(); return; }
			bool ok = true;
			{
				// runs of pieces: one stretch each (pass 2: piece by piece, as they were made)
				for(int q = 0; q < 128 && ok; ) {
					if(!((sh.u_pm[q >> 5] >> (q & 31)) & 1u)) { q++; continue; }
					int q1 = q;
					if(pass != 2) while(q1 + 1 < 128 && ((sh.u_pm[(q1 + 1) >> 5] >> ((q1 + 1) & 31)) & 1u)) q1++;
					int64_t lo_ = (P0 + q) << kRefPieceBits, hi_ = ((P0 + q1 + 1) << kRefPieceBits) - 1;
					if(wide) { lo_ = w_lo; hi_ = hi; }
					if(lo_ < w_lo) lo_ = w_lo;
					if(hi_ > hi) hi_ = hi;
					ok = ref_exact_window(v, lo_, hi_, sh.xw, REF_SYMBOLS);
					q = q1 + 1;
				}
			}
			if(ok && redo_slope) {
				float *phs = reinterpret_cast<float *>(sh.keptw);
				WAVE_FOR(l)
					if(l < 16) phs[l] = v.Phi(b.sync_sample - 3 - 150 + 10 * l);
				WAVE_END
				LANE0
					float p_, f_;
					sync_metric(phs, T, p_, f_);
					sh.u_pwr_db = f_;                                  // (a scalar slot that is free until the frames are written)
				LANE0_END
				vdphi = sh.u_pwr_db; ppm = ppm_of(vdphi, freq);
				WAVE_SYNC();
			}
			WAVE_FOR(l)
				int negd = 0;
				if(ok) for(uint32_t m = mfirst + (uint32_t)l; m <= mlast; m += 64) {
					if(!(sh.sym[m] & 0x80)) continue;
					const int64_t t = b.t_first + (int64_t)m * kSpsDec;
					const float cur = v.Phi(t);
					const float prev = m > 0 ? v.Phi(t - kSpsDec) : (b.prev_n >= 0 ? v.Phi(b.prev_n) : b.prev_phi0);
					int neg = 0;
					const int idx = slice_symbol(cur, prev, vdphi, neg);
					negd += neg - (int)((sh.sym[m] >> 6) & 1);
					sh.sym[m] = (uint8_t)(idx ^ (idx >> 1));
				}
				sh.neg[l] = negd;
			WAVE_END
			LANE0
				int d = 0;
				for(int l = 0; l < 64; l++) d += sh.neg[l];
				sh.u_negs += d;
			LANE0_END
		}
	}
	LANE0
		if(sh.u_negs > 0) VDL2_CNT_ADD(cnt, CNT_SLICER_NEG_IDX, sh.u_negs);
	LANE0_END

	K5_MARK(0);
	// 2. descramble + pack data and FEC octets LSB-first (bitstream.c:70-81,94-107): the scrambling sequence comes an octet at a time
	const uint32_t ntot = octets + fec;
	WAVE_FOR(l)
		for(uint32_t i = l; i < ntot; i += 64) {
			uint32_t o = 0, bit0 = kHdrBits + 8 * i;
			for(int j = 0; j < 8; j++) {
				uint32_t bb = bit0 + j;
				o |= (((uint32_t)sh.sym[bb / 3] >> (2 - bb % 3)) & 1u) << j;
			}
			sh.oct[i] = (uint8_t)(o ^ T.prbs_oct[i]);
		}
		for(uint32_t i = l; i < nblk * 256; i += 64) sh.tab[i] = 0;
	WAVE_END

	K5_MARK(1);
	// 3. de-interleave (closed form of decode.c:135-163): column-major, the last row is shorter
	uint32_t fec_rows = nblk; if(npar_last == 0) fec_rows--;
	uint32_t flast = fec % kRsPar; if(flast == 0) flast = kRsPar;
	WAVE_FOR(l)
		for(uint32_t i = l; i < octets; i += 64) {
			uint32_t row, col;
			if(i < last * nblk) { col = i / nblk; row = i - col * nblk; }
			else { uint32_t r = i - last * nblk; col = last + r / (nblk - 1); row = r % (nblk - 1); }
			sh.tab[row * 256 + col] = sh.oct[i];
		}
		for(uint32_t i = l; i < fec; i += 64) {
			uint32_t row, col;
			if(i < flast * fec_rows) { col = i / fec_rows; row = i - col * fec_rows; }
			else { uint32_t r = i - flast * fec_rows; col = flast + r / (fec_rows - 1); row = r % (fec_rows - 1); }
			sh.tab[row * 256 + kRsK + col] = sh.oct[octets + i];
		}
	WAVE_END

	K5_MARK(2);
	// 4. Reed-Solomon per block (decode.c:305-334, rs.c:32-49)
	int fec_fixed = 0; int failed = 0;
	for(uint32_t r = 0; r < nblk && !failed; r++) {
		const int npar = (r == nblk - 1) ? npar_last : kRsPar;
		uint8_t *row = &sh.tab[r * 256];
		rs_decode_row(row, npar, sh);
		LANE0
			VDL2_CNT_ADD(cnt, CNT_BLOCKS_PROCESSED, 1);
			if(sh.u_ret < 0) VDL2_CNT_ADD(cnt, CNT_ERR_FEC_BAD, 1);
			else VDL2_CNT_ADD(cnt, CNT_BLOCKS_FEC_OK, 1);
		LANE0_END
		if(sh.u_ret < 0) failed = 1;
		else if(sh.u_ret > 0) fec_fixed += sh.u_ret - (kRsPar - npar);
	}
	K5_MARK(3);
	if(failed) { K5_END(); return; }

	// 5. re-serialise the corrected rows LSB-first (decode.c:325-328), truncate to TL bits (decode.c:338-342),
	//    then bitstream_copy_next_frame() (bitstream.c:109-150) for the whole burst at once:
	//    with R(i) = length of the run of 1s ending just before bit i (the reference's `ones`),
	//      stuffed zero  : x[i]=0, R=5          flag terminator : x[i]=0, R=6          error : x[i]=1, R>=6
	//    a segment between terminators with j kept bits before its terminator is a leading flag (j=7,
	//    skipped), an error (j<7) or a frame of j-7 bits; what follows the last terminator is the tail frame.
	//    A word per lane throughout (a burst of 170 octets is 43 words: 43 lanes at work, one step each).
	uint32_t nbits = 8 * octets; if(b.tl_bits < nbits) nbits = b.tl_bits;
	const int nw = (int)((nbits + 31) / 32);
	WAVE_FOR(l)
		for(int w = l; w <= nw; w += 64) {
			uint32_t x = 0;
			for(int k = 0; k < 4; k++) {
				const uint32_t i = 4u * (uint32_t)w + (uint32_t)k;
				if(i < octets) { const uint32_t r = i / kRsK; x |= (uint32_t)sh.tab[r * 256 + (i - r * kRsK)] << (8 * k); }
			}
			sh.xw[w] = x;
		}
	WAVE_END
	WAVE_FOR(l)
		uint32_t err = 0xffffffffu;
		for(int w = l; w < nw; w += 64) {
			const uint64_t z = ((uint64_t)sh.xw[w] << 32) | (w ? sh.xw[w - 1] : 0u);
			const uint64_t s1 = z << 1, s2 = s1 & (z << 2), s3 = s2 & (z << 3), s4 = s3 & (z << 4);
			const uint64_t s5 = s4 & (z << 5), s6 = s5 & (z << 6), s7 = s6 & (z << 7);
			const uint32_t rem = nbits - 32u * (uint32_t)w;
			const uint32_t valid = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
			const uint32_t stuffed = (uint32_t)((~z & s5 & ~s6) >> 32) & valid;
			const uint32_t term = (uint32_t)((~z & s6 & ~s7) >> 32) & valid;
			const uint32_t bad = (uint32_t)((z & s6) >> 32) & valid;
			sh.keptw[w] = valid & ~stuffed;
			sh.termw[w] = term;
			sh.cumk[w] = (uint16_t)popc32(valid & ~stuffed); sh.cumt[w] = (uint16_t)popc32(term);     // counts now, prefix sums below
			if(bad && err == 0xffffffffu) err = 32u * (uint32_t)w + (uint32_t)ctz32(bad);              // the lane's words ascend: its first is its lowest
		}
		sh.lanek[l] = err;
	WAVE_END
	const uint32_t E = wave_min64(sh.lanek);                        // first "seven ones" position (0xffffffff: none)
	WAVE_SYNC();
	// exclusive prefix sums of the per-word counts: a lane sums a block of eight consecutive words, the 64 block sums are scanned
	// over the wavefront, the lane adds its block's offset
	WAVE_FOR(l)
		uint32_t bk = 0, bt = 0;
		for(int q = 0; q < 8; q++) {
			const int w = 8 * l + q;
			if(w < nw) { const uint32_t ck = sh.cumk[w], ct = sh.cumt[w]; sh.cumk[w] = (uint16_t)bk; sh.cumt[w] = (uint16_t)bt; bk += ck; bt += ct; }
		}
		sh.lanek[l] = bk; sh.lanet[l] = bt;
	WAVE_END
	const uint32_t gtotal = wave_excl_scan64(sh.lanek);
	const uint32_t nterm = wave_excl_scan64(sh.lanet);
	WAVE_FOR(l)
		const uint32_t ok = sh.lanek[l], ot = sh.lanet[l];
		for(int q = 0; q < 8; q++) {
			const int w = 8 * l + q;
			if(w < nw) { sh.cumk[w] = (uint16_t)(sh.cumk[w] + ok); sh.cumt[w] = (uint16_t)(sh.cumt[w] + ot); }
		}
		if(l == 0) { sh.cumk[nw] = (uint16_t)gtotal; sh.cumt[nw] = (uint16_t)nterm; }
	WAVE_END
	// terminator list: position and kept bits before it
	WAVE_FOR(l)
		for(int w = l; w < nw; w += 64) {
			uint32_t t = sh.termw[w];
			const uint32_t kw = sh.keptw[w], bk = sh.cumk[w];
			uint32_t r = sh.cumt[w];
			while(t) {
				const int bit = ctz32(t); t &= t - 1;
				if(r < (uint32_t)kMaxTerm) { sh.tpos[r] = (uint16_t)(32 * w + bit); sh.tG[r] = (uint16_t)(bk + (uint32_t)popc32(kw & ((1u << bit) - 1u))); }
				r++;
			}
		}
	WAVE_END
	LANE0
		sh.u_k = 0; sh.u_sprev = 0; sh.u_lastend = 0xffffffffu;
	LANE0_END
	K5_MARK(4);
	// 6. frames out, up to 64 at a time: one lane walks the (few) terminators and lists the frames (bitstream_copy_next_frame()'s
	//    rules), reserves their records and octets with ONE pair of atomics, then the wavefront writes the records (a lane per
	//    frame) and gathers the octets of all of them (a lane per octet)
	int nframes = 0;
	for(;;) {
		LANE0
			uint32_t k = sh.u_k, sprev = sh.u_sprev, lastend = sh.u_lastend;
			int nfr = 0, end = 0;                       // end: 0 list full (more may follow), 2 unstuff error, 3 truncated octets, 4 done
			uint32_t pooltot = 0;
			while(nfr < 64) {
				int kind = 0;
				uint32_t S = 0, L = 0;
				for(; k < nterm && k < (uint32_t)kMaxTerm; k++) {
					const uint32_t tp = sh.tpos[k], g = sh.tG[k];
					if(tp > E) { kind = 2; break; }
					const uint32_t j = g - sprev;
					const uint32_t snext = g + 1;
					if(j == 7) { sprev = snext; continue; }
					if(j < 7) { kind = 2; break; }
					S = sprev; L = j - 7; sprev = snext;
					lastend = tp;
					kind = (L % 8) ? 3 : 1;
					k++;
					break;
				}
				if(kind == 0) {
					// past the last terminator: the tail is processed iff a call is still made (bitstream.c:149, decode.c:345)
					const bool call = lastend == 0xffffffffu || lastend + 1 < nbits;
					if(!call || k == 0xffffffffu) kind = 4;
					else if(E != 0xffffffffu) kind = 2;
					else { S = sprev; L = gtotal - sprev; kind = (L % 8) ? 3 : 1; k = 0xffffffffu; }
				}
				if(kind != 1) { end = kind; break; }
				sh.fr_S[nfr] = S; sh.fr_len[nfr] = L / 8; sh.fr_cum[nfr] = pooltot;
				pooltot += (L / 8 + 3u) & ~3u;
				nfr++;
				if(k == 0xffffffffu) { end = 4; break; }
			}
			sh.fr_cum[nfr] = pooltot;
			sh.u_k = k; sh.u_sprev = sprev; sh.u_lastend = lastend; sh.u_kind = end; sh.u_S = (uint32_t)nfr;
			if(end == 2) VDL2_CNT_ADD(cnt, CNT_ERR_UNSTUFF, 1);
			else if(end == 3) VDL2_CNT_ADD(cnt, CNT_ERR_TRUNCATED_OCTETS, 1);
			if(nfr) {
				VDL2_CNT_ADD(cnt, CNT_MSG_GOOD, nfr);
				// Records and octet space come out of a small reserve the wavefront holds (burst_reserve_*): the two feed-wide counters are
				// touched once per kResSlots frames or so instead of once per burst - a returning atomic on a counter that every
				// wavefront of the chip is after was the longest wait of the whole burst (tens of thousands of clocks).
				if((uint32_t)nfr > sh.res_nslot || pooltot > sh.res_npool) {
					burst_reserve_release(frames, sh);                    // what is left of the old reserve is given up (tombstones, a hole)
					const uint32_t ns = (uint32_t)nfr + (uint32_t)kResSlots, np = pooltot + (uint32_t)kResPool;
#if VDL2_DEVICE_PASS
					sh.res_slot = atomicAdd(&ctl->nframes, ns);
					sh.res_off = atomicAdd(&ctl->pool_used, np);
#else
					sh.res_slot = ctl->nframes; ctl->nframes += ns;
					sh.res_off = ctl->pool_used; ctl->pool_used += np;
#endif
					sh.res_nslot = ns; sh.res_npool = np;
				}
				sh.u_L = sh.res_slot; sh.u_off = sh.res_off;
				sh.res_slot += (uint32_t)nfr; sh.res_nslot -= (uint32_t)nfr; sh.res_off += pooltot; sh.res_npool -= pooltot;
				sh.u_pwr_db = 10.0f * log10f(sh.u_pwr);
				sh.u_capf = sh.res_capf; sh.u_capp = sh.res_capp;
			}
		LANE0_END
		K5_MARK(5);
		const int nfr = (int)sh.u_S, end = sh.u_kind;
		if(nfr) {
			const uint32_t slot0 = sh.u_L, off0 = sh.u_off, capf = sh.u_capf, capp = sh.u_capp;
			WAVE_FOR(l)
				if(l < nfr) {
					const uint32_t slot = slot0 + (uint32_t)l, off = off0 + sh.fr_cum[l], len = sh.fr_len[l];
					const bool ok = slot < capf && off + len <= capp;
					sh.flag_err[l] = ok;
					if(ok) {
						OutFrame &f = frames[slot];
						f.chan = b.chan; f.idx = nframes + l; f.len = len; f.pool_off = off;
						f.synd_weight = b.syndrome >> 8; f.datalen_octets = octets;
						f.num_fec_corrections = fec_fixed;
						f.frame_pwr_dbfs = sh.u_pwr_db;
						f.nf_pwr_dbfs = 0.f; f.nf_upd = b.nf_upd;            // filled in by stamp_noise_floor()
						f.ppm_error = ppm;
						f.burst_ord = b.ord; f.sync_sample = b.sync_sample; f.end_sample = b.end_sample;
					} else {
						ctl->overflow = 1;
						if(slot < capf) { OutFrame &f = frames[slot]; f.chan = -1; f.len = 0; f.pool_off = 0; f.nf_upd = 0; }   // tombstone: skipped downstream
					}
				}
			WAVE_END
			K5_MARK(6);
			// gather kept bits S+8i .. S+8i+7 of frame f into its octet i (LSB first, bitstream.c:70-81), all frames of the list at once:
			// octet j of the list's (4-byte padded) pool space belongs to the frame whose space holds it
			const uint32_t span = sh.fr_cum[nfr];
			WAVE_FOR(l)
				for(uint32_t j = l; j < span; j += 64) {
					int f = 0;
					{ int lo = 0, hi = nfr; while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(sh.fr_cum[mid] <= j) lo = mid; else hi = mid; } f = lo; }
					const uint32_t i = j - sh.fr_cum[f];
					if(i < sh.fr_len[f] && sh.flag_err[f]) {
						const uint32_t q0 = sh.fr_S[f] + 8 * i;
						int lo = 0, hi = nw;              // last word with cumk[w] <= q0
						while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(sh.cumk[mid] <= q0) lo = mid; else hi = mid; }
						int w = lo;
						uint32_t m = sh.keptw[w];
						for(uint32_t r = q0 - sh.cumk[w]; r > 0; r--) m &= m - 1;
						uint32_t o = 0;
						for(int t = 0; t < 8; t++) {
							while(m == 0) { w++; m = sh.keptw[w]; }
							const int bit = ctz32(m); m &= m - 1;
							o |= ((sh.xw[w] >> bit) & 1u) << t;
						}
						pool[off0 + j] = (uint8_t)o;
					}
				}
			WAVE_END
			K5_MARK(7);
			nframes += nfr;
		}
		if(end == 2 || end == 3) { K5_END(); return; }
		if(end == 4) break;
	}
	LANE0
		if(sh.u_pwr > 1.0f) VDL2_CNT_ADD(cnt, CNT_MSG_GOOD_LOUD, 1);
	LANE0_END
	K5_END();
}

}  // namespace vdl2
