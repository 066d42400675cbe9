// vdl2hip.hip - libvdl2hip.so: context management and the C ABI of include/vdl2hip.h.
// Host code only orchestrates: every sample-rate or burst-rate computation runs in the
// kernels of kernels.h.  There is deliberately no CPU fallback: without a HIP device
// vdl2hip_create() fails with VDL2HIP_E_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include "../../include/vdl2hip.h"
#include "kernels.h"
#include "tables.h"

using namespace vdl2;

static_assert(VDL2HIP_NUM_COUNTERS == kNumCounters, "counter enum out of sync");

namespace {

// a delivered frame: its record plus the octet pool of the feed it came from (one allocation per feed, shared by its frames)
struct HostFrame {
	OutFrame f;
	std::shared_ptr<std::vector<uint8_t>> pool;
	const uint8_t *octets() const { return pool && f.pool_off + f.len <= pool->size() ? pool->data() + f.pool_off : nullptr; }
};

#ifndef VDL2_K1_RUN
#define VDL2_K1_RUN 2
#endif
constexpr int kRun = VDL2_K1_RUN;        // decimated outputs per lane in K1 (specialised builds); 2 measured best: dev/gpu_k1_variants.sh
constexpr int kRunGeneric = 2;
constexpr int kHistory = 65536;          // decimated samples kept behind the newest block (> longest burst, 56 090)
constexpr int kNumEv = 12;            // profiling: {start, stop} of K1, K2, K3, K4, K4b, K5
constexpr int kColdParts = 4;         // pieces a cold-start block is copied and channelised in
constexpr size_t kColdMinBytes = 8u << 20;
constexpr uint32_t kRetryScans = 64;       // referee: scans of one launch that may be run again from further back because they had not met their witness (more: published as they are, counted)
constexpr uint32_t kPreScans = 4096;    // referee: stretches around marked candidates one feed may list for the scan ahead of the walk (what does not fit is asked for by the walk itself)
// referee: bursts of one feed that may wait for their scans, stretches they may wait for.  What does not fit is scanned on the spot, by the
// burst's own wavefront, one stretch after the other at 4.4 ms each: with 256 / 512 (rounds 5, 6a/b) a capture full of weak bursts - config4
// WITHOUT its --max-ppm gate: the neighbours' leakage is locked on to and decoded, a symbol in a few hundred within the margin - ran over
// and a 16-block feed took 68 ms instead of 2 (profiles/r06_weak_bursts.txt).  A short feed lists a sixteenth of these.
// The lists grow with the feed: a stretch per 8 192 channel-samples of the feed (a 16 s block of 256 channels: 52 000; a rank's 32
// channels: 6 500), half as many bursts, at least kDeferScans / kDeferBursts; the grids that serve them are sized the same way - a
// workgroup that finds no request is gone within a microsecond.
constexpr uint32_t kDeferBursts = 4096, kDeferScans = 8192, kDeferScansMax = 1u << 17;
static uint32_t defer_scans_for(int64_t D, int C) {
	const uint64_t want = (uint64_t)(D > 0 ? D : 0) * (uint64_t)C / 8192u;
	const uint64_t cap = std::min<uint64_t>(kDeferScansMax, std::max<uint64_t>(kDeferScans, want));
	return (uint32_t)((cap + 63) / 64 * 64);
}
// Streams per priority class.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues per
// priority, round-robin in the order of their creation, and two streams on one queue run one after the other.  With a scan stream and
// a burst stream per slot (four of each), a scan stream shared its queue with the walk stream and burst streams shared theirs with the
// front and the noise-floor stream (all three of the front's priority): every fourth feed's 1.5 ms scan sat in front of the next walks
// (a rank-sized receiver: 2.64 ms per step; 1.91 with GPU_MAX_HW_QUEUES=8 - which in turn cost the 256-channel receiver 4 %:
// profiles/r06_hw_queues_ab.txt).  So: high priority = walk + 3 scan streams, the front's priority = front + noise floor + 2 burst
// streams - four each, nobody shares.  (Feed i + 3's scan, feed i + 2's burst decoder behind feed i's: done long before.)
// The streams of a receiver that is destroyed go to a pool (per device) and the next receiver takes them over: the runtime's mapping
// of streams to queues depends on every stream the process has created so far, and a receiver made after a hundred others had come
// and gone ran 30-60 % slower than the same receiver in a fresh process (bench.py's secondary workloads, round 6).
#ifndef VDL2_SIDE_PRE
#define VDL2_SIDE_PRE 3
#endif
constexpr int kSidePre = VDL2_SIDE_PRE, kSideBurst = 2;
constexpr int kSlots = VDL2HIP_MAX_DRAIN_LAG + 1;   // feeds in flight (vdl2hip_set_drain_lag: at most kSlots - 1 undelivered).  Six: a feed's way through the device of a receiver of few channels - front 1 ms, the scans ahead of the walk 1.5, walk, the next feed's walk (the second walks are queued behind it), burst decoder and its scans 1.5-2.5 - is four to six fronts long (rank-sized receivers, walk ahead: 2.22 / 1.46 / 1.56 ms per step with four slots, 1.76 / 1.29 / 1.45 with six; with 256 channels it is three fronts and four slots were enough)

}  // namespace

struct OutSlot {
	Burst *d_bursts = nullptr; uint32_t *d_nbchan = nullptr;
	OutFrame *d_frames = nullptr; uint8_t *d_pool = nullptr;
	OutMail *d_mail = nullptr; OutCtl *d_ctl = nullptr;   // control block + the first few delivered frames, contiguous (d_ctl = &d_mail->ctl): one small copy brings both
	OutFrame *d_frames_out = nullptr; uint8_t *d_pool_out = nullptr;    // what k_frame_finish delivers (no tombstones, no holes): what the host copies
	EvalChunk *d_log = nullptr; uint32_t *d_nlog = nullptr;   // the walker's evaluation log of this feed (read by K4b)
	uint32_t *d_dq = nullptr; ScanReq *d_sq = nullptr;      // referee, long feeds: the bursts that wait for a scan, the stretches they wait for (counts: d_rqn[1], d_rqn[2])
	RefReq *d_rq = nullptr; uint32_t *d_rqn = nullptr, *d_rqflag = nullptr; RefBad *d_rqbad = nullptr;
	ScanReq *d_retry = nullptr;            // [3][kRetryScans]: the unmet scans of the scans ahead of the walk / of the check / of the burst decoder (counts: d_rqn[4..6])
	ScanReq *d_pq = nullptr; hipEvent_t ev_pre = nullptr;      // referee: the stretches around marked candidates, made exact between the front and the walk (count: d_rqn[3])   // referee, optimistic mode: this feed's decisions to check, its "walk again" flags
	OutMail *h_mail = nullptr;             // pinned
	hipEvent_t done = nullptr, ev_front = nullptr, ev_chan = nullptr, ev_walk = nullptr, ev_nf = nullptr, ev[kNumEv] = {};
	bool pending = false, ev_valid = false, fused = false; int ev_level = 0;
	uint64_t seq = 0;
	// the burst-rate back end of this feed (K4, K4b, K5, frame finish) still has to be queued: launch_back()
	bool back_queued = false; int64_t back_D = 0, back_k0 = 0; hipEvent_t ev_k1 = nullptr;
	bool k1_timed = false;                 // the channeliser launch of this feed carries start/stop events (not on a cold-start feed: its pieces wait for copies in between)
	bool prescan = false;                  // referee: the stretches around this feed's marked candidates are scanned ahead of its walk
	// launch_back() / launch_rest(): the walk and check of this feed are queued, what follows them is not yet (rest_pending); what the
	// second walks need of the first: its arguments, segmentation and speculative walks; has_chk: the walk noted decisions to a list that is checked
	bool rest_pending = false, has_chk = false; int nseg = 1; int64_t seglen = 0; K4Args k4{}; SpecOut *d_spec_of = nullptr;
	hipEvent_t ev_stitch = nullptr, ev_chk = nullptr; uint32_t *d_rqflag2 = nullptr;
	unsigned k5_waves = 0; bool small = false;   // wavefronts of this feed's burst decoder; short feed: its whole back end runs on the front stream
};

struct vdl2hip_ctx {
	vdl2hip_cfg cfg{};
	int C = 0, chan_first = 0, os = 0, fmt = 0, run = kRun, cr = 1;
	bool specialised = false;
	std::vector<uint32_t> freqs, dphi;
	LpfCoeffs lpf{};
	BlockForm bf{};
	hipStream_t stream = nullptr;
	// device memory
	BlockForm *d_bf = nullptr; Lut4 *d_lut = nullptr; Tables *d_tab = nullptr;
	uint32_t *d_dphi = nullptr, *d_freq = nullptr; float *d_ppmthr = nullptr;
	// host-fed input: one device buffer + "copy done" event per slot, filled on a copy stream of its own so that the H2D of
	// block i+1 runs beside the channeliser of block i (process_buf_*() hands over host memory: src/demod.c:356-365)
	uint8_t *d_in[kSlots] = {}; hipEvent_t ev_copied[kSlots] = {}; size_t in_cap = 0;
	hipStream_t stream_copy = nullptr, stream_out = nullptr;
	hipEvent_t pinned_pending = nullptr;   // copy event of the last vdl2hip_feed_pinned() whose source buffer the caller may not touch yet
	// cold start (nothing in flight) of a large page-locked block: the copy goes in kColdParts pieces and the channeliser is launched
	// piece by piece behind them, so that the first block of a stream does not wait for its whole H2D (feed_host)
	struct { int n = 0; uint64_t samples[kColdParts] = {}; hipEvent_t ev[kColdParts] = {}; } cold;
	uint8_t *h_stage = nullptr; size_t stage_cap = 0;   // pinned D2H staging for frame records + octets
	uint8_t *d_carry[2] = {nullptr, nullptr}; int carry_sel = 0; uint32_t ncarry = 0;
	cf32 *d_y = nullptr, *d_pf = nullptr; uint64_t *d_cand = nullptr, *d_flag = nullptr;
	uint32_t cap = 0;
	float4 *d_segend = nullptr; uint32_t nseg_cap = 0;
	float4 *d_qpow = nullptr;
	float4 *d_tcarry[2] = {nullptr, nullptr}; int tcarry_sel = 0;
	unsigned long long *d_segpub = nullptr; uint32_t *d_synctmo = nullptr; bool fuse_k2 = true;   // K2 fused into K1 (one-step look-back between segments)
	WalkState *d_ws = nullptr; unsigned long long *d_cnt = nullptr, *d_acnt = nullptr;
	// The reference's per-channel counters live in TWO rows per channel, one per writer: d_cnt is the burst decoder's (atomic adds from
	// k_burst on the burst streams), d_wcnt = d_cnt + C * kNumCounters the walker's (sync.good, the header outcomes, ppm_reject: plain
	// read-modify-writes on the walk stream, and the only row the walk-again snapshot saves and restores).  A reader adds the two.  With one
	// row, a channel walked again wiped whatever the previous feed's burst decoder had added since the snapshot (round 5's red test).
	unsigned long long *d_wcnt = nullptr;
	NfState *d_nf = nullptr; int64_t *d_scfirst = nullptr, *d_sccum = nullptr;
	float *d_nfring = nullptr, *d_lpbuf = nullptr; NfFeed *d_nffeed = nullptr; uint32_t cap_log = 0, cap_comb = 0, cap_hist = 0, nf_ring = 0;
	uint32_t cap_bursts_chan = 0;
	SpecOut *d_spec[3] = {}; uint32_t *d_segstats = nullptr; int seg_max = 1; int64_t seg_min = 16384;   // segmented walk
	OutSlot slot[kSlots];                  // per-feed output buffers: the fronts of feeds i+1, i+2 run while feed i's back still fills slot i%kSlots
	uint64_t feed_no = 0; int drain_lag = 0;
	hipStream_t stream_pre[kSidePre] = {};        // referee, VDL2HIP_REF_PRESCAN=1: the scans ahead of the walk, a stream per feed in flight (one stream would put them in a row: 4.4 ms each)
	hipStream_t stream_back = nullptr, stream_nf = nullptr, stream_burst[kSideBurst] = {};   // (a burst stream per feed in flight: a burst decoder that waits for the referee - a scan over a whole burst takes milliseconds - does not hold up the next feed's)
	// Experiment switches (only read in builds with -DVDL2_EXPERIMENTS, dev/gpu_run.sh; the measured outcomes are in DESIGN 6).
	// sync_on: 0 = both sync kernels on the front stream (the product); 1 = the exact tier in front of the walk on the walk stream;
	// 2 = both on a stream of their own (stream_sync), beside the channeliser of the next feed.  tiles_force / k3b_wpl: K1 tiles per
	// workgroup segment / K3b words per lane instead of the values chosen from the channel count.
	int sync_on = 0; hipStream_t stream_sync = nullptr; int k3b_wpl = 0, tiles_force = 0; bool show_gaps = false; int ablate = 0;
	OutCtl ctl_template{};                 // the capacities of a feed's output buffers (the counters are reset on the device: reset_out_ctl)
	bool pooled = false;                   // its streams are complete and of the product's priorities: they go to the pool when the receiver is destroyed
	bool avlc_filter = false, failed = false; int debug_force_timeout = 0, debug_force_again = 0;
	// Referee (kernels.h): decisions within the margin of the channeliser's distance from the reference's fp32 scan are taken on the
	// reference's own samples, recomputed from the raw input.  The input of a feed stays where it is (d_in / the caller's device
	// buffer) while its back end runs; what lies before it - up to ref_T samples: the run-up of the scan + the longest burst - is kept
	// in a ring (ref_hist), appended to by every feed (its last min(n, ref_T) samples).  ref_pieces: what of the stream the ring holds,
	// contiguously, newest last: {first absolute sample, count, ring position of the first}.
	bool referee = true; int ref_kinds = 7; bool ref_prescan = false;   /* VDL2HIP_REF_PRESCAN=1: the stretches around marked candidates are made exact ahead of the walk - a rank-sized shard 4.05 -> 2.88 ms per step, 256 channels 7.2 -> 7.85: for receivers of few channels */ int64_t ref_warm = 3 << 16 /* 196 608: kernels.h */, ref_T = 0; uint8_t *d_refhist = nullptr; uint64_t ref_cap = 0, ref_wp = 0;
	struct HistPiece { int64_t s0, n; uint64_t pos; }; std::vector<HistPiece> ref_pieces;
	unsigned long long *d_refdbg = nullptr; int ref_dbg_chan = -1;
	WalkState *d_ws_snap[3] = {}, *d_ws_tmp = nullptr; unsigned long long *d_cnt_snap[3] = {}, *d_cnt_tmp = nullptr; uint32_t rq_cap = 8192; uint32_t sq_alloc = 8192; bool ref_optimistic = true;
	int walk_ahead = 1; bool walk_ahead_auto = true; double walk_ahead_below = 1.1e8; int debug_force_mismatch = 0;   // launch_back(): the walks of the next feed (1) or the next two (2) do not wait for this feed's check
	int ref_retry_mul = 2;                 // a scan that has not met its witness is run again from this many times further back (0: not at all; VDL2HIP_REF_RETRY)
	RefChan *d_ref[kSlots] = {}; unsigned long long *d_refdone = nullptr; uint32_t *d_refdonen = nullptr, *d_refstats = nullptr; uint8_t *d_mix = nullptr;
	bool defer_back = false;               // VDL2HIP_BACKEND=deferred: the back end of feed i is queued behind the channeliser of feed i+1 (launch_back)
	std::vector<uint64_t> statsd_prev;
	std::vector<HostFrame> queue;
	int64_t k_total = 0; uint64_t n_total = 0;
	// profiling
	int profiling = 0;                     // 0 off, 1 channeliser only, 2 every stage
	vdl2hip_stats stats{};
};

#define HIPCHK(expr) do { hipError_t e_ = (expr); if(e_ != hipSuccess) { \
	fprintf(stderr, "vdl2hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return VDL2HIP_E_DEVICE; } } while(0)

static size_t sample_bytes(int fmt) { return fmt == VDL2HIP_FMT_S16LE ? 4 : 2; }

// Every entry point works on the context's own device whatever the calling thread's current device is (a process may hold
// receivers on several GPUs), and leaves the thread's device as it found it.
struct OnDevice {
	int prev = -1; bool switched = false;
	explicit OnDevice(const vdl2hip_ctx *c) { if(c && hipGetDevice(&prev) == hipSuccess && prev != c->cfg.device) switched = hipSetDevice(c->cfg.device) == hipSuccess; }
	~OnDevice() { if(switched) (void)hipSetDevice(prev); }
};

// Launch with the kernel's own start/stop stamped into two events (null: plain launch).  Used instead of hipEventRecord
// pairs, which are separate queue entries and cost a few microseconds of stream time each.
#define LAUNCH_EV(kernel, grid, block, stream, e0, e1, ...) hipExtLaunchKernelGGL(kernel, grid, block, 0u, stream, e0, e1, 0, __VA_ARGS__)
#define EV(i) (prof_all ? ev[i] : (hipEvent_t) nullptr)

template<int OS, int R>
static void launch_chanfir(vdl2hip_ctx *c, const K1Args &a, int cr, size_t lds, hipEvent_t e0, hipEvent_t e1) {
	const int groups = (c->C + cr - 1) / cr;
	K1Args b = a;
	b.gy = (groups + 3) / 4;
	const int nseg8 = (a.seg1 - a.seg0 + 7) / 8 * 8;
	dim3 grid((unsigned)(nseg8 * b.gy)), block(256);
	// e0/e1 (profiling only, else null): the runtime stamps them with the kernel's own start and stop, so the roofline figure is
	// the kernel's duration and not the time the launch spent queued behind other streams' work
	switch(cr) {
		case 4:
			// (unsigned-byte input where the tile prefetch exists: the build that fetches and converts its tiles the way the s16 build does)
			if constexpr(OS == 20 || OS == 10) { if(c->fmt == VDL2HIP_FMT_U8) { hipExtLaunchKernelGGL((k_chanfir<OS, R, 4, true>), grid, block, (uint32_t)lds, c->stream, e0, e1, 0, b); break; } }
			hipExtLaunchKernelGGL((k_chanfir<OS, R, 4>), grid, block, (uint32_t)lds, c->stream, e0, e1, 0, b); break;
		case 2: hipExtLaunchKernelGGL((k_chanfir<OS, R, 2>), grid, block, (uint32_t)lds, c->stream, e0, e1, 0, b); break;
		default: hipExtLaunchKernelGGL((k_chanfir<OS, R, 1>), grid, block, (uint32_t)lds, c->stream, e0, e1, 0, b); break;
	}
}

static int launch_back(vdl2hip_ctx *c, OutSlot &sl, hipEvent_t gate);
static int launch_rest(vdl2hip_ctx *c, OutSlot &sl, struct OutSlot *succ, struct OutSlot *succ2);
static int flush_rest(vdl2hip_ctx *c, const OutSlot *upto);
// A short feed (fewer than two walk segments' worth of samples; the reference's own 320 000-byte blocks are 4 000): launch_back()
static bool feed_is_small(const vdl2hip_ctx *c, int64_t D);

static int collect_slot(vdl2hip_ctx *c, OutSlot &sl) {
	if(!sl.pending) return VDL2HIP_OK;
	if(sl.back_queued) { int r = launch_back(c, sl, nullptr); if(r != VDL2HIP_OK) return r; }   // nothing followed this feed: its back end goes now
	if(sl.rest_pending) { int r = flush_rest(c, &sl); if(r != VDL2HIP_OK) return r; }   // ... nor did the walks that its second walks could have been queued behind (it and what is older, oldest first)
	HIPCHK(hipEventSynchronize(sl.done));
	sl.pending = false;
	if(c->profiling && sl.ev_valid) {
		hipEvent_t *ev = sl.ev;
		float ms = 0.f;
		if(sl.k1_timed && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) { c->stats.chanfir_ms += ms; c->stats.chanfir_launches++; }
		if(sl.ev_level >= 2) {
		if(!sl.fused && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) c->stats.phase_ms += ms;
		if(hipEventElapsedTime(&ms, ev[4], sl.ev_front) == hipSuccess) c->stats.sync_ms += ms;
		if(hipEventElapsedTime(&ms, ev[6], ev[7]) == hipSuccess) c->stats.walk_ms += ms;
		if(sl.small) {        // noise floor and burst decoder were one launch (k_nf_burst): booked under the burst decoder
			if(hipEventElapsedTime(&ms, ev[8], ev[11]) == hipSuccess) c->stats.burst_ms += ms;
		} else {
			if(hipEventElapsedTime(&ms, ev[8], ev[9]) == hipSuccess) c->stats.nf_ms += ms;
			if(hipEventElapsedTime(&ms, ev[10], ev[11]) == hipSuccess) c->stats.burst_ms += ms;
		}
		if(c->show_gaps) {   // development: idle time of the front stream between its kernels
			float g12 = 0, g23 = 0, g31 = -1;
			if(!sl.fused) { (void)hipEventElapsedTime(&g12, ev[1], ev[2]); (void)hipEventElapsedTime(&g23, ev[3], ev[4]); } else (void)hipEventElapsedTime(&g23, ev[1], ev[4]);
			OutSlot &pv = c->slot[(sl.seq + kSlots - 1) % kSlots];
			if(sl.seq > 0 && pv.ev_level >= 2) (void)hipEventElapsedTime(&g31, pv.ev_front, ev[0]);
			float lat = -1, per = -1, fr = -1, wk = -1, w2d = -1, w2b = -1, bb = -1, b2d = -1, w2n = -1;
			(void)hipEventElapsedTime(&w2b, ev[7], ev[10]); (void)hipEventElapsedTime(&bb, ev[10], ev[11]); (void)hipEventElapsedTime(&b2d, ev[11], sl.done); (void)hipEventElapsedTime(&w2n, ev[7], ev[8]);
			fprintf(stderr, "tail feed %llu: walk end -> burst start %.3f (nf start %.3f), burst %.3f, burst end -> done %.3f\n", (unsigned long long)sl.seq, w2b, w2n, bb, b2d);
			(void)hipEventElapsedTime(&lat, ev[0], sl.done); (void)hipEventElapsedTime(&fr, ev[0], sl.ev_front); (void)hipEventElapsedTime(&wk, sl.ev_front, ev[6]); (void)hipEventElapsedTime(&w2d, ev[7], sl.done);
			if(sl.seq > 0 && pv.ev_level >= 2) (void)hipEventElapsedTime(&per, pv.ev[0], ev[0]);
			fprintf(stderr, "gaps feed %llu: K1->K2 %.1f us, K2->K3 %.1f us, K3(prev)->K1 %.1f us | K1 start -> done %.3f ms, front %.3f, front end -> walk start %.3f, walk end -> done %.3f, K1 start (prev) -> K1 start %.3f\n", (unsigned long long)sl.seq, g12 * 1e3, g23 * 1e3, g31 * 1e3, lat, fr, wk, w2d, per);
		}
		}
	}
	if(sl.ev_valid) (void)hipGetLastError();   // (an event query that failed above is not a device error of the next feed)
	sl.ev_valid = false;
	const OutCtl ctl = sl.h_mail->ctl;
	if(ctl.overflow) c->stats.overflow_feeds++;
	c->stats.bursts += std::min(ctl.nbursts, ctl.cap_bursts);
	const uint32_t nf = std::min(ctl.nvalid, ctl.cap_frames);
	if(nf) {
		const uint32_t pool_n = std::min(ctl.pool_out_used, ctl.cap_pool);
		const OutFrame *fr; const uint8_t *po;
		if(nf <= (uint32_t)kMailFrames && pool_n <= (uint32_t)kMailPool) {
			// a handful of frames (the usual case for a block of a live receiver): they have come with the control block
			fr = sl.h_mail->frames; po = sl.h_mail->pool;
		} else {
			// one pinned staging buffer, two asynchronous copies on a stream of their own (the burst stream may already hold the
			// next feeds' kernels), one wait
			const size_t fr_bytes = sizeof(OutFrame) * (size_t)nf, need = fr_bytes + pool_n;
			if(need > c->stage_cap) {
				if(c->h_stage) (void)hipHostFree(c->h_stage);
				c->h_stage = nullptr; c->stage_cap = 0;
				size_t cap = std::max<size_t>(need + need / 2, 1u << 20);
				HIPCHK(hipHostMalloc((void **)&c->h_stage, cap, hipHostMallocDefault));
				c->stage_cap = cap;
			}
			HIPCHK(hipMemcpyAsync(c->h_stage, sl.d_frames_out, fr_bytes, hipMemcpyDeviceToHost, c->stream_out));
			if(pool_n) HIPCHK(hipMemcpyAsync(c->h_stage + fr_bytes, sl.d_pool_out, pool_n, hipMemcpyDeviceToHost, c->stream_out));
			HIPCHK(hipStreamSynchronize(c->stream_out));
			fr = reinterpret_cast<const OutFrame *>(c->h_stage); po = c->h_stage + fr_bytes;
		}
		auto pool = std::make_shared<std::vector<uint8_t>>(po, po + pool_n);
		// frames of one feed are sorted when they are drained, so that feeds can be appended to the queue in order
		c->queue.reserve(c->queue.size() + nf);
		for(uint32_t i = 0; i < nf; i++) {
			if(fr[i].chan < 0 || fr[i].chan >= c->C) continue;
			c->stats.frames++;                                           // frames the decoder produced (before the optional AVLC filter)
			if(!c->avlc_filter || fr[i].avlc_status == AVLC_OK) c->queue.push_back(HostFrame{ fr[i], pool });
		}
	}
	return ctl.overflow ? VDL2HIP_E_OVERFLOW : VDL2HIP_OK;
}

// collect every feed except the `keep` most recent ones, oldest first
static int collect_pending(vdl2hip_ctx *c, int keep = 0) {
	int rc = VDL2HIP_OK;
	for(int pass = 0; pass < kSlots; pass++) {
		OutSlot *oldest = nullptr;
		int npend = 0;
		for(auto &sl : c->slot) if(sl.pending) { npend++; if(!oldest || sl.seq < oldest->seq) oldest = &sl; }
		if(!oldest || npend <= keep) break;
		int r = collect_slot(c, *oldest);
		if(r != VDL2HIP_OK) rc = r;
		if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	}
	return rc;
}

static int feed_common(vdl2hip_ctx *c, const void *dev_in, size_t nbytes, bool in_parts = false) {
	const size_t sb = sample_bytes(c->fmt);
	const uint64_t nnew = nbytes / sb;
	const uint64_t nlogical = c->ncarry + nnew;
	const int64_t D = (int64_t)(nlogical / (uint64_t)c->os);
	const uint32_t nrem = (uint32_t)(nlogical - (uint64_t)D * c->os);
	const int seglen = 64 * c->run;
	if(c->failed) return VDL2HIP_E_DEVICE;
	OutSlot &sl = c->slot[c->feed_no % kSlots];
	{ int r = collect_slot(c, sl); if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r; }   // its buffers are about to be reused
	hipStream_t st = c->stream, sb_ = c->stream_back;
	hipEvent_t *ev = sl.ev;
	const bool prof = c->profiling != 0, prof_all = c->profiling >= 2;

	K1Args a{};
	a.in = dev_in; a.carry = c->d_carry[c->carry_sel]; a.ncarry = c->ncarry; a.nlogical = nlogical;
	a.n0 = c->n_total - c->ncarry; a.k0 = c->k_total; a.D = D;
	a.fmt = c->fmt; a.nchan = c->C; a.os = c->os; a.nseg = 0; a.gy = 1;
	a.dphi = c->d_dphi; a.lut = c->d_lut; a.bf = make_k1_consts(c->bf); a.y = c->d_y; a.seg_end = c->d_segend;
	a.qpow = c->d_qpow; a.cap = c->cap; a.mask = c->cap - 1; a.nseg_cap = c->nseg_cap;
	a.fuse = c->fuse_k2 && 64 * c->run == kFixW; a.carry_in = c->d_tcarry[c->tcarry_sel]; a.carry_out = c->d_tcarry[c->tcarry_sel ^ 1];
	a.bfd = c->d_bf; a.seg_pub = c->d_segpub; a.epoch = (uint32_t)(c->feed_no + 1); a.sync_timeouts = c->d_synctmo;
	a.pub_epoch = a.epoch; a.spin_limit = 4096;         // polls of ~1 us before a consumer helps itself: several workgroup lifetimes
	if(c->debug_force_timeout) { a.pub_epoch = a.epoch ^ 0x40000000u; a.spin_limit = 16; }   // tests only: every look-back gives up and takes the fall-back
	{
		// tiles per workgroup segment: long segments save K2 work, but the grid should still offer several thousand
		// workgroups (measured: dev/gpu_k1_tiles.sh - 2 is best at 8 channels, 8 at 256)
		const int64_t ntile = (D + seglen - 1) / seglen;
		const int groups = (c->C + c->cr - 1) / c->cr, gy = (groups + 3) / 4;
		int64_t tiles = ntile * gy / 6144; if(tiles < 1) tiles = 1; if(tiles > 8) tiles = 8;
		if(c->tiles_force) tiles = c->tiles_force;
		a.tiles = (int)tiles;
		a.nseg = (int)((ntile + tiles - 1) / tiles);
	}

	sl.ev_valid = false;
	if(D > 0) {
		const size_t lds = (size_t)c->run * c->os * 65 * sizeof(float2);   // the tile; the tables are static LDS
		// (a cold-start feed launches the channeliser in pieces that wait for the pieces of the copy: not a kernel duration, not timed)
		const bool cold = in_parts && c->cold.n > 1;
		if(cold) c->stats.cold_start_feeds++;
		sl.k1_timed = prof && !cold;
		hipEvent_t e0 = sl.k1_timed ? ev[0] : nullptr, e1 = sl.k1_timed ? ev[1] : nullptr;
		auto launch = [&](int seg0, int seg1, hipEvent_t s_ev, hipEvent_t p_ev) {
			a.seg0 = seg0; a.seg1 = seg1;
			if(c->specialised) {
				switch(c->os) {
					case 20: launch_chanfir<20, kRun>(c, a, c->cr, lds, s_ev, p_ev); break;
					case 13: launch_chanfir<13, kRun>(c, a, c->cr, lds, s_ev, p_ev); break;
					default: launch_chanfir<10, kRun>(c, a, c->cr, lds, s_ev, p_ev); break;
				}
			} else {
				launch_chanfir<0, kRunGeneric>(c, a, c->cr, lds, s_ev, p_ev);
			}
		};
		if(!in_parts || c->cold.n <= 1) launch(0, a.nseg, e0, e1);
		else {
			// cold start: piece p of the block is on the device when cold.ev[p] fires; the workgroup segments whose input lies in the
			// pieces that have arrived (whole multiples of 8, the XCD mapping) are launched behind it.  A segment reads nothing beyond its
			// own input (its tiles, the next tile's prefetch inside the segment, the look-back to the segment before), so the results are
			// those of one launch.
			const uint64_t seg_in = (uint64_t)a.tiles * (uint64_t)seglen * (uint64_t)c->os;     // logical input samples per segment
			int done = 0;
			for(int p = 0; p < c->cold.n; p++) {
				HIPCHK(hipStreamWaitEvent(st, c->cold.ev[p], 0));
				int ready = a.nseg;
				if(p < c->cold.n - 1) { const uint64_t r = (c->ncarry + c->cold.samples[p]) / seg_in; ready = r >= (uint64_t)a.nseg ? a.nseg : (int)(r & ~7ull); }
				if(ready > done) { launch(done, ready, done == 0 ? e0 : (hipEvent_t) nullptr, ready == a.nseg ? e1 : (hipEvent_t) nullptr); done = ready; }
			}
		}
		if(!a.fuse) {
			K2Args k2{ c->d_y, c->d_segend, c->d_tcarry[c->tcarry_sel], c->d_tcarry[c->tcarry_sel ^ 1], c->d_bf,
			           c->k_total, D, c->cap, c->cap - 1, c->nseg_cap, seglen * a.tiles };
			LAUNCH_EV(k_fixup, dim3((unsigned)((D + 255) / 256), (unsigned)c->C), dim3(256), st, EV(2), EV(3), k2);
		}
		c->tcarry_sel ^= 1;
	}
	if(nrem) hipLaunchKernelGGL(k_carry, dim3(1), dim3(64), 0, st, a, (void *)c->d_carry[c->carry_sel ^ 1], nrem);
	c->carry_sel ^= 1; c->ncarry = nrem;
	// Referee: this feed's hook - the raw input it may read (what the history ring holds of the stream before this block, then the
	// block where it lies) - and the block's tail appended to the ring for the feeds that follow.
	RefChan refv{}; RefChan *ref_dst = nullptr;
	if(c->d_refhist) {
		ref_dst = c->d_ref[c->feed_no % kSlots];
		refv.y = c->d_y; refv.cap = c->cap; refv.mask = c->cap - 1; refv.dphi = c->d_dphi; refv.mix = c->d_mix; refv.lut = c->d_lut;
		refv.A0 = c->lpf.A[0]; refv.A1 = c->lpf.A[1]; refv.A2 = c->lpf.A[2]; refv.B1 = c->lpf.B[1]; refv.B2 = c->lpf.B[2];
		refv.os = c->os; refv.fmt = c->fmt; refv.warm = c->ref_warm; refv.kinds = c->ref_kinds;
		refv.done = c->d_refdone; refv.done_n = c->d_refdonen; refv.stats = c->d_refstats; refv.dbg = c->d_refdbg; refv.dbg_chan = c->ref_dbg_chan;
		int np = 0;
		const int64_t blk_s0 = (int64_t)c->n_total, want0 = blk_s0 - c->ref_T;
		// (at most the two newest pieces matter - an older one ends more than ref_T samples back - and each may wrap once)
		for(size_t i = c->ref_pieces.size() > 2 ? c->ref_pieces.size() - 2 : 0; i < c->ref_pieces.size(); i++) {
			vdl2hip_ctx::HistPiece hp = c->ref_pieces[i];
			if(hp.s0 + hp.n <= want0) continue;
			if(hp.s0 < want0) { hp.pos = (hp.pos + (uint64_t)(want0 - hp.s0)) % c->ref_cap; hp.n -= want0 - hp.s0; hp.s0 = want0; }
			const uint64_t first = std::min<uint64_t>((uint64_t)hp.n, c->ref_cap - hp.pos);
			refv.piece[np++] = RefPiece{ c->d_refhist + hp.pos * sb, hp.s0, (int64_t)first };
			if(first < (uint64_t)hp.n) refv.piece[np++] = RefPiece{ c->d_refhist, hp.s0 + (int64_t)first, hp.n - (int64_t)first };
		}
		refv.piece[np++] = RefPiece{ dev_in, blk_s0, (int64_t)nnew };
		refv.npiece = np;
		// the pieces must be contiguous (a gap = a block longer than ref_T whose head was not kept): drop what lies before the last gap
		int firstp = 0;
		for(int i = 1; i < np; i++) if(refv.piece[i - 1].s0 + refv.piece[i - 1].n != refv.piece[i].s0) firstp = i;
		if(firstp) { for(int i = firstp; i < np; i++) refv.piece[i - firstp] = refv.piece[i]; refv.npiece = np - firstp; }
		if(nnew) {
			const uint64_t w = std::min<uint64_t>(nnew, (uint64_t)c->ref_T);
			hipLaunchKernelGGL(k_ref_hist, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, st, (const uint8_t *)dev_in + (nnew - w) * sb, w, c->d_refhist, c->ref_wp, c->ref_cap, (int)sb);
			if(w < nnew) c->ref_pieces.clear();                          // the head of this block is not kept: nothing before it can be reached any more
			if(!c->ref_pieces.empty() && c->ref_pieces.back().s0 + c->ref_pieces.back().n == blk_s0 + (int64_t)(nnew - w)) c->ref_pieces.back().n += (int64_t)w;
			else c->ref_pieces.push_back(vdl2hip_ctx::HistPiece{ blk_s0 + (int64_t)(nnew - w), (int64_t)w, c->ref_wp });
			c->ref_wp = (c->ref_wp + w) % c->ref_cap;
			// forget what the ring has overwritten (a piece never needs to be longer than the ring is)
			auto &bp = c->ref_pieces.back();
			if((uint64_t)bp.n > c->ref_cap / 2) { const int64_t cut = bp.n - (int64_t)(c->ref_cap / 2); bp.s0 += cut; bp.pos = (bp.pos + (uint64_t)cut) % c->ref_cap; bp.n -= cut; }
			while(c->ref_pieces.size() > 2) c->ref_pieces.erase(c->ref_pieces.begin());
		}
	}
	// deferred mode: the back end of the feed before this one is queued now, behind this feed's channeliser
	if(c->feed_no > 0) {
		OutSlot &pv = c->slot[(c->feed_no - 1) % kSlots];
		if(pv.pending && pv.back_queued) {
			hipEvent_t gate = nullptr;
			if(D > 0) { HIPCHK(hipEventRecord(sl.ev_k1, st)); gate = sl.ev_k1; }
			int r = launch_back(c, pv, gate);
			if(r != VDL2HIP_OK) return r;
		}
	}
	if(D > 0) {
		const int64_t k1 = c->k_total + D, nbase = c->k_total & ~63ll;
		// wavefronts of this feed's burst decoder: each owns kResSlots frame records of the output from the start, so a short block gets few
		sl.k5_waves = (unsigned)std::min<int64_t>(2048, std::max<int64_t>(16, (D * (int64_t)c->C) >> 14));
		sl.k5_waves = (sl.k5_waves + kBurstWaves - 1) / kBurstWaves * kBurstWaves;
		// (a short feed - launch_back: `small` - has no scans ahead of its walk: its one walk notes, the noted stretches are scanned side by side and checked on the front stream)
		sl.prescan = c->referee && c->ref_prescan && !feed_is_small(c, D);
		K3Args k3{ c->d_y, c->d_pf, c->d_cand, c->d_flag, c->d_tab, nbase, k1, c->cap, c->cap - 1, 1, sl.d_ctl, sl.k5_waves, ref_dst, refv, c->cfg.max_ppm, c->d_ppmthr, c->referee ? 1 : 0, sl.d_rqn, sl.d_rqflag, sl.d_rqbad, sl.prescan ? sl.d_pq : nullptr, kPreScans, sl.d_rqflag2 };
		// The exact tier's stop event doubles as "front of this feed done" (what the walk stream waits for): one queue entry less
		// on the front stream than a separate hipEventRecord.  (VDL2HIP_SYNC_ON=walk puts the exact tier in front of the walk on
		// the walk stream, VDL2HIP_SYNC_ON=own both sync kernels on a stream of their own, so that the front stream goes on
		// with the next feed's channeliser: measured, profiles/r02_sync_kernel_streams.txt.)
		hipStream_t s3 = c->sync_on == 2 ? c->stream_sync : st;
		if(c->sync_on == 2) { HIPCHK(hipEventRecord(sl.ev_chan, st)); HIPCHK(hipStreamWaitEvent(s3, sl.ev_chan, 0)); }
		LAUNCH_EV(k_sync_screen, dim3((unsigned)((k1 - nbase + kK3Tile - 1) / kK3Tile), (unsigned)c->C), dim3(kK3Threads), s3, EV(4), c->sync_on == 1 ? sl.ev_chan : (hipEvent_t) nullptr, k3);
		hipStream_t sx = c->sync_on == 1 ? sb_ : s3;
		if(c->sync_on == 1) HIPCHK(hipStreamWaitEvent(sb_, sl.ev_chan, 0));
		const int64_t nwords = ((k1 + 63) >> 6) - (nbase >> 6);
		// words per lane of the exact tier: as many as keep >= 8 workgroups per CU (a wavefront with more words amortises its scan,
		// but the kernel is latency-bound: at 32 channels 1 word per lane - 3 296 workgroups - beat 4 - 832 - by 0.02 ms of a 0.85 ms step,
		// at 256 channels 4 is as good as any; profiles/r03_k3b_forms.txt)
		for(int wpl = kK3bWordsPerLane; wpl >= 1; wpl >>= 1) { k3.wpl = wpl; if(((nwords + 256 * wpl - 1) / (256 * wpl)) * c->C >= 2048 || wpl == 1) break; }
		if(c->k3b_wpl) k3.wpl = c->k3b_wpl;                                   // experiments only (VDL2HIP_K3B_WPL)
		const int64_t wpb = 256 * k3.wpl;                                      // words per block
		// (receivers that scan ahead of the walk work out the windows with one or two unwrap decisions within the margin: sync_metric_ref)
		if(c->referee && c->ref_prescan) LAUNCH_EV(k_sync_exact4<true>, dim3((unsigned)((nwords + wpb - 1) / wpb), (unsigned)c->C), dim3(256), sx, (hipEvent_t) nullptr, sl.ev_front, k3);
		else LAUNCH_EV(k_sync_exact4<false>, dim3((unsigned)((nwords + wpb - 1) / wpb), (unsigned)c->C), dim3(256), sx, (hipEvent_t) nullptr, sl.ev_front, k3);
	}
	if(D <= 0) HIPCHK(hipEventRecord(sl.ev_front, st));
	if(D > 0) { sl.ev_valid = prof; sl.ev_level = c->profiling; sl.fused = a.fuse != 0; if(sl.k1_timed) c->stats.chan_samples += (uint64_t)D * c->os * c->C; } else sl.k1_timed = false;
	sl.back_queued = true; sl.back_D = D; sl.back_k0 = c->k_total;
	sl.pending = true; sl.seq = c->feed_no++;
	c->k_total += D; c->n_total += nnew;
	c->stats.feeds++; c->stats.input_samples += nnew;
	// Eager mode queues this feed's back end right away (it then runs beside the NEXT feed's channeliser); deferred mode leaves it
	// to the next feed call, which queues it behind its own channeliser (above), or to whoever collects this feed first.
	if(!c->defer_back) { int r = launch_back(c, sl, nullptr); if(r != VDL2HIP_OK) return r; }
	HIPCHK(hipGetLastError());
	return VDL2HIP_OK;
}

// The burst-rate back end of one feed (K4 walk, K4b noise floor, K5 burst decoder, frame finish), on three streams of its own so
// that consecutive feeds overlap stage by stage:
//   stream_back   K4   walk(i) -> walk(i+1) -> ...             (each needs the previous one's FSM state)
//   stream_nf     K4b  noise floor of feed i, after walk(i)    (each needs the previous one's NfState)
//   stream_burst  K5   bursts of feed i, after walk(i); the frames get their noise-floor figure when K4b(i) is done
// `gate` (deferred mode): an event of the FOLLOWING feed's front stream - the end of its channeliser - that the walk waits for as
// well, so that these latency-bound kernels run beside the sync screening (few registers: they fit in beside it) instead of taking
// workgroup slots from a channeliser (whose four waves per SIMD own the whole register file: every back-end workgroup keeps one
// channeliser workgroup off its CU for as long as it lives).
// (the kernel is compiled per sample format)
static bool feed_is_small(const vdl2hip_ctx *c, int64_t D) {
	const int nseg = (int)std::min<int64_t>(c->seg_max, D / c->seg_min);
	return D > 0 && D < 2 * c->seg_min && nseg < 2 && !c->defer_back && c->sync_on == 0;
}
#define LAUNCH_SCAN_MULTI(how, ...) do { \
	if(c->fmt == 1) { if(c->os == 20) how((k_ref_scan_multi<1, 20>), __VA_ARGS__); else if(c->os == 10) how((k_ref_scan_multi<1, 10>), __VA_ARGS__); else how((k_ref_scan_multi<1, 0>), __VA_ARGS__); } \
	else { if(c->os == 20) how((k_ref_scan_multi<0, 20>), __VA_ARGS__); else if(c->os == 10) how((k_ref_scan_multi<0, 10>), __VA_ARGS__); else how((k_ref_scan_multi<0, 0>), __VA_ARGS__); } } while(0)
// The walk of feed i + 1 does not wait for the check of feed i ("walk ahead").  walk(i + 1) needs the state walk(i) leaves, and that
// state is final only when the decisions walk(i) noted have been checked - behind a scan of 1.5 ms that no hardware shortens; with a
// front of 1 ms (a rank's 32 channels) that chain was the step.  But the check confirms the walk all but always.  So:
//   walk stream:    ... stitch(i)   again(i-1) redo(i)   stitch(i+1)   again(i) redo(i+1)   stitch(i+2) ...
//   check(i) = scans + verify on the scan stream of its slot, between stitch(i) and again(i), i.e. beside stitch(i+1);
//   again(i): the channels whose decisions did not stand (or that flagged themselves) are walked again from feed i's snapshot, into a
//     scratch row; their end state and counters are compared with what stitch(i+1) started from (feed i+1's snapshot).  The same:
//     stitch(i+1) stands.  Different (a few per thousand second walks): the snapshot of feed i+1 is corrected and the channel flagged;
//   redo(i+1): the flagged channels are stitched once more over feed i+1 from the corrected snapshot, into the live rows.
// stitch(i+2) follows on the same stream: it starts from a state in which feeds <= i are checked - the check of feed i overlaps one
// walk and one front.  launch_back() queues the walk and the check; launch_rest() - called when the NEXT feed's walk has been queued,
// or, if nothing follows (a drain, a short feed, the old schedule with VDL2HIP_WALK_AHEAD=0), at once with `succ` = nullptr: again(i)
// then works on the live rows as it always did - queues the second walks and everything behind them: noise floor, burst decoder,
// frame finish, the copy of the control block.
static int launch_rest(vdl2hip_ctx *c, OutSlot &sl, OutSlot *succ, OutSlot *succ2);

// Does the walk of the feed after this one go ahead of this feed's check?  It pays when a feed's front is shorter than a check (a scan:
// 2.3 ms) - a front is 6.2 ms for 1.68 M decimated samples of 256 channels, 1.45e-8 ms per channel-sample: receivers of 16-64 channels
// on the bench's 16 s blocks (rounds 6a/b), and ANY receiver of >= 16 channels on feeds of a second or so - the drop-in adapter's
// sixteen collected 320 000-byte blocks at 256 channels: 0.155 -> 0.10 ms per block (profiles/r06_block_batch.txt).  With fewer than
// 16 channels the walk itself is longer than the front and the second walks' extra launches cost more than they save (1.22 against
// 1.11 ms at 8).  VDL2HIP_WALK_AHEAD / the debug option fix the depth for every feed.
static int walk_ahead_of(const vdl2hip_ctx *c, int64_t D) {
	if(!c->walk_ahead_auto) return c->walk_ahead;
	return c->C >= 16 && (double)D * (double)c->C < c->walk_ahead_below ? 1 : 0;
}

static int launch_back(vdl2hip_ctx *c, OutSlot &sl, hipEvent_t gate) {
	const int64_t D = sl.back_D, k0 = sl.back_k0;
	const int wa = walk_ahead_of(c, D);
	// long feeds: the walk runs in speculative segments (vdl2_core.h), one wavefront per (channel, segment, grid phase)
	int nseg = (int)std::min<int64_t>(c->seg_max, D / c->seg_min);
	// A short feed (fewer than two walk segments' worth of samples; the reference's own 320 000-byte blocks are 4 000) is a chain of kernels that each run for
	// microseconds: its whole back end goes onto the FRONT stream, behind its own sync kernels - no event hand-offs between streams -
	// with the three noise-floor passes as one kernel.  Whoever follows on the front stream is then behind it anyway.
	// (... and only when both sync kernels ARE on the front stream - sync_on 0, the product; the experiment builds' other placements hand the
	// candidate bitmap over with an event the short cut does not wait for)
	const bool small = feed_is_small(c, D);
	hipStream_t sb_ = small ? c->stream : c->stream_back;
	hipEvent_t *ev = sl.ev;
	const bool prof_all = sl.ev_valid && sl.ev_level >= 2;
	sl.back_queued = false; sl.small = small; sl.has_chk = false; sl.nseg = 1; sl.seglen = D;
	OutSlot &pv = c->slot[(sl.seq + kSlots - 1) % kSlots];
	// (short feeds too, since round 6: a block with a decision within the margin - one in three of the reference's own 320 000-byte
	// blocks at 256 channels - used to wait for a scan by the walker's own wavefront, 3.1 ms; noted, scanned side by side (1.5 ms for all
	// of them) and checked, on the front stream like the rest of a short feed's back end, it is 1.34 -> 0.8 ms per block on average)
	const bool opt = c->referee && c->ref_optimistic && D > 0;
	const uint32_t rq_cap = small ? std::min<uint32_t>(c->rq_cap, 16u * kScanLanes) : c->rq_cap;      // (a short feed: few requests, small grids - it queues these kernels whether or not it notes anything)
	// does this feed's walk go ahead of the check of the feed(s) before?  (long feeds with a check, segmented walks)  If not, what is
	// still waiting for a successor's walk is queued now, oldest first
	const bool ahead = wa && opt && nseg >= 2 && !gate && sl.seq > 0 && pv.pending && pv.rest_pending && pv.has_chk && pv.nseg >= 2;
	if(!ahead) { int r = flush_rest(c, nullptr); if(r != VDL2HIP_OK) return r; }
	if(small) {
		// the walker, the noise floor and the burst list carry state from feed to feed: wait for a predecessor whose back end is on the other streams
		if(sl.seq > 0 && pv.pending && !pv.small && !pv.back_queued) HIPCHK(hipStreamWaitEvent(sb_, pv.done, 0));
	} else {
		HIPCHK(hipStreamWaitEvent(sb_, sl.ev_front, 0));
		if(gate) HIPCHK(hipStreamWaitEvent(sb_, gate, 0));
	}
	if(D <= 0) hipLaunchKernelGGL(k_reset_ctl, dim3(1), dim3(1), 0, sb_, sl.d_ctl, 0u);     // (a feed with a front has had it reset by its last sync kernel)
	hipStream_t sp_ = small ? c->stream : c->stream_pre[sl.seq % kSidePre];
	const int rty = c->ref_retry_mul;
	if(D > 0 && sl.prescan) {
		// Referee: the stretches the exact sync tier has listed (around its marked candidates) are made the reference's own NOW, beside
		// the next feed's front and off the walk stream - the walk of this feed waits for them, the walk of the next one does not
		// (a scan on the walk stream is 1.5 ms that every following feed's walk queues behind: with 8 channels that was the step time)
		// (on a stream of its own: behind this feed's noise floor - which waits for the walk - the next feed's scans would wait for this
		// feed's whole walk chain)
		if(!small) HIPCHK(hipStreamWaitEvent(sp_, sl.ev_front, 0));
		LAUNCH_SCAN_MULTI(hipLaunchKernelGGL, dim3(kPreScans / kScanLanes), dim3(64 * kScanWaves), 0, sp_, c->d_ref[sl.seq % kSlots], (uint32_t)(16 * sl.seq + 8),
		                  (const ScanReq *)sl.d_pq, (const RefReq *) nullptr, (const uint32_t *)(sl.d_rqn + 3), kPreScans, (int64_t)(k0 + D), rty ? sl.d_retry : (ScanReq *) nullptr, sl.d_rqn + 4, kRetryScans, 1);
		// (... and those of them that had not met their witness - a few in ten thousand - again, from further back)
		if(rty) LAUNCH_SCAN_MULTI(hipLaunchKernelGGL, dim3(kRetryScans / kScanLanes), dim3(64 * kScanWaves), 0, sp_, c->d_ref[sl.seq % kSlots], (uint32_t)(16 * sl.seq + 8),
		                  (const ScanReq *)sl.d_retry, (const RefReq *) nullptr, (const uint32_t *)(sl.d_rqn + 4), kRetryScans, (int64_t)(k0 + D), (ScanReq *) nullptr, (uint32_t *) nullptr, 0u, rty);
		if(!small) { HIPCHK(hipEventRecord(sl.ev_pre, sp_)); HIPCHK(hipStreamWaitEvent(sb_, sl.ev_pre, 0)); }
	}
	if(D > 0) {
		const int64_t k1 = k0 + D;
		const int par = (int)(sl.seq % 3);        // feed i's snapshot and speculative walks are still needed when feeds i + 1 and i + 2 are walked: three of each
		sl.k4 = K4Args{ c->d_y, c->d_pf, c->d_cand, c->d_tab, c->d_ws, c->d_wcnt, sl.d_bursts, sl.d_nbchan, c->cap_bursts_chan, sl.d_ctl, c->d_freq,
		           sl.d_log, sl.d_nlog, c->cap_log, k1, c->cfg.max_ppm, c->cap, c->cap - 1, c->chan_first, c->C, c->d_ppmthr, c->referee ? c->d_ref[sl.seq % kSlots] : nullptr, (uint32_t)(16 * sl.seq + 1),
		           opt ? sl.d_rq : nullptr, sl.d_rqn, rq_cap, sl.d_rqflag, c->d_ws_snap[par], c->d_cnt_snap[par], sl.d_rqbad, sl.prescan ? 1 : 0, c->debug_force_again,
		           c->d_ws_tmp, c->d_cnt_tmp, nullptr, nullptr, sl.d_rqflag2, nullptr, nullptr, c->debug_force_mismatch };   // (a short feed's walk notes and is checked like a long one's since round 6a: `opt`)
		const K4Args &k4 = sl.k4;
		sl.d_spec_of = c->d_spec[par];
		if(nseg >= 2) {
			sl.seglen = (D + nseg - 1) / nseg;
			nseg = (int)((D + sl.seglen - 1) / sl.seglen);
			sl.nseg = nseg;
			K4sArgs k4s{ k4, sl.d_spec_of, (uint32_t)(3 * (c->seg_max - 1)), nseg, k0, sl.seglen, c->d_segstats, 0 };
			if(!(c->ablate & 1))
			LAUNCH_EV(k_walk_spec, dim3((unsigned)((1 + 3 * (nseg - 1) + kWalkWaves - 1) / kWalkWaves), (unsigned)c->C), dim3(64 * kWalkWaves), sb_, EV(6), (hipEvent_t) nullptr, k4s);
			if(!(c->ablate & 1))
			hipExtLaunchKernelGGL(k_walk_stitch, dim3((unsigned)((c->C + kStitchWaves - 1) / kStitchWaves)), dim3(64 * kStitchWaves), (unsigned)(sizeof(StitchLds) * kStitchWaves), sb_, (hipEvent_t) nullptr, EV(7), 0, k4s);
		} else {
			LAUNCH_EV(k_walk, dim3((unsigned)c->C), dim3(64), sb_, EV(6), EV(7), k4);
		}
		if(k4.rq) {
			// referee, optimistic mode (long feeds): the decisions within the margin that the walk took are checked, all at once - the
			// stretches first, many side by side (k_ref_scan_multi), then the decisions on them, a wavefront each - on the scan stream of
			// the feed's slot (idle since the scans ahead of the walk): the walk stream goes on with the next feed meanwhile
			// (in the old schedule - receivers of many channels: the front hides the chain - the check stays on the walk stream: on a stream
			// of its own it cost the 256-channel receiver 0.6 ms per step, profiles/r06_walk_ahead_ab.txt)
			sl.has_chk = true;
			hipStream_t sc_ = wa ? sp_ : sb_;
			if(sc_ != sb_) { HIPCHK(hipEventRecord(sl.ev_stitch, sb_)); HIPCHK(hipStreamWaitEvent(sc_, sl.ev_stitch, 0)); }
			LAUNCH_SCAN_MULTI(hipLaunchKernelGGL, dim3(rq_cap / kScanLanes), dim3(64 * kScanWaves), 0, sc_, k4.ref, k4.ref_launch - 1u, (const ScanReq *) nullptr, (const RefReq *)k4.rq, (const uint32_t *)k4.rq_n, rq_cap, k1,
			                  rty ? sl.d_retry + kRetryScans : (ScanReq *) nullptr, sl.d_rqn + 5, kRetryScans, 1);
			if(rty) LAUNCH_SCAN_MULTI(hipLaunchKernelGGL, dim3(kRetryScans / kScanLanes), dim3(64 * kScanWaves), 0, sc_, k4.ref, k4.ref_launch - 1u, (const ScanReq *)(sl.d_retry + kRetryScans), (const RefReq *) nullptr, (const uint32_t *)(sl.d_rqn + 5), kRetryScans, k1,
			                  (ScanReq *) nullptr, (uint32_t *) nullptr, 0u, rty);
			hipLaunchKernelGGL(k_ref_verify, dim3(small ? 64 : 1024), dim3(64), 0, sc_, k4);
			HIPCHK(hipEventRecord(sl.ev_chk, sc_));
		}
	}
	sl.rest_pending = true;
	// the feeds whose second walks have now waited for as many successors' walks as they may (walk_ahead: 1 or 2) ...
	for(;;) {
		OutSlot *old = nullptr;
		for(auto &x : c->slot) if(x.pending && x.rest_pending && (!old || x.seq < old->seq)) old = &x;
		if(!old || old == &sl || sl.seq - old->seq < (uint64_t)wa) break;
		int r = flush_rest(c, old); if(r != VDL2HIP_OK) return r;
	}
	// does the NEXT feed's walk get the chance to go ahead of this feed's check?  Not if there is nothing to check, nor in the old schedule
	if(!(wa && sl.has_chk && sl.nseg >= 2 && !gate && !c->defer_back)) return flush_rest(c, nullptr);
	HIPCHK(hipGetLastError());
	return VDL2HIP_OK;
}

// The feeds whose walk and check are queued and the rest is not (rest_pending), oldest first, up to and including `upto` (nullptr: all
// of them): each with the one or two feeds behind it whose walks are queued as its successors (their walks went ahead of its check).
static int flush_rest(vdl2hip_ctx *c, const OutSlot *upto) {
	for(;;) {
		OutSlot *old = nullptr;
		for(auto &x : c->slot) if(x.pending && x.rest_pending && (!old || x.seq < old->seq)) old = &x;
		if(!old || (upto && old->seq > upto->seq)) return VDL2HIP_OK;
		OutSlot *s1 = &c->slot[(old->seq + 1) % kSlots], *s2 = &c->slot[(old->seq + 2) % kSlots];
		if(!(s1->pending && s1->rest_pending && s1->seq == old->seq + 1)) s1 = nullptr;
		if(!(s1 && s2->pending && s2->rest_pending && s2->seq == old->seq + 2)) s2 = nullptr;
		int r = launch_rest(c, *old, s1, s2); if(r != VDL2HIP_OK) return r;
	}
}

static int launch_rest(vdl2hip_ctx *c, OutSlot &sl, OutSlot *succ, OutSlot *succ2) {
	const int64_t D = sl.back_D, k0 = sl.back_k0;
	const bool small = sl.small;
	hipStream_t sb_ = small ? c->stream : c->stream_back, sn_ = small ? c->stream : c->stream_nf, s5_ = small ? c->stream : c->stream_burst[sl.seq % kSideBurst];

	hipEvent_t *ev = sl.ev;
	const bool prof_all = sl.ev_valid && sl.ev_level >= 2;
	sl.rest_pending = false;
	const int rty = c->ref_retry_mul;
	if(D > 0 && sl.has_chk) {
		// the (rare) channel one of whose decisions did not stand is stitched again
		HIPCHK(hipStreamWaitEvent(sb_, sl.ev_chk, 0));
		K4Args k4 = sl.k4;
		if(succ) { k4.ws_snap_next = succ->k4.ws_snap; k4.cnt_snap_next = succ->k4.cnt_snap; k4.rq_flag2_next = succ->d_rqflag2; k4.rq_flag2_next2 = succ2 ? succ2->d_rqflag2 : nullptr; }
		if(sl.nseg >= 2) {
			K4sArgs k4a{ k4, sl.d_spec_of, (uint32_t)(3 * (c->seg_max - 1)), sl.nseg, k0, sl.seglen, c->d_segstats, succ ? 2 : 1 };
			hipExtLaunchKernelGGL(k_walk_stitch, dim3((unsigned)((c->C + kStitchWaves - 1) / kStitchWaves)), dim3(64 * kStitchWaves), (unsigned)(sizeof(StitchLds) * kStitchWaves), sb_, (hipEvent_t) nullptr, (hipEvent_t) nullptr, 0, k4a);
		} else hipLaunchKernelGGL(k_walk_again, dim3((unsigned)c->C), dim3(64), 0, sb_, k4);      // (never with a successor: launch_back)
	}
	if(!small) {
		HIPCHK(hipEventRecord(sl.ev_walk, sb_));
		HIPCHK(hipStreamWaitEvent(sn_, sl.ev_walk, 0));
		HIPCHK(hipStreamWaitEvent(s5_, sl.ev_walk, 0));
	}
	if(succ) {
		// ... and the next feed once more for the channels whose start state that has corrected (none, all but always: the kernel looks at the flags and ends)
		// (walk ahead by two: the feed after it has been walked as well - the next feed's end state then goes into THAT feed's snapshot, and it is stitched once more too)
		K4sArgs k4r{ succ->k4, succ->d_spec_of, (uint32_t)(3 * (c->seg_max - 1)), succ->nseg, succ->back_k0, succ->seglen, c->d_segstats, succ2 ? 4 : 3 };
		if(succ2) { k4r.k.ws_snap_next = succ2->k4.ws_snap; k4r.k.cnt_snap_next = succ2->k4.cnt_snap; }
		hipExtLaunchKernelGGL(k_walk_stitch, dim3((unsigned)((c->C + kStitchWaves - 1) / kStitchWaves)), dim3(64 * kStitchWaves), (unsigned)(sizeof(StitchLds) * kStitchWaves), sb_, (hipEvent_t) nullptr, (hipEvent_t) nullptr, 0, k4r);
		if(succ2) {
			K4sArgs k4q{ succ2->k4, succ2->d_spec_of, (uint32_t)(3 * (c->seg_max - 1)), succ2->nseg, succ2->back_k0, succ2->seglen, c->d_segstats, 5 };
			hipExtLaunchKernelGGL(k_walk_stitch, dim3((unsigned)((c->C + kStitchWaves - 1) / kStitchWaves)), dim3(64 * kStitchWaves), (unsigned)(sizeof(StitchLds) * kStitchWaves), sb_, (hipEvent_t) nullptr, (hipEvent_t) nullptr, 0, k4q);
		}
	}
	if(D > 0) {
		K4bArgs k4b{ c->d_y, c->d_nf, sl.d_log, sl.d_nlog, c->d_scfirst, c->d_sccum, c->d_nffeed, c->d_lpbuf, c->d_nfring, c->nf_ring - 1,
		             c->cap, c->cap - 1, c->cap_log, c->cap_comb, c->cap_hist, c->C };
		const unsigned nf_lds = (unsigned)(sizeof(NfShared) * kNfWaves), nf_grid = (unsigned)((c->C + kNfWaves - 1) / kNfWaves);
		if(small) {
			// (queued below, together with the burst decoder: k_nf_burst)
		} else {
			hipExtLaunchKernelGGL(k_nf_prepare, dim3(nf_grid), dim3(64 * kNfWaves), nf_lds, sn_, EV(8), (hipEvent_t) nullptr, 0, k4b);
			const unsigned ngrp = (unsigned)std::min<uint64_t>(64, (c->cap_hist + kNfGroup * kNfWaves - 1) / (kNfGroup * kNfWaves));   // workgroups per channel
			if(!(c->ablate & 2))      // (experiment builds: what does a stage cost the front by running beside it?)
			hipLaunchKernelGGL(k_nf_replay, dim3(ngrp, (unsigned)c->C), dim3(64 * kNfWaves), nf_lds, sn_, k4b);
			hipExtLaunchKernelGGL(k_nf_finish, dim3(nf_grid), dim3(64 * kNfWaves), nf_lds, sn_, (hipEvent_t) nullptr, EV(9), 0, k4b);
			HIPCHK(hipEventRecord(sl.ev_nf, sn_));
		}
		K5Args k5{ c->d_y, c->d_tab, c->d_cnt, sl.d_bursts, sl.d_nbchan, c->cap_bursts_chan, c->C,
		           sl.d_frames, sl.d_pool, sl.d_ctl, c->d_freq, c->cap, c->cap - 1, c->referee ? c->d_ref[sl.seq % kSlots] : nullptr, (uint32_t)(16 * sl.seq + 5), BurstDefer{} };
		// referee: a burst that needs a scan is listed by the first pass, the scans run side by side, a second pass decodes the listed bursts.
		// (Short feeds too, since round 6c: their burst wavefronts used to scan on the spot - a weak burst with ten marked symbols held its
		// block for 22 ms, profiles/r06_weak_bursts.txt; their lists and the grids that serve them are a sixteenth of a long feed's.)
		const bool defer5 = c->referee && c->ref_optimistic && ((c->ref_kinds >> REF_SYMBOLS) & 1);
		const uint32_t sq_cap = small ? kDeferScans / 16 : std::min(c->sq_alloc, defer_scans_for(D, c->C)), dq_cap = sq_cap / 2;
		if(defer5) k5.df = BurstDefer{ sl.d_dq, sl.d_rqn + 1, dq_cap, sl.d_sq, sl.d_rqn + 2, sq_cap, 1 };
		if(c->ablate & 4) k5.nchan = 0;      // (experiment builds: no bursts to decode)
		const unsigned k5_lds = (unsigned)((sizeof(BurstShared) + 4 * (kK5MaxChan + 1)) * kBurstWaves);
		hipEvent_t ev_last5 = defer5 ? (hipEvent_t) nullptr : EV(11);
		if(small) hipExtLaunchKernelGGL(k_nf_burst, dim3(nf_grid + sl.k5_waves / kBurstWaves), dim3(64 * kNfWaves), std::max(nf_lds, k5_lds), s5_, EV(8), ev_last5, 0, k4b, k5, (uint32_t)nf_grid);
		else hipExtLaunchKernelGGL(k_burst, dim3(sl.k5_waves / kBurstWaves), dim3(64 * kBurstWaves), k5_lds, s5_, EV(10), ev_last5, 0, k5);
		if(defer5) {
			LAUNCH_SCAN_MULTI(hipLaunchKernelGGL, dim3(sq_cap / kScanLanes), dim3(64 * kScanWaves), 0, s5_, k5.ref, (uint32_t)(16 * sl.seq + 6), (const ScanReq *)sl.d_sq, (const RefReq *) nullptr, (const uint32_t *)(sl.d_rqn + 2), sq_cap, (int64_t)(k0 + D), rty ? sl.d_retry + 2 * kRetryScans : (ScanReq *) nullptr, sl.d_rqn + 6, kRetryScans, 1);
			if(rty) LAUNCH_SCAN_MULTI(hipLaunchKernelGGL, dim3(kRetryScans / kScanLanes), dim3(64 * kScanWaves), 0, s5_, k5.ref, (uint32_t)(16 * sl.seq + 6), (const ScanReq *)(sl.d_retry + 2 * kRetryScans), (const RefReq *) nullptr, (const uint32_t *)(sl.d_rqn + 6), kRetryScans, (int64_t)(k0 + D),
			                  (ScanReq *) nullptr, (uint32_t *) nullptr, 0u, rty);
			K5Args k5b = k5; k5b.df.pass = 2; k5b.ref_launch = (uint32_t)(16 * sl.seq + 7);
			hipExtLaunchKernelGGL(k_burst, dim3(std::max(1u, dq_cap / kBurstWaves / 4)), dim3(64 * kBurstWaves), k5_lds, s5_, (hipEvent_t) nullptr, EV(11), 0, k5b);
		}
		if(!small) HIPCHK(hipStreamWaitEvent(s5_, sl.ev_nf, 0));
		// record chunks of kFrameChunk: enough workgroups for the records the burst decoder's wavefronts own, at most 256
		const unsigned ff_grid = std::min(256u, (sl.k5_waves * (unsigned)kResSlots * 2 / kFrameChunk + kFrameWaves - 1) / kFrameWaves);
		hipLaunchKernelGGL(k_frame_finish, dim3(ff_grid), dim3(64 * kFrameWaves), 0, s5_, sl.d_frames, (const uint8_t *)sl.d_pool, sl.d_mail, (const Tables *)c->d_tab,
		                   c->d_acnt, (const float *)c->d_nfring, c->nf_ring - 1, sl.d_frames_out, sl.d_pool_out);
	}
	// the control block and the first few delivered frames in one copy (collect_slot)
	HIPCHK(hipMemcpyAsync(sl.h_mail, sl.d_mail, sizeof(OutMail), hipMemcpyDeviceToHost, s5_));
	HIPCHK(hipEventRecord(sl.done, s5_));
	HIPCHK(hipGetLastError());
	return VDL2HIP_OK;
}

static void sort_queue(vdl2hip_ctx *c) {
	std::stable_sort(c->queue.begin(), c->queue.end(), [](const HostFrame &a, const HostFrame &b) {
		if(a.f.end_sample != b.f.end_sample) return a.f.end_sample < b.f.end_sample;
		if(a.f.chan != b.f.chan) return a.f.chan < b.f.chan;
		return a.f.idx < b.f.idx;
	});
}

static void fill_frame(const vdl2hip_ctx *c, const HostFrame &h, vdl2hip_frame &f) {
	f.chan = (uint32_t)(h.f.chan + c->chan_first); f.freq = c->freqs[h.f.chan]; f.idx = h.f.idx;
	f.len = h.f.len; f.octets = h.octets();
	f.synd_weight = h.f.synd_weight; f.datalen_octets = h.f.datalen_octets; f.num_fec_corrections = h.f.num_fec_corrections;
	f.frame_pwr_dbfs = h.f.frame_pwr_dbfs; f.nf_pwr_dbfs = h.f.nf_pwr_dbfs; f.ppm_error = h.f.ppm_error;
	f.burst_ord = h.f.burst_ord; f.sync_sample = h.f.sync_sample; f.end_sample = h.f.end_sample;
	f.avlc_status = h.f.avlc_status; f.dst_addr = h.f.dst_addr; f.src_addr = h.f.src_addr;
}

namespace {
// the streams of one receiver (see kSidePre): kept when the receiver goes, taken over by the next one on the same device
struct StreamSet { int device = -1; hipStream_t front = nullptr, copy = nullptr, out = nullptr, back = nullptr, nf = nullptr, pre[kSidePre] = {}, burst[kSideBurst] = {}; };
std::mutex g_pool_mutex;
std::vector<StreamSet> g_pool;
}  // namespace

extern "C" {

int vdl2hip_abi_version(void) { return VDL2HIP_ABI_VERSION; }

const char *vdl2hip_strerror(int err) {
	switch(err) {
		case VDL2HIP_OK: return "ok";
		case VDL2HIP_E_INVAL: return "invalid argument";
		case VDL2HIP_E_NOMEM: return "out of memory";
		case VDL2HIP_E_DEVICE: return "HIP device error";
		case VDL2HIP_E_TOOBIG: return "block larger than max_block_bytes";
		case VDL2HIP_E_OVERFLOW: return "device output buffer overflow";
		default: return "unknown error";
	}
}

void vdl2hip_destroy(vdl2hip_ctx *c) {
	if(!c) return;
	OnDevice dev_guard(c);
	if(c->stream) (void)hipStreamSynchronize(c->stream);
	void *ptrs[] = { c->d_bf, c->d_lut, c->d_tab, c->d_dphi, c->d_freq, c->d_ppmthr, c->d_carry[0], c->d_carry[1], c->d_y, c->d_pf,
	                 c->d_cand, c->d_flag, c->d_segend, c->d_qpow, c->d_tcarry[0], c->d_tcarry[1], c->d_ws, c->d_cnt, c->d_nf, c->d_scfirst, c->d_sccum, c->d_nfring, c->d_lpbuf, c->d_nffeed, c->d_spec[0], c->d_spec[1], c->d_spec[2], c->d_segstats, c->d_acnt, c->d_segpub, c->d_synctmo };
	for(auto &sl : c->slot) {
		void *q[] = { sl.d_bursts, sl.d_nbchan, sl.d_frames, sl.d_pool, sl.d_frames_out, sl.d_pool_out, sl.d_mail, sl.d_log, sl.d_nlog, sl.d_rq, sl.d_rqn, sl.d_rqflag, sl.d_dq, sl.d_sq, sl.d_rqbad, sl.d_pq, sl.d_rqflag2, sl.d_retry };
		for(void *p : q) if(p) (void)hipFree(p);
		if(sl.ev_stitch) (void)hipEventDestroy(sl.ev_stitch);
		if(sl.ev_chk) (void)hipEventDestroy(sl.ev_chk);
		if(sl.h_mail) (void)hipHostFree(sl.h_mail);
		if(sl.done) (void)hipEventDestroy(sl.done);
		if(sl.ev_walk) (void)hipEventDestroy(sl.ev_walk);
		if(sl.ev_front) (void)hipEventDestroy(sl.ev_front);
		if(sl.ev_pre) (void)hipEventDestroy(sl.ev_pre);
		if(sl.ev_chan) (void)hipEventDestroy(sl.ev_chan);
		if(sl.ev_nf) (void)hipEventDestroy(sl.ev_nf);
		if(sl.ev_k1) (void)hipEventDestroy(sl.ev_k1);
		for(int i = 0; i < kNumEv; i++) if(sl.ev[i]) (void)hipEventDestroy(sl.ev[i]);
	}
	if(c->h_stage) (void)hipHostFree(c->h_stage);
	for(auto &p : c->d_in) if(p) (void)hipFree(p);
	for(auto &p : c->d_ref) if(p) (void)hipFree(p);
	{ void *q[] = { c->d_refhist, c->d_refdone, c->d_refdonen, c->d_refstats, c->d_mix, c->d_refdbg, c->d_ws_snap[0], c->d_ws_snap[1], c->d_ws_snap[2], c->d_cnt_snap[0], c->d_cnt_snap[1], c->d_cnt_snap[2], c->d_ws_tmp, c->d_cnt_tmp }; for(void *p : q) if(p) (void)hipFree(p); }
	for(auto &e : c->ev_copied) if(e) (void)hipEventDestroy(e);
	for(auto &e : c->cold.ev) if(e) (void)hipEventDestroy(e);
	for(hipStream_t st_ : { c->stream_copy, c->stream_out, c->stream_sync, c->stream_back, c->stream_nf }) if(st_) (void)hipStreamSynchronize(st_);
	for(auto &sp : c->stream_pre) if(sp) (void)hipStreamSynchronize(sp);
	for(auto &sb5 : c->stream_burst) if(sb5) (void)hipStreamSynchronize(sb5);
	if(c->stream_sync) (void)hipStreamDestroy(c->stream_sync);
	for(void *p : ptrs) if(p) (void)hipFree(p);
	if(c->stream && c->pooled) {
		StreamSet ss; ss.device = c->cfg.device; ss.front = c->stream; ss.copy = c->stream_copy; ss.out = c->stream_out; ss.back = c->stream_back; ss.nf = c->stream_nf;
		for(int i = 0; i < kSidePre; i++) ss.pre[i] = c->stream_pre[i];
		for(int i = 0; i < kSideBurst; i++) ss.burst[i] = c->stream_burst[i];
		std::lock_guard<std::mutex> lk(g_pool_mutex);
		g_pool.push_back(ss);
	} else {
		for(hipStream_t st_ : { c->stream_copy, c->stream_out, c->stream_back, c->stream_nf, c->stream }) if(st_) (void)hipStreamDestroy(st_);
		for(auto &sp : c->stream_pre) if(sp) (void)hipStreamDestroy(sp);
		for(auto &sb5 : c->stream_burst) if(sb5) (void)hipStreamDestroy(sb5);
	}
	delete c;
}

int vdl2hip_create(const vdl2hip_cfg *cfg, vdl2hip_ctx **out) {
	if(!cfg || !out || cfg->struct_size < sizeof(vdl2hip_cfg) || !cfg->freqs || cfg->nchan == 0) return VDL2HIP_E_INVAL;
	if(cfg->oversample == 0 || cfg->oversample > (uint32_t)kMaxOversample) return VDL2HIP_E_INVAL;
	if(cfg->sample_fmt != VDL2HIP_FMT_U8 && cfg->sample_fmt != VDL2HIP_FMT_S16LE) return VDL2HIP_E_INVAL;
	uint32_t first = cfg->chan_first, count = cfg->chan_count ? cfg->chan_count : cfg->nchan - first;
	if(first >= cfg->nchan || first + count > cfg->nchan || count > (uint32_t)kK5MaxChan) return VDL2HIP_E_INVAL;
	*out = nullptr;
	int ndev = 0;
	if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		fprintf(stderr, "vdl2hip: no HIP device available - this library has no CPU path\n");
		return VDL2HIP_E_DEVICE;
	}
	if(cfg->device < 0 || cfg->device >= ndev) return VDL2HIP_E_INVAL;
	struct Restore { int prev = -1; Restore() { (void)hipGetDevice(&prev); } ~Restore() { if(prev >= 0) (void)hipSetDevice(prev); } } restore_device;
	HIPCHK(hipSetDevice(cfg->device));
	vdl2hip_ctx *c = new(std::nothrow) vdl2hip_ctx();
	if(!c) return VDL2HIP_E_NOMEM;
	c->cfg = *cfg; c->cfg.freqs = nullptr;
	c->C = (int)count; c->chan_first = (int)first; c->os = (int)cfg->oversample; c->fmt = (int)cfg->sample_fmt;
	c->freqs.assign(cfg->freqs + first, cfg->freqs + first + count);
	const uint32_t fs = (uint32_t)kSymbolRate * kSps * cfg->oversample;
	c->lpf = design_lpf(8000.f / (float)fs, 0.5f);                // input_lpf_init(), demod.c:45-46,367-370
	c->specialised = (c->os == 10 || c->os == 13 || c->os == 20);
	c->run = c->specialised ? kRun : kRunGeneric;
	c->cr = c->C >= 16 ? 4 : c->C >= 8 ? 2 : 1;                   // channels per wave: keep >= 4 channel groups where possible
#ifdef VDL2_EXPERIMENTS
	if(const char *e = getenv("VDL2HIP_CR")) { int v = atoi(e); if(v == 1 || v == 2 || v == 4) c->cr = v; }
#endif
	c->bf = derive_block_form(c->lpf, c->os, c->run);
	c->dphi.resize(count);
	for(uint32_t i = 0; i < count; i++) c->dphi[i] = nco_step(cfg->centerfreq, c->freqs[i], fs) & 0xffffffu;

	const uint32_t max_bytes = cfg->max_block_bytes ? cfg->max_block_bytes : 320000u;
	const size_t sb = sample_bytes(c->fmt);
	const uint64_t max_samples = max_bytes / sb + c->os;
	const uint64_t dmax = max_samples / c->os + 1;
	uint32_t cap = 1; while(cap < kSlots * dmax + kHistory + 1024) cap <<= 1;   // kSlots feeds may be in flight (fronts of i+1, i+2 over back of i)
	c->cap = cap;
	c->in_cap = max_bytes;
	c->nseg_cap = (uint32_t)(dmax / (64 * c->run) + 2);

	#define DEV_ALLOC(ptr, bytes) do { if(hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) { vdl2hip_destroy(c); return VDL2HIP_E_NOMEM; } } while(0)
	#define DEV_CHK(expr) do { if((expr) != hipSuccess) { vdl2hip_destroy(c); return VDL2HIP_E_DEVICE; } } while(0)
	{
		int prio_low = 0, prio_high = 0;
		DEV_CHK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
		const char *lowp = "nf,burst";
#ifdef VDL2_EXPERIMENTS
		if(getenv("VDL2HIP_NO_PRIO")) prio_low = prio_high = 0;
		if(getenv("VDL2HIP_LOW_PRIO")) lowp = getenv("VDL2HIP_LOW_PRIO");          // list of nf,burst,walk
		if(const char *e = getenv("VDL2HIP_SYNC_ON")) c->sync_on = strcmp(e, "walk") == 0 ? 1 : strncmp(e, "own", 3) == 0 ? 2 : 0;   // front | walk | own | own-high
		if(const char *e = getenv("VDL2HIP_K3B_WPL")) { int v = atoi(e); if(v == 1 || v == 2 || v == 4) c->k3b_wpl = v; }
		if(const char *e = getenv("VDL2HIP_K1_TILES")) { long v = atol(e); if(v >= 1 && v <= 64) c->tiles_force = (int)v; }
		if(const char *e = getenv("VDL2HIP_ABLATE")) c->ablate = (strstr(e, "walk") ? 1 : 0) | (strstr(e, "nf") ? 2 : 0) | (strstr(e, "burst") ? 4 : 0);
		c->show_gaps = getenv("VDL2HIP_GAPS") != nullptr;
#endif
		int prio_front = prio_low;
#ifdef VDL2_EXPERIMENTS
		if(const char *e = getenv("VDL2HIP_FRONT_PRIO")) { if(strcmp(e, "mid") == 0) prio_front = (prio_low + prio_high) / 2; else if(strcmp(e, "high") == 0) prio_front = prio_high; }
		if(getenv("VDL2HIP_GAPS")) fprintf(stderr, "vdl2hip: stream priority range low %d .. high %d, front %d\n", prio_low, prio_high, prio_front);
#endif
		bool from_pool = false;
#ifndef VDL2_EXPERIMENTS
		{
			std::lock_guard<std::mutex> lk(g_pool_mutex);
			for(size_t i = 0; i < g_pool.size(); i++) if(g_pool[i].device == cfg->device) {
				const StreamSet ss = g_pool[i]; g_pool.erase(g_pool.begin() + (long)i);
				c->stream = ss.front; c->stream_copy = ss.copy; c->stream_out = ss.out; c->stream_back = ss.back; c->stream_nf = ss.nf;
				for(int k = 0; k < kSidePre; k++) c->stream_pre[k] = ss.pre[k];
				for(int k = 0; k < kSideBurst; k++) c->stream_burst[k] = ss.burst[k];
				from_pool = true;
				break;
			}
		}
#endif
		if(!from_pool) {
		DEV_CHK(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_front));
		// The walk goes first whenever it competes with the channeliser of a later feed (every later stage waits for it).  The
		// noise-floor and burst streams do not: their many single-wave workgroups, dispatched with priority, each take the
		// register slot of one of the four waves a channeliser workgroup needs on a CU and so keep whole workgroups out; at
		// the front's priority they fill in while the sync kernels (few registers) run.  Measured at 256 channels
		// (profiles/r02_stream_priorities.txt): all three high 7.10 ms/step, walk only 6.91, none 6.93; no difference at 8.
		auto prio_of = [&](const char *name) { return strstr(lowp, name) ? prio_low : prio_high; };
		DEV_CHK(hipStreamCreateWithPriority(&c->stream_back, hipStreamNonBlocking, prio_of("walk")));
		DEV_CHK(hipStreamCreateWithPriority(&c->stream_nf, hipStreamNonBlocking, prio_of("nf")));
		for(auto &sp : c->stream_pre) DEV_CHK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, prio_high));
		for(auto &sb5 : c->stream_burst) DEV_CHK(hipStreamCreateWithPriority(&sb5, hipStreamNonBlocking, prio_of("burst")));
		DEV_CHK(hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
		DEV_CHK(hipStreamCreateWithFlags(&c->stream_out, hipStreamNonBlocking));
		}
#ifndef VDL2_EXPERIMENTS
		c->pooled = true;
#endif
#ifdef VDL2_EXPERIMENTS
		if(c->sync_on == 2) DEV_CHK(hipStreamCreateWithPriority(&c->stream_sync, hipStreamNonBlocking, strcmp(getenv("VDL2HIP_SYNC_ON"), "own-high") == 0 ? prio_high : prio_low));
#endif
	}
	for(auto &sl : c->slot) {
		DEV_CHK(hipEventCreate(&sl.done)); for(int i = 0; i < kNumEv; i++) DEV_CHK(hipEventCreate(&sl.ev[i]));
		DEV_CHK(hipEventCreateWithFlags(&sl.ev_walk, hipEventDisableTiming)); DEV_CHK(hipEventCreateWithFlags(&sl.ev_nf, hipEventDisableTiming));
		DEV_CHK(hipEventCreate(&sl.ev_front)); DEV_CHK(hipEventCreate(&sl.ev_chan)); DEV_CHK(hipEventCreateWithFlags(&sl.ev_k1, hipEventDisableTiming));
	}
	DEV_ALLOC(c->d_bf, sizeof(BlockForm)); DEV_ALLOC(c->d_lut, sizeof(Lut4) * 256); DEV_ALLOC(c->d_tab, sizeof(Tables));
	DEV_ALLOC(c->d_dphi, 4 * count); DEV_ALLOC(c->d_freq, 4 * count); DEV_ALLOC(c->d_ppmthr, 4 * count);
	DEV_ALLOC(c->d_carry[0], 4 * kMaxOversample); DEV_ALLOC(c->d_carry[1], 4 * kMaxOversample);
	const size_t nring = (size_t)count * cap;
	DEV_ALLOC(c->d_y, nring * sizeof(cf32)); DEV_ALLOC(c->d_pf, nring * sizeof(cf32));
	DEV_ALLOC(c->d_cand, nring / 8); DEV_ALLOC(c->d_flag, nring / 8);
	DEV_ALLOC(c->d_segend, (size_t)count * c->nseg_cap * sizeof(float4));
	DEV_ALLOC(c->d_qpow, 64 * sizeof(float4));
	DEV_ALLOC(c->d_tcarry[0], count * sizeof(float4)); DEV_ALLOC(c->d_tcarry[1], count * sizeof(float4));
	DEV_ALLOC(c->d_segpub, (size_t)count * c->nseg_cap * 4 * 8); DEV_ALLOC(c->d_synctmo, 4);
	DEV_CHK(hipMemset(c->d_segpub, 0, (size_t)count * c->nseg_cap * 4 * 8)); DEV_CHK(hipMemset(c->d_synctmo, 0, 4));
	// The fused fix-up makes a workgroup wait for the workgroup of the previous segment (one-step look-back, kernels.h).  The wait is
	// short while the workgroups of a launch are dispatched in order and stay resident - one process per GPU, the deployment this
	// library is built for.  On a GPU time-sliced between PROCESSES the driver saves and restores waves in no particular order and
	// waiting consumers can hold the CUs their producers need: a consumer therefore gives up after a few milliseconds and works the
	// state out itself from the previous segment's last tile (stats.front_sync_timeouts counts those; results are unaffected).
	// VDL2HIP_NO_FUSE=1 selects the separate fix-up kernel k_fixup instead (no inter-workgroup wait at all, bit-identical results,
	// ~3 % slower): worth setting where the GPU is permanently shared, so that no time is spent waiting.
	c->fuse_k2 = getenv("VDL2HIP_NO_FUSE") == nullptr;
	if(const char *e = getenv("VDL2HIP_BACKEND")) c->defer_back = strcmp(e, "deferred") == 0;   // eager (default) | deferred
	DEV_ALLOC(c->d_ws, count * sizeof(WalkState)); DEV_ALLOC(c->d_cnt, 2 * (size_t)count * kNumCounters * 8); c->d_wcnt = c->d_cnt + (size_t)count * kNumCounters;
	DEV_ALLOC(c->d_acnt, (size_t)count * kNumAvlcCounters * 8);
	// a decodable burst occupies >= 22 symbols = 220 decimated samples (header + 3 data + 2 FEC octets)
	c->cap_bursts_chan = (uint32_t)(dmax / 220 + 4);
	uint64_t cap_b = (uint64_t)count * c->cap_bursts_chan;
	uint64_t cap_f = cap_b * 2; if(cap_f < 4096) cap_f = 4096;
	cap_f += 2048 * kResSlots;                                    // the burst decoder's wavefronts each own a first share of the records and octets
	uint64_t cap_p = cap_b * 512; if(cap_p < (1u << 22)) cap_p = 1u << 22; if(cap_p > (1u << 30)) cap_p = 1u << 30;
	c->cap_log = 8192; c->cap_comb = c->cap_log + kNfTail; c->cap_hist = (uint32_t)(dmax / 3000 + 8);
	cap_p += 2048 * kResPool;
	c->ctl_template = OutCtl{ 0, 0, 0, 0, (uint32_t)cap_b, (uint32_t)cap_f, (uint32_t)cap_p, c->cap_log, 0, 0, {0, 0} };
	DEV_ALLOC(c->d_nf, count * sizeof(NfState));
	DEV_ALLOC(c->d_scfirst, (size_t)count * (c->cap_comb + 1) * 8); DEV_ALLOC(c->d_sccum, (size_t)count * (c->cap_comb + 1) * 8);
	// noise-floor history: a frame looks up the value at its burst's sync, at most kSlots feeds + one burst ago
	c->nf_ring = 64; while(c->nf_ring < (kSlots + 2) * c->cap_hist) c->nf_ring <<= 1;
	DEV_ALLOC(c->d_nfring, (size_t)count * c->nf_ring * 4);
	DEV_ALLOC(c->d_lpbuf, (size_t)count * c->cap_hist * 4); DEV_ALLOC(c->d_nffeed, count * sizeof(NfFeed));
	{
		// segments per feed: enough wavefronts to cover the walk's latency, not more than the chip holds at once
		const char *e = getenv("VDL2HIP_SEG_MIN");
		if(e && atoll(e) >= 64) c->seg_min = atoll(e);
		int64_t smax = std::min<int64_t>(16, 8192 / (3 * (int64_t)count));   // measured: 8..24 are within noise of each other at 8 channels
		smax = std::min<int64_t>(smax, dmax / c->seg_min);
		e = getenv("VDL2HIP_SEG_MAX");
		if(e) smax = std::min<int64_t>(smax, atoll(e));
		c->seg_max = (int)std::max<int64_t>(1, smax);
		if(c->seg_max >= 2) for(auto &sp_ : c->d_spec) DEV_ALLOC(sp_, (size_t)count * 3 * (c->seg_max - 1) * sizeof(SpecOut));
		DEV_ALLOC(c->d_segstats, (size_t)count * 2 * 4);
		DEV_CHK(hipMemset(c->d_segstats, 0, (size_t)count * 2 * 4));
	}
	for(auto &sl : c->slot) {
		DEV_ALLOC(sl.d_bursts, cap_b * sizeof(Burst)); DEV_ALLOC(sl.d_nbchan, count * 4);
		DEV_ALLOC(sl.d_frames, cap_f * sizeof(OutFrame)); DEV_ALLOC(sl.d_pool, cap_p); DEV_ALLOC(sl.d_mail, sizeof(OutMail));
		sl.d_ctl = &sl.d_mail->ctl;
		DEV_CHK(hipMemset(sl.d_mail, 0, sizeof(OutMail)));
		DEV_CHK(hipMemcpy(sl.d_ctl, &c->ctl_template, sizeof(OutCtl), hipMemcpyHostToDevice));
		DEV_ALLOC(sl.d_frames_out, cap_f * sizeof(OutFrame)); DEV_ALLOC(sl.d_pool_out, cap_p);
		DEV_ALLOC(sl.d_log, (size_t)count * c->cap_log * sizeof(EvalChunk)); DEV_ALLOC(sl.d_nlog, count * 4);
		DEV_CHK(hipMemset(sl.d_nlog, 0, count * 4));
		DEV_CHK(hipHostMalloc((void **)&sl.h_mail, sizeof(OutMail), hipHostMallocDefault));
		memset(sl.h_mail, 0, sizeof(OutMail));
		DEV_CHK(hipMemset(sl.d_nbchan, 0, count * 4));
	}

	if(const char *e = getenv("VDL2HIP_REFEREE")) c->referee = atoi(e) != 0;
	// Referee, long feeds: where do the scans go?  A receiver of few channels has a front of a fraction of a millisecond and its step is
	// the walk's chain - walk, scan, check, walk again - unless the scans run AHEAD of the walk on a stream per slot (`prescan`); with
	// hundreds of channels the front hides the chain and the 2.5x scans of that mode (every marked candidate, not every visited one)
	// cost more than they save.  VDL2HIP_REF_PRESCAN=0/1 overrides the choice.
	c->ref_prescan = count <= 64;
	if(const char *e = getenv("VDL2HIP_REF_PRESCAN")) c->ref_prescan = atoi(e) != 0;
	if(const char *e = getenv("VDL2HIP_REF_RETRY")) { const int v = atoi(e); if(v == 0 || (v >= 2 && v <= 8)) c->ref_retry_mul = v; }
	if(const char *e = getenv("VDL2HIP_REF_WARM")) { const long long v = atoll(e); if(v >= 1024 && v <= (1ll << 24)) c->ref_warm = v; }
	if(c->referee) {
		c->ref_T = std::max(1, c->ref_retry_mul) * c->ref_warm + (int64_t)(kHistory + 256) * c->os + 4096;      // run-up (of a retry: ref_retry_mul times the configured one) + the longest burst (its symbols are sliced when its last one has arrived)
		c->ref_cap = 1; while(c->ref_cap < (uint64_t)(kSlots + 2) * (uint64_t)c->ref_T) c->ref_cap <<= 1;
		DEV_ALLOC(c->d_refhist, c->ref_cap * sb);
		for(int k = 0; k < kSlots; k++) { DEV_ALLOC(c->d_ref[k], sizeof(RefChan)); DEV_CHK(hipMemset(c->d_ref[k], 0, sizeof(RefChan))); }
		DEV_ALLOC(c->d_refdone, (size_t)count * kRefCache * 8); DEV_ALLOC(c->d_refdonen, (size_t)count * 4); DEV_ALLOC(c->d_refstats, 64); DEV_ALLOC(c->d_mix, count);
		DEV_CHK(hipMemset(c->d_refdone, 0, (size_t)count * kRefCache * 8)); DEV_CHK(hipMemset(c->d_refdonen, 0, (size_t)count * 4)); DEV_CHK(hipMemset(c->d_refstats, 0, 64));
		std::vector<uint8_t> mix(count);
		for(uint32_t i = 0; i < count; i++) mix[i] = cfg->centerfreq != c->freqs[i];       // v->offset_tuning, demod.c:386
		DEV_CHK(hipMemcpy(c->d_mix, mix.data(), count, hipMemcpyHostToDevice));
		if(const char *e = getenv("VDL2HIP_REF_MODE")) c->ref_optimistic = strcmp(e, "sync") != 0;        // optimistic (default) | sync
		if(const char *e = getenv("VDL2HIP_REF_KINDS")) c->ref_kinds = atoi(e) & 7;                          // (development: 1 candidates, 2 headers, 4 symbols)
		for(int k = 0; k < 3; k++) { DEV_ALLOC(c->d_ws_snap[k], count * sizeof(WalkState)); DEV_ALLOC(c->d_cnt_snap[k], (size_t)count * kNumCounters * 8); }
		DEV_ALLOC(c->d_ws_tmp, count * sizeof(WalkState)); DEV_ALLOC(c->d_cnt_tmp, (size_t)count * kNumCounters * 8);
		// Walk ahead (launch_back): for receivers whose front does not hide the chain walk - scans - check (a feed's results then come a
		// feed later: with 256 channels, where the front hides the chain anyway, that extra depth cost 10 % and more)
		c->walk_ahead = (count >= 16 && count <= 64) ? 1 : 0;   // (8 channels: the walk itself is longer than the front, the second walks' extra launches cost more than they save: 1.22 against 1.11 ms)
		if(const char *e = getenv("VDL2HIP_WALK_AHEAD")) { const int v = atoi(e); c->walk_ahead = v < 0 ? 0 : v > 2 ? 2 : v; c->walk_ahead_auto = false; }      // 0: a feed's walk waits for the check of the feed before (round 5's schedule)
		c->sq_alloc = defer_scans_for((int64_t)dmax, (int)count);
		for(auto &sl : c->slot) {
			DEV_ALLOC(sl.d_rq, (size_t)c->rq_cap * sizeof(RefReq)); DEV_ALLOC(sl.d_rqn, 32); DEV_ALLOC(sl.d_retry, 3 * (size_t)kRetryScans * sizeof(ScanReq)); DEV_ALLOC(sl.d_rqflag, (size_t)count * 4);
			DEV_ALLOC(sl.d_dq, (size_t)(c->sq_alloc / 2) * 4); DEV_ALLOC(sl.d_sq, (size_t)c->sq_alloc * sizeof(ScanReq));
			DEV_ALLOC(sl.d_pq, (size_t)kPreScans * sizeof(ScanReq)); DEV_CHK(hipEventCreateWithFlags(&sl.ev_pre, hipEventDisableTiming));
			DEV_ALLOC(sl.d_rqbad, (size_t)count * sizeof(RefBad)); DEV_CHK(hipMemset(sl.d_rqbad, 0, (size_t)count * sizeof(RefBad)));
			DEV_ALLOC(sl.d_rqflag2, (size_t)count * 4); DEV_CHK(hipMemset(sl.d_rqflag2, 0, (size_t)count * 4));
			DEV_CHK(hipEventCreateWithFlags(&sl.ev_stitch, hipEventDisableTiming)); DEV_CHK(hipEventCreateWithFlags(&sl.ev_chk, hipEventDisableTiming));
			DEV_CHK(hipMemset(sl.d_rqn, 0, 32)); DEV_CHK(hipMemset(sl.d_rqflag, 0, (size_t)count * 4));
		}
	}

	Lut4 lut[256]; build_nco_lut(lut);                            // sincosf_lut_init()
	Tables *tab = new Tables; build_tables(*tab);                 // demod_sync_init(), rs_init(), header tables
	std::vector<WalkState> ws(count);
	for(auto &w : ws) { memset(&w, 0, sizeof w); walk_state_init(w); }
	DEV_CHK(hipMemcpy(c->d_bf, &c->bf, sizeof(BlockForm), hipMemcpyHostToDevice));
	DEV_CHK(hipMemcpy(c->d_qpow, c->bf.Qpow, 64 * sizeof(float4), hipMemcpyHostToDevice));
	DEV_CHK(hipMemcpy(c->d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
	DEV_CHK(hipMemcpy(c->d_tab, tab, sizeof(Tables), hipMemcpyHostToDevice));
	delete tab;
	DEV_CHK(hipMemcpy(c->d_dphi, c->dphi.data(), 4 * count, hipMemcpyHostToDevice));
	DEV_CHK(hipMemcpy(c->d_freq, c->freqs.data(), 4 * count, hipMemcpyHostToDevice));
	{
		std::vector<float> thr(count);
		for(uint32_t i = 0; i < count; i++) thr[i] = ppm_gate_threshold(c->freqs[i], cfg->max_ppm);
		DEV_CHK(hipMemcpy(c->d_ppmthr, thr.data(), 4 * count, hipMemcpyHostToDevice));
	}
	DEV_CHK(hipMemcpy(c->d_ws, ws.data(), count * sizeof(WalkState), hipMemcpyHostToDevice));
	{
		std::vector<NfState> nfs(count);
		for(auto &n : nfs) { memset(&n, 0, sizeof n); nf_state_init(n); }
		DEV_CHK(hipMemcpy(c->d_nf, nfs.data(), count * sizeof(NfState), hipMemcpyHostToDevice));
		DEV_CHK(hipMemset(c->d_nfring, 0, (size_t)count * c->nf_ring * 4)); DEV_CHK(hipMemset(c->d_lpbuf, 0, (size_t)count * c->cap_hist * 4));
		DEV_CHK(hipMemset(c->d_nffeed, 0, count * sizeof(NfFeed)));
	}
	DEV_CHK(hipMemset(c->d_y, 0, nring * sizeof(cf32))); DEV_CHK(hipMemset(c->d_pf, 0, nring * sizeof(cf32)));
	DEV_CHK(hipMemset(c->d_cand, 0, nring / 8)); DEV_CHK(hipMemset(c->d_flag, 0, nring / 8));
	DEV_CHK(hipMemset(c->d_tcarry[0], 0, count * sizeof(float4))); DEV_CHK(hipMemset(c->d_tcarry[1], 0, count * sizeof(float4)));
	DEV_CHK(hipMemset(c->d_cnt, 0, 2 * (size_t)count * kNumCounters * 8)); DEV_CHK(hipMemset(c->d_acnt, 0, (size_t)count * kNumAvlcCounters * 8));
	DEV_CHK(hipMemset(c->d_segend, 0, (size_t)count * c->nseg_cap * sizeof(float4)));
	// the generic-oversample build may need more than the default dynamic LDS limit
	const size_t lds = (size_t)c->run * c->os * 65 * sizeof(float2) + 8192;
	if(lds > 65536) { vdl2hip_destroy(c); return VDL2HIP_E_INVAL; }
	DEV_CHK(hipDeviceSynchronize());
	#undef DEV_ALLOC
	#undef DEV_CHK
	*out = c;
	return VDL2HIP_OK;
}

// Host-fed blocks: the copy into the device goes to one of kSlots input buffers on the copy stream, the channeliser of the
// block waits for it on the front stream; the buffer was last read by the channeliser of feed i - kSlots, which
// collect_slot() has seen complete.  So the H2D of block i+1 overlaps the kernels of block i (and i-1, i-2 further down).
static int feed_host(vdl2hip_ctx *c, const void *buf, size_t nbytes, bool wait_copy) {
	if(!c || (!buf && nbytes)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	if(c->failed) return VDL2HIP_E_DEVICE;
	if(nbytes == 0) return VDL2HIP_OK;                             // process_buf_*: len == 0 is a no-op (demod.c:341,358)
	if(nbytes > c->in_cap) return VDL2HIP_E_TOOBIG;
	nbytes -= nbytes % sample_bytes(c->fmt);
	if(c->pinned_pending) { HIPCHK(hipEventSynchronize(c->pinned_pending)); c->pinned_pending = nullptr; }
	const int k = (int)(c->feed_no % kSlots);
	{ int r = collect_slot(c, c->slot[k]); if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r; }
	if(!c->d_in[k]) {
		if(hipMalloc((void **)&c->d_in[k], c->in_cap + 16) != hipSuccess) return VDL2HIP_E_NOMEM;
		HIPCHK(hipEventCreateWithFlags(&c->ev_copied[k], hipEventDisableTiming));
	}
	// (In a stream of blocks this copy costs nothing: 200-step regions run at the same 5.51 ms per 256-channel block host-fed and
	// HBM-resident.  What a short timed region sees is the FIRST copy, which nothing overlaps: 2.6 ms once.  Holding the copy back so that
	// it lands beside the channeliser rather than the sync kernels changes nothing or makes it worse; profiles/r03_h2d_placement.txt)
	// Nothing in flight and a large block from page-locked memory (the first block of a stream, or of a timed region): the copy is cut in
	// kColdParts pieces and the channeliser follows them piece by piece (feed_common), instead of idling for the whole transfer -
	// 2.6 ms for a 134 MB block.  In a running stream the copy of block i+1 is hidden behind the kernels of block i and goes in one piece.
	bool parts = !wait_copy && nbytes >= kColdMinBytes;
	for(auto &sl : c->slot) if(sl.pending) parts = false;
	c->cold.n = 0;
	if(parts) {
		const size_t piece = (nbytes / kColdParts) & ~(size_t)4095;
		size_t off = 0;
		for(int p = 0; p < kColdParts; p++) {
			const size_t len = p == kColdParts - 1 ? nbytes - off : piece;
			if(!c->cold.ev[p]) HIPCHK(hipEventCreateWithFlags(&c->cold.ev[p], hipEventDisableTiming));
			HIPCHK(hipMemcpyAsync(c->d_in[k] + off, (const uint8_t *)buf + off, len, hipMemcpyHostToDevice, c->stream_copy));
			HIPCHK(hipEventRecord(c->cold.ev[p], c->stream_copy));
			off += len;
			c->cold.samples[p] = off / sample_bytes(c->fmt);
		}
		c->cold.n = kColdParts;
	} else if(wait_copy) {
		// vdl2hip_feed(): `buf` is only ours during the call, so the copy is the blocking one - and a copy the host has waited for
		// needs no event for the front stream to wait on (0.190 -> 0.174-0.181 ms per 320 000-byte block against the asynchronous copy
		// + event + wait it replaces; profiles/r04_dropin_feed_path_ab.txt)
		// RELIES ON: ROCm's hipMemcpy() returns when the data is in device memory, for pageable and page-locked sources alike (the
		// front stream is hipStreamNonBlocking - no implicit ordering with the null stream the copy runs on - and nothing else makes
		// it wait).  CUDA only promises that much for page-locked sources; on a HIP back end without it, use the asynchronous branch
		// below (copy stream + event).  The call also waits for whatever else the process has on the legacy default stream.
		HIPCHK(hipMemcpy(c->d_in[k], buf, nbytes, hipMemcpyHostToDevice));
	} else {
		HIPCHK(hipMemcpyAsync(c->d_in[k], buf, nbytes, hipMemcpyHostToDevice, c->stream_copy));
	}
	if(!wait_copy) {
		HIPCHK(hipEventRecord(c->ev_copied[k], c->stream_copy));
		c->pinned_pending = c->ev_copied[k];
		if(!parts) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_copied[k], 0));
	}
	int r = feed_common(c, c->d_in[k], nbytes, parts);
	if(r == VDL2HIP_E_DEVICE) c->failed = true;          // part of the block's work may be queued, part not: the context is out of step with itself
	return r;
}

int vdl2hip_feed(vdl2hip_ctx *c, const void *buf, size_t nbytes) { return feed_host(c, buf, nbytes, true); }

int vdl2hip_feed_pinned(vdl2hip_ctx *c, const void *buf, size_t nbytes) { return feed_host(c, buf, nbytes, false); }

int vdl2hip_feed_device(vdl2hip_ctx *c, const void *dev_buf, size_t nbytes) {
	if(!c || (!dev_buf && nbytes)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	if(nbytes == 0) return VDL2HIP_OK;
	if(nbytes > c->in_cap) return VDL2HIP_E_TOOBIG;
	if(((uintptr_t)dev_buf) % sample_bytes(c->fmt)) return VDL2HIP_E_INVAL;
	nbytes -= nbytes % sample_bytes(c->fmt);
	if(c->failed) return VDL2HIP_E_DEVICE;
	int r = feed_common(c, dev_buf, nbytes);
	if(r == VDL2HIP_E_DEVICE) c->failed = true;
	return r;
}

int vdl2hip_sync(vdl2hip_ctx *c) {
	if(!c) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	if(c->pinned_pending) { HIPCHK(hipEventSynchronize(c->pinned_pending)); c->pinned_pending = nullptr; }
	if(c->failed) return VDL2HIP_E_DEVICE;
	return collect_pending(c);
}

int vdl2hip_drain(vdl2hip_ctx *c, vdl2hip_frame_cb cb, void *user) {
	if(!c) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	if(c->failed) return VDL2HIP_E_DEVICE;
	int r = collect_pending(c, c->drain_lag);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	sort_queue(c);
	int n = 0;
	for(const HostFrame &h : c->queue) {
		if(cb) {
			vdl2hip_frame f{};
			fill_frame(c, h, f);
			cb(&f, user);
		}
		n++;
	}
	c->queue.clear();
	return n;
}

int vdl2hip_drain_packed(vdl2hip_ctx *c, vdl2hip_packed_frame *frames, size_t cap_frames,
		uint8_t *octets, size_t cap_octets, size_t *octets_used) {
	if(!c || (!frames && cap_frames) || (!octets && cap_octets)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	if(c->failed) return VDL2HIP_E_DEVICE;
	int r = collect_pending(c, c->drain_lag);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	sort_queue(c);
	size_t n = 0, used = 0;
	for(const HostFrame &h : c->queue) {
		if(n >= cap_frames || used + h.f.len > cap_octets) break;
		fill_frame(c, h, frames[n].frame);
		frames[n].frame.octets = nullptr;
		frames[n].octets_off = used;
		if(h.f.len && h.octets()) memcpy(octets + used, h.octets(), h.f.len);
		used += h.f.len;
		n++;
	}
	c->queue.erase(c->queue.begin(), c->queue.begin() + (long)n);
	if(octets_used) *octets_used = used;
	return (int)n;
}

// ---- proto3 wire format, hand-rolled (no protobuf runtime in the product) ----
namespace {
struct PbOut {
	uint8_t *p; size_t cap, n; bool ok;
	void byte(uint8_t b) { if(n < cap) p[n] = b; else ok = false; n++; }
	void varint(uint64_t v) { while(v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; } byte((uint8_t)v); }
	void tag(uint32_t field, uint32_t wt) { varint((uint64_t)field << 3 | wt); }
	void u32(uint32_t field, uint32_t v) { if(v) { tag(field, 0); varint(v); } }                       // proto3: defaults are omitted
	void i32(uint32_t field, int32_t v) { if(v) { tag(field, 0); varint((uint64_t)(int64_t)v); } }      // negative int32 -> 10-byte varint
	void i64(uint32_t field, int64_t v) { if(v) { tag(field, 0); varint((uint64_t)v); } }
	void f32(uint32_t field, float v) { uint32_t u; memcpy(&u, &v, 4); if(u) { tag(field, 5); for(int i = 0; i < 4; i++) byte((uint8_t)(u >> (8 * i))); } }
	void bytes(uint32_t field, const uint8_t *d, size_t len) { tag(field, 2); varint(len); for(size_t i = 0; i < len; i++) byte(d[i]); }
};
size_t pack_metadata(PbOut &o, const vdl2hip_frame *f, const char *station_id, int64_t tv_sec, int64_t tv_usec) {
	const size_t start = o.n;
	if(station_id && station_id[0]) o.bytes(1, (const uint8_t *)station_id, strlen(station_id));
	o.u32(2, f->freq); o.u32(3, f->synd_weight); o.u32(4, f->datalen_octets);
	o.f32(5, f->frame_pwr_dbfs); o.f32(6, f->nf_pwr_dbfs); o.f32(7, f->ppm_error);
	o.i32(8, 1 /* metadata->version, src/decode.c:177 */); o.i32(9, f->num_fec_corrections); o.i32(10, f->idx);
	uint8_t tsbuf[24]; PbOut ts{ tsbuf, sizeof tsbuf, 0, true };
	ts.i64(1, tv_sec); ts.i64(2, tv_usec);
	o.bytes(11, tsbuf, ts.n);                                    // the timestamp sub-message is always present (src/fmtr-binary.c:38)
	return o.n - start;
}
}  // namespace

int vdl2hip_pack_raw_frame(const vdl2hip_frame *f, const char *station_id, int64_t tv_sec, int64_t tv_usec, uint8_t *out, size_t cap) {
	if(!f || !out || (f->len && !f->octets)) return VDL2HIP_E_INVAL;
	uint8_t meta[512]; PbOut m{ meta, sizeof meta, 0, true };
	pack_metadata(m, f, station_id, tv_sec, tv_usec);
	if(!m.ok) return VDL2HIP_E_TOOBIG;
	PbOut o{ out, cap, 0, true };
	o.byte(0); o.byte(0);                                         // record length, patched below
	o.bytes(1, meta, m.n);
	if(f->len) o.bytes(2, f->octets, f->len);                    // proto3 omits empty bytes
	if(!o.ok || o.n > 65535) return VDL2HIP_E_TOOBIG;
	out[0] = (uint8_t)(o.n >> 8); out[1] = (uint8_t)o.n;         // htons(payload + 2), src/output-file.c:181-188
	return (int)o.n;
}

int vdl2hip_counters(vdl2hip_ctx *c, uint32_t chan, uint64_t out[VDL2HIP_NUM_COUNTERS]) {
	if(!c || !out || chan < (uint32_t)c->chan_first || chan >= (uint32_t)(c->chan_first + c->C)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	uint64_t w[kNumCounters];
	HIPCHK(hipMemcpy(out, c->d_cnt + (size_t)(chan - c->chan_first) * kNumCounters, 8 * kNumCounters, hipMemcpyDeviceToHost));
	HIPCHK(hipMemcpy(w, c->d_wcnt + (size_t)(chan - c->chan_first) * kNumCounters, 8 * kNumCounters, hipMemcpyDeviceToHost));
	for(int k = 0; k < kNumCounters; k++) out[k] += w[k];      // the burst decoder's row + the walker's
	return VDL2HIP_OK;
}

int vdl2hip_avlc_counters(vdl2hip_ctx *c, uint32_t chan, uint64_t out[VDL2HIP_NUM_AVLC_COUNTERS]) {
	if(!c || !out || chan < (uint32_t)c->chan_first || chan >= (uint32_t)(c->chan_first + c->C)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	HIPCHK(hipMemcpy(out, c->d_acnt + (size_t)(chan - c->chan_first) * kNumAvlcCounters, 8 * kNumAvlcCounters, hipMemcpyDeviceToHost));
	return VDL2HIP_OK;
}

int vdl2hip_set_avlc_filter(vdl2hip_ctx *c, int on) {
	if(!c) return VDL2HIP_E_INVAL;
	c->avlc_filter = on != 0;
	return VDL2HIP_OK;
}

// counter names in the order of the VDL2HIP_CNT_* / VDL2HIP_ACNT_* enums (statsd.c:34-65); the two diagnostics this
// implementation adds (ppm_reject, slicer_neg_idx) are exported under demod.* as well
static const char *const kCounterNames[VDL2HIP_NUM_COUNTERS] = {
	"demod.sync.good", "decoder.crc.good", "decoder.crc.bad", "decoder.errors.no_header", "decoder.errors.too_long",
	"decoder.errors.no_fec", "decoder.errors.data_truncated", "decoder.errors.fec_truncated", "decoder.errors.deinterleave_data",
	"decoder.errors.deinterleave_fec", "decoder.errors.fec_bad", "decoder.errors.bitstream", "decoder.errors.truncated_octets",
	"decoder.errors.unstuff", "decoder.blocks.processed", "decoder.blocks.fec_ok", "decoder.msg.good", "decoder.msg.good_loud",
	"demod.ppm_reject", "demod.slicer_neg_idx" };
static const char *const kAvlcCounterNames[VDL2HIP_NUM_AVLC_COUNTERS] = {
	"avlc.frames.processed", "avlc.errors.too_short", "avlc.frames.good", "avlc.errors.bad_fcs", "avlc.msg.air2gnd",
	"avlc.msg.air2air", "avlc.msg.air2all", "avlc.msg.gnd2air", "avlc.msg.gnd2gnd", "avlc.msg.gnd2all" };

int vdl2hip_statsd_lines(vdl2hip_ctx *c, const char *ns, char *out, size_t cap) {
	if(!c || !ns || !out) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	const size_t per = VDL2HIP_NUM_COUNTERS + VDL2HIP_NUM_AVLC_COUNTERS;
	std::vector<uint64_t> now((size_t)c->C * per);
	{
		std::vector<uint64_t> a(2 * (size_t)c->C * kNumCounters), b((size_t)c->C * kNumAvlcCounters);      // two copies, whatever the channel count
		HIPCHK(hipMemcpy(a.data(), c->d_cnt, a.size() * 8, hipMemcpyDeviceToHost));
		for(size_t k = 0, n = a.size() / 2; k < n; k++) a[k] += a[n + k];                               // the burst decoder's rows + the walker's
		HIPCHK(hipMemcpy(b.data(), c->d_acnt, b.size() * 8, hipMemcpyDeviceToHost));
		for(int ch = 0; ch < c->C; ch++) {
			std::copy(a.begin() + (size_t)ch * kNumCounters, a.begin() + (size_t)(ch + 1) * kNumCounters, now.begin() + ch * per);
			std::copy(b.begin() + (size_t)ch * kNumAvlcCounters, b.begin() + (size_t)(ch + 1) * kNumAvlcCounters, now.begin() + ch * per + kNumCounters);
		}
	}
	const bool first = c->statsd_prev.empty();
	if(first) c->statsd_prev.assign(now.size(), 0);
	std::string txt;
	char line[320];
	for(int ch = 0; ch < c->C; ch++)
		for(size_t k = 0; k < per; k++) {
			const uint64_t d = now[ch * per + k] - c->statsd_prev[ch * per + k];
			if(!first && d == 0) continue;
			const char *name = k < (size_t)kNumCounters ? kCounterNames[k] : kAvlcCounterNames[k - kNumCounters];
			snprintf(line, sizeof line, "%s.%u.%s:%llu|c\n", ns, c->freqs[ch], name, (unsigned long long)d);
			txt += line;
		}
	if(txt.size() + 1 > cap) { if(first) c->statsd_prev.clear(); return VDL2HIP_E_TOOBIG; }
	memcpy(out, txt.c_str(), txt.size() + 1);
	c->statsd_prev = now;
	return (int)txt.size();
}

int vdl2hip_set_profiling(vdl2hip_ctx *c, int on) {
	if(!c) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	c->profiling = on < 0 ? 0 : on > 2 ? 2 : on;
	return r == VDL2HIP_E_OVERFLOW ? VDL2HIP_OK : r;
}

int vdl2hip_get_stats_sized(vdl2hip_ctx *c, vdl2hip_stats *out, size_t size) {
	if(!c || !out) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	{
		std::vector<uint32_t> ss((size_t)c->C * 2);
		if(hipMemcpy(ss.data(), c->d_segstats, ss.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return VDL2HIP_E_DEVICE;
		c->stats.seg_adopted = c->stats.seg_walked = 0;
		uint32_t tmo = 0;
		if(hipMemcpy(&tmo, c->d_synctmo, 4, hipMemcpyDeviceToHost) != hipSuccess) return VDL2HIP_E_DEVICE;
		c->stats.front_sync_timeouts = tmo;
		for(int i = 0; i < c->C; i++) { c->stats.seg_adopted += ss[2 * i]; c->stats.seg_walked += ss[2 * i + 1]; }
	}
	if(c->d_refstats) {
		uint32_t rs[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		if(hipMemcpy(rs, c->d_refstats, sizeof rs, hipMemcpyDeviceToHost) != hipSuccess) return VDL2HIP_E_DEVICE;
		c->stats.referee_scans = rs[0]; c->stats.referee_cached = rs[1]; c->stats.referee_refused = rs[2]; c->stats.referee_short = rs[3];
		c->stats.referee_candidate_scans = rs[4]; c->stats.referee_header_scans = rs[5]; c->stats.referee_symbol_scans = rs[6]; c->stats.referee_rewalks = rs[7]; c->stats.referee_redone_next = rs[8]; c->stats.referee_unmet = rs[9]; c->stats.referee_retried = rs[10];
	}
	memcpy(out, &c->stats, std::min(size, sizeof c->stats));
	return (r == VDL2HIP_E_OVERFLOW || c->failed) ? VDL2HIP_OK : r;
}
int vdl2hip_get_stats(vdl2hip_ctx *c, vdl2hip_stats *out) { return vdl2hip_get_stats_sized(c, out, sizeof(vdl2hip_stats)); }

void *vdl2hip_stream(vdl2hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int vdl2hip_set_drain_lag(vdl2hip_ctx *c, int lag) {
	if(!c || lag < 0 || lag > kSlots - 1) return VDL2HIP_E_INVAL;
	c->drain_lag = lag;
	return VDL2HIP_OK;
}

// test hook (not declared in vdl2hip.h; tests/test_gpu_parity.py): "no_fuse" = run the segment-start fix-up as the separate kernel
// k_fixup instead of inside K1 (bit-identical results); "force_timeout" = make every look-back hand-off fail (producers publish
// under a wrong epoch), so that the fall-back path is taken by every workgroup and can be tested
int vdl2hip_debug_option(vdl2hip_ctx *c, const char *name, long value) {
	if(!c || !name) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	if(strcmp(name, "no_fuse") == 0) { c->fuse_k2 = value == 0; return VDL2HIP_OK; }
	if(strcmp(name, "force_timeout") == 0) { c->debug_force_timeout = value != 0; return VDL2HIP_OK; }
	if(strcmp(name, "force_mismatch") == 0) { c->debug_force_mismatch = value != 0; return VDL2HIP_OK; }   // every channel walked again with the next feed's walk already done is taken to have ended differently: the next feed is redone for it
	if(strcmp(name, "walk_ahead_below") == 0) { c->walk_ahead_below = (double)value; c->walk_ahead_auto = true; return VDL2HIP_OK; }   // (tests: feeds of fewer channel-samples than this let the next walk go ahead)
	if(strcmp(name, "walk_ahead") == 0) { c->walk_ahead = value < 0 ? 0 : value > 2 ? 2 : (int)value; c->walk_ahead_auto = false; return VDL2HIP_OK; }   // feeds whose walks may go ahead of a feed's check: 0, 1, 2
	if(strcmp(name, "force_again") == 0) { c->debug_force_again = value != 0; return VDL2HIP_OK; }   // every channel of every long feed is stitched a second time (the referee's walk-again path)
	if(strcmp(name, "referee") == 0) { if(value && !c->d_refhist) return VDL2HIP_E_INVAL; c->referee = value != 0; return VDL2HIP_OK; }   // (on only where it was on at create: the history ring)
	if(strcmp(name, "ref_debug_chan") == 0) {
		if(!c->d_refdbg) { if(hipMalloc((void **)&c->d_refdbg, 8 * 4001) != hipSuccess) return VDL2HIP_E_NOMEM; }
		if(hipMemset(c->d_refdbg, 0, 8 * 4001) != hipSuccess) return VDL2HIP_E_DEVICE;
		c->ref_dbg_chan = (int)value; return VDL2HIP_OK;
	}
	if(strcmp(name, "ref_kinds") == 0) { c->ref_kinds = (int)value & 7; return VDL2HIP_OK; }
	if(strcmp(name, "ref_warm") == 0) { if(value < 0 || value > c->ref_T - 4096) return VDL2HIP_E_INVAL; c->ref_warm = value; return VDL2HIP_OK; }
	return VDL2HIP_E_INVAL;
}

// test hook (not declared in vdl2hip.h): the referee's scan on [n_lo, n_hi] of one channel, with the raw input the most recent feed could
// reach - the decimated stream there is then the reference's own, to be read back with vdl2hip_read_decimated().  1: done, 0: refused
int vdl2hip_debug_exact_window_many(vdl2hip_ctx *c, uint32_t chan, int64_t n_lo, int64_t n_hi, uint32_t count, int64_t stride, float *ms) {
	if(!c || !c->d_refhist || chan >= (uint32_t)c->C || c->feed_no == 0 || count == 0 || count > 65536) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	int *d_out = nullptr; std::vector<int> h(count, 0);
	if(hipMalloc((void **)&d_out, 4 * (size_t)count) != hipSuccess) return VDL2HIP_E_NOMEM;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
	if(ok) hipExtLaunchKernelGGL(k_ref_probe, dim3(count), dim3(64), 0, c->stream, e0, e1, 0, c->d_ref[(c->feed_no - 1) % kSlots], (int)chan, c->C, n_lo, n_hi, stride, d_out);
	ok = ok && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(h.data(), d_out, 4 * (size_t)count, hipMemcpyDeviceToHost) == hipSuccess;
	float t = 0.f;
	if(ok && ms) { (void)hipEventElapsedTime(&t, e0, e1); *ms = t; }
	if(e0) (void)hipEventDestroy(e0);
	if(e1) (void)hipEventDestroy(e1);
	(void)hipFree(d_out);
	if(!ok) return VDL2HIP_E_DEVICE;
	int n = 0; for(int v : h) n += v;
	return n;            // stretches done (of `count`)
}
// test hook (not declared in vdl2hip.h): the same through k_ref_scan_multi - `count` stretches (chan[i], lo[i], hi[i]) side by side;
// returns the number of scans run, *ms = the kernels' time.  with_retry: as the product's launches do it - the scans that have not met
// their witness are listed and run again from ref_retry_mul times further back (without: they are counted as unmet and published)
int vdl2hip_debug_scan_multi2(vdl2hip_ctx *c, const int32_t *chan, const int64_t *lo, const int64_t *hi, uint32_t count, int with_retry, float *ms) {
	if(!c || !c->d_refhist || c->feed_no == 0 || count == 0 || count > 65536 || !chan || !lo || !hi) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	std::vector<ScanReq> h(count);
	for(uint32_t i = 0; i < count; i++) { if(chan[i] < 0 || chan[i] >= c->C) return VDL2HIP_E_INVAL; h[i] = ScanReq{ chan[i], REF_CANDIDATE, lo[i], hi[i] }; }
	const bool rty = with_retry && c->ref_retry_mul > 0;
	ScanReq *d_sq = nullptr, *d_rt = nullptr; uint32_t *d_n = nullptr;
	if(hipMalloc((void **)&d_sq, sizeof(ScanReq) * (size_t)count) != hipSuccess || hipMalloc((void **)&d_n, 8) != hipSuccess || hipMalloc((void **)&d_rt, sizeof(ScanReq) * (size_t)kRetryScans) != hipSuccess) {
		if(d_sq) (void)hipFree(d_sq);
		if(d_n) (void)hipFree(d_n);
		return VDL2HIP_E_NOMEM;
	}
	uint32_t before[8] = {0}, after[8] = {0};
	const uint32_t nn[2] = { count, 0u };
	hipEvent_t e0 = nullptr, e1 = nullptr;
	bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess
		&& hipMemcpy(d_sq, h.data(), sizeof(ScanReq) * (size_t)count, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_n, nn, 8, hipMemcpyHostToDevice) == hipSuccess
		&& hipMemcpy(before, c->d_refstats, sizeof before, hipMemcpyDeviceToHost) == hipSuccess;
	RefChan *ref = c->d_ref[(c->feed_no - 1) % kSlots];
	if(ok) LAUNCH_SCAN_MULTI(hipExtLaunchKernelGGL, dim3((count + kScanLanes - 1) / kScanLanes), dim3(64 * kScanWaves), 0, c->stream, e0, rty ? (hipEvent_t) nullptr : e1, 0, ref, 0xfffeu,
	                             (const ScanReq *)d_sq, (const RefReq *) nullptr, (const uint32_t *)d_n, count, (int64_t)c->k_total, rty ? d_rt : (ScanReq *) nullptr, d_n + 1, kRetryScans, 1);
	if(ok && rty) LAUNCH_SCAN_MULTI(hipExtLaunchKernelGGL, dim3(kRetryScans / kScanLanes), dim3(64 * kScanWaves), 0, c->stream, (hipEvent_t) nullptr, e1, 0, ref, 0xfffeu,
	                             (const ScanReq *)d_rt, (const RefReq *) nullptr, (const uint32_t *)(d_n + 1), kRetryScans, (int64_t)c->k_total, (ScanReq *) nullptr, (uint32_t *) nullptr, 0u, c->ref_retry_mul);
	ok = ok && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(after, c->d_refstats, sizeof after, hipMemcpyDeviceToHost) == hipSuccess;
	float t = 0.f;
	if(ok && ms) { (void)hipEventElapsedTime(&t, e0, e1); *ms = t; }
	if(e0) (void)hipEventDestroy(e0);
	if(e1) (void)hipEventDestroy(e1);
	(void)hipFree(d_sq); (void)hipFree(d_n); (void)hipFree(d_rt);
	return ok ? (int)(after[0] - before[0]) : VDL2HIP_E_DEVICE;
}
int vdl2hip_debug_scan_multi(vdl2hip_ctx *c, const int32_t *chan, const int64_t *lo, const int64_t *hi, uint32_t count, float *ms) { return vdl2hip_debug_scan_multi2(c, chan, lo, hi, count, 0, ms); }
int vdl2hip_debug_exact_window(vdl2hip_ctx *c, uint32_t chan, int64_t n_lo, int64_t n_hi) { return vdl2hip_debug_exact_window_many(c, chan, n_lo, n_hi, 1, 0, nullptr); }

// test hook (not declared in vdl2hip.h): what the DPP controls the channeliser's scan relies on do on this device
int vdl2hip_debug_u8_levels(float out[512]) {        // out[0..255]: the channeliser's division-free (i - 127.5) / 127.5, out[256..511]: the division (tests)
	float *d = nullptr;
	if(hipMalloc((void **)&d, 512 * sizeof(float)) != hipSuccess) return VDL2HIP_E_NOMEM;
	hipLaunchKernelGGL(k_u8_level_probe, dim3(1), dim3(256), 0, 0, d);
	const bool ok = hipMemcpy(out, d, 512 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
	(void)hipFree(d);
	return ok ? VDL2HIP_OK : VDL2HIP_E_DEVICE;
}

int vdl2hip_debug_dpp_probe(const float in[64], float out[256]) {
	float *d_in = nullptr, *d_out = nullptr;
	if(hipMalloc((void **)&d_in, 64 * 4) != hipSuccess || hipMalloc((void **)&d_out, 256 * 4) != hipSuccess) return VDL2HIP_E_NOMEM;
	int rc = VDL2HIP_OK;
	if(hipMemcpy(d_in, in, 64 * 4, hipMemcpyHostToDevice) != hipSuccess) rc = VDL2HIP_E_DEVICE;
	if(rc == VDL2HIP_OK) {
		hipLaunchKernelGGL(k_dpp_probe, dim3(1), dim3(64), 0, 0, (const float *)d_in, d_out);
		if(hipMemcpy(out, d_out, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = VDL2HIP_E_DEVICE;
	}
	(void)hipFree(d_in); (void)hipFree(d_out);
	return rc;
}

#ifdef VDL2_K1_PROF
int vdl2hip_debug_k1_prof(unsigned long long out[16], int reset) {
	static unsigned long long h[64][16];
	if(hipMemcpyFromSymbol(h, HIP_SYMBOL(vdl2_k1_prof), sizeof h) != hipSuccess) return -3;
	for(int k = 0; k < 16; k++) { out[k] = 0; for(int s = 0; s < 64; s++) out[k] += h[s][k]; }
	if(reset) { memset(h, 0, sizeof h); if(hipMemcpyToSymbol(HIP_SYMBOL(vdl2_k1_prof), h, sizeof h) != hipSuccess) return -3; }
	return 0;
}
#endif

#ifdef VDL2_K5_PROF
int vdl2hip_debug_k5_prof(unsigned long long out[16], int reset) {
	static unsigned long long h[64][16];
	if(hipMemcpyFromSymbol(h, HIP_SYMBOL(vdl2_k5_prof), sizeof h) != hipSuccess) return -3;
	for(int k = 0; k < 16; k++) { out[k] = 0; for(int s = 0; s < 64; s++) out[k] += h[s][k]; }
	if(reset) { memset(h, 0, sizeof h); if(hipMemcpyToSymbol(HIP_SYMBOL(vdl2_k5_prof), h, sizeof h) != hipSuccess) return -3; }
	return 0;
}
int vdl2hip_debug_nf_prof(unsigned long long out[16], int reset) {
	static unsigned long long h[64][16];
	if(hipMemcpyFromSymbol(h, HIP_SYMBOL(vdl2_nf_prof), sizeof h) != hipSuccess) return -3;
	for(int k = 0; k < 16; k++) { out[k] = 0; for(int s = 0; s < 64; s++) out[k] += h[s][k]; }
	if(reset) { memset(h, 0, sizeof h); if(hipMemcpyToSymbol(HIP_SYMBOL(vdl2_nf_prof), h, sizeof h) != hipSuccess) return -3; }
	return 0;
}
int vdl2hip_debug_k4_prof(unsigned long long out[24], int reset) {
	static unsigned long long h[64][24];
	if(hipMemcpyFromSymbol(h, HIP_SYMBOL(vdl2_k4_prof), sizeof h) != hipSuccess) return -3;
	for(int k = 0; k < 24; k++) { out[k] = 0; for(int s = 0; s < 64; s++) out[k] += h[s][k]; }
	if(reset) { memset(h, 0, sizeof h); if(hipMemcpyToSymbol(HIP_SYMBOL(vdl2_k4_prof), h, sizeof h) != hipSuccess) return -3; }
	return 0;
}
#endif

int vdl2hip_get_lpf(vdl2hip_ctx *c, float A[3], float B[3]) {
	if(!c) return VDL2HIP_E_INVAL;
	memcpy(A, c->lpf.A, sizeof c->lpf.A); memcpy(B, c->lpf.B, sizeof c->lpf.B);
	return VDL2HIP_OK;
}

int vdl2hip_get_nco_step(vdl2hip_ctx *c, uint32_t chan, uint32_t *dphi) {
	if(!c || !dphi || chan < (uint32_t)c->chan_first || chan >= (uint32_t)(c->chan_first + c->C)) return VDL2HIP_E_INVAL;
	*dphi = nco_step(c->cfg.centerfreq, c->freqs[chan - c->chan_first], (uint32_t)kSymbolRate * kSps * c->os);
	return VDL2HIP_OK;
}

int vdl2hip_read_decimated(vdl2hip_ctx *c, uint32_t chan, int64_t first, float *dst, size_t cap) {
	if(!c || !dst || chan < (uint32_t)c->chan_first || chan >= (uint32_t)(c->chan_first + c->C)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	if(first < 0 || first > c->k_total || c->k_total - first > (int64_t)c->cap) return VDL2HIP_E_INVAL;
	size_t n = std::min<size_t>(cap, (size_t)(c->k_total - first));
	const cf32 *base = c->d_y + (size_t)(chan - c->chan_first) * c->cap;
	size_t done = 0;
	while(done < n) {
		uint32_t slot = (uint32_t)(first + (int64_t)done) & (c->cap - 1);
		size_t m = std::min<size_t>(n - done, c->cap - slot);
		HIPCHK(hipMemcpy(dst + 2 * done, base + slot, m * sizeof(cf32), hipMemcpyDeviceToHost));
		done += m;
	}
	return (int)n;
}

// development aid: the referee's event log (ref_debug_log) of the channel chosen with debug option "ref_debug_chan": out[4 * i ..] per entry
int vdl2hip_debug_ref_log(vdl2hip_ctx *c, unsigned long long *out, size_t cap_entries) {
	if(!c || !c->d_refdbg) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	(void)collect_pending(c);
	unsigned long long n = 0;
	HIPCHK(hipMemcpy(&n, c->d_refdbg, 8, hipMemcpyDeviceToHost));
	if(n > 1000) n = 1000;
	if(n > cap_entries) n = cap_entries;
	if(n) HIPCHK(hipMemcpy(out, c->d_refdbg + 1, 32 * n, hipMemcpyDeviceToHost));
	return (int)n;
}

// test hook (not declared in vdl2hip.h): the decisions the walk of the LAST feed has asked the referee to check (optimistic mode), as
// (chan, kind, n, k0 of the feed) quadruples of int64; returns how many there were
int vdl2hip_debug_read_requests(vdl2hip_ctx *c, int64_t *out, size_t cap) {
	if(!c || !out) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	if(c->feed_no == 0 || !c->referee) return 0;
	OutSlot &sl = c->slot[(c->feed_no - 1) % kSlots];
	uint32_t n = 0;
	HIPCHK(hipMemcpy(&n, sl.d_rqn, 4, hipMemcpyDeviceToHost));
	const uint32_t m = std::min<uint32_t>(n, c->rq_cap);
	std::vector<RefReq> rq(m);
	if(m) HIPCHK(hipMemcpy(rq.data(), sl.d_rq, (size_t)m * sizeof(RefReq), hipMemcpyDeviceToHost));
	for(uint32_t i = 0; i < m && i < cap; i++) { out[4 * i] = rq[i].chan; out[4 * i + 1] = rq[i].kind; out[4 * i + 2] = rq[i].n; out[4 * i + 3] = sl.back_k0; }
	return (int)n;
}

// test hook (not declared in vdl2hip.h): what the sync kernels left for decimated samples first .. first+count-1 of one channel - the tabulated
// metric {pherr (its sign: the referee's mark), slope} and the candidate bit, one byte per sample
int vdl2hip_debug_read_sync(vdl2hip_ctx *c, uint32_t chan, int64_t first, size_t count, float *pf, uint8_t *cand) {
	if(!c || !pf || !cand || chan < (uint32_t)c->chan_first || chan >= (uint32_t)(c->chan_first + c->C)) return VDL2HIP_E_INVAL;
	OnDevice dev_guard(c);
	int r = collect_pending(c);
	if(r != VDL2HIP_OK && r != VDL2HIP_E_OVERFLOW) return r;
	if(first < 0 || first > c->k_total || c->k_total - first > (int64_t)c->cap) return VDL2HIP_E_INVAL;
	const size_t n = std::min<size_t>(count, (size_t)(c->k_total - first)), ch = chan - c->chan_first;
	std::vector<uint64_t> words(c->cap >> 6);
	HIPCHK(hipMemcpy(words.data(), c->d_cand + ch * (c->cap >> 6), words.size() * 8, hipMemcpyDeviceToHost));
	for(size_t i = 0; i < n; i++) {
		const uint32_t slot = (uint32_t)(first + (int64_t)i) & (c->cap - 1);
		cand[i] = (uint8_t)((words[slot >> 6] >> (slot & 63)) & 1u);
	}
	size_t done = 0;
	while(done < n) {
		const uint32_t slot = (uint32_t)(first + (int64_t)done) & (c->cap - 1);
		const size_t m = std::min<size_t>(n - done, c->cap - slot);
		HIPCHK(hipMemcpy(pf + 2 * done, c->d_pf + ch * c->cap + slot, m * sizeof(cf32), hipMemcpyDeviceToHost));
		done += m;
	}
	return (int)n;
}

}  // extern "C"

#include "group.inc"   // vdl2hip_group_*: one receiver over several GPUs of this process (needs the statics above)
#include "ubench.inc"  // vdl2hip_debug_ubench(): issue and LDS-gather rates measured on the spot (bench.py's roofline line)
