// hostsim.cpp - TEST INFRASTRUCTURE.  Compiles dumpvdl2_amd/csrc/vdl2_core.h (the
// source of the walker and burst-decoder kernels) with plain g++ and runs it on the
// CPU, one "wavefront" = a 64-iteration loop, so the burst-level device logic can be
// unit-tested without a GPU.  Input is the decimated stream of each channel (e.g. the
// oracle's trace); output is the frame list.  Never linked into libvdl2hip.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../dumpvdl2_amd/csrc/vdl2_core.h"
#include "../../dumpvdl2_amd/csrc/tables.h"
#include "../../dumpvdl2_amd/csrc/design.h"

// The referee's hook of the CPU build: the "reference's own samples" come from a trace the test supplies (the oracle's decimated
// stream); the device build works them out from the raw input (kernels.h).
namespace vdl2 {
struct RefChan { const float *exact; int64_t n_exact; cf32 *y; uint32_t mask; int64_t calls, samples; std::vector<uint8_t> *done; };
inline void ref_debug_log(const ChanView &, int tag, int64_t a, float b, float c, float d) { if(getenv("HOSTSIM_REF_LOG")) fprintf(stderr, "reflog %d %lld %.9g %.9g %.9g\n", tag, (long long)a, b, c, d); }
inline bool ref_exact_window(const ChanView &v, int64_t n_lo, int64_t n_hi, void *, int kind) {
	RefChan *r = v.ref;
	if(!r || !r->exact) return false;
	r->calls++;
	if(getenv("HOSTSIM_REF_CALLS")) fprintf(stderr, "refcall y=%p kind %d lo %lld hi %lld\n", (void *)r->y, kind, (long long)n_lo, (long long)n_hi);
	for(int64_t n = n_lo < 0 ? 0 : n_lo; n <= n_hi && n < r->n_exact; n++) { r->y[(uint32_t)n & r->mask] = cf32{ r->exact[2 * n], r->exact[2 * n + 1] }; r->samples++; if(r->done) (*r->done)[n] = 1; }
	return true;
}
// (the device: has ANOTHER launch made the stretch exact - here: has anybody)
inline bool ref_window_done(const ChanView &v, int64_t n_lo, int64_t n_hi) {
	RefChan *r = v.ref;
	if(!r || !r->exact || !r->done) return false;
	for(int64_t n = n_lo < 0 ? 0 : n_lo; n <= n_hi && n < r->n_exact; n++) if(!(*r->done)[n]) return false;
	return true;
}
}

using namespace vdl2;

// metric_contiguous() plus the referee's error figure E (vdl2_core.h: sync_metric_ref)
static void metric_contiguous_ref(const ChanView &v, int64_t n, const Tables &T, float &p, float &f, float &E, float &alo, float &ahi, bool full) {
	float ph[kPreamble], e2[kPreamble];
	for(int i = 0; i < kPreamble; i++) { const int64_t t = n - 150 + 10 * i; ph[i] = v.Phi(t); e2[i] = ref_eps2(v, t); }
	sync_metric_ref(ph, e2, 1, T, p, f, E, alo, ahi, full);
}

struct Sim {
	int nchan; uint32_t cap, mask; float max_ppm;
	std::vector<uint32_t> freqs;
	std::vector<cf32> y, pf; std::vector<uint64_t> cand;
	std::vector<WalkState> st; std::vector<unsigned long long> cnt, acnt;
	std::vector<NfState> nf; std::vector<EvalChunk> log; std::vector<uint32_t> nlog; std::vector<int64_t> scf, scc; std::vector<float> ring, lpbuf;
	uint32_t cap_log = 8192, cap_comb = 8192 + kNfTail, cap_hist = 4096, nf_ring = 16384;
	Tables T;
	int64_t k_total = 0;
	std::vector<Burst> bursts; std::vector<OutFrame> frames; std::vector<uint8_t> pool;
	std::vector<OutFrame> all_frames; std::vector<uint8_t> all_pool;
	OutCtl ctl;
	int64_t seg_min = 0; int seg_max = 1;          // segmented walk (off by default)
	bool two_tier = false; int64_t n_exact = 0, n_total = 0;   // K3's screening rule instead of the exact metric everywhere
	std::vector<SpecOut> spec; uint32_t seg_stats[2] = {0, 0};
	// referee: exact samples [nchan][exact_D] and the per-channel hooks; marg: candidates K3 marked as within the margin
	std::vector<float> exact; int64_t exact_D = 0; std::vector<RefChan> rc; std::vector<std::vector<uint8_t>> rdone; bool prescan = false; std::vector<int64_t> pre_lo, pre_hi; int64_t n_marg = 0, n_cand = 0, n_walk_windows = 0;
	std::vector<float> pe, pa, pb;
	bool optimistic = true; std::vector<RefReq> rq; uint32_t rq_n = 0; std::vector<uint32_t> rq_flag; std::vector<RefBad> rq_bad; std::vector<WalkState> ws_snap; std::vector<unsigned long long> cnt_snap; int64_t n_rewalk = 0, n_requests = 0;
	std::vector<Burst> all_bursts;   // every burst descriptor the walker has emitted (debugging aid)
};

extern "C" {

// sync_metric_unwrap_alts() (vdl2_core.h: the values of the sync metric with one or two unwrap decisions taken the other way, from the
// residuals of the metric as it is) against the metric run again with those decisions forced: `trials` random windows; returns the
// number of windows whose brute-force range is NOT inside the short cut's, *worst = the largest excess of the short cut's range over
// the brute-force one in units of its own slack (1e-3 + 2e-5 |v|)
int hostsim_check_unwrap_alts(int trials, unsigned seed, double *worst) {
	static Tables T; build_tables(T);
	srand(seed);
	auto forced = [&](const float *ph, int f1, int f2) {
		float e[kPreamble]; float mean = 0.f, unwrap = 0.f;
		float prev = mean = e[0] = ph[0] - T.pr_phase[0];
		for(int i = 1; i < kPreamble; i++) {
			const float cur = ph[i] - T.pr_phase[i], diff = cur - prev; prev = cur;
			double step = diff > kPiBelow ? -(2.0f * M_PI) : (diff < -kPiBelow ? (2.0f * M_PI) : 0.0);
			if(i == f1 || i == f2) step = step != 0.0 ? 0.0 : (diff > 0.f ? -(2.0f * M_PI) : (2.0f * M_PI));
			unwrap = (float)((double)unwrap + step); e[i] = cur + unwrap; mean += e[i];
		}
		mean /= kPreamble;
		for(int i = 0; i < kPreamble; i++) e[i] -= mean;
		float slope = 0.f;
		for(int i = 0; i < kPreamble; i++) slope += T.lrx[i] * e[i];
		slope /= T.lr_den;
		float acc = 0.f;
		for(int i = 0; i < kPreamble; i++) { const float r = e[i] - slope * T.lrx[i]; acc += r * r; }
		return acc;
	};
	int bad = 0; double w = 0.0;
	for(int t = 0; t < trials; t++) {
		float ph[kPreamble];
		for(int i = 0; i < kPreamble; i++) ph[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * M_PI);
		int ev[2] = { 1 + rand() % (kPreamble - 1), 1 + rand() % (kPreamble - 1) }; const int nev = 1 + rand() % 2;
		if(nev == 2 && ev[0] == ev[1]) continue;
		if(nev == 2 && ev[0] > ev[1]) std::swap(ev[0], ev[1]);
		float p, f; sync_metric(ph, T, p, f);
		float lo, hi; sync_metric_unwrap_alts(ph, T, ev, nev, p, lo, hi);
		float blo = p, bhi = p;
		for(int c = 1; c < (1 << nev); c++) {
			const float v = forced(ph, (c & 1) ? ev[0] : -1, (c & 2) ? ev[1] : -1);
			blo = std::min(blo, v); bhi = std::max(bhi, v);
		}
		if((lo > blo && !(lo == 0.f)) || hi < bhi) bad++;
		w = std::max(w, std::max((double)blo - lo, (double)hi - bhi) / (1e-3 + 2e-5 * std::max(fabs(bhi), fabs(blo))));
	}
	if(worst) *worst = w;
	return bad;
}

Sim *hostsim_create(int nchan, const uint32_t *freqs, float max_ppm, int cap_log2) {
	Sim *s = new Sim();
	s->nchan = nchan; s->cap = 1u << cap_log2; s->mask = s->cap - 1; s->max_ppm = max_ppm;
	s->freqs.assign(freqs, freqs + nchan);
	s->y.assign((size_t)nchan * s->cap, cf32{0, 0}); s->pf.assign((size_t)nchan * s->cap, cf32{0, 0});
	s->cand.assign((size_t)nchan * (s->cap / 64), 0);
	s->st.resize(nchan); s->cnt.assign((size_t)nchan * kNumCounters, 0); s->acnt.assign((size_t)nchan * kNumAvlcCounters, 0);
	for(auto &w : s->st) { memset(&w, 0, sizeof w); walk_state_init(w); }
	s->nf.resize(nchan); for(auto &n : s->nf) { memset(&n, 0, sizeof n); nf_state_init(n); }
	s->log.resize((size_t)nchan * s->cap_log); s->nlog.assign(nchan, 0); s->scf.resize((size_t)nchan * (s->cap_comb + 1)); s->scc.resize((size_t)nchan * (s->cap_comb + 1));
	s->ring.assign((size_t)nchan * s->nf_ring, 0.f); s->lpbuf.assign((size_t)nchan * s->cap_hist, 0.f);
	build_tables(s->T);
	s->bursts.resize(65536); s->frames.resize(65536); s->pool.resize(1 << 24);
	return s;
}

void hostsim_destroy(Sim *s) { delete s; }

// walk feeds of at least 2*seg_min decimated samples in up to seg_max speculative segments (0 = plain sequential walk)
void hostsim_set_segments(Sim *s, int64_t seg_min, int seg_max) { s->seg_min = seg_min; s->seg_max = seg_max < 1 ? 1 : seg_max > kMaxSeg ? kMaxSeg : seg_max; }
void hostsim_set_two_tier(Sim *s, int on) { s->two_tier = on != 0; }
void hostsim_set_optimistic(Sim *s, int on) { s->optimistic = on != 0; }
void hostsim_set_prescan(Sim *s, int on) { s->prescan = on != 0; }
// referee on: decisions within the margin of the stream's error are taken on `exact` ([nchan][D] complex, the oracle's trace)
void hostsim_set_exact(Sim *s, const float *exact, int64_t D) {
	s->exact.assign(exact, exact + (size_t)s->nchan * D * 2); s->exact_D = D;
	s->rc.resize(s->nchan); s->rdone.assign(s->nchan, std::vector<uint8_t>((size_t)D, 0));
	for(int c = 0; c < s->nchan; c++) s->rc[c] = RefChan{ s->exact.data() + (size_t)c * D * 2, D, &s->y[(size_t)c * s->cap], s->mask, 0, 0, &s->rdone[c] };
	s->pe.assign((size_t)s->nchan * s->cap, 0.f); s->pa.assign((size_t)s->nchan * s->cap, 0.f); s->pb.assign((size_t)s->nchan * s->cap, 0.f);
}
// [0] candidates K3 marked, [1] candidate bits set, [2] exact windows served, [3] samples replaced, [4] windows asked for by the walkers (one feed)
void hostsim_referee_stats(Sim *s, int64_t out[7]) {
	out[0] = s->n_marg; out[1] = s->n_cand; out[2] = out[3] = 0; out[4] = s->n_walk_windows; out[5] = s->n_requests; out[6] = s->n_rewalk;
	for(auto &r : s->rc) { out[2] += r.calls; out[3] += r.samples; }
}
void hostsim_two_tier_stats(Sim *s, int64_t out[2]) { out[0] = s->n_exact; out[1] = s->n_total; }
void hostsim_segment_stats(Sim *s, uint32_t out[2]) { out[0] = s->seg_stats[0]; out[1] = s->seg_stats[1]; }

// y: [nchan][D] complex (re,im) floats, channel-major
int hostsim_feed(Sim *s, const float *yin, int64_t D) {
	const int64_t k0 = s->k_total, k1 = k0 + D;
	for(int c = 0; c < s->nchan; c++) {
		cf32 *y = &s->y[(size_t)c * s->cap];
		cf32 *pf = &s->pf[(size_t)c * s->cap]; uint64_t *cand = &s->cand[(size_t)c * (s->cap / 64)];
		for(int64_t k = k0; k < k1; k++) {
			cf32 v{ yin[((size_t)c * D + (k - k0)) * 2], yin[((size_t)c * D + (k - k0)) * 2 + 1] };
			y[(uint32_t)k & s->mask] = v;
		}
		ChanView cv{ y, pf, cand, s->mask };
		const bool ref_on = !s->rc.empty();
		float *pe = ref_on ? &s->pe[(size_t)c * s->cap] : nullptr, *pa = ref_on ? &s->pa[(size_t)c * s->cap] : nullptr, *pb = ref_on ? &s->pb[(size_t)c * s->cap] : nullptr;
		// sync kernel: whole 64-aligned words covering [k0, k1)
		if(!s->two_tier) {
			for(int64_t n = k0 & ~63ll; n < ((k1 + 63) & ~63ll); n++) {
				cf32 r = (n < k1) ? metric_contiguous(cv, n, s->T) : cf32{kPherrBig, 0.f};
				if(ref_on && n < k1) { float p_, f_, E_, a_, b_; metric_contiguous_ref(cv, n, s->T, p_, f_, E_, a_, b_, s->prescan); pe[(uint32_t)n & s->mask] = E_; pa[(uint32_t)n & s->mask] = a_; pb[(uint32_t)n & s->mask] = b_; }
				pf[(uint32_t)n & s->mask] = r;
			}
		} else {
			// K3's rule (kernels.h:k_sync_screen / k_sync_exact): the screening value, from single-precision phases, everywhere;
			// the exact arithmetic only where the screening value is under kScreenThr or 3 samples either side of such a place,
			// or where the right neighbour has not arrived yet
			const int64_t nb = (k0 & ~63ll) - 3, ne = (k1 + 63) & ~63ll;
			std::vector<float> scr((size_t)(ne + 3 - nb));
			auto screen_at = [&](int64_t n) -> float {
				if(n < 0 || n >= k1) return kPherrBig;
				float ph[kPreamble];
				for(int i = 0; i < kPreamble; i++) { int64_t t = n - 150 + 10 * i; ph[i] = t < 0 ? 0.f : phase_fast(y[(uint32_t)t & s->mask]); }
				float d[kPreamble];
				for(int i = 1; i < kPreamble; i++) d[i] = ph[i] - ph[i - 1];
				ScreenAcc a; screen_begin(a); screen_taps(d, 1, kScreenEarly, a);
				float v = screen_value(a, kScreenEarly);
				if(v < kScreenEarlyThr) { screen_taps(d, kScreenEarly, kPreamble, a); v = screen_value(a, kPreamble); }
				return v;
			};
			for(int64_t n = nb; n < ne + 3; n++) scr[(size_t)(n - nb)] = screen_at(n);
			auto fl = [&](int64_t n) -> bool { return n >= 0 && n < k1 && scr[(size_t)(n - nb)] < kScreenThr; };
			for(int64_t n = k0 & ~63ll; n < ne; n++) {
				cf32 r{kPherrBig, 0.f};
				if(n < k1) {
					const bool need = fl(n - 3) || fl(n) || (n + 3 < k1 ? fl(n + 3) : true);
					// the kernel stores a metric value only where it computed the exact one; everywhere else the ring keeps whatever
					// an earlier lap left there, which the walker must never look at: the simulation puts poison there
					r = need ? metric_contiguous(cv, n, s->T) : cf32{12345.f, 54321.f};
					if(ref_on && need) { float p_, f_, E_, a_, b_; metric_contiguous_ref(cv, n, s->T, p_, f_, E_, a_, b_, s->prescan); pe[(uint32_t)n & s->mask] = E_; pa[(uint32_t)n & s->mask] = a_; pb[(uint32_t)n & s->mask] = b_; }
					s->n_exact += need; s->n_total++;
				}
				pf[(uint32_t)n & s->mask] = r;
			}
		}
		for(int64_t w = k0 >> 6; w < ((k1 + 63) >> 6); w++) {
			uint64_t bits = 0; int64_t wm_first = -1, wm_last = -1;
			for(int b = 0; b < 64; b++) {
				int64_t n = (w << 6) + b;
				if(n >= k1 || n < 3) continue;
				if(!ref_on) { if(is_candidate(pf[(uint32_t)(n - 3) & s->mask].re, pf[(uint32_t)n & s->mask].re)) bits |= 1ull << b; continue; }
				// referee (the device's K3 exact tier does the same, kernels.h): "may fire" in the bitmap, "within the margin" as the sign of pf[n].p
				auto R = [&](int64_t m) -> RefRange {
					if(m < 0) return RefRange{ kPherrBig, kPherrBig };
					const float p = fabsf(pf[(uint32_t)m & s->mask].re);
					return p > 999.f ? RefRange{ kPherrBig, kPherrBig } : ref_pherr_range(p, pa[(uint32_t)m & s->mask], pb[(uint32_t)m & s->mask], pe[(uint32_t)m & s->mask]);
				};
				const int vd = ref_candidate_verdict(R(n), R(n - 3), pf[(uint32_t)(n - 3) & s->mask].im, pe[(uint32_t)(n - 3) & s->mask], R(n - 6), s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm));
				if(getenv("HOSTSIM_DEBUG_AT") && c == atoi(getenv("HOSTSIM_DEBUG_CH")) && llabs(n - atoll(getenv("HOSTSIM_DEBUG_AT"))) <= 9) {
					const RefRange a0 = R(n), a3 = R(n - 3), a6 = R(n - 6);
					fprintf(stderr, "at c=%d n=%lld vd=%d p0=%.6f r6=[%g,%g] r3=[%g,%g] r0=[%g,%g] f3=%g E0=%g palt0=%g\n", c, (long long)n, vd, pf[(uint32_t)n & s->mask].re, a6.lo, a6.hi, a3.lo, a3.hi, a0.lo, a0.hi,
						pf[(uint32_t)(n - 3) & s->mask].im, pe[(uint32_t)n & s->mask], pa[(uint32_t)n & s->mask]);
				}
				if((vd & 2) && getenv("HOSTSIM_DEBUG_REF")) {
					static int shown = 0;
					const RefRange a0 = R(n), a3 = R(n - 3), a6 = R(n - 6);
					if(n > 3000 && shown++ < 40) fprintf(stderr, "marg c=%d n=%lld r6=[%g,%g] r3=[%g,%g] r0=[%g,%g] f3=%g E3=%g\n", c, (long long)n, a6.lo, a6.hi, a3.lo, a3.hi, a0.lo, a0.hi,
						pf[(uint32_t)(n - 3) & s->mask].im, pe[(uint32_t)(n - 3) & s->mask]);
				}
				if((vd & 2) && n > 3000 && getenv("HOSTSIM_REF_REASONS")) {
					static long cnt[6]; static long tot;
					const RefRange r0 = R(n), r3 = R(n - 3), r6 = R(n - 6);
					const bool surely = r3.hi < kSyncThr && r0.lo > r3.hi;
					int why = 0;
					if(!surely) why = !(r3.hi < kSyncThr) ? 0 : 1;
					else if(ref_vertex_marginal(RefRange{ kPherrBig, kPherrBig }, r3, r0)) why = 2;
					else if(r6.lo < kPherrBig && ref_vertex_marginal(r6, r3, r0)) why = 3;
					else why = 4;
					if(r3.hi >= kRefBig || r0.hi >= kRefBig || (r6.lo < kPherrBig && r6.hi >= kRefBig)) why = 5;
					cnt[why]++; tot++;
					if(tot % 5 == 0) fprintf(stderr, "reasons: p3~4 %ld, p0~p3 %ld, vertex(y1 max) %ld, vertex(p6) %ld, gate %ld, cannot tell %ld\n", cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5]);
				}
				if(vd & 1) { bits |= 1ull << b; s->n_cand++; }
				if(vd & 2) { cf32 &q = pf[(uint32_t)n & s->mask]; q.re = -fabsf(q.re); s->n_marg++; if(wm_first < 0) wm_first = n; wm_last = n; }
			}
			cand[(uint32_t)w & (s->mask >> 6)] = bits;
			if(wm_first >= 0 && s->prescan) {      // (the device's exact tier lists the stretch; k_ref_scan_multi makes it exact before the walk)
				int64_t lo = (wm_first - kRefPre) & ~255ll, hi = (wm_last + kRefPost) | 255; if(lo < 0) lo = 0; if(hi > k1 - 1) hi = k1 - 1;
				s->pre_lo.push_back(((int64_t)c << 40) | lo); s->pre_hi.push_back(hi);
			}
		}
	}
	for(size_t i = 0; i < s->pre_lo.size(); i++) {
		const int c = (int)(s->pre_lo[i] >> 40); const int64_t lo = s->pre_lo[i] & ((1ll << 40) - 1);
		ChanView cv{ &s->y[(size_t)c * s->cap], nullptr, nullptr, s->mask }; cv.ref = &s->rc[c];
		(void)ref_exact_window(cv, lo, s->pre_hi[i], nullptr, REF_CANDIDATE);
	}
	s->pre_lo.clear(); s->pre_hi.clear();
	s->k_total = k1;
	memset(&s->ctl, 0, sizeof s->ctl);
	s->ctl.cap_bursts = (uint32_t)s->bursts.size(); s->ctl.cap_frames = (uint32_t)s->frames.size(); s->ctl.cap_pool = (uint32_t)s->pool.size(); s->ctl.cap_log = s->cap_log;
	static WalkShared wsh;
	const bool opt = !s->rc.empty() && s->optimistic;
	if(opt) { s->rq.resize(8192); s->rq_n = 0; s->rq_flag.assign(s->nchan, 0); s->rq_bad.assign(s->nchan, RefBad{}); s->ws_snap.resize(s->nchan); s->cnt_snap.resize((size_t)s->nchan * kNumCounters); }
	const WalkSnap snap{ opt ? s->ws_snap.data() : nullptr, opt ? s->cnt_snap.data() : nullptr };
	std::vector<uint32_t> nb_first(s->nchan, 0);
	for(int c = 0; c < s->nchan; c++) {
		ChanView v{ &s->y[(size_t)c * s->cap], &s->pf[(size_t)c * s->cap], &s->cand[(size_t)c * (s->cap / 64)], s->mask };
		if(!s->rc.empty()) { v.ref = &s->rc[c]; v.ref_pre = s->prescan; }
		if(opt) { v.rq = s->rq.data(); v.rq_n = &s->rq_n; v.rq_cap = (uint32_t)s->rq.size(); v.rq_flag = s->rq_flag.data(); v.rq_bad = &s->rq_bad[c]; }
		nb_first[c] = s->ctl.nbursts;
		EvalLog lg{ &s->log[(size_t)c * s->cap_log], &s->nlog[c] };
		uint32_t nbc = 0;
		int nseg = s->seg_min > 0 ? (int)std::min<int64_t>(s->seg_max, D / s->seg_min) : 1;
		int64_t seglen = D;
		if(nseg >= 2) {
			// same three steps as k_walk_spec / k_walk_stitch
			seglen = (D + nseg - 1) / nseg;
			nseg = (int)((D + seglen - 1) / seglen);
			Burst *bdst = s->bursts.data() + s->ctl.nbursts; const uint32_t bcap = (uint32_t)s->bursts.size() - s->ctl.nbursts;
			s->spec.resize((size_t)3 * (nseg - 1));
			for(int x = 0; x < 3 * (nseg - 1); x++) {
				const int seg = 1 + x / 3, r = x % 3;
				const int64_t b = k0 + (int64_t)seg * seglen, kn = seg + 1 < nseg ? b + seglen : k1;
				spec_walk(c, s->freqs[c], s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm), b, r, kn, s->T, v, &s->spec[x], wsh);
			}
			static StitchShared ssh;
			stitch_channel(c, s->freqs[c], s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm), k0, seglen, nseg, k1, s->T, v, &s->st[c], &s->cnt[(size_t)c * kNumCounters], bdst, bcap, &nbc,
			               &s->ctl, lg, s->spec.data(), wsh, ssh, s->seg_stats, snap);
		} else
		walk_channel(c, s->freqs[c], s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm), k1, s->T, v, &s->st[c], &s->cnt[(size_t)c * kNumCounters], s->bursts.data() + s->ctl.nbursts,
		             (uint32_t)s->bursts.size() - s->ctl.nbursts, &nbc, &s->ctl, lg, wsh, snap);
		if(opt) {
			// the device checks a feed's noted decisions all at once after the walks and walks the (rare) channel again; here channel by channel
			static float vlds[64];
			ChanView vv = v; vv.rq = nullptr; vv.rq_n = nullptr; vv.rq_cap = 0; vv.rq_flag = nullptr;
			const uint32_t nreq = s->rq_n < (uint32_t)s->rq.size() ? s->rq_n : (uint32_t)s->rq.size();
			for(uint32_t i = 0; i < nreq; i++) { s->n_requests++; if(!ref_verify(s->rq[i], s->freqs[c], s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm), k1, s->T, vv, vlds)) { s->rq_flag[c] = 1; RefBad &B = s->rq_bad[c]; if(B.n < (uint32_t)kRefBad) B.at[B.n] = 4 * s->rq[i].n + s->rq[i].kind; B.n++; } }
			s->rq_n = 0;
			if(s->rq_flag[c] && nseg >= 2) {
				s->n_rewalk++;
				static StitchShared ssh2;
				Burst *bdst = s->bursts.data() + s->ctl.nbursts; const uint32_t bcap = (uint32_t)s->bursts.size() - s->ctl.nbursts;
				stitch_channel(c, s->freqs[c], s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm), k0, seglen, nseg, k1, s->T, vv, &s->st[c], &s->cnt[(size_t)c * kNumCounters], bdst, bcap, &nbc,
				               &s->ctl, lg, s->spec.data(), wsh, ssh2, s->seg_stats, snap, true);
			} else if(s->rq_flag[c]) {
				s->n_rewalk++;
				walk_again(c, s->freqs[c], s->max_ppm, ppm_gate_threshold(s->freqs[c], s->max_ppm), k1, s->T, vv, &s->st[c], &s->cnt[(size_t)c * kNumCounters], s->bursts.data() + s->ctl.nbursts,
				           (uint32_t)s->bursts.size() - s->ctl.nbursts, &nbc, &s->ctl, lg, wsh, snap);
			}
		}
		s->ctl.nbursts += nbc;
		static NfShared nsh;
		NfScratch sc{ &s->scf[(size_t)c * (s->cap_comb + 1)], &s->scc[(size_t)c * (s->cap_comb + 1)] };
		NfFeed fd;
		nf_prepare(&s->nf[c], lg, sc, s->cap_comb, &fd, nsh);
		for(int64_t g = 0; fd.u0 + 1 + kNfGroup * g <= fd.u1; g++) nf_replay_group(v, sc, fd, g, &s->lpbuf[(size_t)c * s->cap_hist], s->cap_hist, nsh);
		nf_finish(&s->nf[c], sc, fd, &s->lpbuf[(size_t)c * s->cap_hist], &s->ring[(size_t)c * s->nf_ring], s->nf_ring - 1, s->cap_hist, nsh);
	}
	for(auto &r : s->rc) s->n_walk_windows += r.calls;      // (cumulative below: calls made by the walkers so far)
	static BurstShared bsh;
	s->ctl.nframes = burst_reserve_initial_frames(1); s->ctl.pool_used = burst_reserve_initial_pool(1);   // one "wavefront" decodes everything
	burst_shared_init(s->T, 0, &s->ctl, bsh);
	uint32_t nb = s->ctl.nbursts;
	s->all_bursts.insert(s->all_bursts.end(), s->bursts.begin(), s->bursts.begin() + nb);
	for(uint32_t i = 0; i < nb; i++) {
		const Burst &b = s->bursts[i];
		int c = b.chan;
		ChanView v{ &s->y[(size_t)c * s->cap], &s->pf[(size_t)c * s->cap], &s->cand[(size_t)c * (s->cap / 64)], s->mask };
		if(!s->rc.empty()) { v.ref = &s->rc[c]; v.ref_pre = s->prescan; }
		decode_burst(b, s->freqs[c], s->T, v, &s->cnt[(size_t)c * kNumCounters], s->frames.data(), s->pool.data(), &s->ctl, bsh);
	}
	burst_reserve_done(s->frames.data(), bsh);
	uint32_t nf = s->ctl.nframes < s->ctl.cap_frames ? s->ctl.nframes : s->ctl.cap_frames;
	static FrameShared fsh;
	frame_shared_init(s->T, fsh);
	for(uint32_t i = 0; i < nf; i++) {
		const int c = s->frames[i].chan;
		if(c < 0) continue;                                  // tombstone: a record the burst decoder reserved and did not use
		finish_frame(s->frames[i], s->pool.data(), s->T, &s->acnt[(size_t)c * kNumAvlcCounters], &s->ring[(size_t)c * s->nf_ring], s->nf_ring - 1, fsh);
	}
	for(uint32_t i = 0; i < nf; i++) {
		OutFrame f = s->frames[i];
		if(f.chan < 0) continue;
		uint32_t off = (uint32_t)s->all_pool.size();
		s->all_pool.insert(s->all_pool.end(), s->pool.begin() + f.pool_off, s->pool.begin() + f.pool_off + f.len);
		f.pool_off = off;
		s->all_frames.push_back(f);
	}
	return s->ctl.overflow ? -1 : (int)nf;
}

// debugging aid: (chan, sync_sample, t_first, nsym, tl_bits, syndrome, vdphi_err * 1e6, prev_n) of every burst descriptor so far
int64_t hostsim_bursts(Sim *s, int64_t *out, int64_t cap) {
	int64_t n = 0;
	for(const Burst &b : s->all_bursts) { if(n >= cap) break; int64_t *o = out + 8 * n++; o[0] = b.chan; o[1] = b.sync_sample; o[2] = b.t_first; o[3] = b.nsym; o[4] = b.tl_bits; o[5] = b.syndrome; o[6] = (int64_t)(b.vdphi_err * 1e6f); o[7] = b.prev_n; }
	return n;
}
// what the sync stage left for samples first .. first+count-1 of one channel: {pherr (sign: the referee's mark), slope}, candidate bit
void hostsim_read_sync(Sim *s, int chan, int64_t first, int64_t count, float *pf, uint8_t *cand) {
	for(int64_t i = 0; i < count; i++) {
		const uint32_t slot = (uint32_t)(first + i) & s->mask;
		const cf32 q = s->pf[(size_t)chan * s->cap + slot];
		pf[2 * i] = q.re; pf[2 * i + 1] = q.im;
		cand[i] = (uint8_t)((s->cand[(size_t)chan * (s->cap / 64) + (slot >> 6)] >> (slot & 63)) & 1u);
	}
}
int64_t hostsim_num_frames(Sim *s) { return (int64_t)s->all_frames.size(); }
const OutFrame *hostsim_frames(Sim *s) { return s->all_frames.data(); }
const uint8_t *hostsim_pool(Sim *s) { return s->all_pool.data(); }
void hostsim_avlc_counters(Sim *s, int chan, unsigned long long *out) { memcpy(out, &s->acnt[(size_t)chan * kNumAvlcCounters], sizeof(unsigned long long) * kNumAvlcCounters); }
void hostsim_counters(Sim *s, int chan, unsigned long long *out) { memcpy(out, &s->cnt[(size_t)chan * kNumCounters], sizeof(unsigned long long) * kNumCounters); }
// phase_of() of the device code on n (re, im) pairs, for comparison with libm
void hostsim_phase(const float *reim, float *out, int64_t n) { for(int64_t i = 0; i < n; i++) out[i] = phase_of(cf32{reim[2 * i], reim[2 * i + 1]}); }
// the screening-tier phase of the sync kernel (turns)
void hostsim_phase_fast(const float *reim, float *out, int64_t n) { for(int64_t i = 0; i < n; i++) out[i] = phase_fast(cf32{reim[2 * i], reim[2 * i + 1]}); }
float hostsim_screen_guard() { return kScreenGuard; }
double hostsim_atan2(double y, double x) { return atan2_f64(y, x); }

int hostsim_sizeof_outframe() { return (int)sizeof(OutFrame); }

// got_sync() metric of n phase windows (16 taps each): the exact reference arithmetic and K3's screening form
void hostsim_metric_pairs(const float *ph, int64_t n, float *exact, float *slope, float *screen) {
	static Tables T; static bool init = false;
	if(!init) { build_tables(T); init = true; }
	for(int64_t i = 0; i < n; i++) {
		sync_metric(ph + 16 * i, T, exact[i], slope[i]);
		float pt[16];
		for(int j = 0; j < 16; j++) pt[j] = ph[16 * i + j] * (float)(0.5 / M_PI);     // the screening tier works in turns
		screen[i] = sync_metric_screen(pt);
	}
}
// the early bound the sync kernel tests after kScreenEarly taps
void hostsim_metric_early(const float *ph, int64_t n, float *early) {
	for(int64_t i = 0; i < n; i++) {
		float pt[16];
		for(int j = 0; j < 16; j++) pt[j] = ph[16 * i + j] * (float)(0.5 / M_PI);
		early[i] = sync_metric_screen(pt, kScreenEarly);
	}
}
int hostsim_screen_early_taps() { return kScreenEarly; }

// more of tables.h: preamble phases (units of pi/4 are checked by the caller), Gray map, FCS table, first PRBS bits, RS field
void hostsim_misc_tables(float *pr_phase16, uint8_t *gray8, uint16_t *crc256, uint8_t *prbs64, uint8_t *gf_exp8) {
	static Tables T; build_tables(T);
	memcpy(pr_phase16, T.pr_phase, sizeof T.pr_phase); memcpy(gray8, T.gray, 8); memcpy(crc256, T.crc16, sizeof T.crc16);
	memcpy(prbs64, T.prbs, 64); memcpy(gf_exp8, T.gf_exp, 8);
}

// the header-code tables the walker uses (tables.h), for comparison with the reference's own
void hostsim_header_tables(uint32_t *H, uint32_t *fix, uint32_t *weight) {
	static Tables T; build_tables(T);
	memcpy(H, T.hdr_H, sizeof T.hdr_H); memcpy(fix, T.hdr_fix, sizeof T.hdr_fix); memcpy(weight, T.hdr_weight, sizeof T.hdr_weight);
}

// finish_frame() on one frame's octets: returns avlc_status, fills dst/src (direct fuzzing of the FCS slicing and the address parse)
int hostsim_finish_frame(const uint8_t *octets, uint32_t len, uint32_t *dst, uint32_t *src, unsigned long long *acnt /* [10] */) {
	static Tables T; static bool init = false; static FrameShared fsh;
	if(!init) { build_tables(T); frame_shared_init(T, fsh); init = true; }
	OutFrame f; memset(&f, 0, sizeof f);
	f.len = len; f.pool_off = 0; f.nf_upd = 0;
	float ring[1] = { 2.0f };
	finish_frame(f, octets, T, acnt, ring, 0, fsh);
	*dst = f.dst_addr; *src = f.src_addr;
	return (int)f.avlc_status;
}

// ---- init-time constants of the channeliser (design.h), for known-answer and consistency tests ----
void hostsim_design_lpf(float fc, float ripple, float *A, float *B) { LpfCoeffs c = design_lpf(fc, ripple); memcpy(A, c.A, 12); memcpy(B, c.B, 12); }
uint32_t hostsim_nco_step(uint32_t centerfreq, uint32_t freq, uint32_t fs) { return nco_step(centerfreq, freq, fs); }
void hostsim_nco_lut(float *out /* [256][4] = s, c, ds, dc */) { Lut4 l[256]; build_nco_lut(l); memcpy(out, l, sizeof l); }
int hostsim_sizeof_blockform() { return (int)sizeof(BlockForm); }
void hostsim_block_form(const float *A, const float *B, int os, int run, BlockForm *out) { LpfCoeffs c; memcpy(c.A, A, 12); memcpy(c.B, B, 12); *out = derive_block_form(c, os, run); }

// the burst decoder's RS stage on one 255-octet row (for direct comparison with libfec / the oracle)
int hostsim_rs_decode(uint8_t *row, int npar) {
	static BurstShared sh; static Tables T; static bool init = false;
	if(!init) { build_tables(T); OutCtl ctl{}; burst_shared_init(T, 0, &ctl, sh); init = true; }
	memcpy(sh.tab, row, 255);
	rs_decode_row(sh.tab, npar, sh);
	memcpy(row, sh.tab, 255);
	return sh.u_ret;
}

}
