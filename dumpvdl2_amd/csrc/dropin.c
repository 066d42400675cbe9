/*
 * dropin.c - host adapter (plain C): the reference's hot-path entry points on top of
 * libvdl2hip.so.  See include/vdl2hip_dropin.h.  What each function replaces is cited inline.
 *
 * Threading contract kept from the reference (src/dumpvdl2.c:117-135, src/demod.c:300-301,
 * 342-346,360-364): main() creates two barriers of count N+1 and N threads running
 * process_samples(); the producer calls process_buf_*() per block and, after the last block,
 * waits once more on demods_ready (src/dumpvdl2.c:1170).  The work is placed where the reference
 * places it: process_buf_*() - between the two barriers, on the producer's thread - hands the
 * block to the GPU (the reference converts it to floats there) and returns; the FIRST channel's
 * process_samples() thread then does what the reference's demodulator threads do in that slot,
 * after samples_ready and before it comes back to demods_ready: it waits for the block's frames
 * and pushes them, while the producer is already reading the next block.  The other N-1 threads
 * only take part in the barriers.  So the device works on block i while main() reads block i+1,
 * every frame of a block has been pushed before process_buf_*() of the next block gets past
 * demods_ready, and main()'s final demods_ready wait returns only when the last block's frames
 * are out - before avlc_decoder_shutdown(), as in the reference.
 *
 * Blocks per feed.  The reference's file reader hands over FILE_BUFSIZE = 320 000 bytes at a time
 * (src/dumpvdl2.h:48, src/dumpvdl2.c:353-356): 4 000 decimated samples per channel, microseconds of
 * GPU work behind a fixed chain of kernel launches - and, one block in three at 256 channels, a 2 ms
 * sequential scan of the referee's.  A producer that comes back for the next block at
 * once (a file, a fast pipe) therefore has its blocks COLLECTED: process_buf_*() copies the block
 * into a staging buffer and returns; every `batch` blocks (as many as make a feed long enough to be
 * walked in segments and pipelined: 16 of the reference's s16 blocks at oversample 20, 4 of its u8
 * blocks at 10; VDL2HIP_DROPIN_BATCH=<n> sets it, 1 = every block on its own as before) the buffer
 * goes to the GPU as one feed and the frames of the feed two before it are pushed (drain lag 2).  The
 * results are the same frames in the same order (any chunking gives the same answer:
 * tests/test_gpu_parity.py), pushed up to three batches late.  The end of the stream - a block
 * shorter than the one before, or an empty one: what ends process_iq_file()'s loop - flushes
 * everything before process_buf_*()'s partner thread returns to demods_ready, so main()'s final
 * wait still means "every frame is out".  A producer that stays away for more than LIVE_GAP_US
 * between two calls (an SDR delivering in real time) gets every block fed and delivered on the
 * spot, as before: collecting would only add latency there.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/time.h>
#ifdef VDL2HIP_IN_TREE
#include "dumpvdl2.h"
#include "output-common.h"
#include "decode.h"
#endif
#include "vdl2hip.h"
#include "vdl2hip_dropin.h"

extern pthread_barrier_t demods_ready, samples_ready;   /* src/dumpvdl2.c:66-67 */

#define MAX_CHANNELS 1024
#define BLOCK_MAX (8u << 20)
#define ARRIVALS 256
#define BATCH_MAX 64u
#define BATCH_DECIMATED 64000u     /* decimated samples per feed the collector aims for (two walk segments are 32 768: vdl2hip.hip, feed_is_small) */
#define LIVE_GAP_US 10000L         /* a producer away for longer than this between two blocks is a live source (an SDR's blocks are tens of milliseconds apart; a file on a slow disk is not to be taken for one) */

float *sbuf;                       /* callers allocate it (src/dumpvdl2.c:342,346; src/rtl.c:194); never read here */

static struct {
	uint32_t centerfreq, source_rate, oversample;
	uint32_t freqs[MAX_CHANNELS];
	uint32_t nchan;
	vdl2hip_group *grp;            /* one member per GPU listed in VDL2HIP_DEVICES (default: device 0) */
	uint64_t overflow_seen;
	int fmt;
	void *first_chan;              /* the channel whose process_samples() thread delivers the frames */
	uint64_t samples_total;        /* complex input samples handed over so far */
	/* when each of the last blocks arrived and which decimated samples (105 kS/s clock) it begins with: burst_timestamp */
	struct { int64_t k_first; struct timeval tv; } arrival[ARRIVALS];
	uint64_t nblocks;
	/* the collector (see the head of this file) */
	unsigned char *stage;          /* blocks waiting to go to the GPU as one feed */
	uint32_t stage_cap, stage_len, stage_blocks;
	uint32_t batch;                /* blocks per feed; 1: every block on its own */
	uint32_t prev_len;             /* length of the block before (a shorter one ends the stream) */
	struct timeval left;           /* when process_buf_*() last returned to the producer */
	int fed;                       /* between the barriers of this block: something went to the GPU (or the stream ended) - there is something to deliver */
	int flush;                     /* ... and every frame still on the GPU is wanted now (stream end, live source) */
	int lag;                       /* the group's current drain lag */
#ifndef VDL2HIP_IN_TREE
	float max_ppm;
	char *station_id;
#endif
} G;

#ifndef VDL2HIP_IN_TREE
void vdl2hip_dropin_configure(float max_ppm, char *station_id) { G.max_ppm = max_ppm; G.station_id = station_id; }
static float cfg_max_ppm(void) { return G.max_ppm; }
static char *cfg_station_id(void) { return G.station_id; }
#else
static float cfg_max_ppm(void) { return Config.max_ppm; }
static char *cfg_station_id(void) { return Config.station_id; }
#endif

/* VDL2HIP_DROPIN_TIMING=1: where the wall clock goes, printed at the stream's end (development aid) */
static struct { double wait_demods, feed, wait_samples, drain, stats, push; unsigned long feeds, drains, frames; int on; } T;
static double now_ms(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec * 1e3 + t.tv_usec * 1e-3; }

static void *must_calloc(size_t n, size_t sz) {   /* XCALLOC semantics, src/util.c:32-40 */
	void *p = calloc(n ? n : 1, sz);
	if(!p) { fprintf(stderr, "vdl2hip dropin: calloc(%zu, %zu) failed\n", n, sz); _exit(1); }
	return p;
}

vdl2_channel_t *vdl2_channel_init(uint32_t centerfreq, uint32_t freq, uint32_t source_rate, uint32_t oversample) {
	if(G.nchan >= MAX_CHANNELS) return NULL;
	G.centerfreq = centerfreq; G.source_rate = source_rate; G.oversample = oversample;
	G.freqs[G.nchan++] = freq;
#ifdef VDL2HIP_IN_TREE
	vdl2_channel_t *v = must_calloc(1, sizeof(vdl2_channel_t));
	v->freq = freq; v->oversample = oversample;
#else
	vdl2_channel_t *v = (vdl2_channel_t *)must_calloc(1, 4096);
#endif
	if(!G.first_chan) G.first_chan = v;
	return v;
}

/* the tables and coefficients these set up are derived inside vdl2hip_create() */
void sincosf_lut_init(void) {}
void demod_sync_init(void) {}
void input_lpf_init(uint32_t sample_rate) { G.source_rate = sample_rate; }
void process_buf_uchar_init(void) {}
int rs_init(void) { return 0; }

static void deliver_block(void);

void *process_samples(void *arg) {
	const int delivers = arg == G.first_chan;
	for(;;) {                                  /* same two waits as src/demod.c:300-301 */
		pthread_barrier_wait(&demods_ready);
		pthread_barrier_wait(&samples_ready);
		if(delivers) deliver_block();          /* the slot in which the reference's threads demodulate the block */
	}
	return NULL;
}

static struct timeval arrival_of(int64_t sync_sample) {
	uint64_t n = G.nblocks < ARRIVALS ? G.nblocks : ARRIVALS;
	for(uint64_t i = 0; i < n; i++) {          /* newest first */
		uint64_t j = (G.nblocks - 1 - i) % ARRIVALS;
		if(G.arrival[j].k_first <= sync_sample || i + 1 == n) return G.arrival[j].tv;
	}
	struct timeval now; gettimeofday(&now, NULL); return now;
}

static void push_frame(const vdl2hip_frame *f, void *user) {
	(void)user;
	const double tp0 = T.on ? now_ms() : 0;
	/* decode_frame(), src/decode.c:173-194 */
	vdl2_msg_metadata *m = must_calloc(1, sizeof *m);
	m->version = 1;
	m->station_id = cfg_station_id();
	m->freq = f->freq;
	m->frame_pwr_dbfs = f->frame_pwr_dbfs;
	m->nf_pwr_dbfs = f->nf_pwr_dbfs;
	m->ppm_error = f->ppm_error;
	/* the reference stamps the wall clock when it finds the sync (src/demod.c:246), i.e. while it works through the block the
	 * burst's preamble lies in, right after that block arrived: the arrival time of that block */
	m->burst_timestamp = arrival_of(f->sync_sample);
	m->datalen_octets = f->datalen_octets;
	m->synd_weight = f->synd_weight;
	m->num_fec_corrections = f->num_fec_corrections;
	m->idx = f->idx;
	uint8_t *copy = must_calloc(f->len, 1);
	memcpy(copy, f->octets, f->len);
	octet_string_t *os = must_calloc(1, sizeof *os);   /* octet_string_new(), src/util.c:145-150 */
	os->buf = copy; os->len = f->len;
	avlc_decoder_queue_push(m, os, 0);
	if(T.on) T.push += now_ms() - tp0;
}

/* VDL2HIP_DEVICES=0,1,2,...: the GPUs the channels are spread over (contiguous ranges, src/dumpvdl2.c:117-135 has one worker
 * per channel; here the workers are grouped by device).  A device may be listed more than once. */
static uint32_t parse_devices(int32_t *dev, uint32_t cap) {
	const char *e = getenv("VDL2HIP_DEVICES");
	uint32_t n = 0;
	if(e) {
		char *end;
		while(*e && n < cap) {
			long v = strtol(e, &end, 10);
			if(end == e) break;
			dev[n++] = (int32_t)v;
			e = (*end == ',') ? end + 1 : end;
		}
	}
	if(n == 0) dev[n++] = 0;
	return n;
}

static void gpu_feed(const unsigned char *p, uint32_t len) {
	const double t0 = T.on ? now_ms() : 0;
	for(uint32_t off = 0; off < len; off += BLOCK_MAX) {
		uint32_t n = len - off < BLOCK_MAX ? len - off : BLOCK_MAX;
		int r = vdl2hip_group_feed(G.grp, p + off, n);       /* returns when the copy out of p is complete: buf is only ours during the call */
		if(r != VDL2HIP_OK) { fprintf(stderr, "vdl2hip_group_feed: %s\n", vdl2hip_strerror(r)); _exit(2); }
	}
	if(T.on) { T.feed += now_ms() - t0; T.feeds++; }
}

/* blocks per feed: VDL2HIP_DROPIN_BATCH, or as many of this size as hold BATCH_DECIMATED decimated samples */
static uint32_t pick_batch(uint32_t len, int fmt) {
	const char *e = getenv("VDL2HIP_DROPIN_BATCH");
	uint32_t b;
	if(e && atoi(e) >= 1) b = (uint32_t)atoi(e);
	else {
		uint32_t dec = len / (fmt == VDL2HIP_FMT_S16LE ? 4u : 2u) / (G.oversample ? G.oversample : 1u);
		b = dec ? (BATCH_DECIMATED + dec - 1) / dec : 1;
	}
	if(b > BATCH_MAX) b = BATCH_MAX;
	if(len && b > BLOCK_MAX / len) b = BLOCK_MAX / len;
	return b ? b : 1;
}

static void stage_to_gpu(void) {
	if(G.stage_len) gpu_feed(G.stage, G.stage_len);
	G.stage_len = 0; G.stage_blocks = 0;
}

/* one block from the producer (between the two barriers, on the producer's thread) */
static void take_block(unsigned char *buf, uint32_t len, int fmt, struct timeval now) {
	if(!G.grp) {
		vdl2hip_cfg cfg;
		memset(&cfg, 0, sizeof cfg);
		cfg.struct_size = sizeof cfg;
		cfg.centerfreq = G.centerfreq;
		cfg.oversample = G.oversample;
		cfg.sample_fmt = (uint32_t)fmt;
		cfg.nchan = G.nchan;
		cfg.freqs = G.freqs;
		cfg.max_ppm = cfg_max_ppm();
		cfg.max_block_bytes = BLOCK_MAX;
		int32_t dev[64];
		uint32_t ndev = parse_devices(dev, 64);
		if(ndev > G.nchan) ndev = G.nchan;
		int r = vdl2hip_group_create(&cfg, dev, ndev, &G.grp);
		if(r != VDL2HIP_OK) { fprintf(stderr, "vdl2hip_group_create: %s\n", vdl2hip_strerror(r)); _exit(2); }
		G.fmt = fmt;
		G.batch = pick_batch(len, fmt);
		if(G.batch > 1) { G.stage_cap = G.batch * len; G.stage = must_calloc(G.stage_cap, 1); }
	}
	/* what the producer did since it last left: a file reader is back within microseconds, an SDR in real time */
	const long away_us = G.nblocks ? (now.tv_sec - G.left.tv_sec) * 1000000L + (now.tv_usec - G.left.tv_usec) : 0;
	/* (the very first block goes at once as well: a stream that is one short block long ends without a shorter one) */
	const int live = away_us > LIVE_GAP_US || G.nblocks == 0;
	const int end = len == 0 || len < G.prev_len;           /* what ends process_iq_file()'s loop, src/dumpvdl2.c:353-356 */
	G.prev_len = len;
	if(len) {
		uint64_t slot = G.nblocks % ARRIVALS;
		G.arrival[slot].k_first = (int64_t)(G.samples_total / G.oversample);
		G.arrival[slot].tv = now;
		G.nblocks++;
		G.samples_total += len / (fmt == VDL2HIP_FMT_S16LE ? 4u : 2u);
	}
	if(G.batch <= 1) {                                       /* every block on its own, every frame of it out before the next */
		if(len) { gpu_feed(buf, len); G.fed = 1; G.flush = 1; }
		return;
	}
	if(len && G.stage_len + len > G.stage_cap) {             /* (a block larger than the first one was) */
		stage_to_gpu(); gpu_feed(buf, len); G.fed = 1;
	} else if(len) {
		memcpy(G.stage + G.stage_len, buf, len);
		G.stage_len += len; G.stage_blocks++;
	}
	if(G.stage_blocks >= G.batch || end || live) {
		stage_to_gpu();
		G.fed = 1;
	}
	if(end || live) G.flush = 1;
}

/* wait for the frames of the block(s) handed over and push them (first channel's thread, between samples_ready and demods_ready) */
static void deliver_block(void) {
	if(!G.grp || !G.fed) return;                           /* (a collected block: nothing new on the GPU) */
	/* a feed of collected blocks leaves the two feeds before it on the GPU; the stream's end and a live source want everything */
	const int lag = G.flush ? 0 : 2;
	G.fed = 0; G.flush = 0;
	if(lag != G.lag) { vdl2hip_group_set_drain_lag(G.grp, lag); G.lag = lag; }
	const double t0 = T.on ? now_ms() : 0;
	int r = vdl2hip_group_drain(G.grp, push_frame, NULL);
	if(r < 0) { fprintf(stderr, "vdl2hip_group_drain: %s\n", vdl2hip_strerror(r)); _exit(2); }
	const double t1 = T.on ? now_ms() : 0;
	/* the drain calls only count device-side buffer overflows (bursts or frames dropped): say so once per occurrence.
	 * (vdl2hip_get_stats() collects every feed in flight first - it waits for the device: only where that is wanted anyway) */
	if(lag == 0) {
		uint64_t ov = 0;
		for(uint32_t i = 0; i < vdl2hip_group_size(G.grp); i++) {
			vdl2hip_stats st;
			if(vdl2hip_get_stats(vdl2hip_group_ctx(G.grp, i), &st) == VDL2HIP_OK) ov += st.overflow_feeds;
		}
		if(ov != G.overflow_seen) { fprintf(stderr, "vdl2hip: device output buffers overflowed in %llu block(s): frames were dropped\n", (unsigned long long)(ov - G.overflow_seen)); G.overflow_seen = ov; }
	}
	if(T.on) {
		T.drain += t1 - t0; T.stats += now_ms() - t1; T.drains++; T.frames += (unsigned long)r;
		if(lag == 0) fprintf(stderr, "vdl2hip dropin timing: %llu blocks, %lu feeds, %lu drains, %lu frames; producer: %.1f ms waiting at demods_ready, %.1f in feeds, %.1f waiting at samples_ready; "
				"delivering thread: %.1f ms in drains (%.1f of it pushing frames), %.1f reading stats\n", (unsigned long long)G.nblocks, T.feeds, T.drains, T.frames, T.wait_demods, T.feed, T.wait_samples, T.drain, T.push, T.stats);
	}
}

static void process_buf(unsigned char *buf, uint32_t len, int fmt) {
	/* an empty block: the reference's threads find nothing to do; here it may be the end of a stream some of whose blocks are
	 * still collected or whose last feed's frames are still on the GPU */
	if(len == 0 && (!G.grp || G.batch <= 1)) return;
	struct timeval now;
	gettimeofday(&now, NULL);                              /* (before the wait: that is the delivering thread's time, not the producer's) */
	if(!G.grp) T.on = getenv("VDL2HIP_DROPIN_TIMING") != NULL;
	const double t0 = T.on ? now_ms() : 0;
	pthread_barrier_wait(&demods_ready);
	const double t1 = T.on ? now_ms() : 0;
	take_block(buf, len, fmt, now);
	const double t2 = T.on ? now_ms() : 0;
	pthread_barrier_wait(&samples_ready);
	gettimeofday(&G.left, NULL);
	if(T.on && G.nblocks > 1) { T.wait_demods += t1 - t0; T.wait_samples += now_ms() - t2; }
}

void process_buf_uchar(unsigned char *buf, uint32_t len, void *ctx) {   /* src/demod.c:339-347 */
	(void)ctx;
	process_buf(buf, len, VDL2HIP_FMT_U8);
}

void process_buf_short(unsigned char *buf, uint32_t len, void *ctx) {   /* src/demod.c:356-365 */
	(void)ctx;
	process_buf(buf, len, VDL2HIP_FMT_S16LE);
}
