/*
 * dropin.c - host adapter (plain C): the reference's hot-path entry points on top of
 * libvdl2hip.so.  See include/vdl2hip_dropin.h.  What each function replaces is cited inline.
 *
 * Threading contract kept from the reference (src/dumpvdl2.c:117-135, src/demod.c:300-301,
 * 342-346,360-364): main() creates two barriers of count N+1 and N threads running
 * process_samples(); the producer calls process_buf_*() per block and, after the last block,
 * waits once more on demods_ready (src/dumpvdl2.c:1170).  Here the N threads are parked on the
 * barriers and the producer thread does the work between them: feed the block to the GPU, drain
 * the frames of that block and push them.  Everything has been pushed when process_buf_*()
 * returns, hence also before main()'s final barrier wait and avlc_decoder_shutdown().
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#ifdef VDL2HIP_IN_TREE
#include "dumpvdl2.h"
#include "output-common.h"
#include "decode.h"
#endif
#include "vdl2hip.h"
#include "vdl2hip_dropin.h"

extern pthread_barrier_t demods_ready, samples_ready;   /* src/dumpvdl2.c:66-67 */

#define MAX_CHANNELS 1024
#define BLOCK_MAX (4u << 20)

float *sbuf;                       /* callers allocate it (src/dumpvdl2.c:342,346; src/rtl.c:194); never read here */

static struct {
	uint32_t centerfreq, source_rate, oversample;
	uint32_t freqs[MAX_CHANNELS];
	uint32_t nchan;
	vdl2hip_group *grp;            /* one member per GPU listed in VDL2HIP_DEVICES (default: device 0) */
	uint64_t overflow_seen;
	int fmt;
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
	return v;
#else
	return (vdl2_channel_t *)must_calloc(1, 4096);
#endif
}

/* the tables and coefficients these set up are derived inside vdl2hip_create() */
void sincosf_lut_init(void) {}
void demod_sync_init(void) {}
void input_lpf_init(uint32_t sample_rate) { G.source_rate = sample_rate; }
void process_buf_uchar_init(void) {}
int rs_init(void) { return 0; }

void *process_samples(void *arg) {
	(void)arg;
	for(;;) {                                  /* same two waits as src/demod.c:300-301 */
		pthread_barrier_wait(&demods_ready);
		pthread_barrier_wait(&samples_ready);
	}
	return NULL;
}

static void push_frame(const vdl2hip_frame *f, void *user) {
	(void)user;
	/* decode_frame(), src/decode.c:173-194 */
	vdl2_msg_metadata *m = must_calloc(1, sizeof *m);
	m->version = 1;
	m->station_id = cfg_station_id();
	m->freq = f->freq;
	m->frame_pwr_dbfs = f->frame_pwr_dbfs;
	m->nf_pwr_dbfs = f->nf_pwr_dbfs;
	m->ppm_error = f->ppm_error;
	gettimeofday(&m->burst_timestamp, NULL);    /* the reference stamps wall-clock at sync (src/demod.c:246) */
	m->datalen_octets = f->datalen_octets;
	m->synd_weight = f->synd_weight;
	m->num_fec_corrections = f->num_fec_corrections;
	m->idx = f->idx;
	uint8_t *copy = must_calloc(f->len, 1);
	memcpy(copy, f->octets, f->len);
	octet_string_t *os = must_calloc(1, sizeof *os);   /* octet_string_new(), src/util.c:145-150 */
	os->buf = copy; os->len = f->len;
	avlc_decoder_queue_push(m, os, 0);
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

static void feed_block(unsigned char *buf, uint32_t len, int fmt) {
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
	}
	for(uint32_t off = 0; off < len; off += BLOCK_MAX) {
		uint32_t n = len - off < BLOCK_MAX ? len - off : BLOCK_MAX;
		int r = vdl2hip_group_feed(G.grp, buf + off, n);
		if(r != VDL2HIP_OK) { fprintf(stderr, "vdl2hip_group_feed: %s\n", vdl2hip_strerror(r)); _exit(2); }
	}
	int r = vdl2hip_group_drain(G.grp, push_frame, NULL);
	if(r < 0) { fprintf(stderr, "vdl2hip_group_drain: %s\n", vdl2hip_strerror(r)); _exit(2); }
	/* the drain calls only count device-side buffer overflows (bursts or frames dropped): say so once per occurrence */
	uint64_t ov = 0;
	for(uint32_t i = 0; i < vdl2hip_group_size(G.grp); i++) {
		vdl2hip_stats st;
		if(vdl2hip_get_stats(vdl2hip_group_ctx(G.grp, i), &st) == VDL2HIP_OK) ov += st.overflow_feeds;
	}
	if(ov != G.overflow_seen) { fprintf(stderr, "vdl2hip: device output buffers overflowed in %llu block(s): frames were dropped\n", (unsigned long long)(ov - G.overflow_seen)); G.overflow_seen = ov; }
}

void process_buf_uchar(unsigned char *buf, uint32_t len, void *ctx) {   /* src/demod.c:339-347 */
	(void)ctx;
	if(len == 0) return;
	pthread_barrier_wait(&demods_ready);
	feed_block(buf, len, VDL2HIP_FMT_U8);
	pthread_barrier_wait(&samples_ready);
}

void process_buf_short(unsigned char *buf, uint32_t len, void *ctx) {   /* src/demod.c:356-365 */
	(void)ctx;
	if(len == 0) return;
	pthread_barrier_wait(&demods_ready);
	feed_block(buf, len, VDL2HIP_FMT_S16LE);
	pthread_barrier_wait(&samples_ready);
}
