/*
 * vdl2hip_iqfile - minimal `dumpvdl2 --iq-file` work-alike on top of libvdl2hip.so (plain C host code).
 *
 * It reproduces the part of the reference's command line that feeds the hot path
 * (src/dumpvdl2.c:831-1099 option handling, :168-180 centre-frequency rule, :323-358 file loop):
 *   --iq-file <path|->            raw IQ file, read in FILE_BUFSIZE (320000-byte) blocks; sets oversample 10 and U8
 *   --blocks-per-feed <n>         how many of those blocks go to the GPU as one feed (default: as many as hold 64 000 decimated
 *                                 samples - 16 s16 blocks at oversample 20, 4 u8 blocks at 10; 1 = the reference's block by block).
 *                                 A block is microseconds of GPU work behind a fixed chain of launches (and, now and then, a 2 ms
 *                                 scan of the referee's): collected blocks give the same frames at 5-7x the rate
 *   --sample-format U8|S16_LE     (the reference's token is S16_LE, src/dumpvdl2.c:849)
 *   --oversample <n>  --centerfreq <Hz>  --max-ppm <x>  --station-id <s>
 *   --avlc-filter                 deliver only frames that pass avlc_parse()'s first checks (length, FCS) - src/avlc.c:168-187
 *   --statsd-out <path>           at exit, write the per-channel counters in the reference's statsd names
 *                                 ("dumpvdl2[.<station-id>].<freq>.<counter>:<n>|c", src/statsd.c:34-65,153-160)
 *   --raw-frames-out <path>       write every frame in the reference's raw-frame archive format, so that a stock
 *                                 `dumpvdl2 --raw-frames-file <path>` decodes them through the full protocol stack
 *   freq [freq ...]               channel frequencies in Hz; default: the CSC, 136975000
 * and prints one line per AVLC frame (metadata in the reference's "[S:…] [L:…] [F:…] [#idx]" style + hex octets).
 * Everything after avlc_decoder_queue_push() (AVLC/ACARS/X.25/... decoding, formatters) is out of scope here.
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "vdl2hip.h"

#define FILE_BUFSIZE 320000U        /* src/dumpvdl2.h:48 */
#define FILE_OVERSAMPLE 10          /* src/dumpvdl2.h:49 */
#define CSC_FREQ 136975000U         /* src/dumpvdl2.h:47 */
#define SYMBOL_RATE 10500

static FILE *raw_out;
static const char *station_id;
static unsigned long nframes;

static void on_frame(const vdl2hip_frame *f, void *user) {
	(void)user;
	nframes++;
	printf("%u Hz [%.1f/%.1f dBFS] [%.1f dB] [%.1f ppm] [S:%u] [L:%u] [F:%d] [#%d] len=%u ",
			f->freq, f->frame_pwr_dbfs, f->nf_pwr_dbfs, f->frame_pwr_dbfs - f->nf_pwr_dbfs, f->ppm_error,
			f->synd_weight, f->datalen_octets, f->num_fec_corrections, f->idx, f->len);
	for(uint32_t i = 0; i < f->len; i++) printf("%02x", f->octets[i]);
	printf("\n");
	if(raw_out) {
		static uint8_t rec[70000];
		struct timeval tv; gettimeofday(&tv, NULL);
		int n = vdl2hip_pack_raw_frame(f, station_id, tv.tv_sec, tv.tv_usec, rec, sizeof rec);
		if(n > 0) fwrite(rec, 1, (size_t)n, raw_out);
		else fprintf(stderr, "frame not archived: %s\n", vdl2hip_strerror(n));
	}
}

int main(int argc, char **argv) {
	const char *infile = NULL, *rawpath = NULL, *statsd_path = NULL;
	int avlc_filter = 0;
	uint32_t per_feed = 0;
	uint32_t oversample = 0, centerfreq = 0, fmt = VDL2HIP_FMT_U8, freqs[1024], nfreq = 0;
	float max_ppm = 0.f;
	int fmt_set = 0;
	for(int i = 1; i < argc; i++) {
		const char *a = argv[i];
		#define NEEDARG() do { if(i + 1 >= argc) { fprintf(stderr, "%s needs an argument\n", a); return 1; } } while(0)
		if(!strcmp(a, "--iq-file")) { NEEDARG(); infile = argv[++i]; if(!oversample) oversample = FILE_OVERSAMPLE; }
		else if(!strcmp(a, "--sample-format")) {
			NEEDARG(); i++; fmt_set = 1;
			if(!strcmp(argv[i], "U8")) fmt = VDL2HIP_FMT_U8;
			else if(!strcmp(argv[i], "S16_LE")) fmt = VDL2HIP_FMT_S16LE;
			else { fprintf(stderr, "Unknown sample format\n"); return 1; }
		}
		else if(!strcmp(a, "--oversample")) { NEEDARG(); oversample = (uint32_t)strtoul(argv[++i], NULL, 10); }
		else if(!strcmp(a, "--centerfreq")) { NEEDARG(); centerfreq = (uint32_t)strtoul(argv[++i], NULL, 10); }
		else if(!strcmp(a, "--max-ppm")) { NEEDARG(); max_ppm = strtof(argv[++i], NULL); }
		else if(!strcmp(a, "--station-id")) { NEEDARG(); station_id = argv[++i]; }
		else if(!strcmp(a, "--raw-frames-out")) { NEEDARG(); rawpath = argv[++i]; }
		else if(!strcmp(a, "--statsd-out")) { NEEDARG(); statsd_path = argv[++i]; }
		else if(!strcmp(a, "--avlc-filter")) avlc_filter = 1;
		else if(!strcmp(a, "--blocks-per-feed")) { NEEDARG(); per_feed = (uint32_t)strtoul(argv[++i], NULL, 10); }
		else if(a[0] == '-' && a[1]) { fprintf(stderr, "unknown option %s\n", a); return 1; }
		else if(nfreq < 1024) freqs[nfreq++] = (uint32_t)strtoul(a, NULL, 10);
	}
	(void)fmt_set;
	if(!infile) { fprintf(stderr, "usage: %s --iq-file <file|-> [--sample-format U8|S16_LE] [--oversample n] [--centerfreq Hz] "
			"[--max-ppm x] [--station-id s] [--raw-frames-out file] [--avlc-filter] [--statsd-out file] [--blocks-per-feed n] [freq ...]\n", argv[0]); return 1; }
	if(nfreq == 0) {
		fprintf(stderr, "Warning: frequency not set - using VDL2 Common Signalling Channel as a default (%u Hz)\n", CSC_FREQ);
		freqs[nfreq++] = CSC_FREQ;
	}
	const uint32_t sample_rate = SYMBOL_RATE * 10u * oversample;             /* src/dumpvdl2.c:1073 */
	fprintf(stderr, "Sampling rate set to %u sps\n", sample_rate);
	if(centerfreq == 0) {                                                      /* calc_centerfreq(), src/dumpvdl2.c:168-180 */
		uint32_t lo = freqs[0], hi = freqs[0];
		for(uint32_t i = 0; i < nfreq; i++) { if(freqs[i] < lo) lo = freqs[i]; if(freqs[i] > hi) hi = freqs[i]; }
		if(hi - lo > sample_rate - SYMBOL_RATE * 4) { fprintf(stderr, "Error: given frequencies are too far apart\n"); return 2; }
		centerfreq = lo + (hi - lo) / 2;
	}
	FILE *f = !strcmp(infile, "-") ? stdin : fopen(infile, "r");
	if(!f) { perror("Could not open input file"); return 2; }
	if(rawpath && !(raw_out = fopen(rawpath, "w"))) { perror("Could not open raw frames output"); return 2; }

	vdl2hip_cfg cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.struct_size = sizeof cfg; cfg.centerfreq = centerfreq; cfg.oversample = oversample; cfg.sample_fmt = fmt;
	if(per_feed == 0) {
		const uint32_t dec = FILE_BUFSIZE / (fmt == VDL2HIP_FMT_S16LE ? 4u : 2u) / oversample;      /* decimated samples per block */
		per_feed = dec ? (64000u + dec - 1) / dec : 1;
	}
	if(per_feed > 64) per_feed = 64;
	cfg.nchan = nfreq; cfg.freqs = freqs; cfg.max_ppm = max_ppm; cfg.device = 0; cfg.max_block_bytes = (size_t)per_feed * FILE_BUFSIZE;
	vdl2hip_ctx *rx = NULL;
	int r = vdl2hip_create(&cfg, &rx);
	if(r != VDL2HIP_OK) { fprintf(stderr, "vdl2hip_create: %s\n", vdl2hip_strerror(r)); return 3; }

	if(avlc_filter) vdl2hip_set_avlc_filter(rx, 1);

	unsigned char *buf = malloc((size_t)per_feed * FILE_BUFSIZE);
	if(!buf) { perror("malloc"); return 3; }
	size_t len, held = 0;
	if(per_feed > 1) vdl2hip_set_drain_lag(rx, 2);                              /* the frames of a feed are printed while the next two are on the GPU */
	do {                                                                        /* process_iq_file(), src/dumpvdl2.c:353-356 */
		len = fread(buf + held, 1, FILE_BUFSIZE, f);
		held += len;
		if(held + FILE_BUFSIZE <= (size_t)per_feed * FILE_BUFSIZE && len == FILE_BUFSIZE) continue;      /* room for another block, and there may be one */
		if(len != FILE_BUFSIZE) vdl2hip_set_drain_lag(rx, 0);                   /* the last feed: every frame out */
		if((r = vdl2hip_feed(rx, buf, held)) != VDL2HIP_OK) { fprintf(stderr, "vdl2hip_feed: %s\n", vdl2hip_strerror(r)); return 3; }
		if((r = vdl2hip_drain(rx, on_frame, NULL)) < 0) { fprintf(stderr, "vdl2hip_drain: %s\n", vdl2hip_strerror(r)); return 3; }
		held = 0;
	} while(len == FILE_BUFSIZE);
	free(buf);
	uint64_t cnt[VDL2HIP_NUM_COUNTERS];
	for(uint32_t c = 0; c < nfreq; c++)
		if(vdl2hip_counters(rx, c, cnt) == VDL2HIP_OK)
			fprintf(stderr, "%u Hz: sync.good=%" PRIu64 " crc.good=%" PRIu64 " blocks=%" PRIu64 "/%" PRIu64 " msg.good=%" PRIu64 " fec_bad=%" PRIu64 "\n",
					freqs[c], cnt[VDL2HIP_CNT_SYNC_GOOD], cnt[VDL2HIP_CNT_CRC_GOOD], cnt[VDL2HIP_CNT_BLOCKS_FEC_OK],
					cnt[VDL2HIP_CNT_BLOCKS_PROCESSED], cnt[VDL2HIP_CNT_MSG_GOOD], cnt[VDL2HIP_CNT_ERR_FEC_BAD]);
	fprintf(stderr, "%lu frames\n", nframes);
	{
		vdl2hip_stats st;                                                       /* the drain calls only count buffer overflows: say so */
		if(vdl2hip_get_stats(rx, &st) == VDL2HIP_OK && st.overflow_feeds)
			fprintf(stderr, "warning: device output buffers overflowed in %llu block(s): frames were dropped\n", (unsigned long long)st.overflow_feeds);
	}
	if(statsd_path) {
		static char lines[1 << 20];
		char ns[300];
		if(station_id) snprintf(ns, sizeof ns, "dumpvdl2.%s", station_id); else snprintf(ns, sizeof ns, "dumpvdl2");    /* statsd.c:103-108 */
		int n = vdl2hip_statsd_lines(rx, ns, lines, sizeof lines);
		FILE *so = n >= 0 ? fopen(statsd_path, "w") : NULL;
		if(so) { fwrite(lines, 1, (size_t)n, so); fclose(so); }
		else fprintf(stderr, "statsd counters not written: %s\n", n < 0 ? vdl2hip_strerror(n) : "cannot open file");
	}
	vdl2hip_destroy(rx);
	if(raw_out) fclose(raw_out);
	if(f != stdin) fclose(f);
	return 0;
}
