/*
 * dropin_harness.c - TEST ONLY.  Plays the part of the unmodified dumpvdl2 main() around the
 * drop-in adapter: the call sequence of src/dumpvdl2.c:1086-1099 (channel init, rs_init),
 * :1148-1153 (lut/lpf/sync init, barriers of count N+1, one process_samples thread per channel),
 * process_iq_file() :323-358 (fread FILE_BUFSIZE blocks -> process_buf_short) and the final
 * barrier wait :1170; and of the consumer behind avlc_decoder_queue_push() (src/decode.c:165-171,
 * 523-525: it free()s metadata, frame->buf and frame).  Prints one line per frame.
 *   usage: dropin_harness <iq-file> <oversample> <centerfreq> <freq> [freq...]
 *   environment: HARNESS_U8=1 (the file holds unsigned bytes: process_buf_uchar), HARNESS_MAX_PPM=<x> (--max-ppm), HARNESS_TIMING=1
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "vdl2hip_dropin.h"

#define FILE_BUFSIZE 320000U     /* src/dumpvdl2.h:48 */

pthread_barrier_t demods_ready, samples_ready;   /* src/dumpvdl2.c:66-67 */
static pthread_mutex_t out_lock = PTHREAD_MUTEX_INITIALIZER;

void avlc_decoder_queue_push(vdl2_msg_metadata *m, octet_string_t *frame, int flags) {
	pthread_mutex_lock(&out_lock);
	printf("FRAME freq=%u idx=%d len=%zu S=%u L=%u F=%d pwr=%.3f nf=%.3f ppm=%.3f flags=%d station=%s octets=",
			m->freq, m->idx, frame->len, m->synd_weight, m->datalen_octets, m->num_fec_corrections,
			m->frame_pwr_dbfs, m->nf_pwr_dbfs, m->ppm_error, flags, m->station_id ? m->station_id : "-");
	for(size_t i = 0; i < frame->len; i++) printf("%02x", frame->buf[i]);
	printf("\n");
	pthread_mutex_unlock(&out_lock);
	free(frame->buf); free(frame); free(m);
}

int main(int argc, char **argv) {
	if(argc < 5) { fprintf(stderr, "usage: %s file oversample centerfreq freq...\n", argv[0]); return 2; }
	uint32_t oversample = (uint32_t)atoi(argv[2]), centerfreq = (uint32_t)strtoul(argv[3], NULL, 10);
	int nchan = argc - 4;
	uint32_t sample_rate = 10500u * 10u * oversample;            /* src/dumpvdl2.c:1073 */
	vdl2_channel_t **ch = calloc((size_t)nchan, sizeof *ch);
	vdl2hip_dropin_configure(getenv("HARNESS_MAX_PPM") ? strtof(getenv("HARNESS_MAX_PPM"), NULL) : 0.f, "HARNESS");   /* Config.max_ppm (--max-ppm), Config.station_id */
	for(int i = 0; i < nchan; i++)
		if((ch[i] = vdl2_channel_init(centerfreq, (uint32_t)strtoul(argv[4 + i], NULL, 10), sample_rate, oversample)) == NULL) return 2;
	if(rs_init() < 0) return 3;
	sincosf_lut_init();
	input_lpf_init(sample_rate);
	demod_sync_init();
	pthread_barrier_init(&demods_ready, NULL, (unsigned)nchan + 1);
	pthread_barrier_init(&samples_ready, NULL, (unsigned)nchan + 1);
	pthread_t *th = calloc((size_t)nchan, sizeof *th);
	for(int i = 0; i < nchan; i++) pthread_create(&th[i], NULL, process_samples, ch[i]);

	FILE *f = fopen(argv[1], "r");
	if(!f) { perror("open"); return 2; }
	static unsigned char buf[FILE_BUFSIZE];
	sbuf = calloc(FILE_BUFSIZE / sizeof(int16_t), sizeof(float));
	uint32_t len, nblk = 0;
	struct timeval t0, t1, t2;
	gettimeofday(&t0, NULL); t1 = t0;
	do {
		len = (uint32_t)fread(buf, 1, FILE_BUFSIZE, f);
		if(getenv("HARNESS_U8")) process_buf_uchar(buf, len, NULL);     /* --sample-format U8, the --iq-file default (src/dumpvdl2.c:849) */
		else process_buf_short(buf, len, NULL);
		if(nblk++ == 0) gettimeofday(&t1, NULL);                  /* (the first call creates the receiver: hundreds of milliseconds) */
	} while(len == FILE_BUFSIZE);
	fclose(f);
	pthread_barrier_wait(&demods_ready);                        /* src/dumpvdl2.c:1170 */
	gettimeofday(&t2, NULL);
	if(getenv("HARNESS_TIMING") && nblk > 1)
		fprintf(stderr, "HARNESS %u blocks of %u bytes: %.3f ms per block after the first (which took %.1f ms)\n", nblk, FILE_BUFSIZE,
				((t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_usec - t1.tv_usec) * 1e-3) / (nblk - 1), (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) * 1e-3);
	fflush(stdout);
	return 0;
}
