/*
 * vdl2_oracle.h - TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, strict IEEE, sequential scan per channel) of the
 * dumpvdl2 per-channel DSP + burst decoder hot path.  It exists to check the
 * HIP product path; nothing in the product may include, link or call it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pin: see the header comment of vdl2_oracle.c.
 */
#ifndef VDL2_ORACLE_H
#define VDL2_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VDL2O_FMT_U8 = 0, VDL2O_FMT_S16LE = 1 };

/* per-channel event counters; names follow the reference's statsd counters
 * (statsd.c:34-65, call sites demod.c:245, decode.c:204-373) */
enum {
	VDL2O_CNT_SYNC_GOOD = 0,       /* demod.sync.good */
	VDL2O_CNT_CRC_GOOD,            /* decoder.crc.good (header syndrome == 0) */
	VDL2O_CNT_CRC_BAD,             /* decoder.crc.bad (reserved bits set after correction) */
	VDL2O_CNT_ERR_NO_HEADER,
	VDL2O_CNT_ERR_TOO_LONG,
	VDL2O_CNT_ERR_NO_FEC,
	VDL2O_CNT_ERR_DATA_TRUNCATED,
	VDL2O_CNT_ERR_FEC_TRUNCATED,
	VDL2O_CNT_ERR_DEINTERLEAVE_DATA,
	VDL2O_CNT_ERR_DEINTERLEAVE_FEC,
	VDL2O_CNT_ERR_FEC_BAD,
	VDL2O_CNT_ERR_BITSTREAM,
	VDL2O_CNT_ERR_TRUNCATED_OCTETS,
	VDL2O_CNT_ERR_UNSTUFF,
	VDL2O_CNT_BLOCKS_PROCESSED,
	VDL2O_CNT_BLOCKS_FEC_OK,
	VDL2O_CNT_MSG_GOOD,
	VDL2O_CNT_MSG_GOOD_LOUD,
	VDL2O_CNT_PPM_REJECT,          /* preamble dropped by --max-ppm (demod.c:192); no statsd name */
	VDL2O_CNT_SLICER_NEG_IDX,      /* slicer produced idx < 0 (out-of-bounds table read in the reference) */
	VDL2O_NUM_COUNTERS
};

/* One AVLC frame as handed to avlc_decoder_queue_push() (decode.c:165-194) */
typedef struct {
	int32_t  chan;                 /* channel index */
	uint32_t freq;                 /* metadata->freq */
	int32_t  idx;                  /* frame number within the burst */
	uint32_t len;                  /* frame length, octets */
	uint64_t octets_off;           /* offset into the ctx octet pool */
	uint32_t synd_weight;
	uint32_t datalen_octets;
	int32_t  num_fec_corrections;
	float    frame_pwr_dbfs;
	float    nf_pwr_dbfs;
	float    ppm_error;
	int64_t  burst_ord;            /* ordinal of the burst (successful sync) on this channel */
	int64_t  sync_sample;          /* decimated sample index at which got_sync() fired */
	int64_t  end_sample;           /* decimated sample index at which the burst was decoded */
} vdl2o_frame;

typedef struct vdl2o_ctx vdl2o_ctx;

vdl2o_ctx *vdl2o_create(uint32_t centerfreq, const uint32_t *freqs, int nchan,
		uint32_t oversample, int sample_fmt, float max_ppm);
void vdl2o_destroy(vdl2o_ctx *c);

/* Equivalent of one process_buf_uchar()/process_buf_short() call followed by
 * every channel's process_samples() pass over that block (demod.c:288-365).
 * nthreads <= 1: channels are processed one after another on the caller.
 * nthreads  > 1: channels are spread over that many pthreads (the reference
 * runs one thread per channel; conversion stays serial as in the reference). */
void vdl2o_process(vdl2o_ctx *c, const uint8_t *buf, uint32_t len, int nthreads);

/* A whole capture with persistent threads, fed in block_bytes pieces like process_iq_file() (dumpvdl2.c:323-358).
 * VDL2O_RUN_THREAD_PER_CHANNEL: the reference's threading - one thread per channel + the producer, two barriers of count N+1
 * per block, serial sample conversion on the producer (dumpvdl2.c:117-135, demod.c:300-301,342-346,356-365); nthreads ignored.
 * VDL2O_RUN_WORKQUEUE: nthreads persistent workers, conversion spread over them, channels handed out dynamically.
 * Frames are gathered per block in channel order, exactly as by repeated vdl2o_process() calls.  Returns 0 or < 0. */
enum { VDL2O_RUN_THREAD_PER_CHANNEL = 1, VDL2O_RUN_WORKQUEUE = 2 };
int vdl2o_run(vdl2o_ctx *c, const uint8_t *buf, uint64_t total_len, uint32_t block_bytes, int mode, int nthreads);

size_t vdl2o_num_frames(const vdl2o_ctx *c);
const vdl2o_frame *vdl2o_frames(const vdl2o_ctx *c);
const uint8_t *vdl2o_octets(const vdl2o_ctx *c);
void vdl2o_clear_frames(vdl2o_ctx *c);
void vdl2o_counters(const vdl2o_ctx *c, int chan, uint64_t out[VDL2O_NUM_COUNTERS]);

/* filter coefficients / NCO step as computed at init (KAT hooks) */
void vdl2o_get_lpf(const vdl2o_ctx *c, float A[3], float B[3]);
uint32_t vdl2o_get_dphi(const vdl2o_ctx *c, int chan);
void vdl2o_get_sincos_lut(const vdl2o_ctx *c, float s[257], float co[257]);

/* Optional trace of the decimated stream (lp_re, lp_im interleaved) of one
 * channel; cap = number of complex samples the buffer can hold. */
void vdl2o_trace_decimated(vdl2o_ctx *c, int chan, float *dst, size_t cap);
size_t vdl2o_trace_count(const vdl2o_ctx *c);
/* trace every channel: dst[chan][cap_per_chan] complex (re,im) */
void vdl2o_trace_all(vdl2o_ctx *c, float *dst, size_t cap_per_chan);
int64_t vdl2o_decimated_count(const vdl2o_ctx *c, int chan);

/* Stand-alone pieces, exported for known-answer tests */
int  vdl2o_rs_decode(uint8_t block[255], int fec_octets);          /* rs.c:32-49 */
void vdl2o_rs_encode(const uint8_t data[249], uint8_t parity[6]);   /* generator side (no reference counterpart) */
uint32_t vdl2o_header_decode(uint32_t *word);                       /* decode.c:111-122; returns syndrome */
uint32_t vdl2o_header_parity(uint32_t upper20);                     /* 5 parity bits for a 20-bit field */
uint16_t vdl2o_crc16(const uint8_t *data, uint32_t len, uint16_t init); /* crc.c:21-64 (own bitwise version) */
void vdl2o_chebyshev(float fc, float ripple_pct, float A[3], float B[3]); /* chebyshev.c:67-119, 2 poles */
int  vdl2o_avlc_screen(const uint8_t *buf, uint32_t len, uint32_t *dst, uint32_t *src, int *dir);  /* avlc.c:163-236 up to the addresses */

#ifdef __cplusplus
}
#endif
#endif
