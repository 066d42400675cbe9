/*
 * vdl2hip_dropin.h - the reference's own symbol names, implemented on top of the
 * C ABI of vdl2hip.h, so that an unmodified dumpvdl2 main() (src/dumpvdl2.c) can link
 * the GPU path instead of src/demod.c + src/decode.c's burst half + src/rs.c +
 * src/chebyshev.c + src/bitstream.c.  Every declaration below has the signature the
 * reference declares at the cited line.
 *
 * Build modes of dumpvdl2_amd/csrc/dropin.c:
 *   -DVDL2HIP_IN_TREE  : compiled inside a dumpvdl2 source tree; uses the tree's own
 *                        dumpvdl2.h / output-common.h / decode.h types.
 *   (default)          : stand-alone; uses the layout-compatible types declared here
 *                        (used by tests/dropin_harness.c on the GPU box).
 */
#ifndef VDL2HIP_DROPIN_H
#define VDL2HIP_DROPIN_H
#include <stddef.h>
#include <stdint.h>
#include <sys/time.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef VDL2HIP_IN_TREE
/* src/dumpvdl2.h:421-425 */
typedef struct {
	uint8_t *buf;
	size_t len;
} octet_string_t;

/* src/output-common.h:31-43 */
typedef struct {
	char *station_id;
	uint32_t freq;
	uint32_t synd_weight;
	uint32_t datalen_octets;
	float frame_pwr_dbfs;
	float nf_pwr_dbfs;
	float ppm_error;
	int version;
	int num_fec_corrections;
	int idx;
	struct timeval burst_timestamp;
} vdl2_msg_metadata;

/* opaque stand-in for vdl2_channel_t (src/dumpvdl2.h:321-352); main() only stores the pointer
 * and writes the demod_thread field, so the adapter hands out a zeroed block that is larger */
typedef struct vdl2_channel_s vdl2_channel_t;

/* stand-alone mode has no global Config (src/dumpvdl2.h:205-218): set the two fields the path reads */
void vdl2hip_dropin_configure(float max_ppm, char *station_id);
#endif

/* ---- provided by the adapter (reference declarations: src/dumpvdl2.h:371-388) ---- */
extern float *sbuf;                                                       /* :372 (kept defined, unused) */
vdl2_channel_t *vdl2_channel_init(uint32_t centerfreq, uint32_t freq,
		uint32_t source_rate, uint32_t oversample);                        /* :373, src/demod.c:379 */
void sincosf_lut_init(void);                                               /* :374, src/demod.c:372 */
void input_lpf_init(uint32_t sample_rate);                                 /* :375, src/demod.c:367 */
void demod_sync_init(void);                                                /* :376, src/demod.c:84  */
void process_buf_uchar_init(void);                                         /* :377, src/demod.c:349 */
void process_buf_uchar(unsigned char *buf, uint32_t len, void *ctx);       /* :378, src/demod.c:339 */
void process_buf_short(unsigned char *buf, uint32_t len, void *ctx);       /* :380, src/demod.c:356 */
void *process_samples(void *arg);                                          /* :381, src/demod.c:288 */
int rs_init(void);                                                         /* :387, src/rs.c:27     */

/* ---- consumed by the adapter (defined by the unmodified reference) ---- */
/* src/decode.h:31, src/decode.c:165-171: takes ownership; the consumer free()s all three allocations */
void avlc_decoder_queue_push(vdl2_msg_metadata *metadata, octet_string_t *frame, int flags);
/* src/dumpvdl2.c:66-67: pthread_barrier_t demods_ready, samples_ready (count = channels + 1) */

#ifdef __cplusplus
}
#endif
#endif
