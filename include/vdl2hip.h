/*
 * vdl2hip.h - C ABI of libvdl2hip.so: the MI355X-native replacement for
 * dumpvdl2's per-channel DSP + burst decoder hot path.
 *
 * Boundary it replaces (reference @ v2.6.0, paths relative to the reference root):
 *   input  : process_buf_uchar()/process_buf_short()          src/demod.c:339-365, src/dumpvdl2.h:378-380
 *   setup  : vdl2_channel_init(), input_lpf_init(),
 *            sincosf_lut_init(), demod_sync_init(), rs_init()   src/demod.c:367-392, src/demod.c:84-96, src/rs.c:27-30
 *   work   : process_samples() -> demod() -> got_sync()
 *            -> decode_vdl2_burst() -> decode_frame()            src/demod.c:105-337, src/decode.c:173-384
 *   output : avlc_decoder_queue_push(metadata, frame, flags)     src/decode.c:165-171, src/decode.h:31
 *
 * Plain C: opaque context, plain pointers and sizes, int return codes
 * (0 = ok, negative = VDL2HIP_E_*).  No HIP or torch types appear here.
 * The adapter that re-exports the reference's own symbol names on top of
 * this ABI is include/vdl2hip_dropin.h.
 */
#ifndef VDL2HIP_H
#define VDL2HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDL2HIP_ABI_VERSION 6   /* 2: vdl2hip_frame carries the AVLC verdict; vdl2hip_stats grew; avlc/statsd calls added
                                 * 3: vdl2hip_feed_pinned(); vdl2hip_stats.overflow_feeds; vdl2hip_group_*: one receiver over several
                                 *    GPUs from C (a channeliser look-back that gives up is no longer an error: it falls back)
                                 * 4: vdl2hip_group_set_exchange() / vdl2hip_group_exchange(): striped ingest + all-gather is the group's
                                 *    default exchange, broadcast stays selectable; RCCL is opt-in (VDL2HIP_USE_RCCL=1)
                                 * 5: the referee (decisions within the margin of the channeliser's distance from the reference's fp32 scan are
                                 *    taken on the reference's own samples): vdl2hip_stats grew by referee_*; vdl2hip_get_stats_sized() for callers
                                 *    built against an older vdl2hip_stats.  (Since ABI 4 chanfir_ms / chanfir_launches / chan_samples cover only
                                 *    the TIMED channeliser launches - profiling on, cold-start feeds excluded - not every launch.)
                                 * 6: vdl2hip_stats.referee_redone_next (a feed's walk no longer waits for the check of the feed before) */

/* enum sample_formats, src/dumpvdl2.h:319 */
#define VDL2HIP_FMT_U8     0
#define VDL2HIP_FMT_S16LE  1

#define VDL2HIP_OK            0
#define VDL2HIP_E_INVAL      -1   /* bad argument / configuration */
#define VDL2HIP_E_NOMEM      -2
#define VDL2HIP_E_DEVICE     -3   /* HIP runtime error (no GPU, launch failure ...) */
#define VDL2HIP_E_TOOBIG     -4   /* block larger than max_block_bytes */
#define VDL2HIP_E_OVERFLOW   -5   /* device-side frame/burst buffer exhausted (frames were dropped) */

/* Per-channel counters = the reference's per-channel statsd counters on this
 * path (src/statsd.c:34-65; call sites src/demod.c:245, src/decode.c:204-373),
 * plus two that have no statsd name. */
enum {
	VDL2HIP_CNT_SYNC_GOOD = 0,          /* demod.sync.good */
	VDL2HIP_CNT_CRC_GOOD,               /* decoder.crc.good */
	VDL2HIP_CNT_CRC_BAD,                /* decoder.crc.bad */
	VDL2HIP_CNT_ERR_NO_HEADER,          /* decoder.errors.no_header */
	VDL2HIP_CNT_ERR_TOO_LONG,           /* decoder.errors.too_long */
	VDL2HIP_CNT_ERR_NO_FEC,             /* decoder.errors.no_fec */
	VDL2HIP_CNT_ERR_DATA_TRUNCATED,     /* decoder.errors.data_truncated */
	VDL2HIP_CNT_ERR_FEC_TRUNCATED,      /* decoder.errors.fec_truncated */
	VDL2HIP_CNT_ERR_DEINTERLEAVE_DATA,  /* decoder.errors.deinterleave_data */
	VDL2HIP_CNT_ERR_DEINTERLEAVE_FEC,   /* decoder.errors.deinterleave_fec */
	VDL2HIP_CNT_ERR_FEC_BAD,            /* decoder.errors.fec_bad */
	VDL2HIP_CNT_ERR_BITSTREAM,          /* decoder.errors.bitstream */
	VDL2HIP_CNT_ERR_TRUNCATED_OCTETS,   /* decoder.errors.truncated_octets */
	VDL2HIP_CNT_ERR_UNSTUFF,            /* decoder.errors.unstuff */
	VDL2HIP_CNT_BLOCKS_PROCESSED,       /* decoder.blocks.processed */
	VDL2HIP_CNT_BLOCKS_FEC_OK,          /* decoder.blocks.fec_ok */
	VDL2HIP_CNT_MSG_GOOD,               /* decoder.msg.good */
	VDL2HIP_CNT_MSG_GOOD_LOUD,          /* decoder.msg.good_loud */
	VDL2HIP_CNT_PPM_REJECT,             /* preambles dropped by max_ppm (src/demod.c:192) */
	VDL2HIP_CNT_SLICER_NEG_IDX,         /* slicer index < 0: the reference reads out of bounds there (src/demod.c:264) */
	VDL2HIP_NUM_COUNTERS
};

typedef struct vdl2hip_ctx vdl2hip_ctx;

typedef struct {
	uint32_t struct_size;       /* sizeof(vdl2hip_cfg), for ABI evolution */
	uint32_t centerfreq;        /* Hz; vdl2_channel_init() arg 1 */
	uint32_t oversample;        /* sample rate = 105000 * oversample (src/dumpvdl2.c:1073) */
	uint32_t sample_fmt;        /* VDL2HIP_FMT_* */
	uint32_t nchan;
	const uint32_t *freqs;      /* nchan channel frequencies, Hz */
	float    max_ppm;           /* Config.max_ppm; 0 disables (src/demod.c:192) */
	int32_t  device;            /* HIP device ordinal.  Every call on the context runs on this device whatever the calling thread's
	                             * current device is, and leaves the thread's current device as it found it */
	uint32_t max_block_bytes;   /* largest block a feed call may carry; 0 = 320000 (FILE_BUFSIZE) */
	uint32_t chan_first;        /* multi-GPU sharding: this context decodes channels            */
	uint32_t chan_count;        /*   [chan_first, chan_first+chan_count) of freqs[]; 0 = all     */
} vdl2hip_cfg;

/* One AVLC frame plus the vdl2_msg_metadata the reference attaches to it
 * (src/output-common.h:31-43).  `octets` is only valid during the callback. */
typedef struct {
	uint32_t chan;              /* index into cfg.freqs */
	uint32_t freq;              /* metadata->freq */
	int32_t  idx;               /* metadata->idx: frame number within the burst */
	uint32_t len;               /* frame length in octets (may be 0, as in the reference) */
	const uint8_t *octets;
	uint32_t synd_weight;
	uint32_t datalen_octets;
	int32_t  num_fec_corrections;
	float    frame_pwr_dbfs;
	float    nf_pwr_dbfs;
	float    ppm_error;
	int64_t  burst_ord;         /* ordinal of the burst on its channel (0,1,...) */
	int64_t  sync_sample;       /* decimated-sample index (105 kS/s clock) at which the preamble locked */
	int64_t  end_sample;        /* decimated-sample index at which the burst was complete */
	/* avlc_parse()'s first checks (src/avlc.c:163-199), done on the device: */
	uint32_t avlc_status;       /* VDL2HIP_AVLC_OK / _TOO_SHORT (len < 11) / _BAD_FCS (crc16_ccitt residue != 0xF0B8) */
	uint32_t dst_addr, src_addr;/* parse_dlc_addr() of octets 0-3 / 4-7: addr:24 | type:3 << 24 | status:1 << 27; 0 unless OK */
} vdl2hip_frame;

enum { VDL2HIP_AVLC_OK = 0, VDL2HIP_AVLC_TOO_SHORT = 1, VDL2HIP_AVLC_BAD_FCS = 2 };

/* Per-channel counters of the AVLC front door = the reference's statsd counters of src/decode.c:466 and
 * src/avlc.c:170-233, in this order */
enum {
	VDL2HIP_ACNT_FRAMES_PROCESSED = 0,  /* avlc.frames.processed */
	VDL2HIP_ACNT_ERR_TOO_SHORT,         /* avlc.errors.too_short */
	VDL2HIP_ACNT_FRAMES_GOOD,           /* avlc.frames.good */
	VDL2HIP_ACNT_ERR_BAD_FCS,           /* avlc.errors.bad_fcs */
	VDL2HIP_ACNT_MSG_AIR2GND, VDL2HIP_ACNT_MSG_AIR2AIR, VDL2HIP_ACNT_MSG_AIR2ALL,   /* avlc.msg.* */
	VDL2HIP_ACNT_MSG_GND2AIR, VDL2HIP_ACNT_MSG_GND2GND, VDL2HIP_ACNT_MSG_GND2ALL,
	VDL2HIP_NUM_AVLC_COUNTERS
};

typedef void (*vdl2hip_frame_cb)(const vdl2hip_frame *frame, void *user);

typedef struct {
	uint64_t feeds;             /* feed calls so far */
	uint64_t input_samples;     /* complex input samples consumed */
	uint64_t chan_samples;      /* channel-samples processed by the TIMED launches of the channeliser kernel (profiling on) */
	uint64_t chanfir_launches;  /* timed launches of the channeliser kernel: feeds with profiling on, except cold-start feeds */
	double   chanfir_ms;        /* summed HIP-event time of those launches */
	double   phase_ms, sync_ms, walk_ms, burst_ms;   /* other kernels, same convention */
	double   nf_ms;             /* noise-floor passes */
	uint64_t bursts;            /* bursts handed to the burst decoder */
	uint64_t frames;            /* frames the burst decoder produced (before the optional AVLC filter of vdl2hip_set_avlc_filter) */
	uint64_t seg_adopted;       /* segmented walk: speculative segments adopted ... */
	uint64_t seg_walked;        /* ... and segments walked sequentially because a burst straddled their start */
	uint64_t front_sync_timeouts; /* channeliser workgroups that stopped waiting for their predecessor's filter state and worked it out
	                               * themselves (one more tile of work each).  The state they compute equals the published one up to fp32
	                               * rounding (a sum instead of a scan), i.e. the first 128 decimated outputs of such a segment may differ
	                               * from a run without fall-backs in their last bit (measured with every workgroup forced to fall back:
	                               * <= 2e-7 of the signal's peak, a fiftieth of the reference's own rounding noise): NOT bit-identical, and
	                               * which workgroups fall back depends on scheduling.  0 with one process per GPU; non-zero where the GPU is time-sliced
	                               * between processes - set VDL2HIP_NO_FUSE=1 there if bit-reproducible output is required */
	uint64_t overflow_feeds;    /* feeds in which a device-side burst/frame/octet buffer ran out (bursts or frames were dropped);
	                             * vdl2hip_sync() returns VDL2HIP_E_OVERFLOW for those, the drain calls only count here */
	uint64_t cold_start_feeds;  /* large page-locked blocks fed to an idle receiver: copied and channelised in pieces (vdl2hip_feed_pinned);
	                             * their channeliser launches are not in chanfir_ms / chanfir_launches */
	/* The referee.  The channeliser's samples are what exact arithmetic gives; the reference's own fp32 scan (src/demod.c:302-329) differs
	 * from that by its rounding noise (<= 1.5e-4 of the local amplitude).  A decision of the reference that could come out differently
	 * within that distance - candidate test, parabola vertex, --max-ppm gate (src/demod.c:173-192), a symbol at a slicer boundary
	 * (:256-264) - is taken on the reference's own samples, recomputed by running its scan sequentially over the raw input: */
	uint64_t referee_scans;     /* such scans run (3.3 ms each by one wavefront, or 32 side by side by a workgroup) */
	uint64_t referee_cached;    /* requests for a stretch that had been made exact already */
	uint64_t referee_refused;   /* requests that could not be served (the raw input was no longer held): the decision stayed as it was */
	uint64_t referee_short;     /* scans whose run-up was shorter than configured (early in a stream of short blocks) */
	uint64_t referee_rewalks;   /* channels walked again because a decision taken on the channeliser's samples did not stand on the reference's */
	uint64_t referee_candidate_scans, referee_header_scans, referee_symbol_scans;   /* referee_scans by the decision that asked: a preamble candidate
	                             * (candidate test / vertex / gate), a header symbol, the symbols of a burst */
	uint64_t referee_redone_next; /* ABI 6.  A feed's walk no longer waits for the check of the feed before (it starts from that feed's unchecked end
	                             * state; referee_rewalks counts the channels walked again after a check): how often such a second walk ended in
	                             * a DIFFERENT state or counters, so that the next feed was walked once more for that channel as well */
	uint64_t referee_unmet;     /* ABI 6.  Scans (of those run side by side: all of a long feed's) whose zero-start trajectory had NOT become
	                             * bit-identical to a witness trajectory started elsewhere by the stretch's first output - the run-up (196 608
	                             * input samples; VDL2HIP_REF_WARM) was too short for it to have forgotten its start: that stretch is within the
	                             * reference's rounding noise of the reference's samples, not bit for bit them.  Counted here: those that could
	                             * NOT be run again (below).  A monitor, not a proof: a scan can meet its witness before it meets the reference's
	                             * trajectory (measured share of stretches that are not the reference's bit for bit: DESIGN 5) */
	uint64_t referee_retried;   /* ABI 6.  ... and those that were: listed and scanned again from twice as far back (VDL2HIP_REF_RETRY) */
} vdl2hip_stats;

int  vdl2hip_abi_version(void);
const char *vdl2hip_strerror(int err);

/* = vdl2_channel_init() x nchan + input_lpf_init() + sincosf_lut_init() + demod_sync_init() + rs_init() */
int  vdl2hip_create(const vdl2hip_cfg *cfg, vdl2hip_ctx **out);
void vdl2hip_destroy(vdl2hip_ctx *ctx);

/* = process_buf_uchar()/process_buf_short(): one block of raw IQ from host memory.
 * Returns after the block has been queued on the device (the copy out of `buf` is complete).  The copy runs on a
 * stream of its own into one of six device buffers (one per block in flight), so it overlaps the kernels of the blocks fed before. */
int  vdl2hip_feed(vdl2hip_ctx *ctx, const void *buf, size_t nbytes);
/* Same for page-locked host memory (hipHostMalloc / hipHostRegister), without waiting for the copy: the call only queues.
 * `buf` must stay unmodified until the NEXT vdl2hip_feed*() call or vdl2hip_sync() has returned - i.e. a producer
 * alternating between two pinned buffers never waits for the device.  (A block of 8 MiB or more handed to an idle receiver
 * is copied in four pieces with the channeliser following piece by piece, so that the first block of a stream does not wait
 * for its whole transfer; results do not depend on it.) */
int  vdl2hip_feed_pinned(vdl2hip_ctx *ctx, const void *buf, size_t nbytes);
/* Same, for a block that already lives in this device's memory (e.g. the
 * destination of an RCCL broadcast).  The block must stay valid until it has been drained
 * (vdl2hip_sync(), or a drain that covers it - see vdl2hip_set_drain_lag). */
int  vdl2hip_feed_device(vdl2hip_ctx *ctx, const void *dev_buf, size_t nbytes);

/* Pipelining.  A feed call only queues work: the sample-rate front (K1-K3) of blocks i+1, i+2 may run while the
 * burst-rate back (K4-K5) of block i is still in flight.  By default the drain functions wait for everything (lag 0 =
 * the reference's blocking behaviour).  With lag L (1 .. VDL2HIP_MAX_DRAIN_LAG) they deliver every block except the L most recent ones,
 * so a feed/drain loop keeps L+1 blocks in flight; vdl2hip_sync() always completes everything. */
#ifndef VDL2HIP_MAX_DRAIN_LAG
#define VDL2HIP_MAX_DRAIN_LAG 5       /* feeds that may be under way undelivered: lag 0 .. 5 (ABI 6; 3 before) */
#endif
int  vdl2hip_set_drain_lag(vdl2hip_ctx *ctx, int lag);

/* Wait for all queued blocks; moves finished frames to the host-side queue. */
int  vdl2hip_sync(vdl2hip_ctx *ctx);
/* sync + deliver every queued frame, ordered by (end_sample, chan, idx); returns the number delivered (>= 0). */
int  vdl2hip_drain(vdl2hip_ctx *ctx, vdl2hip_frame_cb cb, void *user);

/* Bulk form of vdl2hip_drain for callers that want no per-frame callback: copies up to `cap_frames`
 * frame records (same order; `octets` is NULL, `octets_off` below locates the payload) and their octets,
 * concatenated, into caller memory.  Returns the number of frames copied; frames that did not fit stay queued. */
typedef struct {
	vdl2hip_frame frame;
	uint64_t octets_off;
} vdl2hip_packed_frame;
int  vdl2hip_drain_packed(vdl2hip_ctx *ctx, vdl2hip_packed_frame *frames, size_t cap_frames,
		uint8_t *octets, size_t cap_octets, size_t *octets_used);

/* Serialise one frame in the reference's raw-frame archive format, i.e. what `--output raw:binary:file:...`
 * writes and `--raw-frames-file` reads back: 2-byte big-endian record length (payload + 2) followed by the
 * proto3 message dumpvdl2.raw_avlc_frame { vdl2_msg_metadata metadata = 1; bytes data = 2; }
 * (proto/dumpvdl2.proto:24-47, src/fmtr-binary.c:28-60, src/output-file.c:176-192, reader
 * src/input-raw_frames_file.c:33-107).  `frame->octets` must be valid.  Needs no GPU.
 * Returns the number of bytes written, or VDL2HIP_E_TOOBIG if `cap` is too small / the record exceeds 65535. */
int  vdl2hip_pack_raw_frame(const vdl2hip_frame *frame, const char *station_id, int64_t tv_sec, int64_t tv_usec,
		uint8_t *out, size_t cap);

int  vdl2hip_counters(vdl2hip_ctx *ctx, uint32_t chan, uint64_t out[VDL2HIP_NUM_COUNTERS]);
int  vdl2hip_avlc_counters(vdl2hip_ctx *ctx, uint32_t chan, uint64_t out[VDL2HIP_NUM_AVLC_COUNTERS]);
/* Deliver only frames that pass the AVLC front door (avlc_parse() returns NULL for the others, src/avlc.c:171,186): with
 * `on` the drain functions skip frames whose avlc_status is not OK.  Counters are unaffected.  Default off: every frame
 * reaches the callback, as every frame reaches avlc_decoder_queue_push() in the reference. */
int  vdl2hip_set_avlc_filter(vdl2hip_ctx *ctx, int on);
/* The reference's statsd traffic (src/statsd.c:34-65,153-160) in aggregate: one "<ns>.<freq>.<counter>:<delta>|c" line per
 * counter that changed since the previous call (all counters, with :0, on the first call, like
 * statsd_initialize_counters_per_channel()).  `ns` is the namespace ("dumpvdl2" or "dumpvdl2.<station_id>").  Returns the
 * number of bytes written (excluding the terminating NUL) or VDL2HIP_E_TOOBIG if `cap` is too small (nothing is consumed). */
int  vdl2hip_statsd_lines(vdl2hip_ctx *ctx, const char *ns, char *out, size_t cap);
int  vdl2hip_set_profiling(vdl2hip_ctx *ctx, int level); /* 0 off; 1 time the channeliser kernel (start/stop events attached to
                                                          * its launch); 2 time every stage the same way (costs ~5 % throughput) */
int  vdl2hip_get_stats(vdl2hip_ctx *ctx, vdl2hip_stats *out);
int  vdl2hip_get_stats_sized(vdl2hip_ctx *ctx, vdl2hip_stats *out, size_t size);   /* writes at most `size` bytes: for a caller built against an older header */
void *vdl2hip_stream(vdl2hip_ctx *ctx);                 /* the hipStream_t all work is queued on */

/* ---- One receiver over several GPUs of this process (src/dumpvdl2.c:117-135: one worker per channel over a shared block;
 * here the workers are grouped by device).  Member k of n decodes channels [k*nchan/n, (k+1)*nchan/n) of cfg->freqs on
 * devices[k]; cfg->device, chan_first and chan_count are ignored/must be 0.  A block handed to vdl2hip_group_feed() is put on
 * every member in one of two ways (vdl2hip_group_set_exchange(), or VDL2HIP_GROUP_EXCHANGE=allgather|broadcast at create):
 *   VDL2HIP_GROUP_ALLGATHER (default)  the block is cut into n stripes; member k copies stripe k from `buf` over ITS OWN PCIe
 *                                      link, then fetches the stripes it lacks from its peers over xGMI, all links at once;
 *   VDL2HIP_GROUP_BROADCAST            the whole block crosses PCIe once, into devices[0], and is sent from there to the others
 *                                      (BASELINE's literal form; bound by that one host copy).
 * Both give every member the same bytes, hence the same frames.  Data moves with hipMemcpyPeerAsync - the path that has run on
 * hardware; a device may be listed more than once ("virtual shards", which is how the path is tested on one GPU).  RCCL
 * (ncclAllGather / ncclBroadcast, loaded at run time) is used only with VDL2HIP_USE_RCCL=1 in the environment: those calls have
 * been built and reviewed but not yet run on a multi-GPU node (tests/test_gpu_parity.py::test_group_over_two_real_gpus covers them
 * where two GPUs are visible); a failing RCCL call falls back to peer copies.  Frames are delivered merged, in vdl2hip_drain()'s
 * order.  A failure part-way through a group feed disables the group (every later call returns VDL2HIP_E_DEVICE). ---- */
typedef struct vdl2hip_group vdl2hip_group;
enum { VDL2HIP_GROUP_ALLGATHER = 0, VDL2HIP_GROUP_BROADCAST = 1 };
int  vdl2hip_group_create(const vdl2hip_cfg *cfg, const int32_t *devices, uint32_t ndev, vdl2hip_group **out);
void vdl2hip_group_destroy(vdl2hip_group *g);
int  vdl2hip_group_feed(vdl2hip_group *g, const void *buf, size_t nbytes);      /* = process_buf_*(), blocking like vdl2hip_feed() */
/* the same from page-locked memory without waiting for the copy (the rule of vdl2hip_feed_pinned(): `buf` stays untouched until the
 * next vdl2hip_group_feed*() or vdl2hip_group_sync() has returned) */
int  vdl2hip_group_feed_pinned(vdl2hip_group *g, const void *buf, size_t nbytes);
int  vdl2hip_group_sync(vdl2hip_group *g);
int  vdl2hip_group_drain(vdl2hip_group *g, vdl2hip_frame_cb cb, void *user);
int  vdl2hip_group_set_drain_lag(vdl2hip_group *g, int lag);
int  vdl2hip_group_counters(vdl2hip_group *g, uint32_t chan, uint64_t out[VDL2HIP_NUM_COUNTERS]);
int  vdl2hip_group_avlc_counters(vdl2hip_group *g, uint32_t chan, uint64_t out[VDL2HIP_NUM_AVLC_COUNTERS]);
uint32_t vdl2hip_group_size(vdl2hip_group *g);
vdl2hip_ctx *vdl2hip_group_ctx(vdl2hip_group *g, uint32_t member);            /* for the per-context calls above (stats, statsd, ...) */
int  vdl2hip_group_uses_rccl(vdl2hip_group *g);                                /* 1: RCCL loaded and in use, 0: peer copies */
int  vdl2hip_group_set_exchange(vdl2hip_group *g, int form);                   /* VDL2HIP_GROUP_ALLGATHER / _BROADCAST, for the feeds that follow */
/* what the last feed did: 0 broadcast by peer copies, 1 broadcast by RCCL, 2 all-gather by peer copies, 3 all-gather by RCCL; -1 before the first feed */
int  vdl2hip_group_exchange(vdl2hip_group *g);

/* Introspection used by the parity tests (host copies of what the kernels use) */
int  vdl2hip_get_lpf(vdl2hip_ctx *ctx, float A[3], float B[3]);      /* = static A/B of src/demod.c:55 */
int  vdl2hip_get_nco_step(vdl2hip_ctx *ctx, uint32_t chan, uint32_t *dphi); /* = v->downmix_dphi */
/* Copy up to `cap` decimated (re,im) pairs of one channel starting at decimated index `first`
 * (must still be inside the device history window); returns the count copied. */
int  vdl2hip_read_decimated(vdl2hip_ctx *ctx, uint32_t chan, int64_t first, float *dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
