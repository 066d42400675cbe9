"""vdl2hip_pack_raw_frame() against an independent proto3 implementation (python protobuf) of the schema in the
reference's proto/dumpvdl2.proto:24-47, plus the 2-byte big-endian record framing of src/output-file.c:176-192."""
import struct

import numpy as np
import pytest


def build_schema():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "dumpvdl2_test.proto"; fd.package = "dumpvdl2"; fd.syntax = "proto3"
    md = fd.message_type.add(); md.name = "vdl2_msg_metadata"
    T = descriptor_pb2.FieldDescriptorProto
    for name, num, typ in [("station_id", 1, T.TYPE_STRING), ("frequency", 2, T.TYPE_UINT32), ("synd_weight", 3, T.TYPE_UINT32),
                           ("datalen_octets", 4, T.TYPE_UINT32), ("frame_pwr_dbfs", 5, T.TYPE_FLOAT), ("nf_pwr_dbfs", 6, T.TYPE_FLOAT),
                           ("ppm_error", 7, T.TYPE_FLOAT), ("version", 8, T.TYPE_INT32), ("num_fec_corrections", 9, T.TYPE_INT32),
                           ("idx", 10, T.TYPE_INT32)]:
        f = md.field.add(); f.name = name; f.number = num; f.type = typ; f.label = T.LABEL_OPTIONAL
    ts = md.nested_type.add(); ts.name = "timestamp"
    for name, num in [("tv_sec", 1), ("tv_usec", 2)]:
        f = ts.field.add(); f.name = name; f.number = num; f.type = T.TYPE_INT64; f.label = T.LABEL_OPTIONAL
    f = md.field.add(); f.name = "burst_timestamp"; f.number = 11; f.type = T.TYPE_MESSAGE; f.label = T.LABEL_OPTIONAL
    f.type_name = ".dumpvdl2.vdl2_msg_metadata.timestamp"
    rf = fd.message_type.add(); rf.name = "raw_avlc_frame"
    f = rf.field.add(); f.name = "metadata"; f.number = 1; f.type = T.TYPE_MESSAGE; f.label = T.LABEL_OPTIONAL; f.type_name = ".dumpvdl2.vdl2_msg_metadata"
    f = rf.field.add(); f.name = "data"; f.number = 2; f.type = T.TYPE_BYTES; f.label = T.LABEL_OPTIONAL
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    desc = pool.FindMessageTypeByName("dumpvdl2.raw_avlc_frame")
    return get(desc) if get else message_factory.MessageFactory(pool).GetPrototype(desc)


@pytest.fixture(scope="module")
def RawFrame():
    return build_schema()


def mk(rng, n, **kw):
    f = dict(chan=1, freq=136975000, idx=int(rng.integers(0, 3)), octets=rng.integers(0, 256, n, dtype=np.uint8).tobytes(),
             synd_weight=int(rng.integers(0, 3)), datalen_octets=n + 5, num_fec_corrections=int(rng.integers(0, 9)),
             frame_pwr_dbfs=float(np.float32(rng.normal(-20, 5))), nf_pwr_dbfs=float(np.float32(rng.normal(-40, 3))),
             ppm_error=float(np.float32(rng.normal(0, 2))))
    f.update(kw)
    return f


def test_records_decode_and_reencode_identically(RawFrame):
    from dumpvdl2_amd import vdl2hip
    rng = np.random.default_rng(5)
    cases = [mk(rng, int(n)) for n in rng.integers(1, 2000, 40)]
    cases += [mk(rng, 11, idx=0, synd_weight=0, num_fec_corrections=0, ppm_error=0.0),        # proto3 defaults are omitted
              mk(rng, 300, num_fec_corrections=-1), mk(rng, 0)]
    for i, f in enumerate(cases):
        sid = None if i % 3 == 0 else f"GS-{i}"
        rec = vdl2hip.pack_raw_frame(f, sid, 1790436287 + i, 123456 if i % 2 else 0)
        (ln,) = struct.unpack(">H", rec[:2])
        assert ln == len(rec)                                   # length counts its own two octets (output-file.c:176)
        m = RawFrame(); m.ParseFromString(rec[2:])
        md = m.metadata
        assert m.data == f["octets"] and md.station_id == (sid or "") and md.frequency == f["freq"]
        assert (md.synd_weight, md.datalen_octets, md.version, md.num_fec_corrections, md.idx) == \
               (f["synd_weight"], f["datalen_octets"], 1, f["num_fec_corrections"], f["idx"])
        assert np.float32(md.frame_pwr_dbfs) == np.float32(f["frame_pwr_dbfs"]) and np.float32(md.ppm_error) == np.float32(f["ppm_error"])
        assert md.HasField("burst_timestamp") and md.burst_timestamp.tv_sec == 1790436287 + i
        assert m.SerializeToString(deterministic=True) == rec[2:]   # canonical field order, same omissions


def test_reader_side_limits():
    from dumpvdl2_amd import vdl2hip
    rng = np.random.default_rng(6)
    with pytest.raises(vdl2hip.Vdl2HipError):
        vdl2hip.pack_raw_frame(mk(rng, 66000))                  # cannot be framed with a 16-bit length
