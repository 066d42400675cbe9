"""The drop-in adapter against the reference tree itself (only where /root/reference is mounted: this container, not the GPU box).

1. every prototype include/vdl2hip_dropin.h declares is, token for token, the one the reference declares
   (src/dumpvdl2.h:371-388, src/decode.h:31), and the stand-alone copies of octet_string_t / vdl2_msg_metadata list the
   reference's fields in the reference's order (src/dumpvdl2.h:421-425, src/output-common.h:31-43);
2. dumpvdl2_amd/csrc/dropin.c goes through the compiler in the mode a dumpvdl2 maintainer would build it in
   (-DVDL2HIP_IN_TREE: the tree's own dumpvdl2.h / output-common.h / decode.h instead of the stand-alone copies).  The tree's
   headers pull in glib / libacars / a cmake-generated config.h, none of which this image has; the test writes six one-line
   forward declarations into a temporary directory for them.  That is a SYNTAX AND TYPE CHECK OF THIS REPO'S ADAPTER, nothing
   more: no reference code is compiled, nothing is linked or run, and no parity claim rests on it.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "dumpvdl2.h")), reason="reference tree not mounted")


def _strip(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _norm(decl):
    decl = re.sub(r"\s+", " ", decl).strip()
    decl = re.sub(r"\(\s*\)", "(void)", decl)            # C: f() in the reference, f(void) here - the same type for a definition
    return re.sub(r"\s*([(),*])\s*", r"\1", decl)


def _prototypes(path, names):
    src = _strip(open(path).read())
    out = {}
    for n in names:
        m = re.search(r"(?:^|[;{}\n])\s*((?:extern\s+)?[A-Za-z_][A-Za-z_0-9 \*]*?\b%s\b\s*(?:\([^;{]*\))?)\s*;" % re.escape(n), src)
        assert m, f"{n} not declared in {path}"
        out[n] = _norm(m.group(1))
    return out


def _struct_fields(path, typedef_name):
    src = _strip(open(path).read())
    m = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*%s\s*;" % re.escape(typedef_name), src)
    assert m, f"{typedef_name} not found in {path}"
    return [re.sub(r"\s+", " ", f).strip() for f in m.group(1).split(";") if f.strip()]


def test_prototypes_are_the_references():
    ours = os.path.join(ROOT, "include", "vdl2hip_dropin.h")
    names = ["sbuf", "vdl2_channel_init", "sincosf_lut_init", "input_lpf_init", "demod_sync_init", "process_buf_uchar_init",
             "process_buf_uchar", "process_buf_short", "process_samples", "rs_init"]
    mine = _prototypes(ours, names + ["avlc_decoder_queue_push"])
    ref = _prototypes(os.path.join(REF, "dumpvdl2.h"), names)
    ref.update(_prototypes(os.path.join(REF, "decode.h"), ["avlc_decoder_queue_push"]))
    assert mine == ref
    assert _struct_fields(ours, "octet_string_t") == _struct_fields(os.path.join(REF, "dumpvdl2.h"), "octet_string_t")
    assert _struct_fields(ours, "vdl2_msg_metadata") == _struct_fields(os.path.join(REF, "output-common.h"), "vdl2_msg_metadata")


STUBS = {
    "config.h": "#define HAVE_PTHREAD_BARRIERS 1\n",
    "glib.h": "typedef struct GAsyncQueue_ GAsyncQueue; typedef void *gpointer;\n",
    "libacars/libacars.h": "typedef struct la_type_descriptor_ la_type_descriptor; typedef struct la_proto_node_ la_proto_node; typedef int la_msg_dir;\n",
    "libacars/vstring.h": "typedef struct la_vstring_ la_vstring;\n",
    "libacars/dict.h": "typedef struct la_dict_ la_dict;\n",
    "libacars/list.h": "typedef struct la_list_ la_list;\n",
}


def test_adapter_compiles_in_tree_mode(tmp_path):
    for rel, txt in STUBS.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text("/* throw-away forward declarations for a syntax check of dropin.c (tests/test_dropin_in_tree.py) */\n" + txt)
    cmd = ["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-DVDL2HIP_IN_TREE", "-I", str(tmp_path), "-I", REF,
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "dumpvdl2_amd", "csrc", "dropin.c")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
