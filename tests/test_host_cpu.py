"""CPU tests of the C host (no GPU needed): the C-ABI library loads and exports every symbol include/mbn.h declares,
the topology table, the loaders kept from the reference, and the from-scratch Keras .h5 reader/writer — validated
against files written by the real libhdf5 (tests/golden/*.h5, generator: tests/golden/make_h5_fixtures.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LIBHDF5 = "/opt/conda/lib/libhdf5.so"


# ----------------------------------------------------------------------------- ABI surface

def test_abi_library_loads_and_exports_every_declared_symbol(pkg):
    lib = pkg.load()                       # libmbn.so: HIP kernels + C-ABI + C host; loading needs no GPU
    names = pkg.declared_symbols()
    assert len(names) >= 45
    for must in ("mbn_convolute", "mbn_depthwise", "mbn_pointwise", "mbn_pool", "mbn_init", "mbn_shutdown",
                 "readSquezeNetKernel", "decode_image", "mbn_h5_open", "mbn_h5_get", "mbn_weights_from_h5",
                 "mbn_net_forward", "mbn_profile_begin"):
        assert must in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    host = pkg.host_lib()                  # the C host alone must not depend on HIP
    assert hasattr(host, "mbn_plan_build") and not hasattr(host, "mbn_depthwise")


def test_layer_ext_struct_matches_header(pkg):
    """ctypes mirror == C struct: compile a 3-line C program against include/mbn.h and compare sizeof/offsetof."""
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "mbn.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(mbn_layer_ext), offsetof(mbn_layer_ext, quirks),
 offsetof(mbn_layer_ext, scale), offsetof(mbn_layer_ext, stream), sizeof(mbn_layer_desc), sizeof(mbn_plan),
 offsetof(mbn_plan, layer));return 0;}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(pkg.REPO_ROOT, "include"), "-o", os.path.join(d, "t"),
                               os.path.join(d, "t.c")])
        vals = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    E, D, P = pkg.LayerExt, pkg.LayerDesc, pkg.Plan
    assert vals == [C.sizeof(E), E.quirks.offset, E.scale.offset, E.stream.offset, C.sizeof(D), C.sizeof(P),
                    P.layer.offset]


def test_no_device_is_an_error_code_not_a_crash(pkg):
    lib = pkg.load()
    n = C.c_int(-1)
    assert lib.mbn_device_count(C.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    assert lib.mbn_init(0, C.byref(h)) == pkg.ENODEVICE      # no fallback path: nothing computes without a GPU
    assert not h.value
    with pytest.raises(pkg.MbnError):
        pkg.Context(0)
    assert lib.mbn_shutdown(None) == 0
    assert lib.mbn_sync(None) == pkg.EINVAL
    assert lib.mbn_depthwise(None, None, None, None, 1, 1, 3, 1, 1, None) == pkg.EINVAL
    assert lib.mbn_strerror(pkg.ENODEVICE).decode() == "no HIP device"


# ----------------------------------------------------------------------------- topology

def test_plan_matches_survey_table(pkg):
    p = pkg.plan_build(1.0, 224, 1000)
    assert p.n_layers == 29
    L = [p.layer[i] for i in range(29)]
    assert [l.kind for l in L] == [1] + [2, 3] * 13 + [4, 5]
    assert [l.index for l in L] == list(range(1, 30))
    # (in_rows, in_ch, out_rows, out_ch, stride)
    table = {1: (224, 3, 112, 32, 2), 2: (112, 32, 112, 32, 1), 3: (112, 32, 112, 64, 1), 4: (112, 64, 56, 64, 2),
             5: (56, 64, 56, 128, 1), 8: (56, 128, 28, 128, 2), 9: (28, 128, 28, 256, 1), 12: (28, 256, 14, 256, 2),
             13: (14, 256, 14, 512, 1), 24: (14, 512, 7, 512, 2), 25: (7, 512, 7, 1024, 1), 26: (7, 1024, 7, 1024, 1),
             27: (7, 1024, 7, 1024, 1), 28: (7, 1024, 1, 1024, 1), 29: (1, 1024, 1, 1000, 1)}
    for idx, want in table.items():
        l = L[idx - 1]
        assert (l.in_rows, l.in_ch, l.out_rows, l.out_ch, l.stride) == want, idx
    # TF-SAME: stride-1 depthwise pads 1, stride-2 on even sizes pads 0 (i.e. bottom/right only)
    assert (L[1].pad_top, L[3].pad_top, L[0].pad_top) == (1, 0, 0)
    # parameter counts: conv/FC 4 210 088 weights + 2 x 10 944 folded BN + 1000 bias
    assert sum(l.w_count for l in L) == 864 + 44640 + 3139584 + 1024000
    assert sum(l.out_ch for l in L if l.scale_offset >= 0) == 10944   # 27 BN layers: 10 944 scales (+ as many shifts)
    assert p.max_act_floats == 112 * 112 * 64
    offs = [l.w_offset for l in L if l.w_count] + [l.scale_offset for l in L if l.scale_offset >= 0]
    assert all(o % 64 == 0 for o in offs)      # 256-byte aligned segments


@pytest.mark.parametrize("alpha,res", [(0.5, 160), (0.25, 128), (0.75, 192), (1.0, 96)])
def test_plan_width_and_resolution(pkg, orc, alpha, res):
    p = pkg.plan_build(alpha, res, 1000)
    o = orc.plan_build(alpha, res, 1000)        # independent restatement of the same table
    assert (p.n_layers, p.blob_floats, p.max_act_floats) == (o.n_layers, o.blob_floats, o.max_act_floats)
    for i in range(p.n_layers):
        a, b = p.layer[i], o.layer[i]
        for f in ("index", "kind", "in_rows", "in_cols", "in_ch", "out_rows", "out_cols", "out_ch", "stride", "pad_top",
                  "pad_left", "w_offset", "w_count", "scale_offset", "shift_offset"):
            assert getattr(a, f) == getattr(b, f), (i, f)
    assert p.layer[0].out_ch == int(32 * alpha) and p.layer[26].out_ch == int(1024 * alpha)
    assert p.layer[27].in_rows == res // 32


def test_plan_rejects_bad_arguments(pkg):
    lib = pkg.host_lib()
    p = pkg.Plan()
    for a, r, c in ((0.0, 224, 1000), (-1.0, 224, 1000), (1.0, 8, 1000), (1.0, 224, 0), (0.01, 224, 10)):
        assert lib.mbn_plan_build(a, r, c, C.byref(p)) == pkg.EINVAL
    assert lib.mbn_plan_build(1.0, 224, 1000, None) == pkg.EINVAL
    # odd feature maps are where TF-SAME and Keras' ZeroPadding2D + 'valid' disagree: only res % 32 == 0 is offered
    for r in (100, 150, 225, 33):
        assert lib.mbn_plan_build(1.0, r, 1000, C.byref(p)) == pkg.EUNSUPPORTED


# ----------------------------------------------------------------------------- loaders kept from the reference

def test_text_weight_loader(pkg, tmp_path):
    lib = pkg.host_lib()
    f = tmp_path / "weights_c.txt"
    f.write_text("1.9 -2.7e0 3 0.999 -0.5\n  1e2\t7.0e-1 300")
    m = np.full(10, -99, np.int32)
    assert lib.mbn_read_text_weights(str(f).encode(), m.ctypes.data, 8) == 0
    assert list(m[:8]) == [1, -2, 3, 0, 0, 100, 0, 300] and m[8] == -99      # double -> int truncates toward zero (B8)
    assert lib.mbn_read_text_weights(str(f).encode(), m.ctypes.data, 9) == pkg.EFORMAT   # fewer tokens than asked
    assert lib.mbn_read_text_weights(b"/nonexistent/w.txt", m.ctypes.data, 1) == pkg.EIO
    x = np.zeros(3, np.float32)
    assert lib.mbn_read_text_weights_f32(str(f).encode(), x.ctypes.data, 3, 1) == 0
    assert np.allclose(x, [-2.7, 3, 0.999])
    # the reference symbol itself: same name/signature, reads ./weights_c.txt, every call from the start of the file
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        a, b = np.zeros(3, np.int32), np.zeros(5, np.int32)
        lib.readSquezeNetKernel(a.ctypes.data, 3)
        lib.readSquezeNetKernel(b.ctypes.data, 5)
        assert list(a) == [1, -2, 3] and list(b) == [1, -2, 3, 0, 0]       # same prefix for every "layer"
        os.remove("weights_c.txt")
        lib.readSquezeNetKernel(a.ctypes.data, 3)                          # missing file: no crash, buffer untouched
        assert list(a) == [1, -2, 3]
    finally:
        os.chdir(cwd)


def test_image_loaders(pkg, tmp_path):
    lib = pkg.host_lib()
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    ppm = str(tmp_path / "Cat_Image0.ppm").encode()
    assert lib.mbn_write_ppm(ppm, rgb.ctypes.data, 224, 224) == 0
    out = np.zeros_like(rgb)
    w, h = C.c_int(), C.c_int()
    assert lib.mbn_read_ppm(ppm, out.ctypes.data, C.byref(w), C.byref(h), 224 * 224) == 0
    assert (w.value, h.value) == (224, 224) and np.array_equal(out, rgb)
    assert lib.mbn_read_ppm(ppm, out.ctypes.data, C.byref(w), C.byref(h), 100) == pkg.EINVAL   # too small a buffer
    # comments + odd whitespace in the header
    p2 = tmp_path / "c.ppm"
    p2.write_bytes(b"P6 # made by hand\n2\t1\n# another\n255\n" + bytes([1, 2, 3, 4, 5, 6]))
    small = np.zeros(6, np.uint8)
    assert lib.mbn_read_ppm(str(p2).encode(), small.ctypes.data, C.byref(w), C.byref(h), 4) == 0
    assert list(small) == [1, 2, 3, 4, 5, 6] and (w.value, h.value) == (2, 1)
    (tmp_path / "bad.ppm").write_bytes(b"P5\n2 2\n255\n....")
    assert lib.mbn_read_ppm(str(tmp_path / "bad.ppm").encode(), small.ctypes.data, C.byref(w), C.byref(h), 4) == pkg.EFORMAT
    # decode_image keeps the reference's behaviour: raw bytes from offset 0, header included (B14)
    frame = np.zeros(224 * 224 * 3, np.uint8)
    assert lib.decode_image(frame.ctypes.data, ppm) == 0
    header = b"P6\n224 224\n255\n"
    assert bytes(frame[:len(header)]) == header
    assert np.array_equal(frame[len(header):], rgb.ravel()[:frame.size - len(header)])
    assert lib.decode_image(frame.ctypes.data, b"/nonexistent.ppm") == pkg.EIO
    # RGB de-interleave (MobileNet.c:218-238)
    r, g, b = (np.zeros(224 * 224, np.uint8) for _ in range(3))
    assert lib.mbn_split_rgb(rgb.ctypes.data, 224 * 224, r.ctypes.data, g.ctypes.data, b.ctypes.data) == 0
    assert np.array_equal(r, rgb[..., 0].ravel()) and np.array_equal(b, rgb[..., 2].ravel())


def test_host_softmax_argmax_u8(pkg, orc):
    lib = pkg.host_lib()
    logits = np.random.default_rng(3).integers(0, 40, 1000, dtype=np.uint8)
    logits[417] = 60
    probs = np.zeros(1000)
    loc, mx = C.c_int(), C.c_double()
    assert lib.mbn_softmax_argmax_u8(logits.ctypes.data, 1000, probs.ctypes.data, C.byref(loc), C.byref(mx)) == 0
    # three independent formulations: numpy (float64 exp / sum), the oracle's restatement of MobileNet.c:2771-2792, and the
    # host's per-distinct-value form; same quotients up to the summation order numpy chooses
    e = np.exp(logits.astype(np.float64))
    want = e / e.sum()
    assert np.allclose(probs, want, rtol=1e-12, atol=0) and loc.value == int(np.argmax(want)) + 1 == 418
    op, oloc, omx = orc.softmax_argmax_u8(logits)
    assert loc.value == oloc and mx.value == omx and np.array_equal(probs, op)   # 1-based like MobileNet.c:2788
    ties = np.zeros(10, np.uint8); ties[[3, 7]] = 5                               # first maximum wins (strict > at :2786)
    tp, tl = np.zeros(10), C.c_int()
    assert lib.mbn_softmax_argmax_u8(ties.ctypes.data, 10, tp.ctypes.data, C.byref(tl), C.byref(mx)) == 0 and tl.value == 4
    assert abs(probs.sum() - 1) < 1e-12
    logits[:] = 0
    logits[0] = 9
    lib.mbn_softmax_argmax_u8(logits.ctypes.data, 1000, probs.ctypes.data, C.byref(loc), C.byref(mx))
    assert loc.value == 1       # defined when class 0 wins (the reference leaves `location` uninitialised, B11)


# ----------------------------------------------------------------------------- .h5 reader / writer

def h5_get(lib, path, name):
    h = C.c_void_p()
    rc = lib.mbn_h5_open(path.encode(), C.byref(h))
    if rc:
        return rc, None
    nd, shp, p = C.c_int(), (C.c_int64 * 8)(), C.POINTER(C.c_float)()
    rc = lib.mbn_h5_get(h, name.encode(), C.byref(nd), shp, C.byref(p))
    a = None
    if rc == 0:
        shape = tuple(shp[i] for i in range(nd.value))
        n = int(np.prod(shape)) if shape else 1
        a = np.ctypeslib.as_array(p, shape=(n,)).reshape(shape).copy()
    lib.mbn_h5_close(h)
    return rc, a


def h5_list(lib, path):
    h = C.c_void_p()
    assert lib.mbn_h5_open(path.encode(), C.byref(h)) == 0
    seen = {}
    CB = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p)

    def cb(p, nd, shp, _):
        seen[p.decode()] = tuple(shp[i] for i in range(nd))
        return 0
    fn = CB(cb)
    assert lib.mbn_h5_visit(h, C.cast(fn, C.c_void_p), None) == 0
    lib.mbn_h5_close(h)
    return seen


@pytest.mark.parametrize("stem", ["keras_like_earliest", "latest_format"])
def test_h5_reader_on_libhdf5_written_files(pkg, stem):
    lib = pkg.host_lib()
    exp = json.load(open(os.path.join(GOLD, stem + ".json")))
    for name, v in exp.items():
        rc, a = h5_get(lib, os.path.join(GOLD, stem + ".h5"), name)
        assert rc == 0, (name, rc)
        assert list(a.shape) == v["shape"]
        assert np.array_equal(a.ravel(), np.float32(v["data"])), name


def test_h5_reader_visits_every_keras_dataset(pkg):
    lib = pkg.host_lib()
    sums = json.load(open(os.path.join(GOLD, "keras_like_earliest_sums.json")))
    seen = h5_list(lib, os.path.join(GOLD, "keras_like_earliest.h5"))
    assert set(seen) == set(sums) and len(seen) == 1 + 4 + 13 * 10 + 2
    for name, (shape, total) in sums.items():
        rc, a = h5_get(lib, os.path.join(GOLD, "keras_like_earliest.h5"), name)
        assert rc == 0 and list(a.shape) == shape
        assert abs(float(np.float64(a).sum()) - total) <= 1e-9 * max(1.0, abs(total)), name


def test_h5_reader_errors(pkg, tmp_path):
    lib = pkg.host_lib()
    p = os.path.join(GOLD, "unsupported.h5")
    assert h5_get(lib, p, "/chunked")[0] == pkg.EUNSUPPORTED
    assert h5_get(lib, p, "/f64")[0] == pkg.EUNSUPPORTED
    rc, ok = h5_get(lib, p, "/ok")
    assert rc == 0 and np.array_equal(ok, np.arange(6, dtype=np.float32).reshape(2, 3))
    assert h5_get(lib, p, "/missing")[0] == pkg.ENOTFOUND
    assert h5_get(lib, p, "/ok/deeper")[0] in (pkg.ENOTFOUND, pkg.EFORMAT)
    assert h5_get(lib, os.path.join(GOLD, "keras_like_earliest.h5"), "/conv1")[0] == pkg.ENOTFOUND   # a group
    assert h5_get(lib, "/nonexistent.h5", "/x")[0] == pkg.EIO
    junk = tmp_path / "junk.h5"
    junk.write_bytes(os.urandom(4096))
    assert h5_get(lib, str(junk), "/x")[0] == pkg.EFORMAT
    tiny = tmp_path / "tiny.h5"
    tiny.write_bytes(b"\x89HDF\r\n\x1a\n")
    assert h5_get(lib, str(tiny), "/x")[0] == pkg.EFORMAT
    # truncated real file: must fail cleanly, never read past the mapping
    data = open(os.path.join(GOLD, "keras_like_earliest.h5"), "rb").read()
    for cut in (100, 700, 5000, len(data) // 2):
        t = tmp_path / ("cut%d.h5" % cut)
        t.write_bytes(data[:cut])
        assert h5_get(lib, str(t), "/conv_preds/conv_preds/bias:0")[0] != 0 or cut > 5000


def test_h5_fuzz_never_crashes(pkg, tmp_path):
    """Bit-flip fuzz of a real file: any return code is fine, a crash is not."""
    lib = pkg.host_lib()
    data = bytearray(open(os.path.join(GOLD, "latest_format.h5"), "rb").read())
    rng = np.random.default_rng(5)
    t = tmp_path / "fuzz.h5"
    for _ in range(300):
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        t.write_bytes(bytes(d))
        for name in ("/top", "/grp/inner/vec:0", "/grp/compact"):
            h5_get(lib, str(t), name)
        h = C.c_void_p()
        if lib.mbn_h5_open(str(t).encode(), C.byref(h)) == 0:
            CB = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p)
            fn = CB(lambda *a: 0)
            lib.mbn_h5_visit(h, C.cast(fn, C.c_void_p), None)
            lib.mbn_h5_close(h)


def _crafted_v2_file(chunk_size_field, body=b""):
    """4 KB file: superblock v2 (8-byte offsets/lengths) whose root object header is a v2 OHDR with an 8-byte
    chunk-size field (flags & 3 == 3) holding `chunk_size_field`."""
    f = bytearray(4096)
    f[0:8] = b"\x89HDF\r\n\x1a\n"
    f[8], f[9], f[10], f[11] = 2, 8, 8, 0
    f[12:20] = (0).to_bytes(8, "little")                   # base address
    f[20:28] = (2 ** 64 - 1).to_bytes(8, "little")         # superblock extension: undefined
    f[28:36] = (4096).to_bytes(8, "little")                # EOF
    f[36:44] = (512).to_bytes(8, "little")                 # root group object header
    f[512:516] = b"OHDR"
    f[516], f[517] = 2, 0x03                               # version 2, flags: 8-byte chunk-0 size
    f[518:526] = int(chunk_size_field).to_bytes(8, "little")
    f[526:526 + len(body)] = body
    return bytes(f)


@pytest.mark.parametrize("chunk0", [2 ** 64 - 4, 2 ** 64 - 1, 2 ** 63, 4096, 3571])
def test_h5_crafted_chunk0_size_is_rejected(pkg, tmp_path, chunk0):
    """an 8-byte chunk-0 size taken straight from the file made `chunk0 + 4` wrap, so the bounds
    check passed with a tiny length and the message walk ran past the mapping. Every oversize value must come back
    as an error code from both entry points (run under ASan by tools/run_asan_cpu.sh)."""
    lib = pkg.host_lib()
    t = tmp_path / "crafted.h5"
    t.write_bytes(_crafted_v2_file(chunk0))
    assert h5_get(lib, str(t), "/x")[0] == pkg.EFORMAT
    w = pkg.Weights()
    assert lib.mbn_weights_from_h5(str(t).encode(), C.c_float(0.0), 224, C.byref(w)) != 0
    # a chunk that does fit (all NIL messages) is a valid empty group: the name is simply not there
    t.write_bytes(_crafted_v2_file(64))
    assert h5_get(lib, str(t), "/x")[0] == pkg.ENOTFOUND


def test_h5_never_written_empty_dataset_has_defined_shape(pkg, tmp_path):
    """a contiguous dataset that was never written (address undefined) with zero elements returned MBN_OK
    without setting ndim/shape/data. Crafted v2 header: dataspace (2, 0), float32 datatype, contiguous layout at
    H5_UNDEF."""
    lib = pkg.host_lib()
    space = bytes([1, 2, 0, 0]) + (2).to_bytes(4, "little")[:0] + bytes(4) + (2).to_bytes(8, "little") + (0).to_bytes(8, "little")
    # v1 dataspace message: version 1, rank 2, flags 0, reserved 5 bytes, then the dims
    space = bytes([1, 2, 0, 0, 0, 0, 0, 0]) + (2).to_bytes(8, "little") + (0).to_bytes(8, "little")
    dtype = bytes([0x11, 0x20, 0x1F, 0x00]) + (4).to_bytes(4, "little") + bytes([0, 0, 32, 0, 23, 8, 0, 23]) + (127).to_bytes(4, "little")
    layout = bytes([3, 1]) + (2 ** 64 - 1).to_bytes(8, "little") + (0).to_bytes(8, "little")

    def msg(t, body):
        return bytes([t]) + len(body).to_bytes(2, "little") + bytes([0]) + body
    body = msg(0x01, space) + msg(0x03, dtype) + msg(0x08, layout)
    t = tmp_path / "empty.h5"
    t.write_bytes(_crafted_v2_file(len(body), body))
    h = C.c_void_p()
    assert lib.mbn_h5_open(str(t).encode(), C.byref(h)) == 0
    ndim, shape, data = C.c_int(-7), (C.c_int64 * 8)(*([-7] * 8)), C.POINTER(C.c_float)()
    rc = lib.mbn_h5_get(h, b"/", C.byref(ndim), shape, C.byref(data))
    lib.mbn_h5_close(h)
    if rc == 0:                     # accepted as an empty dataset: everything must be defined
        assert ndim.value == 2 and list(shape)[:2] == [2, 0] and not data
    else:                           # or refused — but never MBN_OK with garbage outputs
        assert rc in (pkg.EFORMAT, pkg.EUNSUPPORTED, pkg.ENOTFOUND)


def _write(lib, path, items):
    w = C.c_void_p()
    assert lib.mbn_h5_create(path.encode(), C.byref(w)) == 0
    for name, arr in items:
        a = np.ascontiguousarray(arr, np.float32)
        shp = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        assert lib.mbn_h5_put(w, name.encode(), a.ndim, shp, a.ctypes.data) == 0, name
    assert lib.mbn_h5_finish(w) == 0


def test_h5_writer_roundtrip_and_libhdf5_reads_it(pkg, tmp_path):
    lib = pkg.host_lib()
    rng = np.random.default_rng(8)
    items = [("/a/a/kernel:0", rng.normal(size=(3, 3, 3, 8))), ("/a/a/bias:0", rng.normal(size=(8,))),
             ("/b/deep/er/x", rng.normal(size=(5, 7))), ("/top", rng.normal(size=(2,)))]
    items += [("/many/d%03d" % i, rng.normal(size=(i % 5 + 1,))) for i in range(70)]     # > default SNOD capacity
    path = str(tmp_path / "w.h5")
    _write(lib, path, items)
    seen = h5_list(lib, path)
    assert set(seen) == {n for n, _ in items}
    for name, arr in items:
        rc, a = h5_get(lib, path, name)
        assert rc == 0 and np.array_equal(a, np.float32(arr)), name
    # duplicate / malformed names are refused
    w = C.c_void_p()
    assert lib.mbn_h5_create(str(tmp_path / "dup.h5").encode(), C.byref(w)) == 0
    one = np.ones(1, np.float32)
    shp = (C.c_int64 * 1)(1)
    assert lib.mbn_h5_put(w, b"/x", 1, shp, one.ctypes.data) == 0
    assert lib.mbn_h5_put(w, b"/x", 1, shp, one.ctypes.data) == pkg.EINVAL
    assert lib.mbn_h5_put(w, b"/x/y", 1, shp, one.ctypes.data) == pkg.EINVAL
    assert lib.mbn_h5_put(w, b"/", 1, shp, one.ctypes.data) == pkg.EINVAL
    assert lib.mbn_h5_finish(w) == 0
    if not os.path.exists(LIBHDF5):
        pytest.skip("libhdf5 not present: cross-read by the real library skipped")
    h5 = C.CDLL(LIBHDF5)
    hid = C.c_int64
    h5.H5Fopen.restype = h5.H5Dopen2.restype = h5.H5Dget_space.restype = hid
    h5.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    h5.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    h5.H5Dget_space.argtypes = [hid]
    h5.H5Sget_simple_extent_npoints.argtypes = [hid]
    h5.H5Sget_simple_extent_npoints.restype = C.c_int64
    h5.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    h5.H5Dclose.argtypes = h5.H5Fclose.argtypes = h5.H5Sclose.argtypes = [hid]
    h5.H5open()
    nf32 = hid.in_dll(h5, "H5T_NATIVE_FLOAT_g").value
    f = h5.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, "libhdf5 refuses the file written by mbn_h5_create"
    for name, arr in items:
        d = h5.H5Dopen2(f, name.encode(), 0)
        assert d >= 0, name
        sp = h5.H5Dget_space(d)
        n = h5.H5Sget_simple_extent_npoints(sp)
        assert n == np.asarray(arr).size
        out = np.zeros(n, np.float32)
        assert h5.H5Dread(d, nf32, 0, 0, 0, out.ctypes.data) >= 0
        assert np.array_equal(out, np.float32(arr).ravel()), name
        h5.H5Sclose(sp)
        h5.H5Dclose(d)
    h5.H5Fclose(f)


# ----------------------------------------------------------------------------- weights: .h5 -> folded, packed blob

def _expected_blob(lib, path, plan):
    """BN folding + repacking restated in numpy from the datasets of the file."""
    blob = np.zeros(plan.blob_floats, np.float32)

    def get(name):
        rc, a = h5_get(lib, path, name)
        assert rc == 0, name
        return a

    def fold(g, l):
        ga, be, mu, va = (get("/%s/%s/%s:0" % (g, g, k)).astype(np.float64)
                          for k in ("gamma", "beta", "moving_mean", "moving_variance"))
        s = ga / np.sqrt(va + 1e-3)
        blob[l.scale_offset:l.scale_offset + l.out_ch] = s
        blob[l.shift_offset:l.shift_offset + l.out_ch] = be - mu * s

    dw = pw = 0
    for i in range(plan.n_layers):
        l = plan.layer[i]
        if l.kind == 1:
            blob[l.w_offset:l.w_offset + l.w_count] = get("/conv1/conv1/kernel:0").ravel()
            fold("conv1_bn", l)
        elif l.kind == 2:
            dw += 1
            blob[l.w_offset:l.w_offset + l.w_count] = get("/conv_dw_%d/conv_dw_%d/depthwise_kernel:0" % (dw, dw)).ravel()
            fold("conv_dw_%d_bn" % dw, l)
        elif l.kind == 3:
            pw += 1
            k = get("/conv_pw_%d/conv_pw_%d/kernel:0" % (pw, pw))[0, 0]
            blob[l.w_offset:l.w_offset + l.w_count] = k.T.ravel()          # [Cin][Cout] -> [Cout][Cin]
            fold("conv_pw_%d_bn" % pw, l)
        elif l.kind == 5:
            k = get("/conv_preds/conv_preds/kernel:0")[0, 0]
            blob[l.w_offset:l.w_offset + l.w_count] = k.T.ravel()
            blob[l.shift_offset:l.shift_offset + l.out_ch] = get("/conv_preds/conv_preds/bias:0")
    return blob


def test_weights_from_keras_like_h5(pkg):
    lib = pkg.host_lib()
    path = os.path.join(GOLD, "keras_like_earliest.h5")
    hw = pkg.HostWeights(path, alpha=0.0, res=64)           # alpha inferred from conv1's kernel shape
    assert abs(hw.plan.alpha - 0.125) < 1e-6 and hw.plan.classes == 10 and hw.plan.res == 64
    want = _expected_blob(lib, path, hw.plan)
    assert np.allclose(hw.blob, want, rtol=1e-6, atol=1e-7)
    assert np.array_equal(hw.blob[hw.plan.layer[0].w_offset:][:27 * 4], want[hw.plan.layer[0].w_offset:][:27 * 4])
    hw.free()
    w = pkg.Weights()
    assert lib.mbn_weights_from_h5(path.encode(), 1.0, 224, C.byref(w)) == pkg.ESHAPE     # wrong alpha for this file
    assert lib.mbn_weights_from_h5(b"/nonexistent.h5", 1.0, 224, C.byref(w)) == pkg.EIO
    assert lib.mbn_weights_from_h5(os.path.join(GOLD, "latest_format.h5").encode(), 1.0, 224, C.byref(w)) == pkg.ENOTFOUND


def test_synthetic_weights_go_through_the_real_loader(pkg, tmp_path):
    lib = pkg.host_lib()
    path = str(tmp_path / "syn.h5")
    pkg.synthetic_h5(path, alpha=0.25, classes=20, seed=42)
    hw = pkg.HostWeights(path, res=96)
    assert (hw.plan.classes, hw.plan.layer[0].out_ch) == (20, 8)
    want = _expected_blob(lib, path, hw.plan)
    assert np.allclose(hw.blob, want, rtol=1e-6, atol=1e-7)
    # deterministic in the seed, different across seeds, sane statistics (He-normal kernels, gamma in [.5,1.5])
    p2 = str(tmp_path / "syn2.h5")
    pkg.synthetic_h5(p2, alpha=0.25, classes=20, seed=42)
    assert open(path, "rb").read() == open(p2, "rb").read()
    pkg.synthetic_h5(p2, alpha=0.25, classes=20, seed=43)
    assert open(path, "rb").read() != open(p2, "rb").read()
    _, k = h5_get(lib, path, "/conv_pw_13/conv_pw_13/kernel:0")
    assert abs(k.std() - (2.0 / 256) ** 0.5) < 0.01 and abs(k.mean()) < 0.01
    _, g = h5_get(lib, path, "/conv_pw_13_bn/conv_pw_13_bn/gamma:0")
    assert 0.5 <= g.min() and g.max() <= 1.5
    hw.free()


def test_c_host_weight_tooling_cli(pkg, tmp_path):
    """`mobilenet --inspect / --convert`: the weight-format tooling of the C host runs without a GPU."""
    import subprocess
    exe = os.path.join(pkg.PKG_DIR, "mobilenet")
    h5 = os.path.join(GOLD, "keras_like_earliest.h5")
    r = subprocess.run([exe, "--inspect", h5], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "/conv_preds/conv_preds/bias:0" in r.stdout and "(3, 3, 3, 4)" in r.stdout
    assert r.stdout.strip().endswith("61506 float32 parameters")       # raw Keras count at alpha 0.125, 10 classes
    out = str(tmp_path / "blob.txt")
    r = subprocess.run([exe, "--convert", h5, out], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    hw = pkg.HostWeights(h5, res=224)
    vals = np.loadtxt(out, dtype=np.float32, comments="#")
    assert vals.shape == (hw.plan.blob_floats,) and np.array_equal(vals, hw.blob)    # %.9g round-trips fp32 exactly
    # the text form is what the checked text loader reads back
    back = np.zeros(100, np.float32)
    lines = [l for l in open(out) if not l.startswith("#")]
    (tmp_path / "w.txt").write_text("".join(lines))
    assert pkg.host_lib().mbn_read_text_weights_f32(str(tmp_path / "w.txt").encode(), back.ctypes.data, 100, 27 * 4) == 0
    assert np.array_equal(back, hw.blob[27 * 4:27 * 4 + 100])
    hw.free()
    assert subprocess.run([exe, "--inspect", "/nonexistent.h5"], capture_output=True).returncode == 1


def test_udiv_magic_formula():
    """The multiply-high division the fused-block kernels use for (n, y, x) of a flattened pixel index
    (csrc/mbn_internal.h: mbn_udiv_magic): floor(v / d) == ((v * M) >> 32) >> s for every v < 2^31, with
    l = ceil(log2 d), M = ceil(2^(31+l) / d), s = l - 1. Checked here on the formula itself (pure integers): every divisor
    up to 4096 against edge values and random values."""
    import random
    rnd = random.Random(7)
    for d in list(range(2, 4097)) + [5000, 65535, 65536, 65537, (1 << 20) + 3]:
        l = (d - 1).bit_length()
        m = -(-(1 << (31 + l)) // d)
        assert m < (1 << 32), d
        s = l - 1
        vals = [0, 1, d - 1, d, d + 1, 2 * d - 1, (1 << 31) - 1, (1 << 31) - d, ((1 << 31) - 1) // d * d, ((1 << 31) - 1) // d * d - 1]
        vals += [rnd.randrange(1 << 31) for _ in range(64)]
        for v in vals:
            if 0 <= v < (1 << 31):
                assert ((v * m) >> 32) >> s == v // d, (v, d)


def test_pw_gemm_counted_waits_match_the_isa():
    """`s_waitcnt vmcnt(NSTF); s_barrier` at the top of a pw_gemm tile is right only if the previous tile's fast epilogue is
    EXACTLY NSTF buffer stores, all behind the next tile's first LDS-DMA, in the instruction stream the compiler produced. The source
    pins that with sched_barrier(0); tools/check_counted_waits.py reads the gfx950 ISA of csrc/mbn_f32_pw.hip (hipcc -S, ~10 s, no GPU
    needed) and fails if a compiler change merges, splits or interleaves them."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_counted_waits.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "kernels with a counted wait checked" in r.stdout


def _fake_bench_record():
    """run_one's record shape (bench.py), with the sizes of the real headline run: 22 launches, 26 unfused layers, three alternative configs."""
    st = lambda ms: {"launches": 8, "ms": ms, "GBps": 3501.03, "TFLOPs": 128.312345, "frac_hbm": 0.437612, "frac_mfma": 0.815712, "layers": "x" * 90}
    stages = {k: st(0.1 * (i + 1)) for i, k in enumerate(("stem_fused", "block_fused", "depthwise", "pointwise", "pool", "fc"))}
    layers = [{"layers": [i + 1], "stage": "pointwise", "ms": 0.17, "GBps": 1.0, "TFLOPs": 2.0} for i in range(26)]
    roof = {"kernel": "pw_gemm<float> (8 pointwise 1x1 conv launches per step)" + " padding" * 40, "bound": "mfma", "achieved": 128.3123456, "peak": 157.3,
            "unit": "TFLOP/s", "frac": 0.81571234, "traffic": 205277161.0, "traffic_source": {"git_sha": "x" * 12, "command": "y" * 300, "date": "2026-10-04"},
            "avg_launch_ms": 0.1794212, "algorithmic_flops_per_launch": 23018340352.0, "algorithmic_bytes_per_launch": 171704320.0, "launches_per_step": 8,
            "held_clock_ghz": 2.213, "frac_at_held_clock": 0.8846, "held_clock_launches": 40, "dw_x13_frac_hbm": 0.69, "pw_x13_frac_mfma": 0.72,
            "blocks_ms": 0.93, "stem_ms": 0.28, "package_power_w": 1349.0, "power_cap_w": 1400.0}
    base = {"metric": "images/sec MobileNet-V1 1.0x224 fp32, batch 256; per-stage HBM GB/s vs roofline", "value": 89059.123456789, "unit": "images/sec",
            "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 2.87451234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MobileNet-V1 1x224 fp32, batch 256 per GPU, 29 layers, 1000 classes (BASELINE.json configs[2])", "global_batch": 256,
                       "per_gpu_batch": 256, "parallelism": "batch-sharded x1, weights broadcast once over RCCL", "streams": 2, "streams_note": "z" * 200,
                       "device": "AMD Radeon Graphics (gfx950:sramecc+:xnack-)"},
            "roofline": roof, "stages": stages, "layers": layers, "sum_kernel_ms": 2.9, "profiled_steps": 5,
            "event_overhead_us": {"empty_pair": 4.6, "how": "h" * 300}, "h2d_ms_per_batch": 55.0,
            "step_ms": {"median": 2.8, "p10": 2.7, "p90": 2.9, "n": 20, "how": "h" * 100},
            "unfused_stages": {"note": "n" * 200, "stages": {"depthwise": dict(st(0.96), launches=13), "pointwise": dict(st(2.4), launches=13), "conv1": dict(st(0.2), launches=1)},
                               "layers": layers, "sum_kernel_ms": 3.6},
            "pw_emul_alt": {"pw_emul": 6, "value": 115000.0, "what": "w" * 600, "parity_check": {"ok": True}},
            "cpu_baseline": {"value": 46.655550652782544, "unit": "images/sec", "cores": 16, "kind": "port", "sample": "s" * 250,
                             "variants": {"threads%d_batch%d" % (t, b): {"ms_per_image": 1.0, "images_per_sec": 2.0} for t in (1, 16) for b in (1, 8)}, "variants_how": "v" * 80},
            "parity_check": {"images": 64, "max_rel_err": 4.781746733827894e-07, "tolerance": 0.001, "ok": True, "argmax_agree": 64, "against": "a" * 120}}
    base["power"] = {"package_w": 1389.0, "cap_w": 1400.0, "sclk_mhz": 2296.0, "samples": 22, "how": "h" * 100}
    alt = {k: base[k] for k in ("power", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "stages", "sum_kernel_ms", "profiled_steps",
                                "event_overhead_us", "step_ms", "parity_check")}
    base["configs_alt"] = {"bf16_1.0x224_b512": alt, "bf16_0.5x160_b512": alt, "f32_1.0x224_b1": alt}
    return base


def test_bench_line_is_compact_parseable_and_keeps_the_contract(tmp_path):
    """round 4's 20 KB line did not fit the driver's 8 KB stdout tail (BENCH_r04 parsed: null). The printer must turn
    run_one's full record into ONE line under 4 KB that json.loads parses, with the contract's keys, scalar-only `roofline` and
    `cpu_baseline`, flat per-stage triples, and the side file named; a record with 8 ranks still fits."""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    rec = _fake_bench_record()
    assert len(json.dumps(rec)) > 15000                                  # the record itself is the size that broke round 4
    line = bench.compact_line(rec, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity_check", "stages_frac", "configs_alt", "full_record"):
        assert k in out, k
    assert out["metric"] == rec["metric"] and out["steps"] == 20 and out["warmup"] == 5 and out["vs_baseline"] is None
    assert abs(out["value"] - rec["value"]) < 0.1 and abs(out["ms_per_step"] - rec["ms_per_step"]) < 1e-3
    assert all(not isinstance(v, (dict, list)) for v in out["roofline"].values())
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_flops_per_launch", "held_clock_ghz",
              "frac_at_held_clock", "dw_x13_frac_hbm", "pw_x13_frac_mfma"):
        assert k in out["roofline"], k
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-3
    assert set(out["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and out["cpu_baseline"]["kind"] == "port"
    assert out["stages_frac"]["pointwise"] == [0.4, 0.4376, 0.8157] and out["stages_frac"]["unfused_depthwise_x13"][0] == 0.96
    assert out["configs_alt"]["bf16_1.0x224_b512"] == [89059.1, 0.8157, True, 1389.0] and out["roofline"]["package_power_w"] == 1349.0
    assert "layers" not in out and "unfused_stages" not in out and "config" in out and "streams_note" not in out["config"]
    # N = 8: the per-rank evidence rides along and the line still fits
    rec8 = dict(rec, n_gpus=8, ranks=[[r, r, "0000:%02x:00.0" % (5 + r), 256, 0.0574123, 1387.0, 2290.0, 2.281] for r in range(8)], ranks_cols="c" * 150,
                collective_world_size=8, backend="nccl")
    line8 = bench.compact_line(rec8, "gpurun_out/bench_full.json")
    out8 = json.loads(line8)
    assert len(line8) < bench.LINE_LIMIT and len(out8["ranks"]) == 8 and out8["collective_world_size"] == 8
    # an oversized record degrades by dropping optional tables, never by printing an unparseable or oversized line
    rec["stages"] = {"s%d" % i: rec["stages"]["pointwise"] for i in range(150)}
    big = bench.compact_line(rec, "x.json")
    assert len(big) < bench.LINE_LIMIT and "stages_frac" in json.loads(big)["dropped"]
    # the side file round-trips
    args = bench.parse_args(["--record", str(tmp_path / "full.json")])
    path = bench.write_record(args, _fake_bench_record())
    assert json.load(open(path))["layers"][0]["stage"] == "pointwise"


def test_bench_power_sampler_parses_rocm_smi_json(monkeypatch):
    """bench.py's sample_power (round 5: package power and core clock while the step runs back to back, AFTER the timed region) against a canned rocm-smi: the JSON
    shapes of `--showpower --showclocks --json` and `--showmaxpower --json` as the MI355X boxes print them; a missing rocm-smi gives None, never an exception."""
    import importlib
    import shutil
    import subprocess
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    calls = {"n": 0}

    def fake_run(cmd, **kw):
        calls["n"] += 1
        if "--showbus" in cmd:
            out = {"card0": {"PCI Bus": "0000:05:00.0"}, "card1": {"PCI Bus": "0000:26:00.0"}}
        elif "--showmaxpower" in cmd:
            out = {"card0": {"Max Graphics Package Power (W)": "1400.0"}, "card1": {"Max Graphics Package Power (W)": "1000.0"}}
        else:
            out = {"card0": {"fclk clock speed:": "(1250Mhz)", "mclk clock speed:": "(2000Mhz)", "sclk clock speed:": "(%dMhz)" % (2280 + calls["n"] % 3), "sclk clock level:": "1",
                             "Current Socket Graphics Package Power (W)": "%d.0" % (1340 + calls["n"] % 5)},
                   "card1": {"sclk clock speed:": "(132Mhz)", "Current Socket Graphics Package Power (W)": "140.0"}}     # an idle neighbour
        return types.SimpleNamespace(stdout=json.dumps(out), returncode=0)
    monkeypatch.setattr(shutil, "which", lambda name: "/opt/rocm/bin/rocm-smi")
    monkeypatch.setattr(subprocess, "run", fake_run)
    steps = {"n": 0}
    p = bench.sample_power(lambda: steps.__setitem__("n", steps["n"] + 1), lambda: None, seconds=0.9)
    assert p is not None and p["cap_w"] == 1400.0 and 1340 <= p["package_w"] <= 1344 and 2280 <= p["sclk_mhz"] <= 2282 and p["samples"] >= 3
    assert steps["n"] > 0 and p["ms_per_step_while_sampling"] > 0
    # round 6: the card is chosen by the PCI bus id of the GPU the context holds, not "the first one"; an id rocm-smi does not list gives None
    p1 = bench.sample_power(lambda: None, lambda: None, seconds=0.6, bus_id="0000:26:00.0")
    assert p1 is not None and p1["card"] == "card1" and p1["package_w"] == 140.0 and p1["cap_w"] == 1000.0 and p1["sclk_mhz"] == 132.0
    p0 = bench.sample_power(lambda: None, lambda: None, seconds=0.6, bus_id="0000:05:00.0")
    assert p0 is not None and p0["card"] == "card0" and p0["cap_w"] == 1400.0
    assert bench.sample_power(lambda: None, lambda: None, seconds=0.1, bus_id="0000:99:00.0") is None
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert bench.sample_power(lambda: None, lambda: None, seconds=0.1) is None
