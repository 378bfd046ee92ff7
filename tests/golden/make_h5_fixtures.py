#!/usr/bin/env python3
"""Writes the HDF5 fixtures under tests/golden/ with the REAL libhdf5 (ctypes on /opt/conda/lib/libhdf5.so, 1.10.6;
h5py/keras are not installed in the build image), so the from-scratch reader in host/mbn_h5.c is validated against
files it did not write. Run once in the build container; the outputs are committed (small).

  keras_like_earliest.h5  Keras-applications MobileNet-V1 layout, alpha = 0.125, 10 classes,
                          default ("earliest") file format: superblock v0, symbol-table groups, v1 object headers,
                          `weight_names`/`layer_names` attributes like Keras writes them.
  latest_format.h5        libver bounds (latest, latest): superblock v3, v2 object headers, compact link-message groups,
                          one compact-layout dataset.
  unsupported.h5          a chunked dataset and a float64 dataset (reader must answer MBN_EUNSUPPORTED).
  *.json                  the expected dataset values (seeded numpy), so tests do not need libhdf5.
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
h5 = C.CDLL("/opt/conda/lib/libhdf5.so")
hid = C.c_int64
for fn in ("H5Fcreate", "H5Gcreate2", "H5Screate_simple", "H5Dcreate2", "H5Pcreate", "H5Acreate2", "H5Tcopy",
           "H5Screate"):
    getattr(h5, fn).restype = hid
h5.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
h5.H5Gcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid]
h5.H5Screate_simple.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
h5.H5Dcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid, hid]
h5.H5Dwrite.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
h5.H5Pcreate.argtypes = [hid]
h5.H5Pset_libver_bounds.argtypes = [hid, C.c_int, C.c_int]
h5.H5Pset_layout.argtypes = [hid, C.c_int]
h5.H5Pset_chunk.argtypes = [hid, C.c_int, C.POINTER(C.c_uint64)]
h5.H5Acreate2.argtypes = [hid, C.c_char_p, hid, hid, hid, hid]
h5.H5Awrite.argtypes = [hid, hid, C.c_void_p]
h5.H5Tcopy.argtypes = [hid]
h5.H5Tset_size.argtypes = [hid, C.c_size_t]
for fn in ("H5Fclose", "H5Gclose", "H5Sclose", "H5Dclose", "H5Pclose", "H5Aclose", "H5Tclose"):
    getattr(h5, fn).argtypes = [hid]
assert h5.H5open() >= 0


def g(name):
    return hid.in_dll(h5, name).value


F32, F64, NF32, NF64, CS1 = g("H5T_IEEE_F32LE_g"), g("H5T_IEEE_F64LE_g"), g("H5T_NATIVE_FLOAT_g"), \
    g("H5T_NATIVE_DOUBLE_g"), g("H5T_C_S1_g")
FAPL_CLS, DCPL_CLS = g("H5P_CLS_FILE_ACCESS_ID_g"), g("H5P_CLS_DATASET_CREATE_ID_g")


def dims(shape):
    return (C.c_uint64 * len(shape))(*shape)


def put(loc, name, arr, dcpl=0, ftype=F32, mtype=NF32):
    arr = np.ascontiguousarray(arr)
    sp = h5.H5Screate_simple(arr.ndim, dims(arr.shape), None)
    d = h5.H5Dcreate2(loc, name.encode(), ftype, sp, 0, dcpl, 0)
    assert d >= 0, name
    assert h5.H5Dwrite(d, mtype, 0, 0, 0, arr.ctypes.data) >= 0
    h5.H5Dclose(d)
    h5.H5Sclose(sp)


def attr_strings(loc, name, strings):
    n = max(len(s) for s in strings) + 1
    t = h5.H5Tcopy(CS1)
    h5.H5Tset_size(t, n)
    sp = h5.H5Screate_simple(1, dims((len(strings),)), None)
    a = h5.H5Acreate2(loc, name.encode(), t, sp, 0, 0)
    buf = b"".join(s.encode().ljust(n, b"\0") for s in strings)
    assert h5.H5Awrite(a, t, buf) >= 0
    h5.H5Aclose(a)
    h5.H5Sclose(sp)
    h5.H5Tclose(t)


def keras_like(path, alpha=0.125, classes=10, seed=1234):
    rng = np.random.default_rng(seed)
    width = [32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024]
    ch = [int(w * alpha) for w in width]
    f = h5.H5Fcreate(path.encode(), 2, 0, 0)
    expect = {}
    layer_names = []

    def layer(gname, weights):
        layer_names.append(gname)
        g1 = h5.H5Gcreate2(f, gname.encode(), 0, 0, 0)
        attr_strings(g1, "weight_names", ["%s/%s" % (gname, k) for k, _ in weights])
        g2 = h5.H5Gcreate2(g1, gname.encode(), 0, 0, 0)
        for k, a in weights:
            put(g2, k, a)
            expect["/%s/%s/%s" % (gname, gname, k)] = a
        h5.H5Gclose(g2)
        h5.H5Gclose(g1)

    def bn(gname, c):
        layer(gname, [("gamma:0", rng.uniform(0.5, 1.5, c).astype(np.float32)),
                      ("beta:0", rng.normal(0, 0.1, c).astype(np.float32)),
                      ("moving_mean:0", rng.normal(0, 0.1, c).astype(np.float32)),
                      ("moving_variance:0", rng.uniform(0.5, 1.5, c).astype(np.float32))])

    layer("conv1", [("kernel:0", rng.normal(0, 0.27, (3, 3, 3, ch[0])).astype(np.float32))])
    bn("conv1_bn", ch[0])
    for i in range(1, 14):
        cin, cout = ch[i - 1], ch[i]
        layer("conv_dw_%d" % i, [("depthwise_kernel:0", rng.normal(0, 0.47, (3, 3, cin, 1)).astype(np.float32))])
        bn("conv_dw_%d_bn" % i, cin)
        layer("conv_pw_%d" % i, [("kernel:0", rng.normal(0, (2.0 / cin) ** 0.5, (1, 1, cin, cout)).astype(np.float32))])
        bn("conv_pw_%d_bn" % i, cout)
    layer("conv_preds", [("kernel:0", rng.normal(0, 0.1, (1, 1, ch[13], classes)).astype(np.float32)),
                         ("bias:0", rng.normal(0, 0.1, classes).astype(np.float32))])
    attr_strings(f, "layer_names", layer_names)
    attr_strings(f, "backend", ["tensorflow"])
    attr_strings(f, "keras_version", ["2.2.4"])
    h5.H5Fclose(f)
    return expect


def latest_format(path):
    rng = np.random.default_rng(99)
    fapl = h5.H5Pcreate(FAPL_CLS)
    assert h5.H5Pset_libver_bounds(fapl, 2, 2) >= 0      # H5F_LIBVER_V110 == LATEST in 1.10
    f = h5.H5Fcreate(path.encode(), 2, 0, fapl)
    expect = {}
    a = rng.normal(0, 1, (4, 5)).astype(np.float32)
    put(f, "top", a)
    expect["/top"] = a
    g1 = h5.H5Gcreate2(f, b"grp", 0, 0, 0)
    g2 = h5.H5Gcreate2(g1, b"inner", 0, 0, 0)
    b = rng.normal(0, 1, (7,)).astype(np.float32)
    put(g2, "vec:0", b)
    expect["/grp/inner/vec:0"] = b
    dcpl = h5.H5Pcreate(DCPL_CLS)
    h5.H5Pset_layout(dcpl, 0)                             # H5D_COMPACT
    c = rng.normal(0, 1, (2, 3)).astype(np.float32)
    put(g1, "compact", c, dcpl=dcpl)
    expect["/grp/compact"] = c
    h5.H5Pclose(dcpl)
    h5.H5Gclose(g2)
    h5.H5Gclose(g1)
    h5.H5Fclose(f)
    h5.H5Pclose(fapl)
    return expect


def unsupported(path):
    f = h5.H5Fcreate(path.encode(), 2, 0, 0)
    dcpl = h5.H5Pcreate(DCPL_CLS)
    h5.H5Pset_layout(dcpl, 2)                             # H5D_CHUNKED
    h5.H5Pset_chunk(dcpl, 2, dims((2, 2)))
    put(f, "chunked", np.ones((4, 4), np.float32), dcpl=dcpl)
    h5.H5Pclose(dcpl)
    put(f, "f64", np.ones((3,), np.float64), ftype=F64, mtype=NF64)
    put(f, "ok", np.arange(6, dtype=np.float32).reshape(2, 3))
    h5.H5Fclose(f)


def dump(expect, path):
    json.dump({k: {"shape": list(v.shape), "data": [float(x) for x in v.ravel()]} for k, v in expect.items()},
              open(path, "w"))


if __name__ == "__main__":
    e = keras_like(os.path.join(HERE, "keras_like_earliest.h5"))
    # the Keras-like file is checked by checksum + a few datasets (its json would be large): keep 6 datasets
    keep = ["/conv1/conv1/kernel:0", "/conv1_bn/conv1_bn/gamma:0", "/conv_dw_7/conv_dw_7/depthwise_kernel:0",
            "/conv_pw_13_bn/conv_pw_13_bn/moving_variance:0", "/conv_preds/conv_preds/bias:0", "/conv_pw_2/conv_pw_2/kernel:0"]
    dump({k: e[k] for k in keep}, os.path.join(HERE, "keras_like_earliest.json"))
    json.dump({k: [list(v.shape), float(np.float64(v).sum())] for k, v in e.items()},
              open(os.path.join(HERE, "keras_like_earliest_sums.json"), "w"), indent=0)
    dump(latest_format(os.path.join(HERE, "latest_format.h5")), os.path.join(HERE, "latest_format.json"))
    unsupported(os.path.join(HERE, "unsupported.h5"))
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))
