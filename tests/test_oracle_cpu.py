"""Pins the CPU oracle (oracle/mbn_oracle.c). The reference has no tests or golden vectors, so the
oracle is "parity unpinned" by the reference; what pins it here is:
  1. hand-derived known-answer vectors from the text of kernel.cl (tests/golden/kat_literal.json),
  2. an independent numpy formulation of the same semantics on seeded random inputs,
  3. torch.nn.functional.conv2d (an independent implementation, not the reference) for the fp32 mode at every
     SURVEY §2.1 geometry,
  4. committed checksums of the oracle's own outputs (tests/golden/oracle_checksums.json) so it cannot drift silently.
"""
import hashlib
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ----------------------------------------------------------------------------- 1. hand-derived KATs

def _kat_cases():
    return [c for c in json.load(open(os.path.join(GOLD, "kat_literal.json")))["cases"]]


@pytest.mark.parametrize("case", _kat_cases(), ids=lambda c: c["name"])
def test_kat_from_kernel_cl(orc, case):
    k, q = case["kernel"], case["quirks"]
    rows, cols, fs, oc = case["rows"], case["cols"], case["filtersize"], case["op_size"]
    if k == "depthwise":
        got = orc.lit_depthwise(case["input"], case["filter"], rows, cols, fs, case["stride"], oc, quirks=q)
    elif k == "pointwise":
        got = orc.lit_pointwise(case["input"], case["filter"], rows, cols, fs, oc, quirks=q)
    elif k == "pool":
        if "input_fill" in case:
            x = np.concatenate([np.full(rows * cols, v, np.uint8) for v in case["input_fill"]])
        else:
            x = np.array(case["input"], np.uint8)
        got = orc.lit_pool(x, rows, cols, fs, oc, quirks=q)
    else:
        g = np.array(case["input_g"], np.uint8) if "input_g" in case else np.full(rows * cols, case["input_g_fill"], np.uint8)
        b = np.array(case["input_b"], np.uint8) if "input_b" in case else np.full(rows * cols, case["input_b_fill"], np.uint8)
        f = np.array(case["filter"], np.int32) if "filter" in case else np.full(oc * 27, case["filter_fill"], np.int32)
        got = orc.lit_convolute(case["input_r"], g, b, f, rows, cols, fs, case["stride"], oc, quirks=q)
    assert list(got) == case["expected"], case["why"]


# ----------------------------------------------------------------------------- 2. independent numpy formulation

def np_lit_depthwise(x, f, rows, cols, stride, quirks):
    """Vectorised over pixels, explicit over taps; written from SURVEY.md §8a's semantic facts, not from the C."""
    ch, in_rows, in_cols = x.shape
    flat = x.reshape(-1).astype(np.int64)
    ty, tx = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    out = np.zeros((ch, rows, cols), np.uint8)
    carry = np.zeros((rows, cols), np.int64)
    for c in range(ch):
        base = 0 if quirks & 2 else c * in_rows * in_cols
        s = carry.copy() if quirks & 1 else np.zeros((rows, cols), np.int64)
        for i in (-1, 0, 1):
            for j in (-1, 0, 1):
                w = int(f[c, i + 1, j + 1])
                if quirks & 4:
                    yi, xi = ty + i, tx + j
                    idx = yi * cols * stride + xi * stride + base
                    ok = (yi >= 0) & (xi >= 0) & (idx < flat.size)
                else:
                    iy, ix = ty * stride + i, tx * stride + j
                    ok = (iy >= 0) & (ix >= 0) & (iy < in_rows) & (ix < in_cols)
                    idx = base + iy * in_cols + ix
                v = np.where(ok, flat[np.clip(idx, 0, flat.size - 1)], 0)
                s = s + v * w
        s = ((s + 2 ** 31) % 2 ** 32) - 2 ** 31            # int32 wrap
        s = np.maximum(s, 0)
        carry = s
        out[c] = (s % 256).astype(np.uint8)
    return out.reshape(-1)


@pytest.mark.parametrize("quirks", [0, 1, 2, 4, 5, 6, 7])
@pytest.mark.parametrize("geom", [(6, 6, 3, 1), (5, 7, 4, 1), (4, 4, 3, 2), (7, 5, 2, 2)])
def test_literal_depthwise_vs_numpy(orc, quirks, geom):
    rows, cols, ch, stride = geom
    rng = np.random.default_rng(rows * 31 + cols + quirks)
    x = rng.integers(0, 256, (ch, rows * stride, cols * stride), dtype=np.uint8)
    f = rng.integers(-4, 5, (ch, 3, 3), dtype=np.int32)
    got = orc.lit_depthwise(x, f, rows, cols, 3, stride, ch, quirks=quirks)
    assert np.array_equal(got, np_lit_depthwise(x, f, rows, cols, stride, quirks))


@pytest.mark.parametrize("carry", [0, 1])
def test_literal_pointwise_vs_numpy(orc, carry):
    rng = np.random.default_rng(carry)
    cin, cout, rows, cols = 5, 7, 3, 4
    x = rng.integers(0, 256, (cin, rows, cols), dtype=np.uint8)
    f = rng.integers(-3, 4, (cout, cin), dtype=np.int32)
    dots = np.einsum("oc,chw->ohw", f.astype(np.int64), x.astype(np.int64))
    want = np.zeros((cout, rows, cols), np.uint8)
    s = np.zeros((rows, cols), np.int64)
    for o in range(cout):
        s = np.maximum((s if carry else 0) + dots[o], 0)
        want[o] = s % 256
    assert np.array_equal(orc.lit_pointwise(x, f, rows, cols, cin, cout, quirks=carry), want.reshape(-1))


def test_literal_pool_vs_numpy(orc):
    x = np.random.default_rng(0).integers(0, 256, (9, 7, 7), dtype=np.uint8)
    sums = x.reshape(9, -1).astype(np.int64).sum(1)
    assert np.array_equal(orc.lit_pool(x, 7, 7, 7, 9, quirks=0), (sums // 49) % 256)
    assert np.array_equal(orc.lit_pool(x, 7, 7, 7, 9, quirks=1), (np.cumsum(sums) // 49) % 256)
    # filtersize^2 consecutive bytes of the plane, not a 2-D window (kernel.cl:126-128)
    y = np.arange(25, dtype=np.uint8).reshape(1, 5, 5)
    assert orc.lit_pool(y, 5, 5, 3, 1, quirks=0)[0] == sum(range(9)) // 9


def test_int32_wraparound(orc):
    x = np.full((1, 1, 1), 255, np.uint8)
    f = np.array([[2 ** 31 - 1]], np.int32)
    # 255 * (2^31-1) mod 2^32 = 0x7FFFFF01 as signed int32 -> positive -> low byte 0x01
    assert orc.lit_pointwise(x, f, 1, 1, 1, 1, quirks=0)[0] == 0x01


# ----------------------------------------------------------------------------- 3. fp32 mode vs torch (independent)

SURVEY_DW = [(112, 32, 1), (112, 64, 2), (56, 128, 1), (56, 128, 2), (28, 256, 1), (28, 256, 2), (14, 512, 1),
             (14, 512, 2), (7, 1024, 1)]
SURVEY_PW = [(112, 32, 64), (56, 64, 128), (56, 128, 128), (28, 128, 256), (28, 256, 256), (14, 256, 512),
             (14, 512, 512), (7, 512, 1024), (7, 1024, 1024)]


def _torch():
    return pytest.importorskip("torch")


def _tf_same_pad(x, k, s, torch):
    """TF/Keras "SAME": total = max((ceil(n/s)-1)*s + k - n, 0), floor half before, rest after."""
    h = x.shape[2]
    total = max((-(-h // s) - 1) * s + k - h, 0)
    lo, hi = total // 2, total - total // 2
    return torch.nn.functional.pad(x, (lo, hi, lo, hi))


@pytest.mark.parametrize("geom", SURVEY_DW)
def test_f32_depthwise_vs_torch(orc, geom):
    torch = _torch()
    h, c, s = geom
    rng = np.random.default_rng(h + c + s)
    x = rng.uniform(-1, 1, (1, h, h, c)).astype(np.float32)
    f = rng.normal(0, 0.5, (3, 3, c)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(0, 0.1, c).astype(np.float32)
    got = orc.f32_depthwise(x, f, sc, sh, s, orc.ACT_RELU6)
    xt = _tf_same_pad(torch.from_numpy(x).permute(0, 3, 1, 2).double(), 3, s, torch)
    wt = torch.from_numpy(f).permute(2, 0, 1).unsqueeze(1).double()
    y = torch.nn.functional.conv2d(xt, wt, stride=s, groups=c)
    y = (y * torch.from_numpy(sc).double().view(1, -1, 1, 1) + torch.from_numpy(sh).double().view(1, -1, 1, 1)).clamp(0, 6)
    want = y.permute(0, 2, 3, 1).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("geom", SURVEY_PW)
def test_f32_pointwise_vs_torch(orc, geom):
    torch = _torch()
    h, cin, cout = geom
    rng = np.random.default_rng(h + cin + cout)
    x = rng.uniform(-1, 1, (1, h, h, cin)).astype(np.float32)
    f = rng.normal(0, (2 / cin) ** 0.5, (cout, cin)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.normal(0, 0.1, cout).astype(np.float32)
    got = orc.f32_pointwise(x, f, sc, sh, orc.ACT_RELU6)
    y = torch.from_numpy(x).double() @ torch.from_numpy(f).double().T
    want = (y * torch.from_numpy(sc).double() + torch.from_numpy(sh).double()).clamp(0, 6).numpy()
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_f32_conv1_pool_softmax_vs_torch(orc):
    torch = _torch()
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (2, 224, 224, 3)).astype(np.float32)
    f = rng.normal(0, 0.27, (3, 3, 3, 32)).astype(np.float32)
    got = orc.f32_conv(x, f, None, None, 2, orc.ACT_RELU)
    xt = _tf_same_pad(torch.from_numpy(x).permute(0, 3, 1, 2).double(), 3, 2, torch)
    y = torch.nn.functional.conv2d(xt, torch.from_numpy(f).permute(3, 2, 0, 1).double(), stride=2).clamp(min=0)
    want = y.permute(0, 2, 3, 1).numpy()
    assert got.shape == (2, 112, 112, 32) and np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    a = rng.uniform(0, 6, (3, 7, 7, 64)).astype(np.float32)
    assert np.allclose(orc.f32_pool(a), a.astype(np.float64).mean((1, 2)), rtol=1e-6)
    l = rng.normal(0, 3, (4, 1000)).astype(np.float32)
    p, am = orc.f32_softmax(l)
    assert np.allclose(p, torch.softmax(torch.from_numpy(l).double(), 1).numpy(), atol=1e-7)
    assert np.array_equal(am, l.argmax(1))


def test_f32_equals_literal_where_they_coincide(orc):
    """identity BN, ReLU, top/left pad, small integers, quirks off => same numbers in both modes."""
    rng = np.random.default_rng(2)
    ch, h = 6, 9
    x = rng.integers(0, 4, (ch, h, h), dtype=np.uint8)
    f = rng.integers(-1, 3, (ch, 3, 3), dtype=np.int32)
    lit = orc.lit_depthwise(x, f, h, h, 3, 1, ch, quirks=0).reshape(ch, h, h)
    f32 = orc.f32_depthwise(x.transpose(1, 2, 0)[None].astype(np.float32), f.transpose(1, 2, 0).astype(np.float32),
                            None, None, 1, orc.ACT_RELU, pad_top=1, pad_left=1)[0].transpose(2, 0, 1)
    assert np.array_equal(lit.astype(np.float32), f32)
    pin = rng.integers(0, 3, (5, 4, 4), dtype=np.uint8)
    pf = rng.integers(-1, 3, (7, 5), dtype=np.int32)
    plit = orc.lit_pointwise(pin, pf, 4, 4, 5, 7, quirks=0).reshape(7, 4, 4)
    pf32 = orc.f32_pointwise(pin.transpose(1, 2, 0).astype(np.float32), pf.astype(np.float32), None, None, orc.ACT_RELU)
    assert np.array_equal(plit.astype(np.float32), pf32.transpose(2, 0, 1))


# ----------------------------------------------------------------------------- 4. whole net + committed checksums

def _tiny_net(pkg, orc, tmp_path, alpha=0.25, res=64, classes=12, seed=11):
    path = str(tmp_path / "t.h5")
    pkg.synthetic_h5(path, alpha=alpha, classes=classes, seed=seed)
    hw = pkg.HostWeights(path, res=res)
    return hw, orc.plan_build(alpha, res, classes)


def test_net_forward_serial_equals_threaded_and_layerwise(pkg, orc, tmp_path):
    """BASELINE config 1 ("first-5-layers ... plumbing, no GPU") on the CPU restatement, plus consistency of the
    whole-net driver with the per-layer functions."""
    hw, oplan = _tiny_net(pkg, orc, tmp_path)
    imgs = np.random.default_rng(0).uniform(-1, 1, (2, 64, 64, 3)).astype(np.float32)
    out1, layers = orc.net_forward(oplan, hw.blob, imgs, threads=1, keep_layers=True)
    out4, _ = orc.net_forward(oplan, hw.blob, imgs, threads=4)
    assert np.array_equal(out1, out4)
    l5, _ = orc.net_forward(oplan, hw.blob, imgs, last_layer=5)
    assert np.array_equal(l5, layers[4]) and l5.shape == (2, 16, 16, 32)
    # re-derive layers 1..3 by hand from the blob
    L = oplan.layer
    b = hw.blob
    x = orc.f32_conv(imgs, b[L[0].w_offset:][:L[0].w_count].reshape(3, 3, 3, -1), b[L[0].scale_offset:][:L[0].out_ch],
                     b[L[0].shift_offset:][:L[0].out_ch], 2, orc.ACT_RELU6)
    assert np.array_equal(x, layers[0])
    x = orc.f32_depthwise(x, b[L[1].w_offset:][:L[1].w_count].reshape(3, 3, -1), b[L[1].scale_offset:][:L[1].out_ch],
                          b[L[1].shift_offset:][:L[1].out_ch], 1, orc.ACT_RELU6)
    assert np.array_equal(x, layers[1])
    x = orc.f32_pointwise(x, b[L[2].w_offset:][:L[2].w_count].reshape(L[2].out_ch, L[2].in_ch),
                          b[L[2].scale_offset:][:L[2].out_ch], b[L[2].shift_offset:][:L[2].out_ch], orc.ACT_RELU6)
    assert np.array_equal(x, layers[2])
    assert out1.shape == (2, 1, 1, 12) and np.isfinite(out1).all()


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def compute_checksums(orc):
    """Seeded inputs -> oracle outputs -> digests. tests/golden/make_oracle_checksums.py writes this to disk."""
    rng = np.random.default_rng(2024)
    out = {}
    x = rng.integers(0, 256, (8, 14, 14), dtype=np.uint8)
    f = rng.integers(-3, 4, (8, 3, 3), dtype=np.int32)
    for q in (0, 15):
        out["lit_dw_q%d" % q] = _digest(orc.lit_depthwise(x, f, 14, 14, 3, 1, 8, quirks=q))
        out["lit_dw_s2_q%d" % q] = _digest(orc.lit_depthwise(x, f, 7, 7, 3, 2, 8, quirks=q))
    pf = rng.integers(-2, 3, (16, 8), dtype=np.int32)
    out["lit_pw_q0"] = _digest(orc.lit_pointwise(x, pf, 14, 14, 8, 16, quirks=0))
    out["lit_pw_q1"] = _digest(orc.lit_pointwise(x, pf, 14, 14, 8, 16, quirks=1))
    planes = [rng.integers(0, 256, 32 * 32, dtype=np.uint8) for _ in range(3)]
    cf = rng.integers(-2, 3, (4, 3, 3, 3), dtype=np.int32)
    for q in (0, 15):
        out["lit_conv_q%d" % q] = _digest(orc.lit_convolute(*planes, cf, 32, 32, 3, 2, 4, quirks=q))
    out["lit_pool_q15"] = _digest(orc.lit_pool(x[:, :7, :7].copy(), 7, 7, 7, 8, quirks=15))
    # fp32 digests are of values rounded to 1e-4 so they survive libm/compiler differences
    xf = rng.uniform(-1, 1, (1, 28, 28, 16)).astype(np.float32)
    ff = rng.normal(0, 0.5, (3, 3, 16)).astype(np.float32)
    out["f32_dw"] = _digest(np.round(orc.f32_depthwise(xf, ff, None, None, 1, 2), 4))
    out["f32_dw_s2"] = _digest(np.round(orc.f32_depthwise(xf, ff, None, None, 2, 2), 4))
    wf = rng.normal(0, 0.3, (24, 16)).astype(np.float32)
    out["f32_pw"] = _digest(np.round(orc.f32_pointwise(xf, wf, None, None, 2), 4))
    return out


def test_oracle_matches_committed_checksums(orc):
    want = json.load(open(os.path.join(GOLD, "oracle_checksums.json")))
    assert compute_checksums(orc) == want
