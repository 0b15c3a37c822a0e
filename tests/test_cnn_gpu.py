"""GPU parity of the CNN forward: MFMA implicit-GEMM convolutions and the head against a plain
PyTorch CPU float32 restatement (oracle/cnn_oracle.py).  Tolerance: the north star's 1e-3 on
logits (absolute; logits are O(1) with calibrated BatchNorm statistics), checked at 2e-4 here.
Logits-vs-TensorFlow is parity-unpinned (no TF / weights in the container)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LOGIT_ATOL = 2e-4  # north star: 1e-3


@pytest.fixture(scope="module")
def engine():
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3")
    yield eng
    eng.close()


CASES = [
    # (Cin, Cout, H, W, ksize, stride, same)  -- every (channels per group, stride, kernel) of WR-ResNet-22-4
    (2, 16, 20, 24, 3, 1, True),      # conv1_1 (direct kernel)
    (2, 16, 19, 23, 3, 1, True),      # conv1_1 with N*H*W = 874, not a multiple of 64: the last wave's LDS-staged stores
    (16, 64, 17, 21, 3, 1, True),     # res2b0_branch2a
    (16, 64, 17, 21, 1, 1, False),    # shortcut2
    (64, 64, 16, 33, 3, 1, True),     # res2*
    (64, 128, 21, 18, 3, 2, True),    # res3b0_branch2a (stride 2, SAME pads bottom/right)
    (64, 128, 21, 18, 1, 2, False),   # shortcut3
    (128, 128, 10, 19, 3, 1, True),   # res3*
    (128, 256, 20, 17, 3, 3, True),   # res4b0_branch2a (stride 3)
    (128, 256, 20, 17, 1, 3, False),  # shortcut4
    (256, 256, 9, 11, 3, 1, True),    # res4*
    # shapes the network does not have but cpx_conv2d takes on the same kernels (conv_bf3w_kernel: 32 / 64 channels
    # per group in, 32 / 64 out; one or two column slices per tile)
    (64, 128, 19, 35, 3, 1, True),
    (128, 64, 18, 17, 3, 1, True),
    (128, 128, 33, 16, 3, 1, False),  # VALID: no padding, output 31 x 14
]


@pytest.fixture(params=["bf16x3", "f32", "bf16x2", "fp16x2"])
def math(engine, request):
    """The ways of multiplying float32 operands on the matrix cores (include/cpx.h: cpx_set_cnn_math)."""
    engine.set_cnn_math(request.param)
    assert engine.get_cnn_math() == request.param
    yield request.param
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)


@pytest.mark.parametrize("case", CASES)
def test_conv2d_matches_torch(engine, math, case):
    import torch
    import torch.nn.functional as F

    from cpx.ml_tools.wrresnet import ConvDesc, pack_conv

    Cin, Cout, H, W, ks, stride, same = case
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    N = 2
    x = rng.normal(0, 1, size=(N, H, W, Cin)).astype(np.float32)
    k = rng.normal(0, 0.2, size=(ks, ks, Cin // 2, Cout)).astype(np.float32)
    in_s = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    in_b = rng.normal(0, 0.3, Cin).astype(np.float32)
    out_s = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    out_b = rng.normal(0, 0.3, Cout).astype(np.float32)
    Ho = -(-H // stride) if same else (H - ks) // stride + 1
    Wo = -(-W // stride) if same else (W - ks) // stride + 1
    res = rng.normal(0, 1, size=(N, Ho, Wo, Cout)).astype(np.float32)
    for variant in range(4):  # 3 = bias only (how conv1_1 and the shortcuts are called)
        use_pro = variant in (0, 2)
        use_res = variant in (1, 2)
        relu = variant in (0, 2)
        use_scale = variant != 3
        # torch reference
        xt = torch.from_numpy(x).permute(0, 3, 1, 2)
        if use_pro:
            xt = F.relu(xt * torch.from_numpy(in_s)[None, :, None, None] + torch.from_numpy(in_b)[None, :, None, None])
        if same:
            ph = max((Ho - 1) * stride + ks - H, 0)
            pw = max((Wo - 1) * stride + ks - W, 0)
            xt = F.pad(xt, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        y = F.conv2d(xt, torch.from_numpy(k).permute(3, 2, 0, 1).contiguous(), None, stride=stride, groups=2)
        if use_scale:
            y = y * torch.from_numpy(out_s)[None, :, None, None]
        y = y + torch.from_numpy(out_b)[None, :, None, None]
        if use_res:
            y = y + torch.from_numpy(res).permute(0, 3, 1, 2)
        if relu:
            y = F.relu(y)
        want = y.permute(0, 2, 3, 1).numpy()
        assert want.shape == (N, Ho, Wo, Cout)
        # device
        dev = engine.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        xd, wd = up(x), up(pack_conv(k))
        isd, ibd, osd, obd, rd = up(in_s), up(in_b), up(out_s), up(out_b), up(res)
        out = torch.full((N, Ho, Wo, Cout), np.nan, dtype=torch.float32, device=dev)
        ptr = lambda v: C.c_void_p(v.data_ptr())
        d = ConvDesc(N, H, W, Cin, Cout, 2, ks, stride, 1 if same else 0, 1 if relu else 0, ptr(xd), ptr(out), ptr(wd),
                     ptr(isd) if use_pro else None, ptr(ibd) if use_pro else None, ptr(osd) if use_scale else None, ptr(obd),
                     ptr(rd) if use_res else None)
        torch.cuda.synchronize()
        rc = engine.lib.cpx_conv2d(engine.h, C.byref(d))
        assert rc == 0, engine._err()
        engine.synchronize()
        got = out.cpu().numpy()
        scale = max(1.0, float(np.abs(want).max()))
        assert np.isfinite(got).all()
        assert float(np.abs(got - want).max()) <= 2e-5 * scale, (case, variant, float(np.abs(got - want).max()))


def test_bf16x3_is_f32_accurate(engine):
    """The split-operand path is float32 arithmetic, not reduced precision: against a float64 convolution of the
    same float32 inputs its error is the float32 MFMA kernel's (both are dominated by the float32 accumulation),
    on all three stride-1 layer shapes, with wide-range operands."""
    import torch
    import torch.nn.functional as F

    from cpx.ml_tools.wrresnet import ConvDesc, pack_conv

    dev = engine.device
    for Cin, Cout, H, W in ((64, 64, 40, 40), (128, 128, 24, 40), (256, 256, 27, 27)):
        rng = np.random.default_rng(Cin)
        x = (rng.normal(0, 1, size=(2, H, W, Cin)) * np.exp(rng.normal(0, 2, size=(2, H, W, Cin)))).astype(np.float32)
        k = (rng.normal(0, 0.1, size=(3, 3, Cin // 2, Cout)) * np.exp(rng.normal(0, 1, size=(3, 3, Cin // 2, Cout)))
             ).astype(np.float32)
        want = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(k).double().permute(3, 2, 0, 1),
                        None, padding=1, groups=2).permute(0, 2, 3, 1).numpy()
        # magnitude of the sum each output accumulates: the scale float32 rounding errors are relative to
        mag = F.conv2d(torch.from_numpy(np.abs(x)).double().permute(0, 3, 1, 2),
                       torch.from_numpy(np.abs(k)).double().permute(3, 2, 0, 1), None, padding=1,
                       groups=2).permute(0, 2, 3, 1).numpy()
        errs = {}
        for mode in ("f32", "bf16x3", "bf16x2", "fp16x2"):
            engine.set_cnn_math(mode)
            xd = torch.from_numpy(x).to(dev)
            wd = torch.from_numpy(pack_conv(k)).to(dev)
            out = torch.full((2, H, W, Cout), np.nan, dtype=torch.float32, device=dev)
            ptr = lambda v: C.c_void_p(v.data_ptr())
            d = ConvDesc(2, H, W, Cin, Cout, 2, 3, 1, 1, 0, ptr(xd), ptr(out), ptr(wd), None, None, None, None, None)
            torch.cuda.synchronize()
            assert engine.lib.cpx_conv2d(engine.h, C.byref(d)) == 0, engine._err()
            engine.synchronize()
            got = out.cpu().numpy().astype(np.float64)
            assert np.isfinite(got).all()
            errs[mode] = float((np.abs(got - want) / mag).max())
            if mode == "fp16x2":  # (the operands lie inside fp16's range: these are the fp16 kernels' results, not the rerun's)
                assert not engine.cnn_last_overflow()
        engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
        print("max |error| / accumulated magnitude vs float64, %d channels: %s" % (Cin, errs))
        # float32 accumulation of K = 9 * Cin / 2 terms: a few 2^-24 relative to the accumulated magnitude
        assert errs["f32"] < 4e-6 and errs["bf16x3"] < 4e-6, errs
        assert errs["bf16x3"] <= 2.0 * errs["f32"] + 2.0 ** -24, errs
        # two planes rounded to nearest (the 32 / 64-channel layers; 128 per group keeps three): hi is within 2^-8 of
        # the operand, lo within 2^-8 of the rest: each operand is off by at most 2^-16 of itself, the dropped lo x lo
        # term is at most 2^-16 of a product -- 3 x 2^-16 of the accumulated magnitude bounds any sum (the signs average
        # most of it away: the network's logits move by < 1e-6, test_wrresnet_logits_match_oracle)
        assert errs["bf16x2"] <= 3 * 2.0 ** -16, errs
        if Cin <= 128:
            assert errs["bf16x2"] > errs["bf16x3"], errs  # (it IS the other arithmetic)
        # two fp16 planes rounded to nearest: 11 + 11 bits, each operand off by at most 2^-22 of itself (2^-25 absolute
        # of the scaled value where the low plane is subnormal), the dropped lo x lo term at most 2^-22 of a product:
        # the float32 accumulation's own error level, 64 times below bf16x2's
        # -- and, measured, the exact split's own criterion: what makes it the default mode
        assert errs["fp16x2"] <= 2.0 * errs["f32"] + 2.0 ** -24, errs
        assert errs["fp16x2"] < 4e-6 and errs["fp16x2"] < errs["bf16x2"] / 8, errs


def test_fp16x2_out_of_range_reruns_in_bf16x3(engine):
    """fp16 has a range: an activation the scaling cannot bring below 65504 must not saturate silently.  The fp16
    kernel raises the device-side overflow word and the three-plane bf16 kernel, launched right behind it, computes
    the layer again: same bits as the bf16x3 mode, and cpx_cnn_last_overflow says so."""
    import torch

    from cpx.ml_tools.wrresnet import ConvDesc, pack_conv

    dev = engine.device
    for Cin, Cout, H, W, stride in ((64, 64, 20, 24, 1), (128, 128, 20, 17, 1), (256, 256, 9, 11, 1), (64, 128, 21, 18, 2)):
        rng = np.random.default_rng(Cin + stride)
        x = rng.normal(0, 1, size=(2, H, W, Cin)).astype(np.float32)
        k = rng.normal(0, 0.1, size=(3, 3, Cin // 2, Cout)).astype(np.float32)
        Ho, Wo = -(-H // stride), -(-W // stride)
        outs = {}
        for big in (False, True):
            xx = x.copy()
            if big:
                xx[1, H // 2, W // 3, 5] = 1.0e5  # one element beyond fp16's largest finite value
            for mode in ("bf16x3", "fp16x2"):
                engine.set_cnn_math(mode)
                xd = torch.from_numpy(xx).to(dev)
                wd = torch.from_numpy(pack_conv(k)).to(dev)
                out = torch.full((2, Ho, Wo, Cout), np.nan, dtype=torch.float32, device=dev)
                ptr = lambda v: C.c_void_p(v.data_ptr())
                d = ConvDesc(2, H, W, Cin, Cout, 2, 3, stride, 1, 0, ptr(xd), ptr(out), ptr(wd), None, None, None, None, None)
                torch.cuda.synchronize()
                assert engine.lib.cpx_conv2d(engine.h, C.byref(d)) == 0, engine._err()
                engine.synchronize()
                outs[(big, mode)] = out.cpu().numpy()
                if mode == "fp16x2":
                    assert engine.cnn_last_overflow() == big
        engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
        assert np.isfinite(outs[(True, "fp16x2")]).all()
        assert np.array_equal(outs[(True, "fp16x2")], outs[(True, "bf16x3")])       # the rerun IS the bf16x3 kernel
        assert not np.array_equal(outs[(False, "fp16x2")], outs[(False, "bf16x3")])  # in range: the other arithmetic
        assert float(np.abs(outs[(False, "fp16x2")] - outs[(False, "bf16x3")]).max()) <= 2e-5 * max(
            1.0, float(np.abs(outs[(False, "bf16x3")]).max()))


def test_fp16x2_network_overflow_is_loud_and_correct(engine):
    """The whole network in fp16x2 on inputs 1000 times the sensor's range: some layer's activations leave fp16's
    range, the forward finishes on the bf16x3 kernels and the logits still match the float32 restatement."""
    import torch

    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(77)
    x = rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=9), x)
    engine.set_cnn_math("fp16x2")
    net = wr.WRResNetDevice(engine, w, 17)
    logits, _ = net.forward(torch.from_numpy(x).to(engine.device))
    assert not engine.cnn_last_overflow()
    want, _ = co.forward(w, x)
    assert float(np.abs(logits.cpu().numpy() - want).max()) <= LOGIT_ATOL
    xb = x * np.float32(1000.0)
    logits_b, _ = net.forward(torch.from_numpy(xb).to(engine.device))
    assert engine.cnn_last_overflow()
    want_b, _ = co.forward(w, xb)
    got_b = logits_b.cpu().numpy()
    assert np.isfinite(got_b).all()
    assert float(np.abs(got_b - want_b).max()) <= 2e-4 * max(1.0, float(np.abs(want_b).max()))
    # the word belongs to ONE forward: the next one, in range again, runs on the fp16 kernels
    logits_c, _ = net.forward(torch.from_numpy(x).to(engine.device))
    assert not engine.cnn_last_overflow()
    assert torch.equal(logits_c, logits)
    net.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)


@pytest.mark.parametrize("scale", [8.0, 30.0, 100.0, 300.0, 1000.0])
def test_fp16x2_partial_overflow_is_correct(engine, scale):
    """Inputs 8 ... 1000 times the range the BatchNorm statistics were fitted to: depending on the factor none, some or all
    of the blocks leave fp16's range (one overflow word per block; the fused stage-2 blocks rerun as their two guarded
    three-plane launches).  Whatever mix of fp16 and rerun blocks a forward ends up with, its logits are the exact mode's
    to float32 accuracy, and the forward says whether anything was rerun."""
    import torch

    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(515)
    x = rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=21), x)
    xs = torch.from_numpy(x * np.float32(scale)).to(engine.device)
    engine.set_cnn_math("bf16x3")
    net = wr.WRResNetDevice(engine, w, 17)
    want, _ = net.forward(xs)
    want = want.clone()
    engine.set_cnn_math("fp16x2")
    n0 = engine.cnn_overflow_forwards(reset=True)
    got, _ = net.forward(xs)
    reran = engine.cnn_last_overflow()
    assert engine.cnn_overflow_forwards(reset=True) == (1 if reran else 0)
    assert torch.isfinite(got).all()
    tol = 2e-5 * max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= tol, (scale, reran, float((got - want).abs().max()), tol)
    if scale >= 1000.0:
        assert reran
    print("scale", scale, "rerun", reran)
    net.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
    del n0


@pytest.mark.parametrize("fs,n", [(32, 3), (64, 1)])
def test_wrresnet_logits_match_oracle(engine, math, fs, n):
    import torch

    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(5 + fs)
    side = 5 * fs
    x = rng.uniform(0, 255, size=(n, side, side, 2)).astype(np.float32)
    x[:, ::7, :, 1] = 0.0
    w = wr.random_weights(17, seed=3)
    w = co.calibrate_bn(w, x)
    want_logits, want_probs = co.forward(w, x)
    net = wr.WRResNetDevice(engine, w, 17)
    logits, probs = net.forward(torch.from_numpy(x).to(engine.device))
    got = logits.cpu().numpy()
    assert np.isfinite(got).all()
    assert float(np.abs(want_logits).max()) > 0.05  # a non-degenerate comparison
    assert float(np.abs(got - want_logits).max()) <= LOGIT_ATOL, float(np.abs(got - want_logits).max())
    np.testing.assert_allclose(probs.cpu().numpy(), want_probs, atol=1e-4)
    # the single-call native forward (cpx_cnn_forward) and the layer-by-layer building blocks are the same kernels
    l2, p2 = net.forward_layerwise(torch.from_numpy(x).to(engine.device))
    if math == "f32":
        assert torch.equal(l2, logits) and torch.equal(p2, probs)
    else:  # the native call folds the 1x1 shortcuts into the following convolution: another summation order
        assert float((l2 - logits).abs().max()) <= 1e-5 and float((p2 - probs).abs().max()) <= 1e-5
    # a second batch size reuses / regrows the network's activation arena
    l3, _ = net.forward(torch.from_numpy(np.concatenate([x, x])).to(engine.device))
    assert torch.equal(l3[:n], logits) and torch.equal(l3[n:], logits)
    net.close()


def test_fused_shortcut_equals_separate_launch(engine, monkeypatch):
    """cpx_cnn_forward folds the 1x1 shortcut of a stage's first block into the block's second convolution (default
    math); an engine created with CPX_CNN_FUSE_SHORTCUT=0 launches it separately.  Same network, same input: the
    logits differ only by the summation order."""
    import torch

    import cnn_oracle as co
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(21)
    x = rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=5), x)
    engine.set_cnn_math("bf16x3")  # (the fused shortcut exists in every split mode; the exact one isolates the summation order)
    net = wr.WRResNetDevice(engine, w, 17)
    fused, _ = net.forward(torch.from_numpy(x).to(engine.device))
    net.close()
    monkeypatch.setenv("CPX_CNN_FUSE_SHORTCUT", "0")
    eng2 = TrackEngine(model="lepton3")
    monkeypatch.delenv("CPX_CNN_FUSE_SHORTCUT")
    eng2.set_cnn_math("bf16x3")
    net2 = wr.WRResNetDevice(eng2, w, 17)
    separate, _ = net2.forward(torch.from_numpy(x).to(eng2.device))
    net2.close()
    eng2.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
    diff = float((fused.cpu() - separate.cpu()).abs().max())
    assert 0.0 < diff <= 1e-5 or diff == 0.0, diff
    want, _ = co.forward(w, x)
    assert float(np.abs(fused.cpu().numpy() - want).max()) <= LOGIT_ATOL


@pytest.mark.parametrize("side,n", [(160, 3), (150, 2), (37, 1)])
def test_fused_block_matches_the_two_launches(monkeypatch, side, n):
    """fp16x2 runs a stage-2 block past the first as ONE launch (`mid` stays in LDS; CPX_CNN_BLOCK_FUSION=1 fuses exactly
    those); an engine created with CPX_CNN_BLOCK_FUSION=0 runs the two convolutions as two.  Three one-launch forms
    (CPX_BLOCK32_SPLIT): 0 = conv_block32_kernel -- same planes, same products in the same order as the two launches: the
    logits are the SAME BITS; 1 (default) = conv_block32s_kernel and 2 = conv_block32p_kernel, the two convolutions on
    different waves, taps summed kx-major: another float32 order of the same terms, logits within 1e-5 (relative to the largest
    logit).  On whole tiles, ragged ones (150 = 9 x 16 + 6) and a map of three tiles."""
    import torch

    import cnn_oracle as co
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(300 + side)
    x = rng.uniform(0, 255, size=(n, side, side, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=8), x)
    out = {}
    for fusion, split in (("0", None), ("1", "0"), ("1", "1"), ("1", "2")):
        # (CPX_BLOCK32_SPLIT is read once per process by the launcher: the forms other than the first one seen run in a child)
        out[(fusion, split)] = _block_form_logits(w, x, fusion, split)
    two = out[("0", None)]
    assert torch.equal(out[("1", "0")], two), float((out[("1", "0")] - two).abs().max())
    scale = max(1.0, float(two.abs().max()))
    for split in ("1", "2"):
        d = float((out[("1", split)] - two).abs().max())
        assert d <= 1e-5 * scale, (split, d)
    want, _ = co.forward(w, x)
    for k, v in out.items():
        assert float(np.abs(v.numpy() - want).max()) <= LOGIT_ATOL, k


def _block_form_logits(w, x, fusion, split):
    """Logits of one forward in a fresh process with CPX_CNN_BLOCK_FUSION / CPX_BLOCK32_SPLIT set (both are read once)."""
    import pickle
    import subprocess
    import sys
    import tempfile

    import torch

    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "in.pkl"), "wb") as fh:
            pickle.dump((w, x), fh)
        code = (
            "import os, sys, pickle, numpy as np, torch\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from cpx.engine import TrackEngine\n"
            "from cpx.ml_tools import wrresnet as wr\n"
            "w, x = pickle.load(open(%r, 'rb'))\n"
            "eng = TrackEngine(model='lepton3'); eng.set_cnn_math('fp16x2')\n"
            "net = wr.WRResNetDevice(eng, w, 17)\n"
            "eng.conv_timing(True)\n"
            "logits, _ = net.forward(torch.from_numpy(x).to(eng.device))\n"
            "launches = eng.conv_timing()\n"
            "assert not eng.cnn_last_overflow()\n"
            "fusion = os.environ['CPX_CNN_BLOCK_FUSION']\n"
            "# key 'stride 4' = a block launch; 320321 = a stage-2 convolution launched on its own\n"
            "if fusion == '1': assert launches[320324][0] == 2 and launches[320321][0] == 1 and 80324 not in launches, launches\n"
            "else: assert 320324 not in launches and launches[320321][0] == 5, launches\n"
            "pickle.dump(logits.cpu().numpy(), open(%r, 'wb'))\n"
            % (os.path.join(REPO, "classifier-pipeline_amd"), os.path.join(REPO, "oracle"), os.path.join(td, "in.pkl"),
               os.path.join(td, "out.pkl")))
        env = dict(os.environ, CPX_CNN_BLOCK_FUSION=fusion)
        env.pop("CPX_BLOCK32_SPLIT", None)
        if split is not None:
            env["CPX_BLOCK32_SPLIT"] = split
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        with open(os.path.join(td, "out.pkl"), "rb") as fh:
            return torch.from_numpy(pickle.load(fh))


@pytest.mark.parametrize("side,n", [(160, 3), (150, 2), (37, 1)])
def test_fused_first_block_matches_the_launches_it_replaces(engine, monkeypatch, side, n):
    """The default (CPX_CNN_BLOCK_FUSION=2) also runs the stage's FIRST block as one launch: its 8-channel convolution on
    fp16 planes (four taps per K = 32 step) instead of the exact three-plane bf16 form, the 1x1 shortcut inside.  Other
    arithmetic for that one layer, float32-accurate either way: logits within 1e-5 of the unfused block's (relative to the
    largest logit) and inside the oracle's bound."""
    import torch

    import cnn_oracle as co
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(900 + side)
    x = rng.uniform(0, 255, size=(n, side, side, 2)).astype(np.float32)
    x[:, ::5, :, 0] = 0.0
    w = co.calibrate_bn(wr.random_weights(17, seed=13), x)
    engine.set_cnn_math("fp16x2")
    net = wr.WRResNetDevice(engine, w, 17)
    engine.conv_timing(True)
    fused, _ = net.forward(torch.from_numpy(x).to(engine.device))
    launches = engine.conv_timing()
    engine.conv_timing(False)
    assert not engine.cnn_last_overflow()
    assert launches[80324][0] == 1 and launches[320324][0] == 2 and 320321 not in launches and 80321 not in launches, launches
    net.close()
    monkeypatch.setenv("CPX_CNN_BLOCK_FUSION", "1")
    eng2 = TrackEngine(model="lepton3")
    monkeypatch.delenv("CPX_CNN_BLOCK_FUSION")
    eng2.set_cnn_math("fp16x2")
    net2 = wr.WRResNetDevice(eng2, w, 17)
    ref, _ = net2.forward(torch.from_numpy(x).to(eng2.device))
    net2.close()
    eng2.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
    scale = max(1.0, float(ref.abs().max()))
    assert float((fused.cpu() - ref.cpu()).abs().max()) <= 1e-5 * scale, float((fused.cpu() - ref.cpu()).abs().max())
    want, _ = co.forward(w, x)
    assert float(np.abs(fused.cpu().numpy() - want).max()) <= LOGIT_ATOL


@pytest.mark.parametrize("side,n", [(160, 3), (150, 2), (37, 1), (20, 2)])
def test_conv1_inside_the_first_block_is_bitwise_the_launch(engine, monkeypatch, side, n):
    """Round 6: the fused first block of stage 2 computes conv1_1 itself while it stages its patch (conv_block32_kernel<true, true>):
    conv1_kernel's multiply-adds in its order, so the logits are those of the separate launch bit for bit
    (CPX_CNN_FUSE_CONV1=0, read when a handle is created); no conv1 launch is timed in the fused form."""
    import torch

    import cnn_oracle as co
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(950 + side)
    x = rng.uniform(0, 255, size=(n, side, side, 2)).astype(np.float32)
    x[:, ::7, :, 1] = 0.0
    w = co.calibrate_bn(wr.random_weights(17, seed=14), x)
    engine.set_cnn_math("fp16x2")
    net = wr.WRResNetDevice(engine, w, 17)
    engine.conv_timing(True)
    fused, _ = net.forward(torch.from_numpy(x).to(engine.device))
    launches = engine.conv_timing()
    engine.conv_timing(False)
    assert not engine.cnn_last_overflow()
    assert 10081 not in launches and launches[80324][0] == 1, launches
    net.close()
    monkeypatch.setenv("CPX_CNN_FUSE_CONV1", "0")
    eng2 = TrackEngine(model="lepton3")
    monkeypatch.delenv("CPX_CNN_FUSE_CONV1")
    eng2.set_cnn_math("fp16x2")
    net2 = wr.WRResNetDevice(eng2, w, 17)
    eng2.conv_timing(True)
    ref, _ = net2.forward(torch.from_numpy(x).to(eng2.device))
    launches2 = eng2.conv_timing()
    assert launches2[10081][0] == 1 and launches2[80324][0] == 1, launches2
    net2.close()
    eng2.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
    assert torch.equal(fused.cpu(), ref.cpu()), float((fused.cpu() - ref.cpu()).abs().max())
    want, _ = co.forward(w, x)
    assert float(np.abs(fused.cpu().numpy() - want).max()) <= LOGIT_ATOL


def test_fused_block_walks_any_share_of_the_tiles(engine, monkeypatch):
    """conv_block32_kernel is persistent: a workgroup walks its XCD's eighth of the tiles with a stride of the grid.  With
    CPX_BLOCK32_GRID=8 (one workgroup per XCD and group; read at every launch) each one walks 38 tiles of a 3-sample
    batch, with 24 some get 13 and some 12: the logits are the default grid's, bit for bit."""
    import torch

    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(412)
    x = rng.uniform(0, 255, size=(3, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=11), x)
    engine.set_cnn_math("fp16x2")
    net = wr.WRResNetDevice(engine, w, 17)
    xd = torch.from_numpy(x).to(engine.device)
    want, _ = net.forward(xd)
    want = want.clone()
    for grid in ("8", "24", "4096"):
        monkeypatch.setenv("CPX_BLOCK32_GRID", grid)
        got, _ = net.forward(xd)
        assert torch.equal(got, want), grid
    monkeypatch.delenv("CPX_BLOCK32_GRID")
    assert not engine.cnn_last_overflow()
    net.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)


@pytest.mark.parametrize("dense_sizes,activation", [((48, 24), "softmax"), ((40,), "sigmoid"), (None, "softmax")])
def test_head_variants_match_oracle(dense_sizes, activation):
    """KerasModel.build_model's head variants (kerasmodel.py:337-345): hidden Dense(relu) layers of dense_sizes and
    a softmax output, through cpx_cnn_forward and the layer-wise cpx_cnn_head_ex."""
    import torch

    import cnn_oracle as co
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(21)
    x = rng.uniform(0, 255, size=(3, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=5, dense_sizes=dense_sizes, activation=activation), x[:2])
    want_logits, want_probs = co.forward(w, x)
    eng = TrackEngine()
    net = wr.WRResNetDevice(eng, w, 17)
    xd = torch.from_numpy(x).to(eng.device)
    for fwd in (net.forward, net.forward_layerwise):
        logits, probs = fwd(xd)
        assert float(np.abs(logits.cpu().numpy() - want_logits).max()) <= 2e-4
        assert float(np.abs(probs.cpu().numpy() - want_probs).max()) <= 1e-5
    if activation == "softmax":
        assert np.allclose(probs.cpu().numpy().sum(axis=1), 1.0, atol=1e-5)
    net.close()
    eng.close()


def test_calibrate_bn_device_matches_oracle(engine):
    """wrresnet.calibrate_bn_device (the set-up utility bench.py uses to give its synthetic network BatchNorm statistics
    that fit its data) against the oracle's calibration of the same weights on the same batch."""
    import torch

    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(31)
    x = rng.uniform(0, 255, size=(4, 160, 160, 2)).astype(np.float32)
    got = wr.calibrate_bn_device(engine, wr.random_weights(17, seed=4), torch.from_numpy(x).to(engine.device))
    want = co.calibrate_bn(wr.random_weights(17, seed=4), x)
    for k, v in want.items():
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            np.testing.assert_allclose(got[k], v, rtol=2e-3, atol=2e-3, err_msg=k)
    logits, _ = co.forward(got, x)
    assert float(np.abs(logits).max()) < 50.0


def test_tflite_model_runs_in_fp16x2_without_the_rerun(engine, tmp_path):
    """A .tflite model's BatchNorms arrive folded (scale / shift, no statistics): the fp16 range scaling of its layers comes
    from a seeded probe forward at load (WRResNetDevice._measure_activation_bounds).  Its forward in the default math must
    stay on the fp16 kernels (no overflow rerun) and match the float32 restatement of the ORIGINAL weights."""
    import torch

    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr
    from cpx.ml_tools.tflite_reader import load_tflite
    from test_tflite_import_cpu import tflite_of

    rng = np.random.default_rng(41)
    x = rng.uniform(0, 255, size=(3, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(wr.random_weights(17, seed=8), x)
    p = tmp_path / "m.tflite"
    p.write_bytes(tflite_of(w, ()))
    lite = load_tflite(p)
    engine.set_cnn_math("fp16x2")
    net = wr.WRResNetDevice(engine, lite, 17)
    plain = wr.WRResNetDevice(engine, w, 17)
    assert all(b > 0 for b in net.act_bounds) and net.act_bounds != plain.act_bounds
    logits, _ = net.forward(torch.from_numpy(x).to(engine.device))
    assert not engine.cnn_last_overflow()
    want, _ = co.forward(w, x)
    assert float(np.abs(logits.cpu().numpy() - want).max()) <= LOGIT_ATOL
    # smooth inputs (what real crops look like) reach further than the noise probe: still inside the headroom
    xs = np.repeat(np.repeat(rng.uniform(0, 255, size=(2, 20, 20, 2)).astype(np.float32), 8, axis=1), 8, axis=2)
    net.forward(torch.from_numpy(xs).to(engine.device))
    assert not engine.cnn_last_overflow()
    net.close()
    plain.close()
    engine.set_cnn_math(engine.DEFAULT_CNN_MATH)
