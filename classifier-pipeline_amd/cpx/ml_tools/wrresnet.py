"""WR-ResNet-22-4 (grouped pre-activation wide ResNet) weights container and the
device forward pass (reference architecture src/ml_tools/resnet/wr_resnet.py:5-98, head
src/ml_tools/kerasmodel.py:308-350).

Weights are kept in Keras layouts under the Keras layer names so that an exported model can
be converted 1:1:
  conv  '<name>/kernel' [kh, kw, Cin/groups, Cout] (HWIO), '<name>/bias' [Cout]
  bn    '<name>/gamma|beta|moving_mean|moving_variance' [C]      (epsilon = 1e-3)
  dense 'prediction/kernel' [C, n_labels], 'prediction/bias'
Layer names: conv1_1; res{s}b{d}_branch2a|2b, bn{s}b{d}_branch2a|2b for stage s = 2..4, block
d = 0..2; shortcut{s} (the 1x1 projection of each stage's first block); final_bn; prediction.
Head variants of KerasModel.build_model (kerasmodel.py:337-345): hidden Dense(relu) layers
'dense_{i}/kernel|bias' (hyperparams.dense_sizes, in order), and the output activation in
'prediction/activation' ("sigmoid", the multi_label default, or "softmax").
tools/keras_to_npz.py writes this layout from an exported Keras model."""

import ctypes as C
import json

import numpy as np

from .._lib import CpxError, WRResNetParams

BN_EPS = 1e-3  # tf.keras.layers.BatchNormalization default
FILTERS = (16, 64, 128, 256)
GROUPS = 2
BLOCKS = 3


def layer_plan(n_labels):
    """-> list of (name, kind, shape) in forward order."""
    plan = [("conv1_1", "conv", (3, 3, 2 // GROUPS, FILTERS[0]))]
    cin = FILTERS[0]
    for stage in (2, 3, 4):
        f = FILTERS[stage - 1]
        for d in range(BLOCKS):
            b = "b%d" % d
            c_in = cin if d == 0 else f
            plan.append(("bn%d%s_branch2a" % (stage, b), "bn", (c_in,)))
            plan.append(("res%d%s_branch2a" % (stage, b), "conv", (3, 3, c_in // GROUPS, f)))
            plan.append(("bn%d%s_branch2b" % (stage, b), "bn", (f,)))
            plan.append(("res%d%s_branch2b" % (stage, b), "conv", (3, 3, f // GROUPS, f)))
            if d == 0:
                plan.append(("shortcut%d" % stage, "conv", (1, 1, c_in // GROUPS, f)))
        cin = f
    plan.append(("final_bn", "bn", (cin,)))
    plan.append(("prediction", "dense", (cin, n_labels)))
    return plan


def random_weights(n_labels=17, seed=0, dense_sizes=None, activation="sigmoid"):
    """Seeded Glorot-uniform kernels, small random biases and BatchNorm statistics (no checkpoint
    can be downloaded here): same architecture, same layouts."""
    rng = np.random.default_rng(seed)
    w = {}
    plan = layer_plan(n_labels)
    if dense_sizes:
        head = plan.pop()
        width = head[2][0]
        for i, size in enumerate(dense_sizes):
            plan.append(("dense_%d" % i, "dense", (width, int(size))))
            width = int(size)
        plan.append(("prediction", "dense", (width, n_labels)))
    if activation != "sigmoid":
        w["prediction/activation"] = activation
    for name, kind, shape in plan:
        if kind == "conv":
            kh, kw, ci, co = shape
            fan_in, fan_out = kh * kw * ci, kh * kw * co // GROUPS
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w[name + "/kernel"] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            w[name + "/bias"] = rng.normal(0, 0.05, size=co).astype(np.float32)
        elif kind == "bn":
            (c,) = shape
            w[name + "/gamma"] = rng.uniform(0.7, 1.3, size=c).astype(np.float32)
            w[name + "/beta"] = rng.normal(0, 0.1, size=c).astype(np.float32)
            w[name + "/moving_mean"] = rng.normal(0, 0.1, size=c).astype(np.float32)
            w[name + "/moving_variance"] = rng.uniform(0.5, 1.5, size=c).astype(np.float32)
        else:
            ci, co = shape
            lim = np.sqrt(6.0 / (ci + co))
            w[name + "/kernel"] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            w[name + "/bias"] = rng.normal(0, 0.05, size=co).astype(np.float32)
    return w


def save_model(path_base, weights, labels, hyperparams=None, thresholds=None, model_type="thermal"):
    """<path>.npz (weights) + <path>.json (labels / hyperparams sidecar as interpreter.py:23-41 reads it)."""
    np.savez(str(path_base) + ".npz", **weights)
    meta = {"labels": list(labels), "hyperparams": dict(hyperparams or {}), "thresholds": thresholds,
            "type": model_type, "version": "cpx-wr-resnet-22-4"}
    with open(str(path_base) + ".json", "w") as fh:
        json.dump(meta, fh, indent=1)


def load_weights(path):
    z = np.load(str(path))
    return {k: (str(z[k]) if z[k].dtype.kind in "US" else np.asarray(z[k], dtype=np.float32)) for k in z.files}


def head_of(weights):
    """-> (hidden layer names in order, activation) of a weights dict."""
    hidden = []
    while "dense_%d/kernel" % len(hidden) in weights:
        hidden.append("dense_%d" % len(hidden))
    act = str(weights.get("prediction/activation", "sigmoid"))
    if act not in ("sigmoid", "softmax"):
        raise NotImplementedError("output activation %r" % act)
    return hidden, act


def bn_affine(w, name):
    """BatchNorm in inference mode as y = x * scale + shift."""
    scale = w[name + "/gamma"] / np.sqrt(w[name + "/moving_variance"] + np.float32(BN_EPS))
    shift = w[name + "/beta"] - w[name + "/moving_mean"] * scale
    return scale.astype(np.float32), shift.astype(np.float32)


def activation_bound(w, name, sigmas=64.0):
    """Upper bound of relu(BatchNorm(x)) for the fp16x2 math mode's range scaling (include/cpx.h:
    cpx_cnn_set_activation_bounds): the normalised input is within `sigmas` standard deviations, so the output is
    within |beta| + sigmas |gamma|; an input that is not costs a rerun of the layer, never correctness."""
    return float(np.max(np.abs(w[name + "/beta"]) + sigmas * np.abs(w[name + "/gamma"])))


def pack_conv(kernel):
    """Keras HWIO [kh,kw,Cin/g,Cout] -> [g][kh*kw][Cin/g][Cout/g] contiguous."""
    kh, kw, ci, co = kernel.shape
    cog = co // GROUPS
    k = kernel.reshape(kh * kw, ci, GROUPS, cog)
    return np.ascontiguousarray(np.transpose(k, (2, 0, 1, 3)), dtype=np.float32)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "Cin", "Cout", "groups", "ksize", "stride", "pad_same", "relu")] + \
               [(n, C.c_void_p) for n in ("in_dev", "out_dev", "weights_dev", "in_scale_dev", "in_shift_dev",
                                          "out_scale_dev", "out_shift_dev", "residual_dev")]


assert C.sizeof(ConvDesc) == 104


class WRResNetDevice:
    """The network resident on one GPU: the parameters are uploaded once and handed to the native network
    object (cpx_cnn_create); forward() is one cpx_cnn_forward call = 23 kernel launches on the engine's stream."""

    def __init__(self, engine, weights, n_labels):
        self.eng = engine
        self.lib = engine.lib
        self.torch = t = engine.torch
        self.n_labels = n_labels
        dev = engine.device

        def up(a):
            return t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        self.p = {}
        w = weights
        self.p["conv1_1/w"] = up(pack_conv(w["conv1_1/kernel"]))
        self.p["conv1_1/b"] = up(w["conv1_1/bias"])
        for stage in (2, 3, 4):
            for d in range(BLOCKS):
                b = "%db%d" % (stage, d)
                sa, ha = bn_affine(w, "bn%s_branch2a" % b)
                sb, hb = bn_affine(w, "bn%s_branch2b" % b)
                self.p["%s/in_scale" % b] = up(sa)
                self.p["%s/in_shift" % b] = up(ha)
                self.p["%s/wa" % b] = up(pack_conv(w["res%s_branch2a/kernel" % b]))
                # conv2a bias + the following BatchNorm folded: (acc + bias) * sb + hb
                self.p["%s/a_scale" % b] = up(sb)
                self.p["%s/a_shift" % b] = up(w["res%s_branch2a/bias" % b] * sb + hb)
                self.p["%s/wb" % b] = up(pack_conv(w["res%s_branch2b/kernel" % b]))
                self.p["%s/bb" % b] = up(w["res%s_branch2b/bias" % b])
            self.p["sc%d/w" % stage] = up(pack_conv(w["shortcut%d/kernel" % stage]))
            self.p["sc%d/b" % stage] = up(w["shortcut%d/bias" % stage])
        fs, fh = bn_affine(w, "final_bn")
        self.p["final/scale"], self.p["final/shift"] = up(fs), up(fh)
        self.p["dense/w"] = up(w["prediction/kernel"])
        self.p["dense/b"] = up(w["prediction/bias"])
        self.hidden, self.activation = head_of(w)
        if len(self.hidden) > 4:
            raise NotImplementedError("more than 4 hidden dense layers")
        for name in self.hidden:
            self.p[name + "/w"] = up(w[name + "/kernel"])
            self.p[name + "/b"] = up(w[name + "/bias"])
        self._bufs = {}
        self._cnn = None
        # what the 3x3 convolutions' activated inputs can reach, in launch order (fp16x2 range scaling): from the
        # BatchNorm parameters where the archive has them; a model whose BatchNorms arrive folded into scale / shift (a
        # .tflite: moving_variance is the reader's identity marker) says nothing about its activations' spread -- there
        # the bounds are measured on a seeded probe batch, with headroom (_measure_activation_bounds)
        names = ["bn%db%d_branch2%s" % (stage, d, ab) for stage in (2, 3, 4) for d in range(BLOCKS) for ab in "ab"]
        self.act_bounds = [activation_bound(w, n) for n in names]
        folded = [bool(np.allclose(w[n + "/moving_variance"], 1.0 - BN_EPS) and np.all(w[n + "/moving_mean"] == 0))
                  for n in names]
        self._create_native()
        if any(folded):
            measured = self._measure_activation_bounds()
            self.act_bounds = [m if f else b for b, m, f in zip(self.act_bounds, measured, folded)]
            self._set_activation_bounds()

    def _create_native(self):
        ptr = lambda key: self.p[key].data_ptr()
        prm = WRResNetParams()
        prm.n_labels, prm.blocks_per_stage, prm.groups, prm.in_channels = self.n_labels, BLOCKS, GROUPS, 2
        for i, f in enumerate(FILTERS):
            prm.filters[i] = f
        prm.conv1_w, prm.conv1_b = ptr("conv1_1/w"), ptr("conv1_1/b")
        for si, stage in enumerate((2, 3, 4)):
            for d in range(BLOCKS):
                b = "%db%d" % (stage, d)
                blk = prm.block[si][d]
                blk.in_scale, blk.in_shift = ptr("%s/in_scale" % b), ptr("%s/in_shift" % b)
                blk.wa, blk.a_scale, blk.a_shift = ptr("%s/wa" % b), ptr("%s/a_scale" % b), ptr("%s/a_shift" % b)
                blk.wb, blk.bb = ptr("%s/wb" % b), ptr("%s/bb" % b)
            prm.shortcut_w[si], prm.shortcut_b[si] = ptr("sc%d/w" % stage), ptr("sc%d/b" % stage)
        prm.final_scale, prm.final_shift = ptr("final/scale"), ptr("final/shift")
        prm.dense_w, prm.dense_b = ptr("dense/w"), ptr("dense/b")
        prm.n_hidden = len(self.hidden)
        prm.activation = 1 if self.activation == "softmax" else 0
        for k, name in enumerate(self.hidden):
            prm.hidden_sizes[k] = int(self.p[name + "/b"].shape[0])
            prm.hidden_w[k], prm.hidden_b[k] = ptr(name + "/w"), ptr(name + "/b")
        out = C.c_void_p()
        rc = self.lib.cpx_cnn_create(self.eng.h, C.byref(prm), C.byref(out))
        if rc != 0:
            raise CpxError(rc, self.eng._err())
        self._cnn = out
        self._set_activation_bounds()

    def _set_activation_bounds(self):
        bounds = (C.c_float * len(self.act_bounds))(*self.act_bounds)
        rc = self.lib.cpx_cnn_set_activation_bounds(self._cnn, bounds, len(self.act_bounds))
        if rc != 0:
            raise CpxError(rc, self.eng._err())

    def _measure_activation_bounds(self, headroom=32.0, n=2, seed=12345):
        """The largest activated input of every 3x3 convolution on a seeded probe batch (uniform 0..255 tiles through the
        network layer by layer, exact-split math), times `headroom`: deterministic for a given model; an input that goes
        beyond it costs the overflow rerun, never correctness."""
        t = self.torch
        rng = np.random.default_rng(seed)
        x = t.from_numpy(rng.uniform(0, 255, size=(n, 160, 160, 2)).astype(np.float32)).to(self.eng.device)
        prev = self.eng.get_cnn_math()
        self.eng.set_cnn_math("bf16x3")
        seen = []
        try:
            self.forward_layerwise(x, want_probs=False, _probe=seen)
        finally:
            self.eng.set_cnn_math(prev)
        return [max(float(v) * headroom, 1e-3) for v in seen]

    def close(self):
        if self._cnn is not None:
            if self.eng.h:  # a closed engine has already freed its networks
                self.lib.cpx_cnn_destroy(self._cnn)
            self._cnn = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _buf(self, key, shape):
        t = self.torch
        b = self._bufs.get(key)
        n = int(np.prod(shape))
        if b is None or b.numel() < n:
            b = t.empty(n, dtype=t.float32, device=self.eng.device)
            self._bufs[key] = b
        return b[:n].view(*shape)

    def _conv(self, x, out, wkey, N, H, W, Cin, Cout, ksize, stride, same, relu, in_affine=None, out_scale=None,
              out_shift=None, residual=None):
        ptr = lambda v: None if v is None else C.c_void_p(v.data_ptr())
        d = ConvDesc(N, H, W, Cin, Cout, GROUPS, ksize, stride, 1 if same else 0, 1 if relu else 0,
                     ptr(x), ptr(out), ptr(self.p[wkey]),
                     ptr(self.p[in_affine + "/in_scale"]) if in_affine else None,
                     ptr(self.p[in_affine + "/in_shift"]) if in_affine else None,
                     ptr(out_scale), ptr(out_shift), ptr(residual))
        rc = self.lib.cpx_conv2d(self.eng.h, C.byref(d))
        if rc != 0:
            raise CpxError(rc, self.eng._err())

    def forward(self, x, want_probs=True):
        """x: device float32 [N, S, S, 2] (values 0..255, no input scaling) -> (logits, probs) [N, n_labels]."""
        t = self.torch
        N, H, W, cin = x.shape
        assert cin == 2 and x.dtype == t.float32 and x.is_contiguous()
        t.cuda.current_stream(self.eng.device).synchronize()
        logits = t.empty((N, self.n_labels), dtype=t.float32, device=self.eng.device)
        probs = t.empty((N, self.n_labels), dtype=t.float32, device=self.eng.device) if want_probs else None
        rc = self.lib.cpx_cnn_forward(self._cnn, C.c_void_p(x.data_ptr()), N, H, W, C.c_void_p(logits.data_ptr()),
                                      C.c_void_p(probs.data_ptr()) if probs is not None else None)
        if rc != 0:
            raise CpxError(rc, self.eng._err())
        self.eng.synchronize()
        return logits, probs

    def forward_async(self, x, logits, probs=None):
        """Enqueue the forward of x into caller-provided outputs on the engine's stream; no host synchronisation
        (the caller orders x's producer with an event and synchronises the engine before reading)."""
        N, H, W, cin = x.shape
        assert cin == 2 and x.is_contiguous() and logits.is_contiguous()
        for attempt in range(2):
            rc = self.lib.cpx_cnn_forward(self._cnn, C.c_void_p(x.data_ptr()), N, H, W, C.c_void_p(logits.data_ptr()),
                                          C.c_void_p(probs.data_ptr()) if probs is not None else None)
            if rc == -6 and attempt == 0:   # CPX_ERR_NOMEM: the handle's activation buffers are a plain hipMalloc, and what
                # torch's allocator holds in its cache is invisible to it -- hand that back and try once more
                self.eng.synchronize()
                self.torch.cuda.empty_cache()
                continue
            break
        if rc != 0:
            raise CpxError(rc, self.eng._err())

    def forward_layerwise(self, x, want_probs=True, _probe=None):
        """The same network issued layer by layer through cpx_conv2d / cpx_cnn_head (what a caller binding the
        building blocks directly would write); tests compare it with forward()."""
        t = self.torch
        N, H, W, cin = x.shape
        assert cin == 2 and x.dtype == t.float32 and x.is_contiguous()
        t.cuda.current_stream(self.eng.device).synchronize()
        cur = self._buf("act0", (N, H, W, FILTERS[0]))
        self._conv(x, cur, "conv1_1/w", N, H, W, 2, FILTERS[0], 3, 1, True, False, out_shift=self.p["conv1_1/b"])
        c_in, flip = FILTERS[0], 0
        for stage in (2, 3, 4):
            f = FILTERS[stage - 1]
            stride = stage - 1  # wr_block(stride=stage) with stage index 1..3 (wr_resnet.py:27-30)
            for d in range(BLOCKS):
                b = "%db%d" % (stage, d)
                s = stride if d == 0 else 1
                Ho, Wo = -(-H // s), -(-W // s)
                mid = self._buf("mid", (N, Ho, Wo, f))
                if _probe is not None:  # the activated input of branch2a: relu(BatchNorm 2a(block input))
                    self.eng.synchronize()
                    _probe.append(t.relu(cur * self.p["%s/in_scale" % b] + self.p["%s/in_shift" % b]).max().item())
                self._conv(cur, mid, "%s/wa" % b, N, H, W, c_in, f, 3, s, True, True, in_affine=b,
                           out_scale=self.p["%s/a_scale" % b], out_shift=self.p["%s/a_shift" % b])
                if _probe is not None:  # branch2b reads `mid` as it is
                    self.eng.synchronize()
                    _probe.append(mid.max().item())
                if d == 0:
                    sc = self._buf("sc", (N, Ho, Wo, f))
                    self._conv(cur, sc, "sc%d/w" % stage, N, H, W, c_in, f, 1, s, False, False,
                               out_shift=self.p["sc%d/b" % stage])
                    res = sc
                else:
                    res = cur
                flip ^= 1
                nxt = self._buf("act%d" % flip, (N, Ho, Wo, f))
                self._conv(mid, nxt, "%s/wb" % b, N, Ho, Wo, f, f, 3, 1, True, True, out_shift=self.p["%s/bb" % b],
                           residual=res)
                cur, H, W, c_in = nxt, Ho, Wo, f
        logits = t.empty((N, self.n_labels), dtype=t.float32, device=self.eng.device)
        probs = t.empty((N, self.n_labels), dtype=t.float32, device=self.eng.device) if want_probs else None
        from .._lib import HeadDesc

        hd = HeadDesc()
        hd.N, hd.HW, hd.C, hd.L = N, H * W, c_in, self.n_labels
        hd.n_hidden, hd.activation = len(self.hidden), 1 if self.activation == "softmax" else 0
        for k, name in enumerate(self.hidden):
            hd.hidden_sizes[k] = int(self.p[name + "/b"].shape[0])
            hd.hidden_w_dev[k], hd.hidden_b_dev[k] = self.p[name + "/w"].data_ptr(), self.p[name + "/b"].data_ptr()
        hd.in_dev = cur.data_ptr()
        hd.bn_scale_dev, hd.bn_shift_dev = self.p["final/scale"].data_ptr(), self.p["final/shift"].data_ptr()
        hd.dense_w_dev, hd.dense_b_dev = self.p["dense/w"].data_ptr(), self.p["dense/b"].data_ptr()
        hd.logits_dev = logits.data_ptr()
        hd.probs_dev = probs.data_ptr() if probs is not None else None
        rc = self.lib.cpx_cnn_head_ex(self.eng.h, C.byref(hd))
        if rc != 0:
            raise CpxError(rc, self.eng._err())
        self.eng.synchronize()
        return logits, probs


def calibrate_bn_device(engine, weights, x, n_labels=17, var_floor=1e-2):
    """Set-up utility for SYNTHETIC networks (bench.py, probes; a converted model brings its own statistics): set every
    BatchNorm's moving mean / variance to the statistics of the calibration batch x (device float32 [N, S, S, 2]) as
    it flows through the HIP convolutions layer by layer -- what training would have left behind -- so that seeded
    random kernels give O(1) activations and logits instead of magnitudes no trained network has.  Returns a new
    weights dict; the per-channel reductions are torch calls on the calibration tensors (set-up, not the hot path)."""
    t = engine.torch
    w = dict(weights)
    net = WRResNetDevice(engine, w, n_labels)
    dev = engine.device

    def up(a):
        return t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

    def fit(tensor, name):
        engine.synchronize()
        flat = tensor.reshape(-1, tensor.shape[-1]).double()
        mean = flat.mean(dim=0)
        var = flat.var(dim=0, unbiased=False)
        t.cuda.synchronize()
        w[name + "/moving_mean"] = mean.float().cpu().numpy()
        w[name + "/moving_variance"] = (var.float() + var_floor).cpu().numpy()
        return bn_affine(w, name)

    N, H, W, _ = x.shape
    t.cuda.current_stream(dev).synchronize()
    cur = net._buf("act0", (N, H, W, FILTERS[0]))
    net._conv(x, cur, "conv1_1/w", N, H, W, 2, FILTERS[0], 3, 1, True, False, out_shift=net.p["conv1_1/b"])
    c_in, flip = FILTERS[0], 0
    for stage in (2, 3, 4):
        f = FILTERS[stage - 1]
        for d in range(BLOCKS):
            b = "%db%d" % (stage, d)
            s = (stage - 1) if d == 0 else 1
            Ho, Wo = -(-H // s), -(-W // s)
            sa, ha = fit(cur, "bn%s_branch2a" % b)
            net.p["%s/in_scale" % b].copy_(up(sa))
            net.p["%s/in_shift" % b].copy_(up(ha))
            t.cuda.synchronize()
            mid = net._buf("mid", (N, Ho, Wo, f))
            bias_a = up(w["res%s_branch2a/bias" % b])
            net._conv(cur, mid, "%s/wa" % b, N, H, W, c_in, f, 3, s, True, False, in_affine=b, out_shift=bias_a)
            sb, hb = fit(mid, "bn%s_branch2b" % b)
            net.p["%s/a_scale" % b].copy_(up(sb))
            net.p["%s/a_shift" % b].copy_(up(w["res%s_branch2a/bias" % b] * sb + hb))
            t.cuda.synchronize()
            net._conv(cur, mid, "%s/wa" % b, N, H, W, c_in, f, 3, s, True, True, in_affine=b,
                      out_scale=net.p["%s/a_scale" % b], out_shift=net.p["%s/a_shift" % b])
            if d == 0:
                res = net._buf("sc", (N, Ho, Wo, f))
                net._conv(cur, res, "sc%d/w" % stage, N, H, W, c_in, f, 1, s, False, False, out_shift=net.p["sc%d/b" % stage])
            else:
                res = cur
            flip ^= 1
            nxt = net._buf("act%d" % flip, (N, Ho, Wo, f))
            net._conv(mid, nxt, "%s/wb" % b, N, Ho, Wo, f, f, 3, 1, True, True, out_shift=net.p["%s/bb" % b], residual=res)
            cur, H, W, c_in = nxt, Ho, Wo, f
    fit(cur, "final_bn")
    net.close()
    return w
