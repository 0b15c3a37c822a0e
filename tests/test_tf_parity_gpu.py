"""CNN logits against the exported TensorFlow model (BASELINE north_star: within 1e-3).  TensorFlow and the released
weights are absent from the build container, so this test is the hook a maintainer with TensorFlow uses:

    python tools/keras_to_npz.py <model.keras> /tmp/wr --dump-io 8        (wherever TensorFlow is installed)
    CPX_TF_MODEL=/tmp/wr CPX_TF_IO=/tmp/wr_io.npz python -m pytest tests/test_tf_parity_gpu.py -m gpu

Skipped (and the a20 row stays "parity unpinned") while those files do not exist."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not (os.environ.get("CPX_TF_MODEL") and os.environ.get("CPX_TF_IO")),
                    reason="needs a converted TensorFlow model and its input/output dump (tools/keras_to_npz.py --dump-io)")
def test_hip_forward_matches_tensorflow_logits():
    import torch

    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    w = wr.load_weights(os.environ["CPX_TF_MODEL"] + ".npz")
    io = np.load(os.environ["CPX_TF_IO"])
    eng = TrackEngine()
    net = wr.WRResNetDevice(eng, w, int(w["prediction/bias"].shape[0]))
    logits, probs = net.forward(torch.from_numpy(np.ascontiguousarray(io["x"], np.float32)).to(eng.device))
    assert float(np.abs(logits.cpu().numpy() - io["logits"]).max()) <= 1e-3      # the north-star tolerance
    assert float(np.abs(probs.cpu().numpy() - io["probs"]).max()) <= 1e-3
    net.close()
    eng.close()
