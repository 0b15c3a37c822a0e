"""Model server with the reference's wire format (reference src/piclassifier/servemodel.py:17-36):

    GET  /ready    -> {"ready": true}
    POST /predict  body = raw little-endian float32 [N, H, W, C]  ->  raw float32 [N, n_labels]

One request at a time (the reference serves with threads=1: the classifier is not thread safe; neither is a cpx handle).
The predictions come from the HIP network (cpx_cnn_forward); there is no CPU model.

    python -m cpx.servemodel -m model.npz [--port 8123]
"""

import argparse
import json
import logging
from http.server import BaseHTTPRequestHandler, HTTPServer

import numpy as np


def make_server(interpreter, port, host="127.0.0.1"):
    num_inputs, in_shape = interpreter.shape()
    if num_inputs > 1:
        raise ValueError("Not support multiple input models")
    sample_shape = tuple(in_shape[1:])

    class Handler(BaseHTTPRequestHandler):
        def log_message(self, fmt, *args):  # quiet: the reference logs through its own logger
            logging.debug("servemodel: " + fmt, *args)

        def _send(self, code, body, mimetype):
            self.send_response(code)
            self.send_header("Content-Type", mimetype)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)

        def do_GET(self):
            if self.path == "/ready":
                self._send(200, json.dumps({"ready": interpreter is not None}).encode(), "application/json")
            else:
                self._send(404, b"not found", "text/plain")

        def do_POST(self):
            if self.path != "/predict":
                self._send(404, b"not found", "text/plain")
                return
            data = self.rfile.read(int(self.headers.get("Content-Length", 0)))
            per = int(np.prod(sample_shape)) * 4
            if len(data) == 0 or len(data) % per:
                self._send(400, b"body is not a whole number of float32 samples", "text/plain")
                return
            x = np.frombuffer(data, dtype="<f4").reshape((-1,) + sample_shape)
            try:
                predictions = np.ascontiguousarray(interpreter.predict(x), dtype="<f4")
            except Exception as e:  # errors over the wire as text, the server stays up
                logging.error("predict failed", exc_info=True)
                self._send(500, repr(e).encode(), "text/plain")
                return
            self._send(200, predictions.tobytes(), "application/octet-stream")

    server = HTTPServer((host, port), Handler)  # single threaded by construction
    return server


def startup_classifier(interpreter):
    """Classify an empty frame to force the model onto the device (servemodel.py:39-49)."""
    _, in_shape = interpreter.shape()
    interpreter.predict(np.zeros((1,) + tuple(in_shape[1:]), np.float32))


def main(argv=None):
    from .config.config import ModelConfig
    from .ml_tools.interpreter import get_interpreter

    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--model-file", required=True, help="<name>.npz (+ <name>.json sidecar)")
    ap.add_argument("--port", type=int, default=8123)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    interpreter = get_interpreter(ModelConfig.load({"id": 1, "name": "served", "model_file": args.model_file,
                                                    "port": args.port}))
    startup_classifier(interpreter)
    server = make_server(interpreter, args.port)
    logging.info("serving %s on 127.0.0.1:%d", args.model_file, args.port)
    server.serve_forever()


if __name__ == "__main__":
    main()
