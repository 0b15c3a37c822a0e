"""Classify service with the reference's wire format (reference src/classifyservice.py:27-172): a Unix stream socket,
one JSON job per connection

    {"file": "/path/clip.cptv", "cache": null, "track": true, "calculate_thumbnails": true}

answered with the metadata JSON of ClipClassifier.process_file, or {"error": "..."}; the connection is closed after the
answer.  Jobs run one at a time (a cpx handle is not thread safe; the reference starts a thread per connection around a
classifier that is not thread safe either).

    python -m cpx.classifyservice [--service_socket /etc/cacophony/thermal-classifier] [-c classifier.yaml]
"""

import argparse
import json
import logging
import os
import socket
import traceback

from .classify.clipclassifier import ClipClassifier
from .config import Config
from .ml_tools.tools import CustomJSONEncoder


class ClassifyJob:
    def __init__(self, file, cache, track, calculate_thumbnails):
        self.file, self.cache, self.track, self.calculate_thumbnails = file, cache, track, calculate_thumbnails

    @classmethod
    def from_dict(cls, file, cache, track, calculate_thumbnails):
        return cls(file=file, cache=cache, track=track, calculate_thumbnails=calculate_thumbnails)

    def as_dict(self):
        return dict(file=self.file, cache=self.cache, track=self.track, calculate_thumbnails=self.calculate_thumbnails)

    def __repr__(self):
        return "ClassifyJob(%r)" % self.as_dict()


def read_all(sock):
    size = 4096
    data = bytearray()
    while size > 0:
        packet = sock.recv(size)
        data.extend(packet)
        if len(packet) < size:
            break
    return data


def classify_job(clip_classifier, clientsocket, addr):
    job = None
    try:
        raw = read_all(clientsocket).decode()
        if len(raw) == 0:
            logging.info("Client disconnected")
            return
        logging.info("Received job %s", raw)
        try:
            job = ClassifyJob.from_dict(**json.loads(raw))
        except Exception as e:
            logging.error("Could not parse job", exc_info=True)
            clientsocket.sendall(json.dumps({"error": f"Could not parse job {e}"}).encode())
            return
        logging.info("Classifying %s", job)
        meta_data = clip_classifier.process_file(job.file, job.cache, track=job.track,
                                                 calculate_thumbnails=job.calculate_thumbnails)
        clientsocket.sendall(json.dumps(meta_data, cls=CustomJSONEncoder).encode())
    except BrokenPipeError:
        logging.error("Error sending metadata for job %s too %s", getattr(job, "file", None), addr, exc_info=True)
    except Exception:
        logging.error("Error classifying job %s", getattr(job, "file", None), exc_info=True)
        try:
            clientsocket.sendall(json.dumps({"error": f"Error classifying {traceback.format_exc()}"}).encode())
        except OSError:
            pass
    finally:
        try:
            clientsocket.close()
        except OSError:
            pass


class ClassifyService:
    def __init__(self, config):
        self.config = config
        self.clip_classifier = ClipClassifier(config)
        self._sock = None
        self._stop = False

    def run(self, service_socket):
        logging.info("Running on %s", service_socket)
        try:
            os.unlink(service_socket)
        except OSError:
            if os.path.exists(service_socket):
                raise
        sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        sock.bind(service_socket)
        sock.listen(1)
        self._sock = sock
        try:
            while not self._stop:
                logging.info("waiting for jobs")
                try:
                    connection, client_address = sock.accept()
                except OSError:
                    break  # closed by stop()
                classify_job(self.clip_classifier, connection, client_address)
        finally:
            sock.close()

    def stop(self):
        self._stop = True
        if self._sock is not None:
            try:
                self._sock.shutdown(socket.SHUT_RDWR)
            except OSError:
                pass
            self._sock.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--service_socket", default="/etc/cacophony/thermal-classifier", help="Socket name")
    ap.add_argument("-c", "--config-file", help="Path to config file to use")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    service = ClassifyService(Config.load_from_file(args.config_file))
    try:
        service.run(args.service_socket)
    except KeyboardInterrupt:
        logging.info("Keyboard interupt closing down")
    except PermissionError:
        logging.error("Error with permissions", exc_info=True)


if __name__ == "__main__":
    main()
