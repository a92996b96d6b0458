"""Scalar logging with the reference's TensorBoard tags (src/ppo.py:117-118,283-292).  Uses
``torch.utils.tensorboard.SummaryWriter`` when tensorboard is installed; otherwise an in-memory
recorder with the same ``add_scalar/add_text/close`` surface that also appends JSON lines to
``runs/<run_name>/scalars.jsonl``."""
from __future__ import annotations

import json
import os


class ScalarRecorder:
    def __init__(self, log_dir=None, write=True):
        self.scalars = []
        self.texts = []
        self._fh = None
        if log_dir and write:
            os.makedirs(log_dir, exist_ok=True)
            self._fh = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_text(self, tag, text, *a, **k):
        self.texts.append((tag, text))

    def add_scalar(self, tag, value, step):
        value = float(value)
        self.scalars.append((tag, value, int(step)))
        if self._fh:
            self._fh.write(json.dumps({"tag": tag, "value": value, "step": int(step)}) + "\n")

    def series(self, tag):
        return [(s, v) for (t, v, s) in self.scalars if t == tag]

    def close(self):
        if self._fh:
            self._fh.close()
            self._fh = None


def make_writer(log_dir, write=True):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir) if write else ScalarRecorder(None, False)
    except Exception:
        return ScalarRecorder(log_dir, write)
