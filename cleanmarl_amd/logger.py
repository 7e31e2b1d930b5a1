"""Scalar logging with the reference's tags and x-axis (cleanmarl/mappo_multienvs.py:357-362, 461-468, 605-612, 642-650).

Uses torch.utils.tensorboard.SummaryWriter when tensorboard is installed (W&B then piggybacks through
``sync_tensorboard=True`` exactly like the reference, :350-356); otherwise degrades to a JSONL file
``runs/<name>/scalars.jsonl`` with the same (tag, value, step) triples.
"""
import json
import os
import time


class ScalarWriter:
    def __init__(self, logdir):
        self.logdir = logdir
        os.makedirs(logdir, exist_ok=True)
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter  # noqa: WPS433 (optional dependency)
            self._tb = SummaryWriter(logdir)
        except Exception:  # tensorboard not installed
            self._tb = None
        self._f = open(os.path.join(logdir, "scalars.jsonl"), "a")
        self.history = []

    def add_scalar(self, tag, value, step):
        value, step = float(value), int(step)
        self.history.append((tag, value, step))
        self._f.write(json.dumps({"tag": tag, "value": value, "step": step, "wall": time.time()}) + "\n")
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def add_text(self, tag, text):
        with open(os.path.join(self.logdir, tag + ".md"), "w") as f:
            f.write(text)
        if self._tb is not None:
            self._tb.add_text(tag, text)

    def close(self):
        self._f.close()
        if self._tb is not None:
            self._tb.close()
