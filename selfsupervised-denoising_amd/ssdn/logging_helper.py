"""Logging setup of the CLI / trainer (stands where /root/reference/ssdn/ssdn/logging_helper.py:16-88 stands): root logger ->
console + `<run_dir>/log.txt`.  colorlog / colored_traceback are optional dependencies of the reference that are absent here;
plain formatting is used.  `ScalarWriter` is the TensorBoard stand-in: the same `add_scalar(tag, value,
step)` calls land in `<run_dir>/scalars.csv`, and ALSO in a real SummaryWriter when tensorboard is importable."""
import logging
import os
import sys

_FMT = "%(asctime)s %(levelname)-8s %(name)s: %(message)s"
_file_handlers = {}


def setup(log_dir: str = None, filename: str = "log.txt", level=logging.INFO):
    root = logging.getLogger()
    root.setLevel(level)
    if not any(isinstance(h, logging.StreamHandler) and getattr(h, "_ssdn", False) for h in root.handlers):
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter(_FMT, "%H:%M:%S"))
        h._ssdn = True
        root.addHandler(h)
    if log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
        path = os.path.abspath(os.path.join(log_dir, filename))
        if path not in _file_handlers:
            fh = logging.FileHandler(path)
            fh.setFormatter(logging.Formatter(_FMT))
            root.addHandler(fh)
            _file_handlers[path] = fh


class ScalarWriter:
    def __init__(self, log_dir: str, purge_step: int = None):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.csv")
        if purge_step is not None and os.path.exists(self.path):       # resuming: drop records from the resume point onwards
            keep = [ln for i, ln in enumerate(open(self.path)) if i == 0 or int(ln.split(",")[1]) < purge_step]
            open(self.path, "w").writelines(keep)
        if not os.path.exists(self.path):
            open(self.path, "w").write("tag,step,value\n")
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir=log_dir, purge_step=purge_step)
        except Exception:
            pass

    def add_scalar(self, tag: str, value, step: int):
        v = float(value)
        with open(self.path, "a") as f:
            f.write("%s,%d,%.9g\n" % (tag, int(step), v))
        if self._tb is not None:
            self._tb.add_scalar(tag, v, step)

    def close(self):
        if self._tb is not None:
            self._tb.close()
