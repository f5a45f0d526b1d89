"""CPUTimer -- stage timing decorator with the reference's print format (utils/timer.py:14-31)."""
import functools
from time import perf_counter


class CPUTimer:
    def __init__(self, name="", synchronize=False):
        self.name, self.synchronize = name, synchronize

    def __call__(self, fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            t = perf_counter()
            out = fn(*a, **k)
            if self.synchronize:
                import torch
                torch.cuda.synchronize()
            print(">>> %s %.4f >>>" % (self.name, perf_counter() - t))
            return out
        return wrapped
