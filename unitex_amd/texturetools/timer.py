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


class Encoders:
    """The image codecs of a stage (PIL's PNG / JPEG encoders: C code that releases the GIL) side by side on host threads.  A stage that writes
    several artefacts submits them and leaves the `with` block only when all are on disk, so the file hand-off between stages (reference
    pipeline.py: every stage takes and leaves paths) is unchanged; only the encoders of ONE stage overlap each other and the GPU work the stage
    issues meanwhile.  On the reference operating point the codecs were 1.1 s of a 16.7 s mesh (profiles/r02_host_profile_pipeline.log)."""

    def __init__(self, workers=8):
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.jobs = []

    def submit(self, fn, *a, **k):
        f = self.pool.submit(fn, *a, **k)
        self.jobs.append(f)
        return f

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        try:
            for f in self.jobs:
                f.result()          # re-raises an encoder's exception in the stage that owns it
        finally:
            self.pool.shutdown(wait=True)
        return False
