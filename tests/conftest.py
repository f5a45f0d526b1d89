import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


_OPTION_DEFAULTS = None


@pytest.fixture(autouse=True)
def _launch_options_do_not_leak():
    """The library's launch options (`utx_set_option`) are process-global.  A test that leaves one changed silently moves every LATER test onto another
    kernel (round 6: a `finally` that reset UTX_ATTN_Q64 to its round-5 value 0 put the rest of an alphabetical run on the 8 x 32 attention loop).  So: the
    options as the library first reports them are the defaults; a test that ends with any other value FAILS, and the defaults are put back for the next."""
    global _OPTION_DEFAULTS
    import os
    env_before = {k: v for k, v in os.environ.items() if k.startswith("UTX_")}      # the Python host reads its UTX_* switches at use: a variable left set is the same kind of leak
    try:
        from unitex_amd import _lib
        now = _lib.get_options()
    except Exception:
        yield
        _check_env(env_before)
        return
    if _OPTION_DEFAULTS is None:
        _OPTION_DEFAULTS = dict(now)
    yield
    after = _lib.get_options()
    leaked = {k: (after[k], v) for k, v in _OPTION_DEFAULTS.items() if after.get(k) != v}
    for k, v in _OPTION_DEFAULTS.items():
        if after.get(k) != v:
            _lib.set_option(k, v)
    env_leaked = _restore_env(env_before)
    assert not leaked, "test left launch options changed (now, default): %r" % leaked
    assert not env_leaked, "test left UTX_* environment variables changed (now, before): %r" % env_leaked


def _restore_env(before):
    import os
    now = {k: v for k, v in os.environ.items() if k.startswith("UTX_")}
    changed = {k: (now.get(k), before.get(k)) for k in set(now) | set(before) if now.get(k) != before.get(k)}
    for k, (_, was) in changed.items():
        if was is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = was
    return changed


def _check_env(before):
    changed = _restore_env(before)
    assert not changed, "test left UTX_* environment variables changed (now, before): %r" % changed
