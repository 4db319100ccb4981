import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def ft_switches():
    """Sets the library's BT_FT_* test switches for one test: the library reads them from the environment when it is
    loaded and again only on bt_internal_reload_env(), so the fixture sets, reloads, and puts both back afterwards."""
    import os

    from bullet_amd import lib
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            saved.setdefault(k, os.environ.get(k))
            os.environ[k] = str(v)
        lib.load().bt_internal_reload_env()

    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    lib.load().bt_internal_reload_env()
