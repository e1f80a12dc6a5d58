import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fr():
    """The product package (ctypes binding of libfleetrec.so).  Built on demand when hipcc is present."""
    import __graft_entry__ as g
    mod = g.load_package()
    if not os.path.exists(mod.LIB_PATH):
        g.build()
    return mod


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    import __graft_entry__ as g
    return g.load_oracle()


@pytest.fixture(scope="session")
def gpu(fr):
    if fr.device_count() < 1:
        pytest.fail("gpu-marked test running without a visible HIP device (no CPU fallback exists)")
    return 0


@pytest.fixture(scope="session")
def ctxs(fr, gpu):
    """Full-size Model A / B / C contexts (1.4 / 15.1 / 63.2 GB of tables), created lazily, shared by every tests/test_gpu_*.py module of the
    session (a test that changes a context's precision, chain width or tables puts them back in a `finally`)."""
    from gpu_helpers import SEED_TABLES, SEED_WEIGHTS
    cache = {}

    def get(which):
        if which not in cache:
            m = fr.Model.builtin(which)
            c = fr.Context(m, device=gpu)
            c.fill_tables(fr.FILL_HASH, SEED_TABLES)
            c.fill_weights(fr.WEIGHTS_UNIFORM, SEED_WEIGHTS)
            cache[which] = (m, c)
        return cache[which]

    yield get
    for m, c in cache.values():
        c.close()


def free_port_block(n=1):
    """A base port such that base .. base+n-1 could all be bound right now (the GPU host's network namespace may be shared with
    other jobs, so fixed port numbers are not safe)."""
    import random
    import socket
    for _ in range(200):
        base = random.randint(20000, 60000 - n)
        socks = []
        try:
            for i in range(n):
                sk = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                sk.bind(("0.0.0.0", base + i))
                socks.append(sk)
            return base
        except OSError:
            continue
        finally:
            for sk in socks:
                sk.close()
    raise RuntimeError("no free block of %d ports found" % n)
