import json
import os
import sys

import pytest

try:                       # torch ships its own HIP runtime: load it BEFORE libbgn_amd.so pulls in /opt/rocm's, so that
    import torch  # noqa: F401   # the process holds one runtime (a torch.cuda init after the engine's found no GPU)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
KEYS = ["toy64", "k256", "k512", "k1024"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_fixture(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", params=KEYS)
def fx(request):
    return load_fixture(request.param)


def oracle_key(fx):
    """Rebuild the oracle's PublicKey / SecretKey from a fixture."""
    import bgn_ref as R
    p, n = int(fx["p"], 16), int(fx["n"], 16)
    P = R.elem_from_bytes(bytes.fromhex(fx["P"]), p)
    Q = R.elem_from_bytes(bytes.fromhex(fx["Q"]), p)
    pk = R.PublicKey(p=p, n=n, l=fx["l"], P=P, Q=Q, MsgSpace=fx["msg_space"], Deterministic=True,
                     PolyBase=fx["poly_base"])
    sk = R.SecretKey(Key=int(fx["q1"], 16))
    return pk, sk


_ENGINES = {}
_PENDING = {}        # options of the running test (engopts below): applied to every engine the test looks up


def _option_name(name):
    return name[4:].lower() if name.startswith("BGN_") else name


def _apply_pending(eng):
    eng.reset_options()
    for k, v in _PENDING.items():
        eng.set_option(k, v)


def engine_key(fx):
    """bgn_amd PublicKey / SecretKey for a fixture (one engine per key per session)."""
    import bgn_amd
    name = fx["name"]
    if name not in _ENGINES:
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]),
                               bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"])
        sk = bgn_amd.SecretKey(int(fx["q1"], 16))
        _ENGINES[name] = (pk, sk)
    _apply_pending(_ENGINES[name][0].engine)
    return _ENGINES[name]


class EngineOptions:
    """What a test uses to select kernels and alternatives: every setting goes through bgn_ctx_set_option
    (include/bgn_amd.h) on the engines of the session — those that exist and those the test looks up later — and is
    taken back when the test ends.  The library does not read the environment after a context exists."""

    def __init__(self):
        self.extra = []

    def _engines(self):
        return [pk.engine for pk, _ in _ENGINES.values()] + self.extra

    def register(self, eng):
        """An engine the test created itself."""
        self.extra.append(eng)
        for k, v in _PENDING.items():
            eng.set_option(k, v)
        return eng

    def set(self, name, value):
        _PENDING[_option_name(name)] = int(value)
        for e in self._engines():
            e.set_option(_option_name(name), int(value))

    def unset(self, name):
        _PENDING.pop(_option_name(name), None)
        for e in self._engines():
            _apply_pending(e)

    def force(self, kernel):
        """'coop', 'quad' or 'lane' for every batch size of every operation; None: dispatch by size."""
        names = ("coop_max", "coop_max_l2", "coop_max_dec", "quad_max", "quad_max_l2", "quad_max_dec", "quad_max_pow",
                 "quad_max_mc", "quad_min")
        if kernel is None:
            for k in names:
                self.unset(k)
            return
        big = 1 << 40
        self.set("quad_min", 0)
        for k in names[:3]:
            self.set(k, big if kernel == "coop" else 0)
        for k in names[3:8]:
            self.set(k, big if kernel == "quad" else 0)


@pytest.fixture
def engopts():
    _PENDING.clear()
    o = EngineOptions()
    yield o
    _PENDING.clear()
    for e in o._engines():
        try:
            e.reset_options()
        except Exception:
            pass
