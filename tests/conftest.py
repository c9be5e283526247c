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


def engine_key(fx):
    """bgn_amd PublicKey / SecretKey for a fixture (one engine per key per session)."""
    import bgn_amd
    name = fx["name"]
    if name not in _ENGINES:
        pk = bgn_amd.PublicKey(int(fx["p"], 16), int(fx["n"], 16), fx["l"], bytes.fromhex(fx["P"]),
                               bytes.fromhex(fx["Q"]), fx["msg_space"], True, fx["poly_base"])
        sk = bgn_amd.SecretKey(int(fx["q1"], 16))
        _ENGINES[name] = (pk, sk)
    return _ENGINES[name]
