"""bgn_amd — MI355X-native batched BGN homomorphic-op engine.

Host-side mirror of sachaservan/bgn's PublicKey / SecretKey / Ciphertext
surface (bgn.go, ciphertext.go, poly.go, gadgets.go) on top of the C ABI in
include/bgn_amd.h.  All arithmetic runs in hand-written HIP kernels
(bgn_amd/csrc); there is no CPU fallback.
"""
from .api import (Ciphertext, DecryptionProof, Engine, MultiEngine, NewDecryptionProof, PolyCiphertext,  # noqa: F401
                  ProofOfPlaintextKnowledge, PublicKey, SecretKey)
from ._lib import BgnError  # noqa: F401
