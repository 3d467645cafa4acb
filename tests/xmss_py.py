"""TEST INFRASTRUCTURE: the XMSS / WOTS scheme of leanmultisig_amd/xmss.py (crates/xmss restated) with every hash computed by
the CPU oracle, whose permutation is pinned by the reference's KAT.  Checked by tests/test_xmss_witness.py: verify(sign(..))
accepts, tampering rejects, the encoding has the target sum."""
import numpy as np

from leanmultisig_amd.xmss import (CHAIN_LENGTH, LOG_LIFETIME, MESSAGE_LEN, NUM_CHAIN_HASHES, P, PP_LEN, RANDOMNESS_LEN, TARGET_SUM,  # noqa: F401
                                   TWEAK_CHAIN, TWEAK_ENCODING, TWEAK_MERKLE, TWEAK_WOTS_PK, V, W, XMSS_DIGEST_LEN, make_tweak)
from leanmultisig_amd.xmss import Xmss as _Xmss


class Xmss(_Xmss):
    def __init__(self, orc):
        super().__init__(compress=lambda x: orc.poseidon16_compress(np.ascontiguousarray(x, dtype=np.uint32).reshape(-1, 16)))
        self.orc = orc
