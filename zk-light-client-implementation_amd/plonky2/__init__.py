"""plonky2 prover front-end of the MI355X backend (host side).

Mirrors the surface the reference uses on the un-vendored plonky2 fork for the signature-aggregation path:
`CircuitBuilder` / `CircuitData::prove` (near_bft_finality/src/prove_crypto/ed25519.rs:26-39,60,100,
recursion.rs:36,94-95) and the proof files written by near_bft_finality/src/bin/prove_block.rs:320-458.
"""
from . import gates  # noqa: F401
from .builder import CircuitBuilder, CircuitData, Target, standard_recursion_config, wide_ecc_config  # noqa: F401
from . import serialization  # noqa: F401
from .prover import Prover, poseidon_gate_rows, HASH_GL, HASH_BN128  # noqa: F401
