"""zk-light-client-implementation_amd -- MI355X (gfx950) kernels for the NEAR
zk-light-client signature-aggregation hot path, behind the C ABI of
``include/zklc.h`` (``lib/libzklc_mi355.so``).

Import as ``import zklc_amd`` (alias module at the repo root; the directory
name carries the reference's hyphenated name and is not a Python identifier).

Host-side mirror of the reference interface for this path:
  * ``Context``                      -- one per process-per-GPU rank (zklc_ctx)
  * ``signatures.generate_signed_message`` / ``signatures.verify_approvals``
        near_bft_finality/src/prove_block_data/signatures.rs:24-39, 56-123
  * ``signatures.ApprovalProver`` / ``prove_bft.BlockProver``  -- `prove_approvals` / `prove_block_bft`, the reference's
        sequential drivers (signatures.rs:43-141, prove_bft/bft.rs:38-500)
  * ``pipeline.BlockPipeline``       -- the same DAG with several proofs in flight per GPU (the measured path of bench.py)
There is no CPU fallback anywhere in this package.
"""
from ._lib import ZklcError, load, LIB_PATH, declared_symbols  # noqa: F401
from .context import Context  # noqa: F401
from . import signatures  # noqa: F401
from . import distributed  # noqa: F401
from . import header_bphash  # noqa: F401
from . import primitives  # noqa: F401
from . import keys_stakes  # noqa: F401
from . import prove_bft  # noqa: F401
from . import pipeline  # noqa: F401
