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
import os as _os

# Hardware queues.  The HIP runtime maps the normal-priority streams of a process onto GPU_MAX_HW_QUEUES queues (default 4); streams
# that share a queue are serialised.  BlockPipeline keeps six of them busy (three Ed25519 provers, the witness producer, keys /
# stakes, the caller's own) beside three high-priority ones: with 4 queues two prover streams shared one (profiles/r06e_*: all
# signature-proof kernels on two queues).  8 queues: steady state 5.33 -> 5.02 s per block (profiles/r06g_*).  Read by the runtime
# when it initialises, so it must be in the environment before the first HIP call of the process: set here at import, never
# overriding the caller's own value.  A Rust / Go host exports it itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import ZklcError, load, LIB_PATH, declared_symbols  # noqa: F401,E402
from .context import Context  # noqa: F401,E402
from . import signatures  # noqa: F401
from . import distributed  # noqa: F401
from . import header_bphash  # noqa: F401
from . import primitives  # noqa: F401
from . import keys_stakes  # noqa: F401
from . import prove_bft  # noqa: F401
from . import pipeline  # noqa: F401
