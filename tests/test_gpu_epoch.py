"""The 6-block (epoch) branch of `prove_block_bft` on the GPU: near_bft_finality/src/prove_bft/bft.rs:206-255,317-561, the path of
bin/prove_epoch.rs:222-232, on the reference's epoch data (tests/golden/block_window_epoch_CRTZ.json: Block_0 of epoch CRTZ.. and
Block_n-1 of epoch HPi5.., borsh headers pinned by their block hashes, each block with its own validator set and epoch ancestors).
Two finality proofs -- 80 and 75 approval signatures -- nine header-hash chains, two keys / stakes proofs, the joins; both final
proofs are accepted by the verifier restatement and carry [1, hash(block), hash(ancestor), hash(ancestor)]."""
import json

import pytest

from conftest import load_golden
from oracle import plonky2_verifier as V

pytestmark = pytest.mark.gpu


def test_epoch_blocks_proof_on_the_gpu(zctx, block_prover):
    import time
    w = load_golden("block_window_epoch_CRTZ.json")
    hx = bytes.fromhex
    blocks = []
    for blk in w["blocks"]:
        f = {k: hx(blk[k]) for k in ("hash", "prev_hash", "epoch_id", "last_ds_final_hash", "last_final_hash")}
        f["height"] = blk["height"]
        f["approvals"] = [hx(a) for a in blk["approvals"]]
        blocks.append((f, hx(blk["bytes"])))
    bp = block_prover
    bp.counts, bp.seconds = {}, {}
    t0 = time.time()
    b0, bn_1 = bp.prove_block_bft(hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                                  hx(w["ep1_first_block"]["hash"]), blocks, [hx(v) for v in w["validators"]],
                                  ep3_last_block_bytes=hx(w["ep3_last_block"]["bytes"]), ep3_last_block_hash=hx(w["ep3_last_block"]["hash"]),
                                  validators_n_1=[hx(v) for v in w["validators_n_1"]])
    print("epoch blocks: %.1f s (circuits of the shapes not seen earlier in the session are built in Python); counts %s" % (time.time() - t0, bp.counts))
    import conftest
    conftest.STASH["epoch_CRTZ"] = (b0, bn_1)
    for proof in (b0, bn_1):
        V.verify(json.loads(json.dumps(proof[2])), proof[1], proof[0])
    assert b0[2]["public_inputs"] == [1] + list(hx(w["blocks"][4]["hash"])) + list(hx(w["ep2_last_block"]["hash"])) + \
        list(hx(w["ep1_first_block"]["hash"]))
    assert bn_1[2]["public_inputs"] == [1] + list(hx(w["blocks"][5]["hash"])) + list(hx(w["ep3_last_block"]["hash"])) + \
        list(hx(w["ep2_last_block"]["hash"]))
    assert bp.counts["prove_header_hash"] == 9 and bp.counts["prove_approvals"] == 2
