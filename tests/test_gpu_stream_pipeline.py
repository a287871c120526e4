"""zklc_amd.pipeline.BlockPipeline -- the measured path of bench.py -- against the sequential driver (prove_bft.BlockProver, the
reference's `prove_block_bft` restated: near_bft_finality/src/prove_bft/bft.rs:38-500): the SAME nodes of the SAME DAG, so the proofs
must be IDENTICAL (the prover is deterministic: no blinding in the reference's configs), on the reference's mainnet window (5 blocks)
and epoch data (6 blocks, two approval sets); consecutive blocks overlapped (`prove_stream`) give the same bytes again; the
Poseidon-BN128 wrap is accepted by the verifier restatement; an invalid approval raises like the reference's panic."""
import json

import pytest

import conftest
from conftest import load_golden
from oracle import plonky2_verifier as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pipe(zctx):
    from zklc_amd.pipeline import BlockPipeline
    p = BlockPipeline(0, wrap=True)
    yield p
    p.close()


def _serial(block_prover, w, key):
    got = conftest.STASH.get(key)
    if got is None:          # the sequential tests stash their results when they ran earlier in the session
        a, kw = w.bft_args()
        got = conftest.STASH[key] = block_prover.prove_block_bft(*a, w.validators, **kw)
    return got


def _same(p, q):
    from zklc_amd.plonky2 import HASH_GL, serialization as S
    assert p[0] == q[0] and p[1] == q[1], "circuit (common / verifier data) differs"
    assert S.proof_to_bytes(p[2], p[0], HASH_GL) == S.proof_to_bytes(q[2], q[0], HASH_GL), "proof bytes differ"


def test_pipelined_block_is_byte_identical_to_the_sequential_driver(pipe, block_prover):
    import time
    from zklc_amd.pipeline import BlockWindow
    from zklc_amd.plonky2 import HASH_BN128, serialization as S
    w = BlockWindow.from_fixture(load_golden("block_window_HPi5.json"))
    t0 = time.time()
    res = pipe.prove_block_bft(w)
    t1 = time.time()
    assert res.block_n_1 is None and res.block[2]["public_inputs"] == w.expected_public_inputs()[0]
    serial, _ = _serial(block_prover, w, "HPi5")
    _same(res.block, serial)
    wrc, wraw = res.wrap
    wj = S.proof_from_bytes(wraw, wrc.common, HASH_BN128)
    V.verify(json.loads(json.dumps(wj)), wrc.verifier_only, wrc.common)
    assert wj["public_inputs"] == w.expected_public_inputs()[0] and len(wraw) == 127968
    # three consecutive blocks, overlapped: every one complete and identical
    done = []
    t2 = time.time()
    rs = pipe.prove_stream([w] * 3, done.append)
    t3 = time.time()
    assert len(rs) == 3 and done == rs
    for r in rs:
        _same(r.block, serial)
        assert r.wrap[1] == wraw
        assert r.t0 < r.t_signatures < r.t_done
    assert rs[1].t0 < rs[0].t_done, "block 2's signature stage must start before block 1 is complete"
    # the same through the reference-shaped entry point
    a, kw = w.bft_args()
    old, block_prover._pipeline = block_prover._pipeline, pipe
    try:
        bi, none = block_prover.prove_block_bft(*a, w.validators, pipelined=True, **kw)
    finally:
        block_prover._pipeline = old
    assert none is None
    _same(bi, serial)
    print("pipelined block: first call %.1f s (circuits built), then 3 overlapped blocks in %.1f s" % (t1 - t0, t3 - t2))


def test_pipelined_epoch_blocks_are_byte_identical(pipe, block_prover):
    """6-block branch: two approval sets (80 and 75 signatures, two validator lists) through the same signature stage"""
    from zklc_amd.pipeline import BlockWindow
    w = BlockWindow.from_fixture(load_golden("block_window_epoch_CRTZ.json"))
    res = pipe.prove_block_bft(w)
    want = w.expected_public_inputs()
    assert res.block[2]["public_inputs"] == want[0] and res.block_n_1[2]["public_inputs"] == want[1]
    assert len(res.aggregates) == 2
    b0, bn_1 = _serial(block_prover, w, "epoch_CRTZ")
    _same(res.block, b0)
    _same(res.block_n_1, bn_1)


def test_invalid_approval_raises_and_the_pipeline_survives(pipe):
    from conftest import near_set_arrays
    from zklc_amd.signatures import InvalidSignature
    msg, approvals, validators = near_set_arrays(load_golden("ed25519_near_c1_small.json"))
    bad = list(approvals)
    i = next(k for k, a in enumerate(bad) if len(a) == 66)
    bad[i] = bad[i][:10] + bytes([bad[i][10] ^ 1]) + bad[i][11:]
    with pytest.raises(InvalidSignature):
        pipe.prove_approvals(msg, bad, validators)
    (rc, raw), valid_keys = pipe.prove_approvals(msg, approvals, validators)
    from zklc_amd.plonky2 import HASH_GL, serialization as S
    proof = S.proof_from_bytes(raw, rc.common, HASH_GL)
    V.verify(json.loads(json.dumps(proof)), rc.verifier_only, rc.common)
    import hashlib
    assert proof["public_inputs"] == list(hashlib.sha256(valid_keys).digest())
