#!/usr/bin/env python3
"""Shapes of every circuit of a Block_i proof, computed on the CPU (no GPU, no proofs): the DAG of zklc_amd/prove_bft.py
(= near_bft_finality/src/prove_bft/{block_finality,bft}.rs) is walked with provers that BUILD each circuit -- the reference's
leaf circuits and the in-circuit verifier of every `recursive_proof` -- and hand on the public inputs a proof would carry.
Prints rows used / padded degree per node and checks the last recursion (bin/prove_block.rs:279-287, Poseidon-BN128 config)
against the reference's golden final proof: near_bft_finality/proofs/random/CGZP.../common_data.json (degree_bits 12, 97 public
inputs, 13 gate types) -- the shape a gnark verifier circuit compiled from that file accepts.

    python tools/dag_shapes.py [--no-ed25519]     # ~3-5 minutes of host Python (SHA-256 circuits of 2^17 / 2^18 rows)
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zklc_amd  # noqa: E402,F401
from zklc_amd.plonky2 import builder as B, recursion as R  # noqa: E402
from zklc_amd import signatures as SG  # noqa: E402

SHAPES = []


def rows_used(data):
    return sum(1 for g, _ in data.builder.rows if g.id() != "NoopGate")


class ShapeProver:
    def __init__(self, ctx, data, hasher=0):
        self.data, self.hasher, self.common = data, hasher, data.common_data()

    def verifier_data(self):
        return {"constants_sigmas_cap": [{"elements": [0] * 4} for _ in range(16)], "circuit_digest": {"elements": [0] * 4}}

    def prove(self, wires, pis):
        return {"public_inputs": [int(x) for x in pis]}

    def close(self):
        pass


def shape_recursive_proof(self, first, second=None, public_inputs=None, raw=False):
    inners = [first] + ([second] if second is not None else [])
    pis = [int(x) for x in (public_inputs or [])]
    t0 = time.time()
    n_before = len(self._cache)
    rc = self.circuit_for([c for c, _, _ in inners], len(pis))
    if len(self._cache) != n_before:
        SHAPES.append(("recursive_proof(%s)%s" % (", ".join("2^%d" % c["fri_params"]["degree_bits"] for c, _, _ in inners),
                                                   " + %d PI" % len(pis) if pis else ""),
                       rows_used(rc.data), rc.data.degree_bits, len(rc.data.gates), len(rc.data.builder._const_targets), time.time() - t0))
    return rc, {"public_inputs": pis}


class ShapeApprovals:
    """prove_approvals (signatures.rs:43-141) as shapes: the Ed25519 circuit, the fold's two shapes, the closing proof"""

    def __init__(self, ed=True):
        from zklc_amd.plonky2 import HASH_GL
        self.recursion = R.RecursionProver(None, HASH_GL)
        self.ed = ed

    def prove_approvals(self, msg, approvals, validators):
        pos, pks, _ = SG.slice_approvals(approvals, validators)
        vk = b"".join(bytes([p & 0xFF]) + pk.tobytes() for p, pk in zip(pos, pks))
        if self.ed:
            from zklc_amd.plonky2 import CircuitBuilder, wide_ecc_config, ed25519_circuit as E
            t0 = time.time()
            b = CircuitBuilder(wide_ecc_config())
            E.ed25519_circuit(b, 8 * len(msg))
            data = b.build()
            SHAPES.append(("ed25519_circuit(%d-byte message)" % len(msg), rows_used(data), data.degree_bits, len(data.gates),
                           len(b._const_targets), time.time() - t0))
            ed = (data.common_data(), None, {"public_inputs": [0] * data.num_public_inputs})
        else:
            ed = (json.load(open(os.path.join(ROOT, "tests", "golden", "ed25519_common_2p18.json"))), None, {"public_inputs": []})
        agg = ed
        for _ in range(3):
            rc, p = self.recursion.recursive_proof(agg, ed)
            agg = (rc.common, None, p)
        rc, p = self.recursion.recursive_proof(agg, None, list(hashlib.sha256(vk).digest()))
        return (rc, p), vk

    def close(self):
        pass


def main():
    B.CircuitData.prover = lambda self, ctx, hasher=0: ShapeProver(ctx, self, hasher)
    R.RecursionProver.recursive_proof = shape_recursive_proof
    from zklc_amd.plonky2 import HASH_BN128, sha256 as SHA
    from zklc_amd.prove_bft import BlockProver
    from zklc_amd import primitives as PR, keys_stakes as KS
    # leaf provers report their circuits as they build them
    for cls, meth in ((SHA.Sha256Prover, "circuit_for"),):
        orig = getattr(cls, meth)

        def wrapped(self, n, _orig=orig):
            k = len(self._circuits)
            t0 = time.time()
            ent = _orig(self, n)
            if len(self._circuits) != k:
                SHAPES.append(("sha256_proof_u32(%d bytes)" % n, rows_used(ent[0]), ent[0].degree_bits, len(ent[0].gates),
                               len(ent[0].builder._const_targets), time.time() - t0))
            return ent
        setattr(cls, meth, wrapped)
    w = json.load(open(os.path.join(ROOT, "tests", "golden", "block_window_HPi5.json")))
    hx = bytes.fromhex
    blocks = []
    for blk in w["blocks"]:
        f = {k: hx(blk[k]) for k in ("hash", "prev_hash", "epoch_id", "last_ds_final_hash", "last_final_hash")}
        f["height"] = blk["height"]
        f["approvals"] = [hx(a) for a in blk["approvals"]]
        blocks.append((f, hx(blk["bytes"])))
    validators = [hx(v) for v in w["validators"]]
    bp = BlockProver(None, ShapeApprovals(ed="--no-ed25519" not in sys.argv))
    orig_p = PR.PrimitiveProver._prove

    def prim(self, key, build, values):
        k = len(self._cache)
        r = orig_p(self, key, build, values)
        if len(self._cache) != k:
            d = self._cache[key][0]
            SHAPES.append(("primitive %s" % (key,), rows_used(d), d.degree_bits, len(d.gates), len(d.builder._const_targets), 0.0))
        return r
    PR.PrimitiveProver._prove = prim
    bi, _ = bp.prove_block_bft(hx(w["ep2_last_block"]["bytes"]), hx(w["ep2_last_block"]["hash"]), hx(w["ep1_first_block"]["bytes"]),
                               hx(w["ep1_first_block"]["hash"]), blocks, validators)
    for e in bp.keys._cache:
        SHAPES.append(("keys / stakes circuit (%d validators, %d valid keys)" % (len(e[0][1]), len(e[0][0])), rows_used(e[1]),
                       e[1].degree_bits, len(e[1].gates), len(e[1].builder._const_targets), 0.0))
    if "--dump-inner" in sys.argv:       # the Block_i circuit's common data: the input shape of the wrap circuit (tools/wrap_instance.py)
        with open(sys.argv[sys.argv.index("--dump-inner") + 1], "w") as f:
            json.dump(bi[0], f, separators=(",", ":"))
    wrap = R.RecursionProver(None, HASH_BN128)
    wrc, _ = wrap.recursive_proof(bi, None, list(bi[2]["public_inputs"]))
    print("%-52s %9s %7s %6s %7s %7s" % ("circuit (first occurrence in the DAG)", "rows", "degree", "gates", "consts", "build s"))
    for name, rows, db, ng, nc, dt in SHAPES:
        print("%-52s %9d   2^%-3d %6d %7d %7.1f" % (name, rows, db, ng, nc, dt))
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "plonky2_near_random_CGZP.json")))["common_data"]
    same = {k: wrc.common[k] for k in golden} == golden
    print("Block_i proof (recursion for the three hashes, bft.rs:376-387): 2^%d; BN128 wrap: 2^%d, %d public inputs" % (
        bi[0]["fri_params"]["degree_bits"], wrc.data.degree_bits, wrc.data.num_public_inputs))
    print("wrap common_data == golden common_data.json (all %d keys of the reference file held by the fixture): %s" % (len(golden), same))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
