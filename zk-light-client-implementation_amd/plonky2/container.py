"""Circuit containers: a built circuit as a flat, versioned, language-neutral file (include/zklc.h section b'', csrc/container.cpp).

The reference builds a circuit once and proves with it many times (`get_ed25519_circuit_targets` / `ed25519_proof_reuse_circuit`,
near_bft_finality/src/prove_crypto/ed25519.rs:18-42 / :44-66; `recursive_proof`, recursion.rs:36-94 / :95).  A container is that
build -> prove seam as a file: the sections are exactly the arguments of `zklc_plonky2_circuit_create` and
`zklc_plonky2_witness_program_create`, so a program that only knows the C ABI (tests/c_abi/prove_from_file.c) proves from it, and a
Rust shim beside plonky2's `CircuitBuilder` can emit one.  The file is written and parsed by the library itself
(`zklc_plonky2_container_write`, `zklc_container_open`); this module only maps `CircuitData` to the sections and back.

What the host mirror needs on top of the native sections travels in three sections of its own (tags >= ZKLC_SEC_HOST_FIRST, ignored
by the library): a JSON note (config, gate ids, the caller's `aux` tree with targets replaced by table indices), the table of target
keys and the indices of the witness program's input targets.  Loading executes nothing: JSON + integer arrays (the cache of rounds
2-5 stored serialised Python objects).
"""
import ctypes
import json

import numpy as np

from .. import _lib
from . import gates as G
from .builder import CircuitData, Target, TargetRange, fri_reduction_arity_bits

SEC_PARAMS, SEC_GATES, SEC_GATE_EXTRA, SEC_K_IS, SEC_CONSTANTS, SEC_SIGMAS = 1, 2, 3, 4, 5, 6
SEC_WP_DIMS, SEC_WP_CODE, SEC_WP_PARAMS, SEC_WP_INPUT_SLOTS, SEC_WP_WIRE_SLOT, SEC_WP_WIRE_INDEX, SEC_WP_PI_SLOTS = 16, 17, 18, 19, 20, 21, 22
SEC_INPUT_VALUES = 32
SEC_HOST_META, SEC_HOST_TARGET_KEYS, SEC_HOST_INPUT_TARGETS = 0x1000, 0x1001, 0x1002
VERIFY = 1


class EntryC(ctypes.Structure):
    _fields_ = [("tag", ctypes.c_uint32), ("elem_bytes", ctypes.c_uint32), ("data", ctypes.c_void_p), ("bytes", ctypes.c_uint64)]


class DimsC(ctypes.Structure):
    _fields_ = [("code_len", ctypes.c_uint64), ("n_params", ctypes.c_uint64), ("n_wire_entries", ctypes.c_uint64),
                ("n_slots", ctypes.c_uint32), ("n_inputs", ctypes.c_uint32), ("num_wires", ctypes.c_uint32), ("n_rows", ctypes.c_uint32),
                ("n_pi", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class ContainerError(ValueError):
    pass


def _check(rc, what):
    if rc != 0:
        raise ContainerError("%s: status %d (%s)" % (what, rc, _lib.load().zklc_strerror(rc).decode()))


class Container:
    """A container file mapped read-only (zklc_container_open).  `section(tag)` returns a numpy view of the mapped bytes (no copy,
    not writeable) that stays valid while this object is alive."""

    def __init__(self, path, verify=True):
        self._lib = _lib.load()
        self.path = str(path)
        h = ctypes.c_void_p()
        _check(self._lib.zklc_container_open(self.path.encode(), VERIFY if verify else 0, ctypes.byref(h)), "open %s" % self.path)
        self._h = h

    def tags(self):
        out, e = [], EntryC()
        for i in range(self._lib.zklc_container_count(self._h)):
            _check(self._lib.zklc_container_entry_at(self._h, i, ctypes.byref(e)), "entry")
            out.append(int(e.tag))
        return out

    def section(self, tag, dtype=np.uint8, required=True):
        e = EntryC()
        rc = self._lib.zklc_container_find(self._h, tag, ctypes.byref(e))
        if rc != 0:
            if required:
                _check(rc, "section %d of %s" % (tag, self.path))
            return None
        dt = np.dtype(dtype)
        if e.bytes % dt.itemsize:
            raise ContainerError("section %d of %s: %d bytes are not a whole number of %s" % (tag, self.path, e.bytes, dt))
        if e.bytes == 0:
            return np.zeros(0, dtype=dt)
        buf = (ctypes.c_uint8 * e.bytes).from_address(e.data)
        a = np.frombuffer(buf, dtype=dt)
        a.flags.writeable = False
        return a                          # valid while this Container is: its owner (CircuitData._container) keeps it alive

    def release_pages(self):
        self._lib.zklc_container_release_pages(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zklc_container_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ aux trees <-> JSON
def _encode_aux(obj, table, memo):
    """nested dict / list / tuple of Targets, TargetRanges and plain values -> a JSON tree; every distinct Target OBJECT gets one
    index into `table` (identity is what the witness dictionaries key on, so it has to survive the round trip)"""
    if isinstance(obj, Target):
        i = memo.get(id(obj))
        if i is None:
            i = memo[id(obj)] = len(table)
            table.append(obj)
        return {"$t": i}
    if isinstance(obj, TargetRange):
        return {"$r": [obj.row, obj.col, obj.n]}
    if isinstance(obj, dict):
        if not all(isinstance(k, str) and not k.startswith("$") for k in obj):
            return {"$d": [[_encode_aux(k, table, memo), _encode_aux(v, table, memo)] for k, v in obj.items()]}
        return {k: _encode_aux(v, table, memo) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return {"$u": [_encode_aux(v, table, memo) for v in obj]}
    if isinstance(obj, list):
        return [_encode_aux(v, table, memo) for v in obj]
    if isinstance(obj, (bytes, bytearray)):
        return {"$b": bytes(obj).hex()}
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    raise TypeError("circuit container: cannot store a %s in the aux tree" % type(obj).__name__)


def _decode_aux(node, targets):
    if isinstance(node, dict):
        if "$t" in node:
            return targets[node["$t"]]
        if "$r" in node:
            return TargetRange(*node["$r"])
        if "$u" in node:
            return tuple(_decode_aux(v, targets) for v in node["$u"])
        if "$b" in node:
            return bytes.fromhex(node["$b"])
        if "$d" in node:
            return {_decode_aux(k, targets): _decode_aux(v, targets) for k, v in node["$d"]}
        return {k: _decode_aux(v, targets) for k, v in node.items()}
    if isinstance(node, list):
        return [_decode_aux(v, targets) for v in node]
    return node


def _target_from_key(k):
    from .builder import VIRTUAL_BASE
    k = int(k)
    return Target(idx=k - VIRTUAL_BASE) if k >= VIRTUAL_BASE else Target(k >> 8, k & 255)


# ------------------------------------------------------------------------------------------------ CircuitData -> file
def native_arguments(data, hasher=0):
    """the argument blocks of zklc_plonky2_circuit_create for `data` (also what Prover uploads): (ParamsC, GateC array, extra u64 array,
    k_is u64 array).  Everything a container's native sections hold besides the matrices."""
    from .prover import GateC, ParamsC
    from .builder import P, root_of_unity
    cfg = data.config
    fri = cfg["fri_config"]
    p = ParamsC()
    p.degree_bits, p.num_wires, p.num_routed_wires = data.degree_bits, cfg["num_wires"], cfg["num_routed_wires"]
    p.num_constants, p.num_selectors, p.num_challenges = data.num_constants, len(data.groups), cfg["num_challenges"]
    p.rate_bits, p.cap_height, p.proof_of_work_bits = fri["rate_bits"], fri["cap_height"], fri["proof_of_work_bits"]
    p.num_query_rounds = fri["num_query_rounds"]
    p.quotient_degree_factor, p.num_partial_products = data.quotient_degree_factor, data.num_partial_products
    p.num_gate_constraints, p.num_public_inputs = data.num_gate_constraints, data.num_public_inputs
    p.hasher, p.num_gates, p.num_arities = hasher, len(data.gates), len(data.fri_arity_bits)
    for i, a in enumerate(data.fri_arity_bits):
        p.arity_bits[i] = a
    gates = (GateC * len(data.gates))()
    extra = []
    for i, g in enumerate(data.gates):
        gates[i].type = g.code
        for k in range(4):
            gates[i].p[k] = g.params[k]
        s, e = data.groups[data.selector_indices[i]]
        gates[i].selector_index, gates[i].group_start, gates[i].group_end = data.selector_indices[i], s, e
        gates[i].extra_off = len(extra)
        if g.code == G.COSET_INTERPOLATION:   # barycentric weights, then the subgroup points
            w = root_of_unity(g.subgroup_bits)
            extra += list(g.weights) + [pow(w, j, P) for j in range(1 << g.subgroup_bits)]
    return p, gates, np.array(extra, dtype=np.uint64), np.array(data.k_is, dtype=np.uint64)


def write_circuit(path, data, aux=None, note=None):
    """`data`: a CircuitData (its witness program compiled or not); `aux`: whatever the caller needs back with the circuit (the
    input targets by name, ...): nested dicts / lists / tuples of Targets and plain values."""
    lib = _lib.load()
    p, gates, extra, kis = native_arguments(data)
    consts = np.ascontiguousarray(data.constants, dtype=np.uint64)
    sig = np.ascontiguousarray(data.sigmas, dtype=np.uint64)
    assert consts.shape == (data.num_constants, data.n) and sig.shape == (data.config["num_routed_wires"], data.n)
    table, memo = [], {}
    pr = data._program
    meta = {"format": "zklc circuit container, host note v1", "config": data.config, "gates": [g.id() for g in data.gates],
            "num_public_inputs": data.num_public_inputs, "has_program": pr is not None, "note": note}
    dims = None
    arrs = {}
    if pr is not None:
        in_idx = []
        for t in pr["input_targets"]:
            _encode_aux(t, table, memo)
            in_idx.append(memo[id(t)])
        arrs = {k: np.ascontiguousarray(pr[k], dtype=dt) for k, dt in (("code", np.uint32), ("params", np.int64), ("input_slots", np.uint32),
                                                                        ("wire_slot", np.uint32), ("wire_index", np.uint32), ("pi_slots", np.uint32))}
        dims = DimsC(len(arrs["code"]), len(arrs["params"]), len(arrs["wire_slot"]), int(pr["n_slots"]), len(arrs["input_slots"]),
                     data.config["num_wires"], data.n, len(arrs["pi_slots"]), 0)
        arrs["input_targets"] = np.array(in_idx, dtype=np.uint32)
    meta["aux"] = _encode_aux(aux, table, memo)
    keys = np.array([t.k for t in table], dtype=np.int64)
    mj = np.frombuffer(json.dumps(meta, separators=(",", ":")).encode(), dtype=np.uint8)
    host = [(SEC_HOST_META, 1, mj), (SEC_HOST_TARGET_KEYS, 8, keys)]
    if pr is not None:
        host.append((SEC_HOST_INPUT_TARGETS, 4, arrs["input_targets"]))
    ents = (EntryC * len(host))()
    for i, (tag, el, a) in enumerate(host):
        ents[i] = EntryC(tag, el, a.ctypes.data if a.size else None, a.nbytes)

    def ptr(name):
        a = arrs.get(name)
        return a.ctypes.data if a is not None and a.size else None
    rc = lib.zklc_plonky2_container_write(
        str(path).encode(), ctypes.byref(p), gates, extra.ctypes.data if extra.size else None, extra.size, kis.ctypes.data,
        consts.ctypes.data, sig.ctypes.data, ctypes.byref(dims) if dims is not None else None, ptr("code"), ptr("params"),
        ptr("input_slots"), ptr("wire_slot"), ptr("wire_index"), ptr("pi_slots"), ents, len(host))
    _check(rc, "write %s" % path)


def write_input_values(path, values):
    """a witness-input file: uint64 [n_witnesses, n_inputs] in the order of the program's inputs (section ZKLC_SEC_INPUT_VALUES)"""
    a = np.ascontiguousarray(values, dtype=np.uint64)
    e = (EntryC * 1)(EntryC(SEC_INPUT_VALUES, 8, a.ctypes.data if a.size else None, a.nbytes))
    _check(_lib.load().zklc_container_write(str(path).encode(), e, 1), "write %s" % path)


# ------------------------------------------------------------------------------------------------ file -> CircuitData
def read_circuit(path, verify=True):
    """-> (CircuitData, aux).  The matrices and the program arrays are views of the mapped file (read-only, no copy); the
    CircuitData keeps the mapping alive (`data._container`) and hands it to the prover, which creates the GPU circuit straight
    from the file's sections (`zklc_plonky2_circuit_create_from_container`)."""
    from .prover import ParamsC, GateC
    c = Container(path, verify=verify)
    lib = c._lib
    p, d = ParamsC(), DimsC()
    has_prog = SEC_WP_DIMS in c.tags()
    _check(lib.zklc_plonky2_container_params(c._h, ctypes.byref(p), ctypes.byref(d) if has_prog else None), "sections of %s" % path)
    meta = json.loads(bytes(c.section(SEC_HOST_META)).decode())
    gate_recs = c.section(SEC_GATES, dtype=np.dtype([("type", "<u4"), ("p", "<u4", 4), ("selector_index", "<u4"), ("group_start", "<u4"),
                                                     ("group_end", "<u4"), ("extra_off", "<u4")]))
    assert ctypes.sizeof(GateC) == gate_recs.dtype.itemsize
    cfg = meta["config"]
    gates = [G.gate_from_id(s) for s in meta["gates"]]
    n = 1 << p.degree_bits
    fri = cfg["fri_config"]
    # the note must describe the circuit the native sections hold: both are checked against each other, never trusted alone
    if (len(gates) != p.num_gates or cfg["num_wires"] != p.num_wires or cfg["num_routed_wires"] != p.num_routed_wires or
            cfg["num_challenges"] != p.num_challenges or fri["rate_bits"] != p.rate_bits or fri["cap_height"] != p.cap_height or
            fri["proof_of_work_bits"] != p.proof_of_work_bits or fri["num_query_rounds"] != p.num_query_rounds or
            meta["num_public_inputs"] != p.num_public_inputs or
            any(g.code != int(r["type"]) or tuple(g.params) != tuple(int(x) for x in r["p"]) for g, r in zip(gates, gate_recs))):
        raise ContainerError("%s: the host note does not describe the circuit of the native sections" % path)
    data = CircuitData.__new__(CircuitData)
    data.builder = data._plan = data._trace = None
    data._container = c
    data.config = cfg
    data.n, data.degree_bits = n, int(p.degree_bits)
    data.quotient_degree_factor = int(p.quotient_degree_factor)
    data.gates = gates
    data.selector_indices = [int(r["selector_index"]) for r in gate_recs]
    groups = {}
    for r in gate_recs:
        groups[int(r["selector_index"])] = (int(r["group_start"]), int(r["group_end"]))
    data.groups = [groups[i] for i in range(int(p.num_selectors))]
    data.num_constants = int(p.num_constants)
    data.constants = c.section(SEC_CONSTANTS, np.uint64).reshape(data.num_constants, n)
    data.sigmas = c.section(SEC_SIGMAS, np.uint64).reshape(int(p.num_routed_wires), n)
    data.num_gate_constraints = int(p.num_gate_constraints)
    data.num_partial_products = int(p.num_partial_products)
    data.k_is = [int(x) for x in c.section(SEC_K_IS, np.uint64)]
    data.fri_arity_bits = [int(p.arity_bits[i]) for i in range(int(p.num_arities))]
    if data.fri_arity_bits != fri_reduction_arity_bits(cfg, data.degree_bits):
        raise ContainerError("%s: FRI reduction arities differ from the configuration's strategy" % path)
    data.num_public_inputs = int(p.num_public_inputs)
    keys = c.section(SEC_HOST_TARGET_KEYS, np.int64)
    targets = [_target_from_key(k) for k in keys]
    data._program = None
    if has_prog:
        idx = c.section(SEC_HOST_INPUT_TARGETS, np.uint32)
        if len(idx) != d.n_inputs or (len(idx) and int(idx.max()) >= len(targets)):
            raise ContainerError("%s: input target table" % path)
        data._program = {
            "code": c.section(SEC_WP_CODE, np.uint32), "params": c.section(SEC_WP_PARAMS, np.int64),
            "input_targets": [targets[i] for i in idx], "input_slots": _opt(c, SEC_WP_INPUT_SLOTS),
            "wire_slot": _opt(c, SEC_WP_WIRE_SLOT), "wire_index": _opt(c, SEC_WP_WIRE_INDEX), "pi_slots": _opt(c, SEC_WP_PI_SLOTS),
            "n_slots": int(d.n_slots)}
    return data, _decode_aux(meta["aux"], targets)


def _opt(c, tag):
    a = c.section(tag, np.uint32, required=False)
    return np.zeros(0, dtype=np.uint32) if a is None else a
