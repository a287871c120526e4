"""Circuit containers (include/zklc.h section b'', csrc/container.cpp, zklc_amd/plonky2/container.py): the flat file between
`builder.build()` and `data.prove(pw)` (near_bft_finality/src/prove_crypto/ed25519.rs:26-39 -> :60, recursion.rs:94 -> :95).
CPU tests: the file functions are host code.  What the GPU does with a container is in tests/test_gpu_c_abi.py."""
import ctypes
import hashlib
import os
import struct
import subprocess

import numpy as np
import pytest

from zklc_amd import _lib
from zklc_amd.plonky2 import container as C
from zklc_amd.plonky2 import sha256 as SHA
from zklc_amd.plonky2.builder import CircuitData, Target

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sha_circuit():
    msg = bytes(range(70))
    data, words = SHA.sha256_circuit(len(msg))
    data.witness_program(list(words))
    return data, words, msg


def test_round_trip_is_the_same_circuit_and_program(tmp_path, sha_circuit):
    data, words, msg = sha_circuit
    path = tmp_path / "sha256.zkcc"
    data.save(path, {"words": words, "n": 7, "t": (words[0], [words[1], None]), "raw": b"\x01\x02"})
    d2, aux = CircuitData.load(path)
    assert d2.builder is None and d2._container is not None
    for name in ("constants", "sigmas"):
        a, b = getattr(data, name), getattr(d2, name)
        assert a.shape == b.shape and np.array_equal(a, b) and not b.flags.writeable
    assert d2.common_data() == data.common_data() and [g.id() for g in d2.gates] == [g.id() for g in data.gates]
    assert (d2.n, d2.degree_bits, d2.groups, d2.selector_indices, d2.num_partial_products) == \
        (data.n, data.degree_bits, data.groups, data.selector_indices, data.num_partial_products)
    for k in ("code", "params", "input_slots", "wire_slot", "wire_index", "pi_slots"):
        assert np.array_equal(data._program[k], d2._program[k]), k
    assert d2._program["n_slots"] == data._program["n_slots"]
    # the aux tree comes back with the program's input targets as the SAME objects (witness dictionaries key on identity)
    assert all(a is b for a, b in zip(aux["words"], d2._program["input_targets"]))
    assert aux["n"] == 7 and aux["raw"] == b"\x01\x02" and aux["t"][0] is aux["words"][0] and aux["t"][1] == [aux["words"][1], None]
    assert [t.k for t in aux["words"]] == [t.k for t in words]
    # the witness from the loaded program == the witness from the built one == sha256
    want = [int.from_bytes(hashlib.sha256(msg).digest()[4 * i:4 * i + 4], "big") for i in range(8)]
    w1, p1 = data.generate_witness_native([SHA.sha256_witness(words, msg)])
    w2, p2 = d2.generate_witness_native([SHA.sha256_witness(aux["words"], msg)])
    assert [int(x) for x in p2[0]] == want and np.array_equal(w1, w2) and np.array_equal(p1, p2)


def test_the_library_runs_the_program_straight_from_the_file(tmp_path, sha_circuit):
    """zklc_plonky2_witness_run_from_container (host interpreter over the mapped sections) == CircuitData.generate_witness_native;
    the parameter blocks the library reports are the ones written"""
    data, words, msg = sha_circuit
    path = tmp_path / "c.zkcc"
    data.save(path, {"words": words})
    lib = _lib.load()
    c = C.Container(path)
    from zklc_amd.plonky2.prover import ParamsC
    p, d = ParamsC(), C.DimsC()
    assert lib.zklc_plonky2_container_params(c._h, ctypes.byref(p), ctypes.byref(d)) == 0
    assert (p.degree_bits, p.num_wires, p.num_routed_wires, p.num_public_inputs) == (data.degree_bits, 135, 80, 8)
    assert (d.n_inputs, d.n_pi, d.num_wires, d.n_rows) == (len(words), 8, 135, data.n)
    vals = np.array([SHA.padded_words(msg), SHA.padded_words(bytes(70))], dtype=np.uint64)
    wires = np.zeros((2, 135, data.n), dtype=np.uint64)
    pis = np.zeros((2, 8), dtype=np.uint64)
    status = np.zeros(2, dtype=np.int32)
    err = ctypes.create_string_buffer(400)
    rc = lib.zklc_plonky2_witness_run_from_container(c._h, vals.ctypes.data, 2, wires.ctypes.data, pis.ctypes.data, status.ctypes.data, err, 2)
    assert rc == 0 and not status.any()
    w_ref, p_ref = data.generate_witness_native(None, input_values=vals)
    assert np.array_equal(wires, w_ref) and np.array_equal(pis, p_ref)
    assert [int(x) for x in pis[1]] == [int.from_bytes(hashlib.sha256(bytes(70)).digest()[4 * i:4 * i + 4], "big") for i in range(8)]
    c.close()


def _sections(path):
    raw = open(path, "rb").read()
    magic, ver, n, size, th = struct.unpack_from("<8sIIQQ", raw, 0)
    tab = [struct.unpack_from("<IIQQQ", raw, 64 + 32 * i) for i in range(n)]
    return raw, magic, ver, n, size, tab


def test_layout_is_the_documented_one(tmp_path, sha_circuit):
    """header / table / 64-byte aligned payload as include/zklc.h describes them -- what a writer in another language has to emit (or,
    simpler, call zklc_plonky2_container_write for)"""
    data, words, _ = sha_circuit
    path = tmp_path / "c.zkcc"
    data.save(path, None)
    raw, magic, ver, n, size, tab = _sections(path)
    assert magic == b"ZKLCCIRC" and ver == 1 and size == len(raw)
    tags = [t[0] for t in tab]
    assert tags[:5] == [C.SEC_PARAMS, C.SEC_GATES, C.SEC_K_IS, C.SEC_CONSTANTS, C.SEC_SIGMAS] and C.SEC_WP_CODE in tags
    assert all(off % 64 == 0 and off + nb <= len(raw) for _, _, off, nb, _ in tab)
    by = {t[0]: t for t in tab}
    _, el, off, nb, _ = by[C.SEC_SIGMAS]
    assert el == 8 and nb == 80 * data.n * 8 and np.array_equal(np.frombuffer(raw, "<u8", 80 * data.n, off).reshape(80, -1), data.sigmas)
    _, el, off, nb, _ = by[C.SEC_PARAMS]
    assert struct.unpack_from("<3I", raw, off) == (data.degree_bits, 135, 80)


@pytest.mark.parametrize("what", ["magic", "version", "truncated", "extended", "payload", "table"])
def test_damaged_files_are_refused(tmp_path, sha_circuit, what):
    data, words, _ = sha_circuit
    path = tmp_path / "c.zkcc"
    data.save(path, None)
    raw, _, _, _, _, tab = _sections(path)
    b = bytearray(raw)
    if what == "magic":
        b[0] ^= 1
    elif what == "version":
        b[8] = 9
    elif what == "truncated":
        b = b[:-64]
    elif what == "extended":
        b += bytes(64)
    elif what == "payload":
        off = [t for t in tab if t[0] == C.SEC_SIGMAS][0][2]
        b[off + 1000] ^= 0x10
    elif what == "table":
        b[64 + 8] ^= 1                      # a section offset
    bad = tmp_path / "bad.zkcc"
    bad.write_bytes(bytes(b))
    with pytest.raises(C.ContainerError):
        C.Container(bad, verify=True)
    if what == "payload":                   # only the checksum pass reads the payload
        C.Container(bad, verify=False).close()


def test_sections_that_contradict_each_other_are_refused(tmp_path, sha_circuit):
    """a file whose checksums are fine but whose sections do not fit the parameters (a writer's bug, or a hostile file) must not
    reach the create functions: ZKLC_ERR_FORMAT from zklc_plonky2_container_params"""
    data, words, _ = sha_circuit
    lib = _lib.load()
    p, gates, extra, kis = C.native_arguments(data)
    pr = data._program
    dims = C.DimsC(len(pr["code"]), len(pr["params"]), len(pr["wire_slot"]), pr["n_slots"], len(pr["input_slots"]), 135, data.n, len(pr["pi_slots"]), 0)

    def write(path, sig=None, wire_index=None, pi_slots=None, dims_=None, code=None):
        sig = data.sigmas if sig is None else sig
        code_ = pr["code"] if code is None else code
        wi = pr["wire_index"] if wire_index is None else wire_index
        ps = pr["pi_slots"] if pi_slots is None else pi_slots
        e = [(C.SEC_PARAMS, ctypes.sizeof(p), ctypes.addressof(p), ctypes.sizeof(p)),
             (C.SEC_GATES, ctypes.sizeof(gates[0]), ctypes.addressof(gates), ctypes.sizeof(gates)),
             (C.SEC_K_IS, 8, kis.ctypes.data, kis.nbytes), (C.SEC_CONSTANTS, 8, data.constants.ctypes.data, data.constants.nbytes),
             (C.SEC_SIGMAS, 8, sig.ctypes.data, sig.nbytes),
             (C.SEC_WP_DIMS, ctypes.sizeof(dims), ctypes.addressof(dims_ or dims), ctypes.sizeof(dims)),
             (C.SEC_WP_CODE, 4, code_.ctypes.data, code_.nbytes), (C.SEC_WP_PARAMS, 8, pr["params"].ctypes.data, pr["params"].nbytes),
             (C.SEC_WP_INPUT_SLOTS, 4, pr["input_slots"].ctypes.data, pr["input_slots"].nbytes),
             (C.SEC_WP_WIRE_SLOT, 4, pr["wire_slot"].ctypes.data, pr["wire_slot"].nbytes), (C.SEC_WP_WIRE_INDEX, 4, wi.ctypes.data, wi.nbytes),
             (C.SEC_WP_PI_SLOTS, 4, ps.ctypes.data, ps.nbytes)]
        ents = (C.EntryC * len(e))(*[C.EntryC(*x) for x in e])
        assert lib.zklc_container_write(str(path).encode(), ents, len(e)) == 0

    def params_rc(path, want_dims=True):
        c = C.Container(path)
        from zklc_amd.plonky2.prover import ParamsC
        pp, dd = ParamsC(), C.DimsC()
        rc = lib.zklc_plonky2_container_params(c._h, ctypes.byref(pp), None)
        wires = np.zeros((1, 135, data.n), dtype=np.uint64)
        st = np.zeros(1, dtype=np.int32)
        vals = np.zeros((1, len(words)), dtype=np.uint64)
        pis = np.zeros((1, 8), dtype=np.uint64)
        rc2 = lib.zklc_plonky2_witness_run_from_container(c._h, vals.ctypes.data, 1, wires.ctypes.data, pis.ctypes.data, st.ctypes.data, None, 1)
        c.close()
        return rc, rc2
    good = tmp_path / "good.zkcc"
    write(good)
    assert params_rc(good) == (0, 0)
    FORMAT = -6
    short = tmp_path / "short.zkcc"
    write(short, sig=np.ascontiguousarray(data.sigmas[:79]))
    assert params_rc(short)[0] == FORMAT
    oob = tmp_path / "oob.zkcc"
    wi = pr["wire_index"].copy()
    wi[5] = 135 * data.n
    write(oob, wire_index=wi)
    assert params_rc(oob) == (0, FORMAT)
    ps = pr["pi_slots"].copy()
    ps[0] = pr["n_slots"]
    write(oob, pi_slots=ps)
    assert params_rc(oob) == (0, FORMAT)
    d2 = C.DimsC(len(pr["code"]) - 1, len(pr["params"]), len(pr["wire_slot"]), pr["n_slots"], len(pr["input_slots"]), 135, data.n, len(pr["pi_slots"]), 0)
    write(oob, dims_=d2)
    assert params_rc(oob) == (0, FORMAT)
    # the instruction words are walked before any interpreter sees them: a slot that does not exist, an instruction that runs past
    # the end of the code, parameters past the end of the parameter array
    code = pr["code"].copy()
    n_in = int(code[2])
    code[4 + n_in] = pr["n_slots"] + 5                       # first output slot of the first instruction
    write(oob, code=code)
    assert params_rc(oob) == (0, FORMAT)
    code = pr["code"].copy()
    code[3] += 1                                             # one more output than the stream holds: the tiling breaks
    write(oob, code=code)
    assert params_rc(oob) == (0, FORMAT)
    code = pr["code"].copy()
    code[1] = len(pr["params"]) + 1
    write(oob, code=code)
    assert params_rc(oob) == (0, FORMAT)
    # duplicate tags and the library's own tags as "host" sections are refused by the writers
    e2 = (C.EntryC * 2)(C.EntryC(0x1000, 1, kis.ctypes.data, 8), C.EntryC(0x1000, 1, kis.ctypes.data, 8))
    assert lib.zklc_container_write(str(tmp_path / "dup.zkcc").encode(), e2, 2) == -1
    assert lib.zklc_container_write(b"/nonexistent-dir/x.zkcc", e2, 1) == -5


def test_input_value_files(tmp_path):
    vals = np.arange(24, dtype=np.uint64).reshape(3, 8) * np.uint64(0x0101010101010101)
    path = tmp_path / "in.zkcc"
    C.write_input_values(path, vals)
    c = C.Container(path)
    assert c.tags() == [C.SEC_INPUT_VALUES] and np.array_equal(c.section(C.SEC_INPUT_VALUES, np.uint64).reshape(3, 8), vals)
    c.close()


def test_a_recursion_circuit_round_trips_with_its_target_tree(tmp_path):
    """the circuit of `recursive_proof` (recursion.rs:16-97) over a small inner circuit: its targets are a nested tree (proof
    targets, verifier-data targets, public inputs) whose leaves must come back as the objects the program's input list holds"""
    from zklc_amd.plonky2 import CircuitBuilder
    from zklc_amd.plonky2 import recursion as R
    b = CircuitBuilder()
    x = b.add_virtual_public_input()
    b.connect(b.mul(x, x), b.add_virtual_public_input())
    inner = b.build()
    data, targets = R.recursive_circuit([inner.common_data()], 2)
    data.witness_program([t for t in _leaves(targets)])
    path = tmp_path / "rec.zkcc"
    data.save(path, targets)
    d2, t2 = CircuitData.load(path)
    ids = {id(t) for t in d2._program["input_targets"]}
    leaves = list(_leaves(t2))
    assert len(leaves) == len(list(_leaves(targets))) == len(ids) and all(id(t) in ids for t in leaves)
    assert [t.k for t in leaves] == [t.k for t in _leaves(targets)]
    assert np.array_equal(d2.sigmas, data.sigmas) and d2.common_data() == data.common_data()


def _leaves(tree):
    if isinstance(tree, Target):
        yield tree
    elif isinstance(tree, dict):
        for v in tree.values():
            yield from _leaves(v)
    elif isinstance(tree, (list, tuple)):
        for v in tree:
            yield from _leaves(v)


def test_native_caller_builds_and_fails_loudly_without_a_gpu(tmp_path, sha_circuit):
    """tests/c_abi/prove_from_file.c compiles with plain gcc against include/zklc.h and the shared library (no HIP headers, no
    Python); it reads both files on the host and -- on a box without a GPU -- stops at zklc_init with the library's own message"""
    exe = tmp_path / "prove_from_file"
    lib_dir = os.path.join(ROOT, "zk-light-client-implementation_amd", "lib")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "prove_from_file.c"), "-o", str(exe), "-L", lib_dir, "-lzklc_mi355",
                           "-Wl,-rpath," + lib_dir])
    data, words, msg = sha_circuit
    data.save(tmp_path / "c.zkcc", None)
    C.write_input_values(tmp_path / "in.zkcc", np.array([SHA.padded_words(msg)], dtype=np.uint64))
    r = subprocess.run([str(exe), str(tmp_path / "c.zkcc"), str(tmp_path / "in.zkcc"), str(tmp_path / "proof.bin")], capture_output=True, text=True)
    assert "2^%d rows x 135 wires" % data.degree_bits in r.stderr and "1 witness(es)" in r.stderr
    if not os.path.exists("/dev/kfd"):
        assert r.returncode == 3 and "zklc_init" in r.stderr and "no usable gfx950 device" in r.stderr
        assert not (tmp_path / "proof.bin").exists()
    r = subprocess.run([str(exe), str(tmp_path / "in.zkcc"), str(tmp_path / "in.zkcc"), str(tmp_path / "p")], capture_output=True, text=True)
    assert r.returncode == 3 and "zklc_plonky2_container_params" in r.stderr
