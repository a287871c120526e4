"""Opt-in on-disk cache of built circuits (the reference rebuilds every circuit per call; here a circuit is built once
per process -- and, with this cache, once per machine).

Building the reference's circuits is host Python and takes minutes for a cold process (the per-signature Ed25519 circuit ~60 s, a
SHA-256 circuit over a 5 KB inner_rest ~25 s, seven of them in a block window); everything a prover needs afterwards is plain data:
the gate list, the constants and sigma matrices, the witness program (instruction words, slot tables) and the input targets.

    ZKLC_CIRCUIT_CACHE=<dir>   enables it; unset = no file is read or written.

Processes that miss the same entry at the same time (the ranks of a multi-GPU job on a cold cache) build it ONCE: the first takes an
advisory lock beside the entry, the others wait for it and load the result.

A cache entry is ONE pickle (protocol 5, numpy buffers inline) of (CircuitData without its builder, aux) so that the Target objects
shared between `aux` and the witness program's input list stay the same objects.  The file name carries the caller's key and a
digest of this package's circuit-building sources: any change to them invalidates every entry.  The directory is a LOCAL cache and
must be trusted like the code itself (pickle executes what it loads); nothing received from a peer or the network ever goes
through it (the multi-GPU wire format is distributed.encode_obj).
"""
import hashlib
import os
import pickle
import tempfile

_SOURCES_DIGEST = None


def _sources_digest():
    global _SOURCES_DIGEST
    if _SOURCES_DIGEST is None:
        h = hashlib.sha256()
        here = os.path.dirname(os.path.abspath(__file__))
        pkg = os.path.dirname(here)
        files = sorted(os.path.join(here, f) for f in os.listdir(here) if f.endswith(".py"))
        files += sorted(os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py"))
        # the cached witness programs are interpreted by native code: a change of the opcode encoding must invalidate them too
        files += [os.path.join(pkg, "csrc", f) for f in ("plonky2_witness_ops.h", "plonky2_witness.cpp", "plonky2_witness_dev.hip")]
        for f in files:
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
        _SOURCES_DIGEST = h.hexdigest()[:16]
    return _SOURCES_DIGEST


def cache_dir():
    d = os.environ.get("ZKLC_CIRCUIT_CACHE")
    return d if d else None


def load_or_build(name, key, build):
    """build() -> (CircuitData, aux) with the witness program already compiled (`data.witness_program(...)`); `key`: anything with a
    stable repr that identifies the circuit among those called `name`.  Returns (data, aux, from_cache)."""
    d = cache_dir()
    if d is None:
        data, aux = build()
        return data, aux, False
    tag = hashlib.sha256(repr((name, key)).encode()).hexdigest()[:16]
    path = os.path.join(d, "%s-%s-%s.circuit" % (name, tag, _sources_digest()))
    got = _load(path)
    if got is not None:
        return got[0], got[1], True
    # one builder per entry and machine: the ranks of a multi-GPU job start together on a cold cache, and eight concurrent builds
    # of the same circuit are eight times the host memory and time of one.  The others wait on an advisory lock and load the entry.
    lock = _lock(path)
    try:
        if lock is not None:
            got = _load(path)
            if got is not None:
                return got[0], got[1], True
        return _build_and_store(d, path, build)
    finally:
        if lock is not None:
            lock.close()                       # releases the flock


def _load(path):
    if not os.path.exists(path):
        return None
    try:
        with open(path, "rb") as f:
            return pickle.load(f)
    except Exception as e:          # a truncated or foreign file: rebuild and replace it -- and say so, once per entry
        import sys
        print("zklc circuit cache: entry %s discarded (%s: %s), rebuilding" % (os.path.basename(path), type(e).__name__, e),
              file=sys.stderr)
        return None


def _lock(path):
    """exclusive advisory lock on <entry>.lock, or None when the directory cannot hold one (read-only, no flock: build unlocked)"""
    try:
        import fcntl
        os.makedirs(os.path.dirname(path), exist_ok=True)
        f = open(path + ".lock", "a")
        fcntl.flock(f, fcntl.LOCK_EX)
        return f
    except (OSError, ImportError):
        return None


def _build_and_store(d, path, build):
    data, aux = build()
    assert data._program is not None, "compile the witness program before caching a circuit"
    tmp = None
    try:                                       # the cache is an optimisation: a read-only or full directory must not stop a proof
        os.makedirs(d, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=d, suffix=".tmp")
        with os.fdopen(fd, "wb") as f:
            pickle.dump((data, aux), f, protocol=5)
        os.replace(tmp, path)              # atomic: concurrent processes never see a partial entry
    except OSError as e:
        import sys
        print("zklc circuit cache: entry %s not written (%s: %s)" % (os.path.basename(path), type(e).__name__, e), file=sys.stderr)
        if tmp and os.path.exists(tmp):
            try:
                os.unlink(tmp)
            except OSError:
                pass
    return data, aux, False
