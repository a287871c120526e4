"""Opt-in on-disk cache of built circuits (the reference rebuilds every circuit per call; here a circuit is built once
per process -- and, with this cache, once per machine).

Building the reference's circuits is host Python and takes minutes for a cold process (the per-signature Ed25519 circuit ~60 s, a
SHA-256 circuit over a 5 KB inner_rest ~25 s, seven of them in a block window); everything a prover needs afterwards is plain data:
the gate list, the constants and sigma matrices, the witness program (instruction words, slot tables) and the input targets.

    ZKLC_CIRCUIT_CACHE=<dir>   enables it; unset = no file is read or written.

Processes that miss the same entry at the same time (the ranks of a multi-GPU job on a cold cache) build it ONCE: the first takes an
advisory lock beside the entry, the others wait for it and load the result.

A cache entry is ONE circuit container (plonky2/container.py; include/zklc.h section b''; csrc/container.cpp): the flat, versioned,
language-neutral file whose sections are the arguments of `zklc_plonky2_circuit_create` / `zklc_plonky2_witness_program_create`,
plus the host mirror's own sections (configuration, gate ids, the caller's target tree as JSON with table indices).  Loading maps the
file and parses JSON: nothing in an entry is executed, the matrices are never copied into the Python heap, and the prover creates the
GPU circuit straight from the mapped sections.  The file name carries the caller's key and a digest of this package's
circuit-building sources: any change to them invalidates every entry.
"""
import hashlib
import os

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
        files += [os.path.join(pkg, "csrc", f) for f in ("plonky2_witness_ops.h", "plonky2_witness.cpp", "plonky2_witness_dev.hip",
                                                         "container.cpp")]
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
    path = _entry_path(name, key)
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
        from .container import read_circuit
        return read_circuit(path, verify=os.environ.get("ZKLC_CIRCUIT_CACHE_VERIFY", "1") != "0")
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
    try:                                       # the cache is an optimisation: a read-only or full directory must not stop a proof
        os.makedirs(d, exist_ok=True)
        data.save(path, aux, note="circuit cache entry %s" % os.path.basename(path))      # atomic inside the library (temp + rename)
    except (OSError, ValueError) as e:
        import sys
        print("zklc circuit cache: entry %s not written (%s: %s)" % (os.path.basename(path), type(e).__name__, e), file=sys.stderr)
        return data, aux, False
    # continue with the entry just written, not with the builder's object graph (millions of Python objects per circuit): the
    # process that built a circuit then holds what every later process holds -- views of the mapped file
    got = _load(path)
    if got is None:
        return data, aux, False
    del data, aux
    return got[0], got[1], False


# ------------------------------------------------------------------------------------------------ out-of-process builds
def _entry_path(name, key):
    d = cache_dir()
    tag = hashlib.sha256(repr((name, key)).encode()).hexdigest()[:16]
    return os.path.join(d, "%s-%s-%s.circuit" % (name, tag, _sources_digest()))


def _job_key(kind, arg):
    """the (name, key) pair `build_cached` of the circuit's module uses -- kept beside it so that a miss here is a miss there"""
    from .builder import standard_recursion_config, wide_ecc_config
    if kind == "ed25519":
        return "ed25519", (int(arg), sorted(wide_ecc_config().items(), key=str))
    if kind == "sha256":
        from .sha256 import block_num_of
        return "sha256", (block_num_of(int(arg)), sorted(standard_recursion_config().items(), key=str))
    raise ValueError("prewarm: unknown circuit kind %r" % (kind,))


def build_job(kind, arg):
    """what a prewarm worker runs (also: `python -m zklc_amd.plonky2.circuit_cache ed25519 41`): build one circuit into the cache"""
    if kind == "ed25519":
        from .ed25519_circuit import build_cached
    elif kind == "sha256":
        from .sha256 import build_cached
    else:
        raise ValueError("unknown circuit kind %r" % (kind,))
    return build_cached(int(arg))[2]


def prewarm(jobs, processes=None, timeout_s=900):
    """jobs: [(kind, arg)] with kind in {"ed25519", "sha256"} and arg = the message length in bytes.  The entries missing from the
    cache are built by child interpreters (one per entry, at most `processes` at a time; CPU only -- no GPU context is created in
    them), largest first.  Returns {"missing": n, "built": n, "failed": [...], "seconds": s}; never raises."""
    import subprocess
    import sys
    import time
    t0 = time.perf_counter()
    rep = {"missing": 0, "built": 0, "failed": [], "seconds": 0.0}
    try:
        if cache_dir() is None:
            rep["skipped"] = "ZKLC_CIRCUIT_CACHE is not set"
            return rep
        seen, todo = set(), []
        for kind, arg in jobs:
            name, key = _job_key(kind, arg)
            path = _entry_path(name, key)
            if path in seen or os.path.exists(path):
                continue
            seen.add(path)
            todo.append((kind, int(arg), path))
        rep["missing"] = len(todo)
        if not todo:
            return rep
        todo.sort(key=lambda j: (j[0] != "ed25519", -j[1]))          # the longest builds first
        cores = len(os.sched_getaffinity(0))
        procs_max = max(1, min(len(todo), processes or max(1, cores // 2), 12))
        os.makedirs(cache_dir(), exist_ok=True)
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        code = ("import sys; sys.path.insert(0, %r); import zklc_amd; from zklc_amd.plonky2 import circuit_cache as C; "
                "C.build_job(sys.argv[1], sys.argv[2])" % root)
        env = dict(os.environ, OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        import tempfile
        running, pending = [], list(todo)
        deadline = time.perf_counter() + timeout_s

        def tail(f):
            # a child's stderr goes to a file, not a pipe: a pipe nobody reads blocks the child after 64 KB of warnings
            try:
                f.seek(0)
                return f.read()[-300:]
            except (OSError, ValueError):
                return ""
            finally:
                f.close()
        while pending or running:
            if time.perf_counter() > deadline and pending:        # out of time: what has not started is not started any more
                rep["failed"] += [(job[0], job[1], "timeout (not started)") for job in pending]
                pending = []
            while pending and len(running) < procs_max:
                job = pending.pop(0)
                errf = tempfile.TemporaryFile(mode="w+")
                running.append((job, subprocess.Popen([sys.executable, "-c", code, job[0], str(job[1])], env=env,
                                                      stdout=subprocess.DEVNULL, stderr=errf, text=True), errf))
            time.sleep(0.2)
            still = []
            for job, p, errf in running:
                if p.poll() is None:
                    if time.perf_counter() > deadline:
                        p.kill()
                        p.wait()
                        rep["failed"].append((job[0], job[1], "timeout: " + tail(errf)))
                    else:
                        still.append((job, p, errf))
                    continue
                err = tail(errf)
                if p.returncode == 0 and os.path.exists(job[2]):
                    rep["built"] += 1
                else:
                    rep["failed"].append((job[0], job[1], err))
            running = still
    except Exception as e:          # an optimisation: the in-process build on first use remains
        rep["failed"].append(("prewarm", 0, repr(e)[:300]))
    rep["seconds"] = time.perf_counter() - t0
    return rep


if __name__ == "__main__":
    import sys
    print(build_job(sys.argv[1], sys.argv[2]))
