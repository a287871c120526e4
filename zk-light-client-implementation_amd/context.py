"""Thin object wrapper over the zklc C ABI (include/zklc.h)."""
import ctypes

import numpy as np

from . import _lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _host_u8(x, name):
    """bytes / bytearray / numpy -> contiguous uint8 numpy array (no copy when possible)."""
    if isinstance(x, (bytes, bytearray, memoryview)):
        return np.frombuffer(x, dtype=np.uint8)
    a = np.ascontiguousarray(x)
    if a.dtype != np.uint8:
        raise TypeError("%s must be uint8" % name)
    return a.reshape(-1)


class Context:
    """One zklc_ctx: a device, a stream, staging buffers and constant tables."""

    def __init__(self, device_id=0, high_priority=False):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.zklc_init_priority(ctypes.byref(h), int(device_id), 1 if high_priority else 0)
        if rc != 0:
            raise _lib.ZklcError(rc)
        self._h = h
        self.device_id = int(device_id)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zklc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise _lib.ZklcError(rc, self._lib.zklc_last_hip_error(self._h).decode())

    def stream_ptr(self):
        """the context's own hipStream_t (an int usable as the `stream` argument of the *_dev methods)"""
        return self._lib.zklc_stream(self._h)

    def synchronize(self):
        self._check(self._lib.zklc_synchronize(self._h))

    # ---- (a) Ed25519 ---------------------------------------------------
    def ed25519_verify_batch(self, pks, sigs, msg, msg_stride=0, msg_len=None):
        """Host-pointer path.  pks: n*32 bytes, sigs: n*64 bytes, msg: shared
        message (msg_stride=0) or n messages of msg_len bytes at msg_stride.
        Returns a uint8 numpy array of n 0/1 flags."""
        pk = _host_u8(pks, "pks")
        sg = _host_u8(sigs, "sigs")
        if pk.size % 32 or sg.size != (pk.size // 32) * 64:
            raise ValueError("pks must be n*32 bytes and sigs n*64 bytes")
        n = pk.size // 32
        m = _host_u8(msg, "msg")
        if msg_len is None:
            if msg_stride:
                raise ValueError("msg_len required with msg_stride")
            msg_len = m.size
        need = msg_len if not msg_stride else (msg_stride * n)
        if m.size < need:
            raise ValueError("message buffer too small")
        ok = np.zeros(n, dtype=np.uint8)
        self._check(self._lib.zklc_ed25519_verify_batch(
            self._h, pk.ctypes.data, sg.ctypes.data, m.ctypes.data if m.size else None, msg_len, msg_stride, n,
            ok.ctypes.data))
        return ok

    def ed25519_verify_batch_dev(self, d_pks, d_sigs, d_msgs, msg_len, msg_stride, n, d_ok, stream=None):
        """Device-pointer path: torch uint8 CUDA tensors (or raw int pointers); enqueue only."""
        self._check(self._lib.zklc_ed25519_verify_batch_dev(
            self._h, _stream_ptr(stream), _dev_ptr(d_pks), _dev_ptr(d_sigs), _dev_ptr(d_msgs), msg_len, msg_stride, n,
            _dev_ptr(d_ok)))

    def sha512_batch(self, data, stride, length, n):
        d = _host_u8(data, "data")
        if n and d.size < stride * (n - 1) + length:
            raise ValueError("input buffer too small")
        out = np.zeros((n, 64), dtype=np.uint8)
        self._check(self._lib.zklc_sha512_batch(self._h, d.ctypes.data if d.size else None, stride, length, n,
                                                out.ctypes.data))
        return out

    def sha512_batch_dev(self, d_in, stride, length, n, d_out, stream=None):
        self._check(self._lib.zklc_sha512_batch_dev(self._h, _stream_ptr(stream), _dev_ptr(d_in), stride, length, n,
                                                    _dev_ptr(d_out)))


    # ---- (b) Goldilocks ------------------------------------------------
    def gl_ntt(self, data, flags=0, coset_shift=0):
        """data: uint64 numpy array [batch, 2^log_n] (poly-major), transformed copy returned."""
        a = np.ascontiguousarray(data, dtype=np.uint64)
        if a.ndim == 1:
            a = a[None, :]
        batch, n = a.shape
        log_n = n.bit_length() - 1
        if 1 << log_n != n:
            raise ValueError("transform length must be a power of two")
        out = a.copy()
        self._check(self._lib.zklc_gl_ntt(self._h, out.ctypes.data, log_n, batch, flags, coset_shift))
        return out.reshape(np.shape(data))

    def gl_ntt_dev(self, d_data, log_n, batch, flags=0, coset_shift=0, stream=None):
        self._check(self._lib.zklc_gl_ntt_dev(self._h, _stream_ptr(stream), _dev_ptr(d_data), log_n, batch, flags, coset_shift))

    def gl_lde(self, coeffs, rate_bits, coset_shift=7, flags=0):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64)
        if a.ndim == 1:
            a = a[None, :]
        batch, n = a.shape
        log_n = n.bit_length() - 1
        if 1 << log_n != n:
            raise ValueError("polynomial length must be a power of two")
        out = np.zeros((batch, n << rate_bits), dtype=np.uint64)
        self._check(self._lib.zklc_gl_lde(self._h, a.ctypes.data, log_n, rate_bits, batch, coset_shift, out.ctypes.data, flags))
        return out

    def gl_lde_dev(self, d_coeffs, log_n, rate_bits, batch, coset_shift, d_out, flags=0, stream=None):
        self._check(self._lib.zklc_gl_lde_dev(self._h, _stream_ptr(stream), _dev_ptr(d_coeffs), log_n, rate_bits, batch,
                                              coset_shift, _dev_ptr(d_out), flags))

    def poseidon_gl_permute(self, states):
        a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 12).copy()
        self._check(self._lib.zklc_poseidon_gl_permute(self._h, a.ctypes.data, a.shape[0]))
        return a

    def poseidon_gl_permute_dev(self, d_states, n, stream=None):
        self._check(self._lib.zklc_poseidon_gl_permute_dev(self._h, _stream_ptr(stream), _dev_ptr(d_states), n))

    def gl_merkle_tree_words(self, log_leaves, cap_height):
        return int(self._lib.zklc_gl_merkle_tree_words(log_leaves, cap_height))

    def gl_merkle_commit(self, mat, cap_height):
        """mat: uint64 [width, n_leaves] (poly-major).  Returns (cap [2^cap_height, 4], levels list of [m, 4] arrays)."""
        a = np.ascontiguousarray(mat, dtype=np.uint64)
        width, n = a.shape
        log_leaves = n.bit_length() - 1
        if 1 << log_leaves != n:
            raise ValueError("leaf count must be a power of two")
        words = self.gl_merkle_tree_words(log_leaves, cap_height)
        tree = np.zeros(words, dtype=np.uint64)
        self._check(self._lib.zklc_gl_merkle_commit(self._h, a.ctypes.data, n, log_leaves, width, cap_height, tree.ctypes.data))
        levels, off = [], 0
        for l in range(log_leaves - cap_height + 1):
            m = n >> l
            levels.append(tree[off:off + 4 * m].reshape(m, 4))
            off += 4 * m
        return levels[-1], levels

    def gl_merkle_commit_dev(self, d_mat, stride, log_leaves, width, cap_height, d_tree, stream=None):
        self._check(self._lib.zklc_gl_merkle_commit_dev(self._h, _stream_ptr(stream), _dev_ptr(d_mat), stride, log_leaves, width,
                                                        cap_height, _dev_ptr(d_tree)))


    def poseidon_bn254_permute(self, states):
        """states: uint64 [n, 4, 4] (regular-form Fr as 4 LE u64).  Returns the permuted copy."""
        a = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 16).copy()
        self._check(self._lib.zklc_poseidon_bn254_permute(self._h, a.ctypes.data, a.shape[0]))
        return a.reshape(-1, 4, 4)

    def poseidon_bn254_permute_dev(self, d_states, n, stream=None):
        self._check(self._lib.zklc_poseidon_bn254_permute_dev(self._h, _stream_ptr(stream), _dev_ptr(d_states), n))

    def bn254_merkle_commit(self, mat, cap_height):
        """Poseidon-BN254 Merkle tree over a poly-major Goldilocks matrix [width, n_leaves]; same return shape as gl_merkle_commit."""
        a = np.ascontiguousarray(mat, dtype=np.uint64)
        width, n = a.shape
        log_leaves = n.bit_length() - 1
        if 1 << log_leaves != n:
            raise ValueError("leaf count must be a power of two")
        words = self.gl_merkle_tree_words(log_leaves, cap_height)
        tree = np.zeros(words, dtype=np.uint64)
        self._check(self._lib.zklc_bn254_merkle_commit(self._h, a.ctypes.data, n, log_leaves, width, cap_height, tree.ctypes.data))
        levels, off = [], 0
        for l in range(log_leaves - cap_height + 1):
            m = n >> l
            levels.append(tree[off:off + 4 * m].reshape(m, 4))
            off += 4 * m
        return levels[-1], levels

    def bn254_merkle_commit_dev(self, d_mat, stride, log_leaves, width, cap_height, d_tree, stream=None):
        self._check(self._lib.zklc_bn254_merkle_commit_dev(self._h, _stream_ptr(stream), _dev_ptr(d_mat), stride, log_leaves, width,
                                                           cap_height, _dev_ptr(d_tree)))

    # ---- (c) BN254 -----------------------------------------------------
    def bn254_g1_msm(self, points, scalars):
        """points: uint64 [n, 8] (gnark Montgomery affine), scalars: uint64 [n, 4] (regular form).
        Returns (uint64[8] affine result, is_infinity)."""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
        sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        if pts.shape[0] != sc.shape[0]:
            raise ValueError("points and scalars must have the same length")
        out = np.zeros(8, dtype=np.uint64)
        inf = ctypes.c_uint32(0)
        self._check(self._lib.zklc_bn254_g1_msm(self._h, pts.ctypes.data if pts.size else None, sc.ctypes.data if sc.size else None,
                                                pts.shape[0], out.ctypes.data, ctypes.addressof(inf)))
        return out, bool(inf.value)

    def bn254_g1_msm_workspace_bytes(self, n):
        return int(self._lib.zklc_bn254_g1_msm_workspace_bytes(n))

    def bn254_g1_msm_dev(self, d_points, d_scalars, n, d_out, d_inf, d_workspace, workspace_bytes, stream=None):
        self._check(self._lib.zklc_bn254_g1_msm_dev(self._h, _stream_ptr(stream), _dev_ptr(d_points), _dev_ptr(d_scalars), n,
                                                    _dev_ptr(d_out), _dev_ptr(d_inf), _dev_ptr(d_workspace), workspace_bytes))
    # ---- fixed-base form (a Groth16 proving key's bases: the table is built once, include/zklc.h)
    def bn254_msm_fixed_table(self, d_points, n, group=1, stream=None):
        """-> a torch uint8 tensor in HBM holding the table of 2^(c w) P_i for the n affine points at d_points (G1: uint64 [n, 8],
        group=2: uint64 [n, 16]); pass it to bn254_msm_fixed_dev.  windows x n x 64 (128) bytes."""
        import torch
        lib = self._lib
        size_fn, build_fn = ((lib.zklc_bn254_g1_msm_fixed_table_bytes, lib.zklc_bn254_g1_msm_fixed_table_dev) if group == 1 else
                             (lib.zklc_bn254_g2_msm_fixed_table_bytes, lib.zklc_bn254_g2_msm_fixed_table_dev))
        nbytes = int(size_fn(n))
        table = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda:%d" % self.device_id)
        off = (-table.data_ptr()) % 256
        table = table[off:off + nbytes]
        self._check(build_fn(self._h, _stream_ptr(stream), _dev_ptr(d_points), n, table.data_ptr(), nbytes))
        return table

    def bn254_msm_fixed_dev(self, table, d_scalars, n, d_out, d_inf, d_workspace, workspace_bytes, group=1, stream=None):
        fn = self._lib.zklc_bn254_g1_msm_fixed_dev if group == 1 else self._lib.zklc_bn254_g2_msm_fixed_dev
        self._check(fn(self._h, _stream_ptr(stream), table.data_ptr(), _dev_ptr(d_scalars), n, _dev_ptr(d_out), _dev_ptr(d_inf),
                       _dev_ptr(d_workspace), workspace_bytes))

    def bn254_g2_msm(self, points, scalars):
        """points: uint64 [n, 16] (gnark Montgomery affine G2: X.A0, X.A1, Y.A0, Y.A1), scalars: uint64 [n, 4].
        Returns (uint64[16] affine result, is_infinity)."""
        pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 16)
        sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        if pts.shape[0] != sc.shape[0]:
            raise ValueError("points and scalars must have the same length")
        out = np.zeros(16, dtype=np.uint64)
        inf = ctypes.c_uint32(0)
        self._check(self._lib.zklc_bn254_g2_msm(self._h, pts.ctypes.data if pts.size else None, sc.ctypes.data if sc.size else None,
                                                pts.shape[0], out.ctypes.data, ctypes.addressof(inf)))
        return out, bool(inf.value)

    def bn254_pairing_check(self, g1, g2, k, want_gt=False):
        """g1: uint64 [batch, k, 8], g2: uint64 [batch, k, 16] (gnark Montgomery affine) -> (is_one uint32[batch], gt uint64[batch, 48]).
        PRECONDITION (include/zklc.h): the kernel checks neither that the points are on the curve nor that the G2 points lie in
        the r-torsion subgroup, and an all-zero encoding is the point at infinity -- like gnark-crypto's `PairingCheck`, whose
        callers (`groth16.Verify`) validate the proof's points when they are decoded.  A verifier built on this call must do the
        same (`zklc_amd.formats.decompress_proof` / `oracle.bn254` have the curve and subgroup checks) or it accepts invalid-curve
        and small-subgroup points."""
        a = np.ascontiguousarray(g1, dtype=np.uint64).reshape(-1, k, 8)
        b = np.ascontiguousarray(g2, dtype=np.uint64).reshape(-1, k, 16)
        batch = a.shape[0]
        assert b.shape[0] == batch
        ok = np.zeros(batch, dtype=np.uint32)
        gt = np.zeros((batch, 48), dtype=np.uint64) if want_gt else None
        self._check(self._lib.zklc_bn254_pairing_check(self._h, a.ctypes.data, b.ctypes.data, k, batch, ok.ctypes.data,
                                                       gt.ctypes.data if want_gt else None))
        return ok, gt

    def bn254_fr_ntt(self, data, flags=0, coset=0):
        """data: uint64 [n, 4] Fr elements in gnark Montgomery layout -> transformed copy"""
        a = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, 4).copy()
        n = a.shape[0]
        assert n & (n - 1) == 0 and n
        self._check(self._lib.zklc_bn254_fr_ntt(self._h, a.ctypes.data, n.bit_length() - 1, flags, coset))
        return a


def _dev_ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    if _is_torch(t):
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("device tensors must be contiguous CUDA tensors")
        return t.data_ptr()
    raise TypeError("expected a CUDA tensor or an integer device pointer")


def _stream_ptr(s):
    if s is None:
        return None
    if isinstance(s, int):
        return s
    return s.cuda_stream  # torch.cuda.Stream
