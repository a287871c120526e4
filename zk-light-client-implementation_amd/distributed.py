"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

  * approval sets / signatures are independent units -> plain index sharding, no collective
  * the MSM shards the (point, scalar) arrays by index; every rank reduces its shard to ONE
    affine point, the partials are ALL-GATHERed (world x 68 bytes -- RCCL cannot add curve
    points, and at this size the xGMI ring is latency-, not bandwidth-bound) and every rank
    adds the `world` partials locally with a unit-scalar MSM.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous, balanced index range of `rank` (the first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def msm_sharded(local_msm, combine_msm, points_shard, scalars_shard, group=None, device=None):
    """local_msm(points, scalars) -> (uint64[8] affine, is_inf) on this rank's shard;
    combine_msm(points[k,8], scalars[k,4]) the same function used on the gathered partials.
    Returns the full result on every rank."""
    import torch
    import torch.distributed as dist
    out, inf = local_msm(points_shard, scalars_shard)
    if not dist.is_initialized():
        return out, inf
    world = dist.get_world_size(group)       # an initialised world of ONE still runs the collective (the RCCL smoke of a 1-GPU box)
    mine = torch.zeros(9, dtype=torch.int64, device=device)
    mine[:8] = torch.from_numpy(np.asarray(out, dtype=np.uint64).view(np.int64).copy()).to(mine.device)
    mine[8] = 1 if inf else 0
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    parts = torch.stack(gathered).cpu().numpy()
    pts = parts[:, :8].copy().view(np.uint64)
    pts[parts[:, 8] != 0] = 0            # (0, 0) is the point at infinity in the gnark layout
    ones = np.zeros((world, 4), dtype=np.uint64)
    ones[:, 0] = 1
    return combine_msm(pts, ones)


def gather_bytes(chunks, group=None, device=None):
    """ALL-GATHER of variable-length byte strings: every rank contributes a list of `bytes` and gets the list of all ranks'
    lists (rank order).  Two collectives: the lengths, then the zero-padded payloads (RCCL has no ragged gather; a signature proof
    is ~190 KB, so a padded all_gather of a few MB per rank is far below what xGMI moves in a millisecond)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [list(chunks)]
    world = dist.get_world_size(group)
    lens = torch.tensor([len(chunks)] + [len(c) for c in chunks], dtype=torch.int64, device=device)
    n_max = torch.tensor([lens.numel()], dtype=torch.int64, device=device)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX, group=group)
    pad = torch.zeros(int(n_max), dtype=torch.int64, device=device)
    pad[:lens.numel()] = lens
    all_lens = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(all_lens, pad, group=group)
    all_lens = [t.cpu().tolist() for t in all_lens]
    total = max(sum(t[1:1 + t[0]]) for t in all_lens)
    buf = torch.zeros(max(total, 1), dtype=torch.uint8, device=device)
    blob = b"".join(chunks)
    if blob:
        buf[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(buf.device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = []
    for t, b in zip(all_lens, bufs):
        raw, off, items = b.cpu().numpy().tobytes(), 0, []
        for ln in t[1:1 + t[0]]:
            items.append(raw[off:off + ln])
            off += ln
        out.append(items)
    return out


def prove_signatures_sharded(prove_one, n, group=None, device=None):
    """One block's signature proofs over the GPUs of a node: signature i is proven by rank i mod world
    (`prove_one(i) -> bytes`, no data dependence between signatures: signatures.rs:70-123), the proofs are all-gathered, and every
    rank returns the n proofs in signature order -- the order the left fold of signatures.rs:97-105 consumes them in, so the
    aggregate is the one a single GPU would produce."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = [prove_one(i) for i in range(rank, n, world)]
    parts = gather_bytes(mine, group=group, device=device)
    out = [None] * n
    for r, items in enumerate(parts):
        for k, p in enumerate(items):
            out[r + k * world] = p
    assert all(p is not None for p in out)
    return out


# ---------------------------------------------------------------------------------- one block over the GPUs of a node
def _comm_device(device):
    """tensors of a collective live on the GPU with RCCL and on the host with gloo"""
    import torch.distributed as dist
    return device if dist.get_backend() == "nccl" else None


def encode_obj(obj):
    """Length-framed wire form of what the ranks exchange -- proof triples (common_data dict, verifier_only dict, proof as
    `to_bytes` bytes or in the proof.json schema) and dicts / lists of them: [u32 n][JSON text of n bytes][u32 k][k x (u64 len, raw
    bytes)].  The JSON holds the structure (dict, list, tuple as {"__t": [..]}, int, str, None); byte strings travel raw and are
    referenced as {"__b": index}.  Decoding builds plain data only: nothing a peer sends is ever executed (no Python object serialisation)."""
    import json
    import struct
    blobs = []

    def enc(x):
        if isinstance(x, (bytes, bytearray, memoryview)):
            blobs.append(bytes(x))
            return {"__b": len(blobs) - 1}
        if isinstance(x, tuple):
            return {"__t": [enc(v) for v in x]}
        if isinstance(x, list):
            return [enc(v) for v in x]
        if isinstance(x, dict):
            assert all(isinstance(k, str) and not k.startswith("__") for k in x), "dict keys must be plain strings"
            return {k: enc(v) for k, v in x.items()}
        if x is None or isinstance(x, (bool, int, str, float)):
            return x
        if hasattr(x, "item"):          # numpy scalar
            return x.item()
        raise TypeError("encode_obj: unsupported type %r" % type(x))
    text = json.dumps(enc(obj), separators=(",", ":")).encode()
    out = [struct.pack("<I", len(text)), text, struct.pack("<I", len(blobs))]
    for bl in blobs:
        out += [struct.pack("<Q", len(bl)), bl]
    return b"".join(out)


def decode_obj(raw):
    import json
    import struct
    raw = bytes(raw)
    n, = struct.unpack_from("<I", raw, 0)
    tree = json.loads(raw[4:4 + n].decode())
    off = 4 + n
    k, = struct.unpack_from("<I", raw, off)
    off += 4
    blobs = []
    for _ in range(k):
        ln, = struct.unpack_from("<Q", raw, off)
        off += 8
        if off + ln > len(raw):
            raise ValueError("decode_obj: truncated frame")
        blobs.append(raw[off:off + ln])
        off += ln
    if off != len(raw):
        raise ValueError("decode_obj: trailing bytes")

    def dec(x):
        if isinstance(x, dict):
            if set(x) == {"__b"}:
                return blobs[x["__b"]]
            if set(x) == {"__t"}:
                return tuple(dec(v) for v in x["__t"])
            return {k: dec(v) for k, v in x.items()}
        if isinstance(x, list):
            return [dec(v) for v in x]
        return x
    return dec(tree)


def send_obj(obj, dst, group=None, device=None):
    """point-to-point transfer of a proof triple or a dict of them (~150-200 KB each): length, then the frame of encode_obj"""
    import torch
    import torch.distributed as dist
    dev = _comm_device(device)
    raw = encode_obj(obj)
    dist.send(torch.tensor([len(raw)], dtype=torch.int64, device=dev), dst, group=group)
    payload = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    dist.send(payload.to(dev) if dev is not None else payload, dst, group=group)


def recv_obj(src, group=None, device=None):
    import torch
    import torch.distributed as dist
    dev = _comm_device(device)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.recv(n, src, group=group)
    buf = torch.zeros(int(n[0]), dtype=torch.uint8, device=dev)
    dist.recv(buf, src, group=group)
    return decode_obj(buf.cpu().numpy().tobytes())


def tree_fold(local, combine, group=None, device=None):
    """Binary-tree aggregation of the ranks' partial aggregates (the opt-in replacement of the serial left fold of
    signatures.rs:97-105 across GPUs).  Rank r holds `local` = the aggregate of ITS contiguous run of signature proofs (None if it
    has none); at step s = 1, 2, 4, .. rank r with r % 2s == s sends its aggregate to rank r - s, which combines
    (lower-rank aggregate first: the leaves stay in signature order) with `combine(a, b)` = one `recursive_proof(a, b)`.
    log2(world) exchanges of one proof each over xGMI point-to-point links; rank 0 returns the block's aggregate, the others None."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    step = 1
    while step < world:
        if rank % (2 * step) == step:
            send_obj(local, rank - step, group, device)
            return None
        if rank % (2 * step) == 0 and rank + step < world:
            other = recv_obj(rank + step, group, device)
            if local is None:
                local = other
            elif other is not None:
                local = combine(local, other)
        step *= 2
    return local


def all_ok(ok, group=None, device=None):
    """every rank's success flag AND-ed over the ranks (one MIN all-reduce of an int32): the strong form calls this before every
    exchange, so that a rank that failed makes ALL ranks raise instead of feeding a partial aggregate to the others"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t[0]))


class RemoteRankFailed(RuntimeError):
    """another rank of the strong form reported a failure (its own exception is raised there)"""


def gather_objects(obj, dst=0, group=None, device=None):
    """every rank's object on rank `dst` (list in rank order; None elsewhere): the header / keys-stakes proofs made by the other
    ranks travel to the rank that joins the DAG"""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return [obj]
    if rank == dst:
        return [obj if r == dst else recv_obj(r, group, device) for r in range(world)]
    send_obj(obj, dst, group, device)
    return None


def assign_jobs(names, world, skip_rank0_first=True):
    """header proofs -> ranks, round robin starting from the LAST rank so that rank 0 (which also joins the DAG and wraps) gets a
    job only when there are more jobs than other ranks"""
    out = {}
    for k, name in enumerate(names):
        out[name] = (world - 1 - k) % world if skip_rank0_first else k % world
    return out
