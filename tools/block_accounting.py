#!/usr/bin/env python3
"""Block-level accounting of the headline metric: what ONE Block_i proof costs the GPU end to end.

Inputs (all produced on the GPU box by tools/gpu_block_accounting.sh):
  * a rocprofv3 --kernel-trace CSV of `bench.py` (overlapped blocks, the driver's command shape),
  * the bench_detail.json of that run (block_i.timed_region_clock_ns: the timed region in every host clock),
  * optionally a rocprofv3 --pmc counter_collection CSV of the same command (kernels serialised by the counter collection;
    the instruction counts are the same -- the prover is deterministic) with its own bench_detail.json.

Output: one JSON with, per block of the timed region,
  wall_s                      the bench's own seconds per block
  busy_union_s                time with at least one kernel resident (union of the kernel intervals)
  idle_s                      wall - union: no kernel on the GPU (host waits, launch gaps)
  sum_kernel_s                sum of the kernel durations (> union when streams overlap)
  gaps                        histogram of the idle intervals
  by_kernel                   calls / summed seconds / share, top entries
  valu_wave_instr_per_block   SQ_INSTS_VALU summed over the region's dispatches / blocks   (PMC pass)
  valu_lane_instr_per_block   x 64
  frac_of_issue_limit         lane instructions / (wall_s x 39.3 T/s)   -- the multi-pass integer issue limit of DESIGN section 3
  sum_exclusive_kernel_s      sum over dispatches of GRBM_GUI_ACTIVE / sclk: the kernels back to back, nothing overlapped (PMC pass)
  latency_bound_s             exclusive seconds of dispatches whose grid is < 1 wave per SIMD (cannot fill the chip)

    python tools/block_accounting.py --trace kernel_trace.csv --detail bench_detail.json [--pmc counters.csv,detail.json[,kernel_trace.csv] ...]
"""
import argparse
import collections
import csv
import json
import re
import sys

import numpy as np

ISSUE_LIMIT_MULTI = 39.3e12        # lane-instructions / s: 1024 SIMDs x 64 lanes x 2.4 GHz / 4 cycles


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name[:60]


def pick_clock(regions, lo, hi):
    """the host clock whose timed region lies inside the trace's time span"""
    best = None
    for k, (a, b) in regions.items():
        if lo <= a and b <= hi + 1e9:
            best = k if best is None else best
    return best


def read_trace(path):
    names, start, end, queue = [], [], [], []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            names.append(short(r["Kernel_Name"]))
            start.append(int(r["Start_Timestamp"]))
            end.append(int(r["End_Timestamp"]))
            queue.append(int(r.get("Queue_Id", 0) or 0))
    return np.array(names), np.array(start, dtype=np.int64), np.array(end, dtype=np.int64), np.array(queue)


def union_and_gaps(start, end, t0, t1):
    o = np.argsort(start, kind="stable")
    s, e = np.clip(start[o], t0, t1), np.clip(end[o], t0, t1)
    run_end = np.maximum.accumulate(e)
    # a new busy interval begins where a kernel starts after everything before it has ended
    new = np.ones(len(s), dtype=bool)
    new[1:] = s[1:] > run_end[:-1]
    idx = np.flatnonzero(new)
    seg_start = s[idx]
    seg_end = np.append(run_end[idx[1:] - 1], run_end[-1])
    busy = int((seg_end - seg_start).sum())
    gaps = np.concatenate([[seg_start[0] - t0], seg_start[1:] - seg_end[:-1], [t1 - seg_end[-1]]]).astype(np.int64)
    return busy, gaps[gaps > 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--detail", required=True)
    ap.add_argument("--pmc", nargs="*", default=[], help="csv,detail[,kernel_trace] per counter pass")
    ap.add_argument("--out", default=None)
    ap.add_argument("--compact", default=None, help="also write the region's dispatches as a compressed .npz (offline analysis)")
    a = ap.parse_args()

    det = json.load(open(a.detail))
    blk = det.get("block_i") or det["stages"]["prove"]["block_i"]
    steps = int(blk["blocks_timed"])
    wall = float(blk["seconds_per_block"])
    names, start, end, queue = read_trace(a.trace)
    clock = pick_clock(blk["timed_region_clock_ns"], int(start.min()), int(end.max()))
    if clock is None:
        sys.exit("no host clock of the timed region falls inside the trace (trace %d..%d, regions %r)" %
                 (start.min(), end.max(), blk["timed_region_clock_ns"]))
    t0, t1 = blk["timed_region_clock_ns"][clock]
    inside = (end > t0) & (start < t1)
    n_, s_, e_, q_ = names[inside], start[inside], end[inside], queue[inside]
    busy, gaps = union_and_gaps(s_, e_, t0, t1)
    dur = (np.minimum(e_, t1) - np.maximum(s_, t0)).astype(np.int64)
    rep = {"clock": clock, "blocks": steps, "wall_s": wall, "region_s": (t1 - t0) / 1e9, "dispatches_per_block": int(inside.sum()) / steps,
           "busy_union_s": busy / 1e9 / steps, "idle_s": (t1 - t0 - busy) / 1e9 / steps, "sum_kernel_s": int(dur.sum()) / 1e9 / steps,
           "queues": int(len(set(q_.tolist())))}
    rep["idle_frac"] = rep["idle_s"] / (rep["region_s"] / steps)
    edges = [0, 10e3, 50e3, 200e3, 1e6, 5e6, 20e6, 1e12]
    labels = ["<10us", "10-50us", "50-200us", "0.2-1ms", "1-5ms", "5-20ms", ">20ms"]
    rep["gaps"] = {lab: {"count_per_block": int(((gaps >= lo) & (gaps < hi)).sum()) / steps,
                         "s_per_block": float(gaps[(gaps >= lo) & (gaps < hi)].sum()) / 1e9 / steps}
                   for lab, lo, hi in zip(labels, edges, edges[1:])}
    agg = collections.defaultdict(lambda: [0, 0])
    for k, d in zip(n_.tolist(), dur.tolist()):
        agg[k][0] += 1
        agg[k][1] += d
    tot = sum(v[1] for v in agg.values())
    rep["by_kernel"] = [{"kernel": k, "calls_per_block": v[0] / steps, "s_per_block": v[1] / 1e9 / steps, "share_of_sum": v[1] / tot}
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]]
    if a.compact:
        uniq, inv = np.unique(n_, return_inverse=True)
        np.savez_compressed(a.compact, names=uniq, kernel=inv.astype(np.int16), start=(s_ - t0), end=(e_ - t0), queue=q_.astype(np.int16),
                            region=np.array([0, t1 - t0]), blocks=steps)

    # ---- PMC passes: one rocprofv3 run per counter group, each given as  csv,detail[,kernel_trace]  (its own timed region)
    if a.pmc:
        tot = collections.defaultdict(float)
        per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
        small_gui = 0.0
        pm = {}
        for spec in a.pmc:
            parts = spec.split(",")
            path, pdet = parts[0], json.load(open(parts[1]))
            pblk = pdet.get("block_i") or pdet["stages"]["prove"]["block_i"]
            psteps = int(pblk["blocks_timed"])
            stamp = {}
            if len(parts) > 2:                       # timestamps from the pass's own kernel trace, joined by dispatch id
                with open(parts[2], newline="") as f:
                    for r in csv.DictReader(f):
                        stamp[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
            recs = []
            with open(path, newline="") as f:
                for r in csv.DictReader(f):
                    if "Start_Timestamp" in r and r["Start_Timestamp"]:
                        se = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
                    else:
                        se = stamp.get(r["Dispatch_Id"])
                        if se is None:
                            continue
                    recs.append((se[0], se[1], short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"]),
                                 int(r.get("Grid_Size", 0) or 0), r["Dispatch_Id"]))
            if not recs:
                sys.exit("PMC pass %s: no rows with a timestamp" % path)
            lo, hi = min(x[0] for x in recs), max(x[1] for x in recs)
            pclock = pick_clock(pblk["timed_region_clock_ns"], lo, hi)
            if pclock is None:
                sys.exit("PMC pass %s: no clock matches" % path)
            p0, p1 = pblk["timed_region_clock_ns"][pclock]
            sclk = float(pblk.get("telemetry_mean", {}).get("sclk_mhz", 0) or 0) * 1e6
            disp = set()
            ser_ns = small_ns = 0
            for s0, e0, k, cn, cv, grid, did in recs:
                if e0 <= p0 or s0 >= p1:
                    continue
                if did not in disp:
                    ser_ns += e0 - s0              # kernels run one at a time under counter collection: their own durations
                    if grid and grid < 64 * 1024:
                        small_ns += e0 - s0
                disp.add(did)
                tot[cn] += cv / psteps
                per_kernel[k][cn] += cv / psteps
                if cn == "GRBM_GUI_ACTIVE" and grid and grid < 64 * 1024:      # fewer than one wave per SIMD: cannot fill the chip
                    small_gui += cv / psteps
            for cn in {x[3] for x in recs}:
                pm.setdefault("passes", {})[cn] = {"blocks": psteps, "dispatches_per_block": len(disp) / psteps, "sclk_hz": sclk,
                                                   "sum_serialised_kernel_s": ser_ns / 1e9 / psteps,
                                                   "sum_serialised_small_grid_s": small_ns / 1e9 / psteps,
                                                   "wall_s_per_block_under_pmc": pblk["seconds_per_block"]}
        for cn, v in tot.items():
            pm[cn + "_per_block"] = v
        if "SQ_INSTS_VALU" in tot:
            lanes = tot["SQ_INSTS_VALU"] * 64
            pm["valu_lane_instr_per_block"] = lanes
            pm["frac_of_issue_limit"] = lanes / (wall * ISSUE_LIMIT_MULTI)
            pm["floor_s_at_issue_limit"] = lanes / ISSUE_LIMIT_MULTI
        if "GRBM_GUI_ACTIVE" in tot:
            sclk = pm["passes"]["GRBM_GUI_ACTIVE"]["sclk_hz"] or 2.4e9
            pm["sum_exclusive_kernel_s"] = tot["GRBM_GUI_ACTIVE"] / sclk
            pm["latency_bound_exclusive_s"] = small_gui / sclk
        top = sorted(per_kernel.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:30]
        pm["by_kernel"] = [dict(kernel=k, **{c + "_per_block": v for c, v in d.items()}) for k, d in top]
        rep["pmc"] = pm
    js = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(js)
    print(js[:6000])


if __name__ == "__main__":
    main()
