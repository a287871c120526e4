#!/usr/bin/env python3
"""Reduce bench.py's ZKLC_BENCH_SAMPLE file (JSON lines [t, {thread name: "file:line"}], one per 20 ms of the timed region):
per pipeline thread the share of samples by location, and for the Ed25519 prover threads the long stretches spent anywhere but in
the proving call -- with where the OTHER threads were meanwhile.       python tools/thread_samples.py gpurun_out/r06u_samples.jsonl"""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1])]
print("%d samples over %.1f s" % (len(rows), rows[-1][0] - rows[0][0]))
by = collections.defaultdict(collections.Counter)
for t, r in rows:
    for th, loc in r.items():
        by[th][loc] += 1
for th in sorted(by):
    tot = sum(by[th].values())
    print("%-28s %5d samples: %s" % (th, tot, ", ".join("%s %.0f%%" % (k, 100.0 * v / tot) for k, v in by[th].most_common(4))))
workers = sorted(th for th in by if th.startswith("zklc-ed_worker"))
prove_loc = {th: by[th].most_common(1)[0][0] for th in workers}
for th in workers:
    run, runs = None, []
    for t, r in rows:
        loc = r.get(th)
        if loc is not None and loc != prove_loc[th]:
            if run is None or run[2] != loc:
                if run is not None:
                    runs.append(run)
                run = [t, t, loc]
            run[1] = t
        elif run is not None:
            runs.append(run)
            run = None
    if run is not None:
        runs.append(run)
    runs = sorted((r for r in runs if r[1] - r[0] >= 0.2), key=lambda r: r[0])
    print("\n%s: %d stretches >= 0.2 s outside %s" % (th, len(runs), prove_loc[th]))
    for a, b, loc in runs[:14]:
        mid = min(rows, key=lambda x: abs(x[0] - (a + b) / 2))[1]
        others = {k: v for k, v in mid.items() if k != th}
        print("   %.2f-%.2f s (%.2f s) at %s | meanwhile: %s" % (a, b, b - a, loc, ", ".join("%s@%s" % (k.replace("zklc-", ""), v) for k, v in sorted(others.items()))))
