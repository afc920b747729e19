#!/usr/bin/env python3
"""Summary of a combiner batch log (BBP_BATCH_LOG=path; one line per batch: start ms, duration ms, kind, target, size, prove batches
already in flight at its start, requests left queued): python tools/batch_log.py LOG [LOG ...]"""
import sys
for path in sys.argv[1:]:
    rows = [l.split() for l in open(path) if l.strip()]
    pr = [(float(r[0]), float(r[1]), int(r[4]), int(r[5]), int(r[6])) for r in rows if r[2] == "prove"]
    if not pr:
        print(path, "no prove batches")
        continue
    pr.sort()
    span = pr[-1][0] + pr[-1][1] - pr[0][0]
    n = sum(p[2] for p in pr)
    sizes = [p[2] for p in pr]
    print("%s: %d prove batches, %d proofs in %.0f ms = %.0f/s; sizes min/median/max %d/%d/%d" % (path, len(pr), n, span, n / span * 1e3, min(sizes), sorted(sizes)[len(sizes) // 2], max(sizes)))
    print("   sequence (size@inflight): " + " ".join("%d@%d" % (p[2], p[3]) for p in pr[:60]))
    gaps = [b[0] - a[0] for a, b in zip(pr, pr[1:])]
    print("   per-batch ms/proof by size class: " + ", ".join("%s: %.1f us (%d)" % (lab, 1e3 * sum(p[1] for p in g) / max(1, sum(p[2] for p in g)), len(g))
          for lab, g in (("<512", [p for p in pr if p[2] < 512]), ("512-1279", [p for p in pr if 512 <= p[2] < 1280]), ("1280-2047", [p for p in pr if 1280 <= p[2] < 2048]), (">=2048", [p for p in pr if p[2] >= 2048])) if g))
