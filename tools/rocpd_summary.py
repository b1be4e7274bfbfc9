#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a per-kernel stats table (name, calls, total/avg/min/max us, %)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, (end - start) as dur from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total_ms | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for name, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| {short} | {n} | {t / 1e6:.3f} | {t / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * t / total:.2f} |")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
