#!/usr/bin/env python3
"""Per process (= rank) of a rocprofv3 --kernel-trace run over the multi-GPU bench: device time of the attention
kernels (rfa::*), of RCCL's kernels (ncclDevKernel* / rccl*), the part of the RCCL time that lies UNDER an attention
kernel on the timeline (hidden) and the part outside any (exposed).  usage: summarize_overlap.py <dir with *_results.db>"""
import glob
import os
import sqlite3
import sys


def intervals(c, kd, ks, like):
    q = (f"select d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id where " +
         " or ".join(f"s.kernel_name like '%{p}%'" for p in like) + " order by d.start")
    return c.execute(q).fetchall()


def union(iv):
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def overlap(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True))
    print(f"{'database':40s} {'attention_ms':>13s} {'rccl_ms':>10s} {'rccl_hidden_ms':>15s} {'rccl_exposed_ms':>16s} {'hidden_frac':>12s}")
    for db in dbs:
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        t = lambda key: [x for x in tabs if key in x][0]
        try:
            kd, ks = t("kernel_dispatch"), t("kernel_symbol")
        except IndexError:
            continue
        att = union(intervals(c, kd, ks, ["rfa"]))
        rc = union(intervals(c, kd, ks, ["nccl", "rccl"]))
        ta, tr = sum(b - a for a, b in att), sum(b - a for a, b in rc)
        hid = overlap(att, rc)
        print(f"{os.path.basename(db)[:40]:40s} {ta / 1e6:13.3f} {tr / 1e6:10.3f} {hid / 1e6:15.3f} {(tr - hid) / 1e6:16.3f} "
              f"{(hid / tr if tr else 0):12.3f}")


if __name__ == "__main__":
    main()
