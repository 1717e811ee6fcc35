#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`
on ROCm 7.2) into a per-kernel table: calls, avg/min/max/total device time, share.
With --pmc also averages every collected counter per kernel.
usage: summarize_rocpd.py <results.db> [--pmc]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda key: [x for x in tabs if key in x][0]
    kd, ks = t("kernel_dispatch"), t("kernel_symbol")
    rows = c.execute(
        f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"
    ).fetchall()
    tot = sum(r[5] for r in rows) or 1
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:20]:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:10.1f} {r[3]/1e3:10.1f} {r[4]/1e3:10.1f} {r[5]/1e6:10.2f} {100*r[5]/tot:6.1f}")
    if "--pmc" in sys.argv:
        pe, pi = t("pmc_event"), t("info_pmc")
        q = (f"select s.kernel_name, p.name, avg(e.value), count(*) from {pe} e join {pi} p on e.pmc_id=p.id "
             f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name "
             f"order by s.kernel_name, p.name")
        print("\nper-dispatch counter averages")
        gui = {}
        for r in c.execute(q):
            if "rfa" in r[0]:
                print(f"{r[0][:60]:60s} {r[1]:36s} {r[2]:18.1f} (n={r[3]})")
                if r[1] == "GRBM_GUI_ACTIVE":
                    gui[r[0]] = r[2]
        # effective shader clock of a kernel IN THIS (profiled) PASS = GRBM_GUI_ACTIVE (cycles an XCD was busy with the
        # dispatch, averaged over the counter's instances) / the dispatch's duration in the same database
        # (MI355X_MICROARCH.md, "DVFS give-back": the chip clocks to its power budget; profiled passes run a few % lower)
        dur = {r[0]: r[2] for r in rows}
        for name, cyc in gui.items():
            if dur.get(name):
                print(f"{name[:60]:60s} {'EFFECTIVE_CLOCK_GHZ':36s} {cyc / dur[name]:18.3f} (GRBM_GUI_ACTIVE / avg duration {dur[name] / 1e3:.1f} us)")


if __name__ == "__main__":
    main()
