"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count/total/avg (us)."""
import re
import sqlite3
import sys


def main(path, top=18):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                      "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>5s}")
    for r in rows[:top]:
        n = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:62]
        print(f"{n:62s} {r[1]:7d} {r[2]/1e3:9.2f} {r[3]:8.1f} {r[4]:8.1f} {r[5]:8.1f} {100*r[2]/tot:5.1f}")
    print(f"total kernel time {tot/1e3:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 18)
