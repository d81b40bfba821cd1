#!/usr/bin/env python
"""Average PMC counter values per dispatch of kernels matching a substring:  pmc_kernel.py <db> <substr>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select counter_name, count(*), avg(counter_value) from pmc_events where name like ? group by counter_name",
                   ('%' + sys.argv[2] + '%',)).fetchall()
for n, c, v in rows:
    print(f'{n:40s} n={c:4d} avg={v:16.1f}')
