"""dev: average of every counter per kernel-name pattern from a rocprofv3 --pmc run: pmc_dump.py <dir> <name-pattern>"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
for name, avg, n in c.execute("select counter_name, avg(counter_value), count(*) from pmc_events where name like ? group by counter_name",
                              ("%" + sys.argv[2] + "%",)):
    print(f"{name:36s} {avg:16.1f}  ({n} launches)")
