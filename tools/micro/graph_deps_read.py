import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])          # kernels per replay
tr = tr[-n * 3:-n * 2] if len(tr) >= 3 * n else tr[-n:]
t0 = int(tr[0]["Start_Timestamp"])
for r in tr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f dur %6.1f q%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")))
