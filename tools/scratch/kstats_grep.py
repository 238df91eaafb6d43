import csv, sys
pat = sys.argv[2].split("|")
for r in csv.DictReader(open(sys.argv[1])):
    if any(p in r["Name"] for p in pat):
        print(f'{r["Name"].replace("(anonymous namespace)::","")[:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:8.2f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
