import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%8.1f us x%-4s %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], n[:90]))
