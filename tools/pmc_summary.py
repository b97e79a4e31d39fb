"""Summarise a `rocprofv3 --pmc ... --output-format csv` run: per kernel, the average of every collected counter."""
import csv
import glob
import sys


def main(prof_dir, substr=""):
    acc = {}
    for f in glob.glob(prof_dir + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if substr and substr not in k:
                continue
            d = acc.setdefault(k, {})
            d.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print(k[:110])
        for c, v in sorted(d.items()):
            print("   %-32s calls %5d  avg %16.1f" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
