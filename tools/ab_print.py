"""Print the A/B lines of tools/longctx_ab.py outputs compactly: python tools/ab_print.py files..."""
import json
import sys

for f in sys.argv[1:]:
    print("==", f)
    for line in open(f):
        try:
            d = json.loads(line)
        except Exception:
            continue
        print({k: d[k] for k in ("fs", "fused", "grouped", "splits", "ms_per_token", "us_per_layer_vs_first", "status") if k in d})
