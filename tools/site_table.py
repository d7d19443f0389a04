"""Print the per-site table of a `bench.py --profile-json` file."""
import json
import sys

d = json.load(open(sys.argv[1]))
tot = sum(s["ms"] for s in d["sites"])
for s in sorted(d["sites"], key=lambda s: -s["ms"]):
    print("%-16s %-38s n=%3d %8.1f us %5.1f%% avg %6.1f us %7.1f TF/s" % (
        s["site"], s["kernel"], s["launches"], s["ms"] * 1e3, 100 * s["ms"] / tot, s["ms"] * 1e3 / s["launches"],
        s["flops"] / s["ms"] / 1e9 if s["ms"] else 0))
print("sum of event-timed sites: %.1f us; frame: %.3f ms" % (tot * 1e3, d["line"]["ms_per_step"]))
